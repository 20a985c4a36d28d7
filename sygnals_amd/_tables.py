"""Host-side (float64 NumPy) construction of the constant tables the kernels consume:
analysis windows, twiddles, the Slaney mel filterbank and its block-sparse MFMA packing,
DCT-II rows, lifter, spectral-contrast band plan.  Pure host logic, no GPU needed.

Behaviour follows what the reference obtains from librosa / SciPy at
sygnals/core/features/manager.py:184-227, cepstral.py:106-115 and
frequency_domain.py:200-207 (librosa>=0.10 semantics).
"""
from __future__ import annotations

import functools

import numpy as np
import scipy.signal

WAVES = 16          # default waves per workgroup of the fused kernel (stft_mel.hip): 8 or 16
MAXW = 16
MAX_BANDS = 16      # SYG_MAX_BANDS


def analysis_window(window, win_length: int, n_fft: int) -> np.ndarray:
    """Periodic window (fftbins=True) of win_length, centre-padded to n_fft, float64."""
    if isinstance(window, np.ndarray) or isinstance(window, (list,)):
        w = np.asarray(window, dtype=np.float64)
        if w.ndim != 1 or w.shape[0] != win_length:
            raise ValueError(f"window array must be 1-D of length win_length={win_length}")
    else:
        w = scipy.signal.get_window(window, win_length, fftbins=True).astype(np.float64)
    if win_length > n_fft:
        raise ValueError(f"win_length={win_length} must be <= n_fft={n_fft}")
    if win_length < n_fft:
        lp = (n_fft - win_length) // 2
        w = np.concatenate([np.zeros(lp), w, np.zeros(n_fft - win_length - lp)])
    return w


def twiddles(n: int) -> np.ndarray:
    """W_n^k = exp(-2*pi*i*k/n), k = 0..n-1, as float32 [n, 2] (evaluated in float64)."""
    k = np.arange(n, dtype=np.float64)
    ang = -2.0 * np.pi * k / n
    return np.stack([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32)


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f * (3.0 / 200.0)
    with np.errstate(divide="ignore"):
        log = 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) * (27.0 / np.log(6.4))
    return np.where(f >= 1000.0, log, lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), m * (200.0 / 3.0))


def mel_filterbank(sr: float, n_fft: int, n_mels: int = 128, fmin: float = 0.0, fmax=None) -> np.ndarray:
    """Slaney-scale, Slaney-normalised triangular filterbank, float32 [n_mels, 1+n_fft//2].

    librosa stores the basis in float32 (and applies the area normalisation to the float32
    triangles); both roundings are reproduced because the float64 reference pipeline
    multiplies by exactly these float32 values.
    """
    if fmax is None:
        fmax = sr / 2.0
    if n_mels < 1:
        raise ValueError("n_mels must be >= 1")
    freqs = np.fft.rfftfreq(n_fft, 1.0 / sr)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    d = np.diff(edges)
    ramps = edges[:, None] - freqs[None, :]
    up = -ramps[:-2] / d[:-1, None]
    down = ramps[2:] / d[1:, None]
    tri = np.maximum(0.0, np.minimum(up, down)).astype(np.float32)
    enorm = 2.0 / (edges[2:] - edges[:-2])
    return (tri.astype(np.float64) * enorm[:, None]).astype(np.float32)


MEL_MIN_STEPS = 28      # the kernel issues 7 groups of 4 MFMA steps unconditionally
MEL_MAX_STEPS = 1088    # positions of one skewed power row (P_STRIDE 1090, a multiple of 4 below it)


def row_pos(k):
    """Word position of bin k inside a skewed LDS power row of the fused kernel: one pad word after every 16 bins."""
    return k + (k >> 4)


def pack_mel_plan(basis: np.ndarray, waves: int = WAVES):
    """Block-sparse packing of a [M, F] filterbank for v_mfma_f32_4x4x1_16b_f32 (sixteen 4 x 4 blocks per instruction).

    The rows are taken in groups of four; only the non-zero column range of a group is kept and cut into chunks; chunk c
    is slot c: wave c // 4 runs its four slots side by side, one ROW POSITION per slot and MFMA step.  A slot walks
    consecutive words of the kernel's skewed power rows (row_pos: a pad word after every 16 bins, which gets a zero
    weight), so the kernel addresses its B operands with one base and immediate offsets.  `steps` (positions per slot)
    is the smallest multiple of 4 >= 28 for which the chunks fit the 4 * waves slots.  Returns (wpacked float32, plan
    int32):

      wpacked  [waves][steps / 4][64 lanes][4]  A operands, four steps per 16-byte load: lane l of step i carries
               basis[4 g + (l & 3)][bin at position p0 + i] of its slot s = l >> 4 (the four frame groups (l >> 2) & 3
               see the same weight), zero at pad positions, past the chunk and past the last mel row; then 8 groups of
               zero rows; then the int32 tables (bit patterns) [slot p0 | slot group | group first slot | group slot
               count], 64 entries each
      plan     {2, waves, steps, n_groups, table_off (in floats)}
    """
    basis = np.asarray(basis, dtype=np.float32)
    M, F = basis.shape
    if waves not in (8, 16):
        raise ValueError("waves must be 8 or 16")
    ng = (M + 3) // 4
    if ng > 64:
        raise ValueError(f"n_mels={M} needs {ng} groups of four rows (max 64: n_mels <= 256)")
    nslots = 4 * waves
    rng = []
    for g in range(ng):
        cols = np.flatnonzero(np.any(basis[4 * g:4 * g + 4] != 0, axis=0))
        rng.append((int(cols[0]), int(cols[-1]) + 1) if cols.size else (0, 0))

    def chunks_of(a, b, steps):
        """Chunks of the position range [row_pos(a), row_pos(b - 1)] in pieces of `steps` positions."""
        if b <= a:
            return []
        p, pe = row_pos(a) & ~1, row_pos(b - 1) + 1          # even start: the kernel reads two positions per LDS word pair
        return [(q, min(q + steps, pe)) for q in range(p, pe, steps)]

    # every non-empty group needs a slot of its own whatever the chunk length; positions per slot are bounded by a row
    n_nonempty = sum(1 for a, b in rng if b > a)
    if n_nonempty > nslots:
        raise ValueError(f"filterbank has {n_nonempty} non-empty groups of four mel rows, a {waves}-wave plan holds "
                         f"{nslots} slots (n_mels <= {4 * nslots} at most): use the dense filterbank path")
    steps = MEL_MIN_STEPS
    while sum(len(chunks_of(a, b, steps)) for a, b in rng) > nslots:
        steps += 4
        if steps > MEL_MAX_STEPS:
            raise ValueError(f"filterbank does not fit {nslots} slots of at most {MEL_MAX_STEPS} row positions")
    slot_p0 = np.zeros(64, np.int32); slot_g = -np.ones(64, np.int32)
    g_first = np.zeros(64, np.int32); g_cnt = np.zeros(64, np.int32)
    wts = np.zeros((nslots, steps, 4), np.float32)             # [slot][step][row]
    c = 0
    for g, (a, b) in enumerate(rng):
        g_first[g] = c
        for (p0, p1) in chunks_of(a, b, steps):                 # ascending positions = ascending bins
            slot_p0[c] = p0; slot_g[c] = g
            for i in range(p1 - p0):
                p = p0 + i
                if p % 17 == 16:                                # pad word of the skewed row
                    continue
                k = p - p // 17
                if a <= k < b:                                  # (a chunk may start one position below its range)
                    rows = basis[4 * g:min(4 * g + 4, M), k]
                    wts[c, i, :rows.shape[0]] = rows
            c += 1
        g_cnt[g] = c - g_first[g]
    assert c <= nslots
    lane = np.arange(64)
    # [wave][step][lane] = wts[wave * 4 + (lane >> 4)][step][lane & 3]
    A = wts.reshape(waves, 4, steps, 4)[:, lane >> 4, :, lane & 3]          # -> [64 lanes, waves, steps]
    A = np.transpose(A, (1, 2, 0))                                          # [waves][steps][64]
    # device layout: [wave][group of 4 steps][lane][step in group]
    A = A.reshape(waves, steps // 4, 4, 64).transpose(0, 1, 3, 2).reshape(-1)
    tail = np.zeros(8 * 64 * 4, np.float32)
    table = np.concatenate([slot_p0, slot_g, g_first, g_cnt]).astype(np.int32)
    wpacked = np.concatenate([A.astype(np.float32), tail, table.view(np.float32)])
    plan = np.array([2, waves, steps, ng, A.size + tail.size], dtype=np.int32)
    return np.ascontiguousarray(wpacked), plan


SEG_SLOTS = 128         # lane slots of the segment-sum projection: 2 passes x 64 lanes
SEG_LEAD_MAX = 4        # window words that may lie before the piece (the kernel's first SEG_LEAD_MAX steps test `lead <= i`)
SEG_WINDOW = 17         # row words a lane reads per pass


def _distinct_banks(options):
    """Maximum bipartite matching lane -> LDS bank: options[l] = candidate banks of lane l, in order of preference.
    Returns one bank per lane; unmatched lanes (more lanes than free banks among their candidates) take their first."""
    owner = {}

    def augment(u, seen):
        for b in options[u]:
            if b in seen:
                continue
            seen.add(b)
            if b not in owner or augment(owner[b], seen):
                owner[b] = u
                return True
        return False

    for u in range(len(options)):
        augment(u, set())
    got = {u: b for b, u in owner.items()}
    return [got.get(u, options[u][0]) for u in range(len(options))]


def pack_mel_segments(sr: float, n_fft: int, n_mels: int, fmin: float = 0.0, fmax=None, basis=None, n_pass: int = 2,
                      row_base: int = 0, block: int = 16, window: int = None):
    """Piece table of the fused kernel's per-wave mel projection by SEGMENT SUMS (stft_mel.hip, MODE 6).

    A triangular filterbank is piecewise linear in the bin index: between two neighbouring band edges e[s], e[s+1]
    (segment s) the only non-zero weights are the rising side of band s and the falling side of band s - 1, both affine
    in k.  With T0 = sum p[k] and T1 = sum i' p[k] over a run of bins (i' = distance from the run's LAST bin) the run's
    contribution to band s is aR T0 + bR T1 and to band s - 1 aF T0 + bF T1 -- two sums per run instead of two weights
    per bin, and no weight matrix.  The bins of a segment are cut at the 16-bin blocks of the kernel's skewed power rows
    (row_pos: no pad word inside a piece); one lane sums one piece (<= 16 bins), the pieces of a segment sit in
    neighbouring lanes of one 16-lane row (a "run") and are combined by two segmented scans on the DPP path: rising
    sums towards the run's last lane, falling sums towards its first lane, so that band s = R(run s, last lane) +
    F(run s + 1, first lane) meets in neighbouring lanes.  The pieces are spread evenly over the eight 16-lane rows.

    A lane reads a WINDOW of 17 consecutive row words that starts `lead` words before its piece (0 .. SEG_LEAD_MAX, as
    far as the piece still fits): the 32 lanes that share an LDS access are given windows in distinct banks wherever a
    matching exists (the piece starts alone collide: segment starts fall anywhere between the block starts, which are
    17 words apart).

    n_pass: passes of 64 lanes (2: the frame-length-2048 kernel, 128 lane slots; 4: the frame-length-4096 kernel).
    block / window: the rows are cut at `block`-bin boundaries (16, or 8 for the frame-length-256 kernel, whose lanes then
    read windows of 9 words instead of 17: window = block + 1).
    row_base: words in front of bin 0 inside the kernel's row (0: the frame-length-2048 / 4096 kernels; 4 where the first
    piece may be shorter than the window's entry steps and needs room for its lead).
    Returns float32 [n_pass][2][64 lanes][4] (bit patterns for the integers; the kernel keeps the table in LDS and
    reads two 16-byte words per lane and pass):
        q0 = (BYTE offset of the window inside a power row | (lead + bins in the piece) << 16 | lead << 24 (idle lane:
              0 bins, lead 7),
              band stored by this lane or -1,
              links of the rising scan, one BYTE per step: 1 = the lane 1 / 2 / 4 / 8 below belongs to the same run,
              links of the falling scan: the lane 1 / 2 / 4 / 8 above)
        q1 = (aR, bR, aF, bF)
    Raises ValueError when the filterbank does not fit the 128 slots / 16-lane runs, or when `basis` (the float32
    matrix the oracle multiplies by) is not reproduced by the affine pieces to 2e-7 of its largest weight.
    """
    F = n_fft // 2 + 1
    SLOTS = 64 * int(n_pass)
    if block not in (8, 16):
        raise ValueError("block must be 8 or 16")
    WIN = block + 1 if window is None else int(window)

    def rp(k):
        return row_pos(k) + row_base

    fmax = sr / 2.0 if fmax is None else fmax
    e = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    d = np.diff(e)
    if not np.all(d > 0):
        raise ValueError("band edges must increase")
    en = 2.0 / (e[2:] - e[:-2])
    df = sr / n_fft
    seg = np.searchsorted(e, np.arange(F) * df, side="right") - 1     # e[s] <= f_k < e[s + 1]
    runs = [[] for _ in range(n_mels + 1)]
    for blk in range((F + block - 1) // block):
        ks = np.arange(blk * block, min(blk * block + block, F))
        for s in np.unique(seg[ks]):
            if s < 0 or s > n_mels:
                continue
            kk = ks[seg[ks] == s]
            k0, k1 = int(kk[0]), int(kk[-1]) + 1
            fl = df * (k1 - 1)                       # frequency of the piece's last bin (i' = 0)
            aR = en[s] * (fl - e[s]) / d[s] if s <= n_mels - 1 else 0.0
            bR = -en[s] * df / d[s] if s <= n_mels - 1 else 0.0
            aF = en[s - 1] * (e[s + 1] - fl) / d[s] if s >= 1 else 0.0
            bF = en[s - 1] * df / d[s] if s >= 1 else 0.0
            runs[s].append((k0, k1, aR, bR, aF, bF))
    idle = (0, 0, 0.0, 0.0, 0.0, 0.0)
    runs = [run if run else [idle] for run in runs]       # a segment without a bin still separates its neighbours
    if max(len(r) for r in runs) > 16:
        raise ValueError("a segment spans more than 16 blocks")
    # runs -> rows of 16 lanes, in order; a run stays inside one row, the idle lanes at a row's end join the run before
    # them (its rising total then waits in the row's last lane).  How many lanes of each row are used is free: a small
    # deterministic search over row fillings keeps the one whose windows collide least.
    n_rows = SLOTS // 16
    total = sum(len(r) for r in runs)

    lens = [len(r) for r in runs]

    def rows_needed(first):
        """Rows a greedy packing of runs[first:] takes (a run never straddles a row)."""
        rows, used = 0, 16
        for n in lens[first:]:
            if used + n > 16:
                rows, used = rows + 1, 0
            used += n
        return rows

    def fill(rng):
        """Rows of 16 lanes, a random feasible number of runs per row (rng None: as many as fit).  Idle lanes go to the
        row's end (they join the last run: its rising total must reach lane 15) or to the row's start (they join the first
        run, whose falling total must reach lane 0); with every piece within 7 lanes of where its run's total is read
        the scans need three steps instead of four (`short`).  Returns (slots, short)."""
        slots, nxt, short = [], 0, max(lens) <= 8
        for row in range(n_rows):
            first, used, takes = nxt, 0, []
            while nxt < len(runs) and used + lens[nxt] <= 16:
                used += lens[nxt]; nxt += 1
                if rows_needed(nxt) <= n_rows - row - 1:
                    takes.append((nxt, used))
            if first == len(runs):
                slots += [(idle, -2 - row)] * 16
                continue
            nxt, used = takes[-1] if rng is None else takes[int(rng.integers(len(takes)))]
            spare = 16 - used
            lo = max(0, spare - (8 - lens[first]))                   # idle lanes the row's start cannot take
            hi = min(spare, 8 - lens[nxt - 1]) if nxt - 1 > first else min(spare, 8 - lens[first])
            if lo <= hi:
                at_end = hi if rng is None else int(rng.integers(lo, hi + 1))
            else:
                at_end, short = spare, False
            slots += [(idle, first)] * (spare - at_end)
            for r in range(first, nxt):
                slots += [(pc, r) for pc in runs[r]]
            slots += [(idle, nxt - 1)] * at_end
        return slots, short

    def leads_of(n, q):
        """Leads a piece of n bins at row position q may take: the window holds 17 words, its first SEG_LEAD_MAX words are
        entered by `lead <= i` alone, so the piece must not end before them."""
        return [ld for ld in range(max(0, SEG_LEAD_MAX - n), min(SEG_LEAD_MAX, WIN - n) + 1) if q - ld >= 0]

    def windows(slots):
        """Leads per group of 32 lanes (one LDS access) so that the windows start in distinct banks where a matching
        exists; returns (window starts, leads, LDS cycles of the 2 x 17 reads)."""
        start, lead, cycles = [0] * SLOTS, [0] * SLOTS, 0
        for g in range(SLOTS // 32):
            lanes = [j for j in range(32 * g, 32 * g + 32) if slots[j][0][1] > slots[j][0][0]]
            opts = []
            for j in lanes:
                k0, k1 = slots[j][0][:2]
                opts.append([(rp(k0) - ld) % 32 for ld in leads_of(k1 - k0, rp(k0))])
                if not opts[-1]:
                    raise ValueError("a piece at the row's start is shorter than the window's entry steps")
            for j, b in zip(lanes, _distinct_banks(opts)):
                k0, k1 = slots[j][0][:2]
                lead[j] = next(ld for ld in leads_of(k1 - k0, rp(k0)) if (rp(k0) - ld) % 32 == b)
                start[j] = rp(k0) - lead[j]
            for j in range(32 * g, 32 * g + 32):     # idle lanes re-read a neighbour's window (same address: no conflict)
                if j not in lanes:
                    start[j] = start[lanes[0]] if lanes else 0
            st = np.unique([start[j] for j in range(32 * g, 32 * g + 32)])
            cycles += WIN * int(np.bincount(st % 32, minlength=32).max())
        return start, lead, cycles

    if total > SLOTS:
        raise ValueError(f"filterbank needs {total} lane slots, the projection has {SLOTS}")
    if rows_needed(0) > n_rows:
        raise ValueError(f"filterbank needs more than the {SLOTS} lane slots of the projection")
    best = None
    rng = np.random.default_rng(12345)
    for trial in range(1500):
        slots, short = fill(None if trial == 0 else rng)
        start, lead, cycles = windows(slots)
        cost = cycles + (0 if short else 4)          # (a fourth scan step costs about as much as two LDS cycles)
        if best is None or cost < best[0]:
            best = (cost, slots, start, lead)
        ideal = WIN * (SLOTS // 32)
        if cost == ideal or (trial >= 300 and cost <= ideal + 4):     # (conflict-free with a fourth scan step is as good)
            break
    _, slots, start, lead = best
    run_of = np.array([r for _, r in slots])
    tab = np.zeros((int(n_pass), 2, 64, 4), np.float32)
    ti = tab.view(np.int32)
    for j, (pc, r) in enumerate(slots):
        p, l = divmod(j, 64)
        k0, k1, aR, bR, aF, bF = pc
        n = k1 - k0
        ti[p, 0, l, 0] = (4 * start[j]) | (((lead[j] + n) if n else 0) << 16) | ((lead[j] if n else 7) << 24)
        last = (j + 1 == SLOTS) or run_of[j + 1] != r
        ti[p, 0, l, 1] = r if (last and 0 <= r <= n_mels - 1) else -1
        live = [jj for jj in range(16 * (j // 16), 16 * (j // 16) + 16) if run_of[jj] == r and slots[jj][0][1] > slots[jj][0][0]]
        tab[p, 1, l] = (aR, bR, aF, bF)
        for i, dd in enumerate((1, 2, 4, 8)):
            # a link is only set where something non-zero can arrive through it (then a layout whose pieces all lie within
            # 7 lanes of their totals has no step-8 link, and the kernel skips that step)
            if j - dd >= 0 and (j - dd) // 16 == j // 16 and run_of[j - dd] == r and live and min(live) <= j - dd:
                ti[p, 0, l, 2] |= 1 << (8 * i)
            if j + dd < SLOTS and (j + dd) // 16 == j // 16 and run_of[j + dd] == r and live and max(live) >= j + dd:
                ti[p, 0, l, 3] |= 1 << (8 * i)
    if basis is not None:
        Wr = segments_weights(tab, n_mels, F, row_base)
        tol = 2e-7 * float(np.max(np.abs(basis)))
        if Wr.shape != basis.shape or float(np.max(np.abs(Wr - basis))) > tol:
            raise ValueError("the filterbank is not reproduced by affine pieces")
    return np.ascontiguousarray(tab)


def pack_mel_segments_rows(sr: float, n_fft: int, n_mels: int, fmin: float = 0.0, fmax=None, basis=None, rows: int = 2,
                           row_words: int = 568, n_pass: int = 2, row_base: int = 4, block: int = 16):
    """Piece table for a wave that projects `rows` power rows of one frame length in ONE call (the frame-length-1024
    kernel: two frames per transform): the single-row table of pack_mel_segments(..., n_pass) repeated per row, the window
    offsets moved by the row's offset (row_words words per row) and the band words tagged with the row (band | row << 8).
    A row's last lane never stores a band (its last run is the segment above the last band), so nothing leaks from one
    row's passes into the next's.  Returns float32 [rows * n_pass][2][64][4]."""
    one = pack_mel_segments(sr, n_fft, n_mels, fmin, fmax, basis=basis, n_pass=n_pass, row_base=row_base, block=block)
    if n_mels > 255 or (rows - 1) * row_words * 4 + 0xFFFF // 8 > 0xFFFF:
        raise ValueError("too many bands / rows for the packed table words")
    out = np.concatenate([one.copy() for _ in range(rows)], axis=0)
    oi = out.view(np.int32)
    for r in range(rows):
        blk = oi[r * n_pass:(r + 1) * n_pass]
        w0 = blk[:, 0, :, 0]
        blk[:, 0, :, 0] = (w0 & ~0xFFFF) | ((w0 & 0xFFFF) + 4 * r * row_words)
        bw = blk[:, 0, :, 1]
        blk[:, 0, :, 1] = np.where(bw >= 0, bw | (r << 8), -1)
        if int(blk[n_pass - 1, 0, 63, 1]) >= 0:
            raise ValueError("a row's last lane stores a band")
    return np.ascontiguousarray(out)


def segments_read_cycles(tab: np.ndarray, window: int = 17) -> int:
    """LDS cycles of one wave's n_pass x 17 window reads (ds_read_b32: the 32 lanes of a half wave share a cycle unless
    two of them address different words of one bank); 34 per pass = conflict-free."""
    ti = tab.view(np.int32)
    total = 0
    for p in range(tab.shape[0]):
        for h in range(2):
            st = (ti[p, 0, 32 * h:32 * h + 32, 0] & 0xFFFF) // 4
            for i in range(window):
                total += int(np.bincount(np.unique(st + i) % 32, minlength=32).max())
    return total


def _seg_slot(tab: np.ndarray, j: int, row_base: int = 0):
    """Fields of lane slot j: (first bin, bins, band, rising links [4], falling links [4], aR, bR, aF, bF)."""
    ti = tab.view(np.int32)
    p, l = divmod(j, 64)
    w0 = int(ti[p, 0, l, 0])
    hi = (w0 >> 16) & 0xFF
    ld = ((w0 >> 24) & 7) if hi else 0
    pos = (w0 & 0xFFFF) // 4 + ld - row_base          # the piece's first bin sits `lead` words into the window
    lr, lf = int(ti[p, 0, l, 2]), int(ti[p, 0, l, 3])
    return (pos - pos // 17, hi - ld, int(ti[p, 0, l, 1]), [(lr >> (8 * i)) & 1 for i in range(4)],
            [(lf >> (8 * i)) & 1 for i in range(4)], tab[p, 1, l, 0], tab[p, 1, l, 1], tab[p, 1, l, 2], tab[p, 1, l, 3])


def segments_weights(tab: np.ndarray, n_mels: int, F: int, row_base: int = 0) -> np.ndarray:
    """The [n_mels, F] weight matrix a piece table stands for (float64; host check of pack_mel_segments): the kernel's
    data flow -- piece sums, the two segmented scans, the neighbour add -- applied to the rows of the identity."""
    NS = 64 * tab.shape[0]
    slots = [_seg_slot(tab, j, row_base) for j in range(NS)]
    R = np.zeros((NS, F)); Fv = np.zeros((NS, F))
    for j, (k0, n, _, lr, lf, aR, bR, aF, bF) in enumerate(slots):
        ip = np.arange(n - 1, -1, -1.0)
        R[j, k0:k0 + n] = float(aR) + float(bR) * ip
        Fv[j, k0:k0 + n] = float(aF) + float(bF) * ip
    for i, dd in enumerate((1, 2, 4, 8)):
        Rn, Fn = R.copy(), Fv.copy()
        for j in range(NS):
            if slots[j][3][i]:
                Rn[j] += R[j - dd]
            if slots[j][4][i]:
                Fn[j] += Fv[j + dd]
        R, Fv = Rn, Fn
    W = np.zeros((n_mels, F))
    for j in range(NS):
        b = slots[j][2]
        if b >= 0:
            W[b] = R[j] + (Fv[j + 1] if j + 1 < NS else 0.0)
    return W


def segments_project(tab: np.ndarray, P: np.ndarray, n_mels: int, row_base: int = 0) -> np.ndarray:
    """float32 emulation of the kernel's projection of ONE power row P [F] through a piece table: same sums, same order
    (piece sums by running prefix, segmented scans in steps 1, 2, 4, 8) -- host-side model for the CPU tests."""
    f32 = np.float32
    P = np.asarray(P, f32)
    NS = 64 * tab.shape[0]
    slots = [_seg_slot(tab, j, row_base) for j in range(NS)]
    R = np.zeros(NS, f32); Fv = np.zeros(NS, f32)
    for j, (k0, n, _, lr, lf, aR, bR, aF, bF) in enumerate(slots):
        c = f32(0); t1 = f32(0)
        for i in range(n):
            if i > 0:
                t1 = f32(t1 + c)
            c = f32(c + P[k0 + i])
        R[j] = f32(f32(bR) * t1 + f32(f32(aR) * c))         # (an fma in the kernel: one rounding less)
        Fv[j] = f32(f32(bF) * t1 + f32(f32(aF) * c))
    for i, dd in enumerate((1, 2, 4, 8)):
        Rn = R.copy(); Fn = Fv.copy()
        for j in range(NS):
            if slots[j][3][i]:
                Rn[j] = f32(R[j] + R[j - dd])
            if slots[j][4][i]:
                Fn[j] = f32(Fv[j] + Fv[j + dd])
        R, Fv = Rn, Fn
    out = np.zeros(n_mels, f32)
    for j in range(NS):
        b = slots[j][2]
        if b >= 0:
            out[b] = f32(R[j] + (Fv[j + 1] if j + 1 < NS else f32(0)))
    return out


def dct_matrix(n_out: int, n_in: int, dct_type: int = 2, norm="ortho") -> np.ndarray:
    """Rows k < n_out of scipy.fft.dct(type=dct_type, norm=norm) as a float32 matrix [n_out, n_in]."""
    import scipy.fft
    if n_out > n_in:
        raise ValueError(f"n_mfcc={n_out} cannot exceed n_mels={n_in}")
    D = scipy.fft.dct(np.eye(n_in), axis=0, type=dct_type, norm=norm)[:n_out]
    return np.ascontiguousarray(D.astype(np.float32))


def lifter_weights(n_mfcc: int, lifter: float):
    if lifter < 0:
        raise ValueError(f"MFCC lifter={lifter} must be a non-negative number")
    if lifter == 0:
        return None
    return (1.0 + (lifter / 2.0) * np.sin(np.pi * np.arange(1, n_mfcc + 1) / lifter)).astype(np.float32)


def contrast_plan(freqs: np.ndarray, sr: float, n_bands: int = 6, fmin: float = 200.0, quantile: float = 0.02):
    """Band bin ranges [lo, hi) and tail sizes k of librosa.feature.spectral_contrast.

    Octave edges [0, fmin, 2 fmin, ...]; bands after the first include the bin below;
    the last band runs to Nyquist; every other band drops its top bin; k is taken from
    the bin count *before* that drop.
    """
    freqs = np.atleast_1d(np.asarray(freqs, dtype=np.float64))
    if n_bands < 1 or int(n_bands) != n_bands:
        raise ValueError("n_bands must be a positive integer")
    if n_bands + 1 > MAX_BANDS:
        raise ValueError(f"n_bands must be <= {MAX_BANDS - 1}")
    if not 0.0 < quantile < 1.0:
        raise ValueError("quantile must lie in the range (0, 1)")
    if fmin <= 0:
        raise ValueError("fmin must be a positive number")
    octa = np.concatenate([[0.0], fmin * 2.0 ** np.arange(n_bands + 1)])
    if np.any(octa[:-1] >= 0.5 * sr):
        raise ValueError("Frequency band exceeds Nyquist. Reduce either fmin or n_bands.")
    lo = np.zeros(MAX_BANDS, np.int32); hi = np.zeros(MAX_BANDS, np.int32); kk = np.zeros(MAX_BANDS, np.int32)
    for b in range(n_bands + 1):
        idx = np.flatnonzero((freqs >= octa[b]) & (freqs <= octa[b + 1]))
        first, last = int(idx[0]), int(idx[-1])
        if b > 0:
            first -= 1
        if b == n_bands:
            last = len(freqs) - 1
        count = last - first + 1
        if b < n_bands:
            last -= 1
        lo[b], hi[b], kk[b] = first, last + 1, max(int(np.rint(quantile * count)), 1)
    return np.concatenate([[n_bands + 1], lo, hi, kk]).astype(np.int32)


def butter_padlen(sos: np.ndarray) -> int:
    sos = np.asarray(sos, dtype=np.float64)
    ntaps = 2 * sos.shape[0] + 1
    ntaps -= min(int((sos[:, 2] == 0).sum()), int((sos[:, 5] == 0).sum()))
    return 3 * ntaps
