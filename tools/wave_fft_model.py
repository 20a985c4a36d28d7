"""NumPy lane/register model of the wave-per-frame 2048-point real FFT used by
sygnals_amd/csrc/stft_mel.hip.  Validates the 16x16x4 decomposition, the LDS
index swizzles and the mirror-pair unit mapping before the HIP code is run.
Development aid only (not imported by the package)."""
import numpy as np

M = 1024  # complex points


def W(N, e):
    return np.exp(-2j * np.pi * e / N)


def swz1(c, b):          # exchange 1: y[c][b]  -> LDS complex index
    return c * 64 + (b ^ (4 * (c & 7)))


def swz2(c, cp, bp):     # exchange 2: t[c][c'][b'] -> LDS complex index (group of 4 contiguous)
    return c * 64 + 4 * (cp ^ (c & 7) ^ ((c >> 3) << 3)) + bp


def unit_groups(u):
    """unit u (0..127) -> (c, c', cm, cm') primary and mirror group."""
    if u < 112:
        c, cp = 1 + u // 16, u % 16
        return c, cp, 16 - c, 15 - cp
    if u < 120:
        cp = u - 112
        return 8, cp, 8, 15 - cp
    if u < 127:
        cp = u - 119
        return 0, cp, 0, 16 - cp
    return 0, 0, 0, 8


def model(x, win):
    xw = x * win
    z = xw[0::2] + 1j * xw[1::2]
    lane = np.arange(64)
    # pass 1: lane b, regs a: z[64a+b]
    v = np.stack([z[64 * a + lane] for a in range(16)], axis=0)          # [a][lane]
    y = np.stack([sum(v[a] * W(16, a * c) for a in range(16)) for c in range(16)], 0)
    y = y * np.stack([W(1024, lane * c) for c in range(16)], 0)
    lds = np.zeros(1024, complex)
    for c in range(16):
        lds[swz1(c, lane)] = y[c]
    # pass 2: lane=(c=lane>>2, b'=lane&3), regs a': y[c][4a'+b']
    c_l, bp_l = lane >> 2, lane & 3
    u_ = np.stack([lds[swz1(c_l, 4 * a + bp_l)] for a in range(16)], 0)
    t = np.stack([sum(u_[a] * W(16, a * cp) for a in range(16)) for cp in range(16)], 0)
    t = t * np.stack([W(64, bp_l * cp) for cp in range(16)], 0)
    lds2 = np.zeros(1024, complex)
    for cp in range(16):
        lds2[swz2(c_l, cp, bp_l)] = t[cp]
    # pass 3 + real split: lane handles units lane and lane+64
    X = np.full(1025, np.nan + 0j)
    for l in range(64):
        for u in (l, l + 64):
            c, cp, cm, cmp_ = unit_groups(u)
            g = np.array([lds2[swz2(c, cp, b)] for b in range(4)])
            gm = np.array([lds2[swz2(cm, cmp_, b)] for b in range(4)])
            G = np.array([sum(g[b] * W(4, b * d) for b in range(4)) for d in range(4)])
            Gm = np.array([sum(gm[b] * W(4, b * d) for b in range(4)) for d in range(4)])
            k = c + 16 * cp + 256 * np.arange(4)
            km = cm + 16 * cmp_ + 256 * np.arange(4)
            if u == 127:
                a, h = G, Gm
                ka, kh = k, km
                pairs = [(a[0], a[0], ka[0]), (a[1], a[3], ka[1]), (h[0], h[3], kh[0]), (h[1], h[2], kh[1]),
                         (a[2], a[2], ka[2])]
            else:
                pairs = [(G[d], Gm[3 - d], k[d]) for d in range(4)]
                for d in range(4):
                    assert (k[d] + km[3 - d]) % 1024 == 0, (u, k, km)
            for zk, zm, kk in pairs:
                E = 0.5 * (zk + np.conj(zm)); O = -0.5j * (zk - np.conj(zm))
                w = np.exp(-1j * np.pi * kk / M)
                X[kk] = E + w * O
                X[M - kk] = np.conj(E - w * O)
    return X


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    x = rng.normal(size=2048)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(2048) / 2048)
    X = model(x, win)
    ref = np.fft.rfft(x * win)
    assert not np.isnan(X).any()
    print("max err", np.abs(X - ref).max())
    # all 128 units cover all 256 groups exactly once
    seen = set()
    for u in range(128):
        c, cp, cm, cmp_ = unit_groups(u)
        seen.add((c, cp)); seen.add((cm, cmp_))
    print("groups covered", len(seen))
