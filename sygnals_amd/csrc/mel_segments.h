// Mel projection of one LDS power row by one wave, by SEGMENT SUMS (shared by stft_mel.hip MODE 6 / 7 and
// stft_mel_w4096.hip).  Included inside namespace syg { namespace { ... } } after common.h.
#pragma once
// The projection (sygnals_amd/_tables.py: pack_mel_segments builds the table):
// A triangular filterbank is piecewise affine in the bin index, so a run of bins inside one segment (between two band
// edges) contributes  a T0 + b T1  to the band rising there and  a' T0 + b' T1  to the band falling there, with
// T0 = sum p, T1 = sum i' p (i' = distance from the run's last bin): two sums per piece, no weight matrix, and nothing a
// second wave has to see -- the projection needs no workgroup barrier and no partial tiles.
//   * a lane sums one piece (<= 16 bins inside one 16-bin block of the skewed row: no pad word inside) per pass, two
//     passes; it reads a window of 17 row words that starts up to 4 words before the piece -- the host picks the leads so
//     that the 32 lanes of an LDS access start in 32 different banks (piece starts alone collide: 144 instead of 68 LDS
//     cycles per frame); T0 and T1 come from a running prefix (T1 += c; c += p) under the EXEC mask lead <= i < hi
//     (v_cmpx), three vector instructions per word and no constants;
//   * the pieces of a segment sit in neighbouring lanes of one DPP row: rising contributions are summed towards the
//     run's last lane, falling ones towards its first lane (segmented scans in steps 1, 2, 4, 8; the per-lane link
//     weights 0 / 1 come with the table), so band s = R(last lane of run s) + F(first lane of run s + 1) meets in
//     neighbouring lanes (wave_shl:1; the lane after lane 63 of pass 0 is lane 0 of pass 1);
//   * the lane at a run's end stores the band to the clip's mel matrix column and keeps the clip maximum.
// Every sum has a fixed order: results do not depend on scheduling.
constexpr int DPP_ROW_SHL1 = 0x101, DPP_ROW_SHL2 = 0x102, DPP_ROW_SHL4 = 0x104, DPP_ROW_SHL8 = 0x108, DPP_WAVE_SHL1 = 0x130;
constexpr int DPP_WAVE_ROL1 = 0x134;
// a lane's window: 17 consecutive row words, the piece occupies words [lead, hi) of it; the first TRI_LEAD_MAX steps
// enter by `lead <= i` from the full mask (the host keeps hi > i there), the rest leave by `i < hi`
constexpr int TRI_LEAD_MAX = 4;      // == sygnals_amd._tables.SEG_LEAD_MAX
#define SYG_TRI_HEAD(i) \
  "s_mov_b64 exec, %[sv]\n\tv_cmpx_ge_i32_e32 vcc, " #i ", %[lead]\n\tv_add_f32_e32 %[t1], %[t1], %[c]\n\tv_add_f32_e32 %[c], %[c], %[p" #i "]\n\t"
#define SYG_TRI_STEP(i) \
  "v_cmpx_lt_i32_e32 vcc, " #i ", %[hi]\n\tv_add_f32_e32 %[t1], %[t1], %[c]\n\tv_add_f32_e32 %[c], %[c], %[p" #i "]\n\t"
__device__ __forceinline__ void tri_piece_sums(const float (&pw)[17], int lead, int hi, float& c, float& t1) {
  unsigned long long sv;
  c = 0.f; t1 = 0.f;
  asm volatile(
      "s_mov_b64 %[sv], exec\n\t"
      SYG_TRI_HEAD(0) SYG_TRI_HEAD(1) SYG_TRI_HEAD(2) SYG_TRI_HEAD(3)
      "s_mov_b64 exec, %[sv]\n\t"
      SYG_TRI_STEP(4) SYG_TRI_STEP(5) SYG_TRI_STEP(6) SYG_TRI_STEP(7) SYG_TRI_STEP(8) SYG_TRI_STEP(9) SYG_TRI_STEP(10)
      SYG_TRI_STEP(11) SYG_TRI_STEP(12) SYG_TRI_STEP(13) SYG_TRI_STEP(14) SYG_TRI_STEP(15) SYG_TRI_STEP(16)
      "s_mov_b64 exec, %[sv]\n\t"
      "s_nop 1"
      : [c] "+v"(c), [t1] "+v"(t1), [sv] "=&s"(sv)
      : [lead] "v"(lead), [hi] "v"(hi), [p0] "v"(pw[0]), [p1] "v"(pw[1]), [p2] "v"(pw[2]), [p3] "v"(pw[3]), [p4] "v"(pw[4]),
        [p5] "v"(pw[5]), [p6] "v"(pw[6]), [p7] "v"(pw[7]), [p8] "v"(pw[8]), [p9] "v"(pw[9]), [p10] "v"(pw[10]),
        [p11] "v"(pw[11]), [p12] "v"(pw[12]), [p13] "v"(pw[13]), [p14] "v"(pw[14]), [p15] "v"(pw[15]), [p16] "v"(pw[16])
      : "vcc");
}
// the same for a window of 9 words (pieces of <= 8 bins: the frame-length-256 kernel cuts its rows at 8-bin blocks)
__device__ __forceinline__ void tri_piece_sums(const float (&pw)[9], int lead, int hi, float& c, float& t1) {
  unsigned long long sv;
  c = 0.f; t1 = 0.f;
  asm volatile(
      "s_mov_b64 %[sv], exec\n\t"
      SYG_TRI_HEAD(0) SYG_TRI_HEAD(1) SYG_TRI_HEAD(2) SYG_TRI_HEAD(3)
      "s_mov_b64 exec, %[sv]\n\t"
      SYG_TRI_STEP(4) SYG_TRI_STEP(5) SYG_TRI_STEP(6) SYG_TRI_STEP(7) SYG_TRI_STEP(8)
      "s_mov_b64 exec, %[sv]\n\t"
      "s_nop 1"
      : [c] "+v"(c), [t1] "+v"(t1), [sv] "=&s"(sv)
      : [lead] "v"(lead), [hi] "v"(hi), [p0] "v"(pw[0]), [p1] "v"(pw[1]), [p2] "v"(pw[2]), [p3] "v"(pw[3]), [p4] "v"(pw[4]),
        [p5] "v"(pw[5]), [p6] "v"(pw[6]), [p7] "v"(pw[7]), [p8] "v"(pw[8])
      : "vcc");
}
#undef SYG_TRI_STEP
#undef SYG_TRI_HEAD
static_assert(TRI_LEAD_MAX == 4, "tri_piece_sums unrolls four entry steps");

// segl: the piece table in LDS, [NPASS][2][64 lanes] 16-byte words (layout: pack_mel_segments); store(word, value) is
// called in the lanes that hold a band, with the table's band word (the caller decides what it carries: MODE 6 / 7 turn
// it into the band's byte offset inside the clip's mel matrix when they copy the table to LDS).
// (measured and dropped in MODE 6: the window words of the table kept in registers across the transform, 148.4 against
// 144.4 us on one box; both windows read before the first sums: no difference)
// scan8: some lane of the table has a step-8 link (wave-uniform; the host lays the runs out so that three steps suffice
// where the filterbank allows it).  NPASS: 2 or 4 passes of 64 lanes.
template <int NPASS, int W = 17, typename Store>
__device__ __forceinline__ void tri_project(const float* __restrict__ prow, const float4* __restrict__ segl, int la,
                                            bool scan8, Store&& store) {
  static_assert(NPASS == 2 || NPASS == 4, "passes come in pairs (the scans interleave four chains)");
  static_assert(W == 17 || W == 9, "window lengths with a tri_piece_sums form");
  float R[NPASS], F[NPASS];
  int band[NPASS];
#pragma unroll
  for (int pp = 0; pp < NPASS; pp += 2) {
    float4 qa[2], qc[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      qa[p] = segl[((pp + p) * 2 + 0) * 64 + la];
      qc[p] = segl[((pp + p) * 2 + 1) * 64 + la];
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const unsigned w0 = (unsigned)__float_as_int(qa[p].x);
      const float* src = reinterpret_cast<const float*>(reinterpret_cast<const char*>(prow) + (w0 & 0xFFFFu));
      float pw[W];
#pragma unroll
      for (int i = 0; i < W; ++i) pw[i] = src[i];
      float c, t1;
      tri_piece_sums(pw, (int)(w0 >> 24), (int)((w0 >> 16) & 0xFFu), c, t1);
      R[pp + p] = fmaf(qc[p].y, t1, qc[p].x * c);
      F[pp + p] = fmaf(qc[p].w, t1, qc[p].z * c);
      band[pp + p] = __float_as_int(qa[p].y);
    }
    // segmented scans: x += link * x(lane -/+ d), d = 1, 2, 4, 8 -- the link bytes become 0.0 / 1.0 and multiply the
    // neighbour inside the DPP instruction; the four chains (rising / falling of the two passes) are interleaved so that
    // no instruction reads a register written less than two instructions before (DPP hazard)
    {
      const int lr0 = __float_as_int(qa[0].z), lf0 = __float_as_int(qa[0].w);
      const int lr1 = __float_as_int(qa[1].z), lf1 = __float_as_int(qa[1].w);
      float t0, t1, t2, t3;
#define SYG_SCAN_STEP(N, SHR, SHL)                                                                             \
    "v_cvt_f32_ubyte" #N " %[t0], %[lr0]\n\tv_cvt_f32_ubyte" #N " %[t1], %[lf0]\n\t"                         \
    "v_cvt_f32_ubyte" #N " %[t2], %[lr1]\n\tv_cvt_f32_ubyte" #N " %[t3], %[lf1]\n\t"                         \
    "v_fmac_f32_dpp %[r0], %[r0], %[t0] " SHR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                  \
    "v_fmac_f32_dpp %[f0], %[f0], %[t1] " SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                  \
    "v_fmac_f32_dpp %[r1], %[r1], %[t2] " SHR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                  \
    "v_fmac_f32_dpp %[f1], %[f1], %[t3] " SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      asm volatile("s_nop 1\n\t"
                   SYG_SCAN_STEP(0, "row_shr:1", "row_shl:1") SYG_SCAN_STEP(1, "row_shr:2", "row_shl:2")
                   SYG_SCAN_STEP(2, "row_shr:4", "row_shl:4")
                   "s_nop 1"
                   : [r0] "+v"(R[pp]), [f0] "+v"(F[pp]), [r1] "+v"(R[pp + 1]), [f1] "+v"(F[pp + 1]), [t0] "=&v"(t0),
                     [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                   : [lr0] "v"(lr0), [lf0] "v"(lf0), [lr1] "v"(lr1), [lf1] "v"(lf1));
      if (scan8)
        asm volatile(SYG_SCAN_STEP(3, "row_shr:8", "row_shl:8")
                     "s_nop 1"
                     : [r0] "+v"(R[pp]), [f0] "+v"(F[pp]), [r1] "+v"(R[pp + 1]), [f1] "+v"(F[pp + 1]), [t0] "=&v"(t0),
                       [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                     : [lr0] "v"(lr0), [lf0] "v"(lf0), [lr1] "v"(lr1), [lf1] "v"(lf1));
#undef SYG_SCAN_STEP
    }
  }
  // the falling total of the next run: one lane up.  The LAST pass rotated left by one lane puts its lane 0 into lane 63
  // (that lane never stores a band: the last run is the segment above the last band); every other pass is shifted left
  // by one lane, and its lane 63 -- where the shift finds nothing (`old` operand, bound_ctrl off) -- keeps lane 0 of the
  // pass after it, taken from that pass's rotated copy.
  int nx[NPASS];
  nx[NPASS - 1] = __builtin_amdgcn_update_dpp(0, __float_as_int(F[NPASS - 1]), DPP_WAVE_ROL1, 0xF, 0xF, false);
#pragma unroll
  for (int p = NPASS - 2; p >= 0; --p) {
    const int rot = (p == NPASS - 2) ? nx[NPASS - 1]
                                     : __builtin_amdgcn_update_dpp(0, __float_as_int(F[p + 1]), DPP_WAVE_ROL1, 0xF, 0xF, false);
    nx[p] = __builtin_amdgcn_update_dpp(rot, __float_as_int(F[p]), DPP_WAVE_SHL1, 0xF, 0xF, false);
  }
#pragma unroll
  for (int p = 0; p < NPASS; ++p)
    if (band[p] >= 0) store(band[p], R[p] + __int_as_float(nx[p]));
}
