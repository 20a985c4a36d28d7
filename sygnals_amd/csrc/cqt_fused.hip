// The whole constant-Q transform of compute_cqt (sygnals/core/dsp.py:231-289 -> librosa.cqt's recursive octave scheme) in
// ONE launch for its usual shape: hop_length 512, one early decimation, up to seven octaves of <= 16 filters at frame
// length 256 that share one operand table (the octaves' bases are one matrix: the sqrt(2) of the decimator and the
// scalings of the plan cancel).  The separate kernels (cqt.hip: three chained decimation passes that write every level to
// HBM, one framed matrix product per octave that reads it back) move 2.6 GB for a stream of 0.69 GB; here the stream is
// read once and only the transform leaves the chip.
//
// A workgroup owns a segment of the signal and streams through it in steps of 8192 samples; eight of its twelve waves
// decimate, four multiply.  Per step:
//   1  the step's samples go to LDS (requested one step ahead)
//   2  level l = 1 .. n_oct is decimated from level l - 1 inside LDS (x_l[n] = s sum_j h[j] x_(l-1)[2n + 20 - j], the
//      41-tap half-band of cqt.hip: ten symmetric odd-offset pairs and the centre; samples outside a level's length are
//      zero, as the level-by-level zero padding of the separate kernels makes them), each level a fixed 32 samples
//      behind the one above so that every level gains exactly 8192 / 2^l samples per step.  A level is kept twice: as
//      float32 split into even and odd samples (the next level's input: the stride-2 reads become contiguous) with the
//      history the filter needs in front, and as three bfloat16 planes hi + mid + lo (the matrix product's B operand,
//      split once per sample) over the window its next sixteen frames cover.
//   3  the sixteen frames of every octave that have become complete are multiplied -- frame t of every octave is centred
//      on sample 512 t of the input, an octave's block lags the step by what its frame length and the decimator delays
//      need (512 ... 20480 input samples).  A multiplying wave holds one 16-row tile of the 32 x 256 operand table in
//      registers for the whole launch (three terms: 96 VGPRs) and multiplies that tile of at most one octave per phase
//      (v_mfma_f32_16x16x32_bf16, the six term pairs of weight >= 2^-24 as in cqt_bf16x3_kernel: 48 matrix instructions,
//      two accumulation chains, operand reads one k-step ahead); the block leaves as the separate kernels' rows.
//   4  the windows slide: the tail a level's next block still needs moves to the front (each array in a phase in which
//      nothing else touches it).
// Four phases (barriers) per step.  The decimating waves run two level chains side by side -- 1, 2, 3 on the step's
// samples, 4 ... 7 on level 3 as the previous step left it (they and their octaves are one step behind); a level's octave
// is multiplied in the phase after the one that completed its window, by the multiplying wave of each SIMD beside that
// SIMD's two decimating waves (a wave alone issues one vector instruction per four cycles: two keep the vector pipe busy,
// the third brings the matrix instructions; a matrix instruction holds the vector issue for half its length, so the two
// kinds of work overlap by about half).
// A segment is entered 24576 samples early (the deepest octave's window and the decimators' delays: the first levels
// computed from an empty history are wrong only in a region no stored frame reads) and left 4 steps late; segments of
// ~83 steps, one per CU.  LDS 133 KiB: float32 levels 0 .. 6 (66.7 KiB), planes of levels 1 .. 7 (66.4 KiB; 16-byte
// chunk c of a window at slot c + (c >> log2(hop / 8)): the sixteen frames of a ds_read_b128 group fall on sixteen bank
// groups, checked by enumeration).  150 VGPRs: three waves per SIMD.
#include <string.h>
#include "common.h"

namespace syg {
namespace {

typedef __bf16 cf_v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 cf_v4bf __attribute__((ext_vector_type(4)));
typedef __bf16 cf_v2bf __attribute__((ext_vector_type(2)));
typedef float cf_v4f __attribute__((ext_vector_type(4)));

constexpr int CF_NT = 512;               // lanes that decimate (waves 0 .. 7)
constexpr int CF_BLOCK = 768;            // lanes per workgroup: + four waves that multiply (one per SIMD)
constexpr int CF_STEP = 8192;            // input samples per step (sixteen frames of every octave)
constexpr int CF_MAXOCT = 7;
constexpr int CF_LEAD = 3;               // steps a segment is entered early / left late

__host__ __device__ constexpr int cf_new(int lv) { return CF_STEP >> lv; }                     // new samples of level lv per step
__host__ __device__ constexpr int cf_hist(int lv) { return lv == 0 ? 84 : 52; }                // float32 history in front
// Level 0's even / odd arrays are read by level 1 with a lane stride of 32 bytes (eight outputs per lane): 16-byte chunk c
// sits at c + c / 16, so that the sixteen lanes of a ds_read_b128 group fall on sixteen bank groups instead of eight
// (float index f -> cf_sk0(f)); the other levels' readers move 16 or 8 bytes per lane and need no skew.
__host__ __device__ constexpr int cf_sk0(int f) { return f + ((f >> 6) << 2); }
__host__ __device__ constexpr int cf_np_plain(int lv) { return ((cf_hist(lv) + cf_new(lv)) / 2 + 3) & ~3; }
__host__ __device__ constexpr int cf_np(int lv) {          // pairs (padded; level 0 with its skew)
  return lv == 0 ? ((cf_sk0(cf_np_plain(0)) + 4 + 3) & ~3) : cf_np_plain(lv);
}
__host__ __device__ constexpr int cf_hop(int lv) { return 512 >> lv; }                         // frame hop of octave lv - 1
// input samples an octave's frame block lags the step: a multiple of 512 >= (127 + 32) 2^lv
__host__ __device__ constexpr int cf_lag(int lv) {
  return lv == 1 ? 512 : lv == 2 ? 1024 : lv == 3 ? 1536 : lv == 4 ? 2560 : lv == 5 ? 5120 : lv == 6 ? 10240 : 20480;
}
__host__ __device__ constexpr int cf_kept(int lv) { return (cf_lag(lv) >> lv) + 96; }          // window samples kept per step
__host__ __device__ constexpr int cf_win(int lv) { return cf_kept(lv) + cf_new(lv); }
__host__ __device__ constexpr int cf_sh(int lv) { return lv <= 5 ? 6 - lv : 31; }              // log2(hop / 8), none below hop 16
__host__ __device__ constexpr int cf_slots(int lv) {
  return cf_sh(lv) < 31 ? cf_win(lv) / 8 + ((cf_win(lv) / 8) >> cf_sh(lv)) + 1 : cf_win(lv) / 8;
}
// LDS map (floats): E / O of levels 0 .. 6, then the planes of levels 1 .. 7 (16-byte slots)
__host__ __device__ constexpr int cf_off_f32(int lv) { return lv == 0 ? 0 : cf_off_f32(lv - 1) + 2 * cf_np(lv - 1); }
__host__ __device__ constexpr int cf_off_pl(int lv) {          // in floats
  return lv == 1 ? cf_off_f32(7) : cf_off_pl(lv - 1) + 3 * cf_slots(lv - 1) * 4;
}
// (compile-time tables: called with a template argument inside device code these recursive functions would otherwise be
// emitted as real -- recursive -- device functions)
template <int LV> struct CfC {
  static constexpr int OFF_F32 = cf_off_f32(LV), OFF_PL = cf_off_pl(LV < 1 ? 1 : LV), NP = cf_np(LV), SLOTS = cf_slots(LV < 1 ? 1 : LV);
};
constexpr int CF_LDS_FLOATS = cf_off_pl(8);
static_assert(cf_kept(1) == 352 && cf_kept(3) == 288 && cf_kept(7) == 256, "window arithmetic");
// what the schedule relies on, level by level
constexpr bool cf_level_ok(int lv) {
  return cf_lag(lv) % 512 == 0 && cf_lag(lv) >= (127 + 32) * (1 << lv)          // a block's last frame ends inside the samples a step has
         && cf_kept(lv) % 8 == 0 && cf_new(lv) % 8 == 0                          // whole 16-byte chunks slide and arrive
         && 15 * cf_hop(lv) + 256 <= cf_win(lv)                                  // sixteen frames inside the window
         && (cf_sh(lv) == 31 || (cf_new(lv) / 8) % (1 << cf_sh(lv)) == 0)        // the slide moves skewed slots by a constant
         && (cf_sh(lv) == 31 ? cf_kept(lv) / 8 : cf_kept(lv) / 8 + ((cf_kept(lv) / 8 - 1) >> cf_sh(lv))) <= 64   // one wave slides an array
         && cf_hist(lv < 7 ? lv : 0) / 2 <= 64 && 2 * 10 + 32 <= cf_hist(lv < 7 ? lv : 0);   // history: ten pairs behind + the 32-sample lag
}
static_assert(cf_level_ok(1) && cf_level_ok(2) && cf_level_ok(3) && cf_level_ok(4) && cf_level_ok(5) && cf_level_ok(6) && cf_level_ok(7),
              "level schedule");
static_assert(CF_STEP == 16 * 512 && CF_NT * 16 == CF_STEP, "sixteen frames per block; sixteen samples per decimating lane");
static_assert(CF_LDS_FLOATS * 4 <= 160 * 1024, "LDS");

#ifdef SYG_CQF_STAMP
__device__ unsigned long long cqf_stamp[8 * 256];      // development build: per-workgroup phase sums (100 MHz ticks)
#define CQF_T(k) do { if (tid == 0) { const unsigned long long n_ = wall_clock64(); tsum[k] += n_ - tlast; tlast = n_; } } while (0)
#else
#define CQF_T(k) do { } while (0)
#endif

struct CfParams {
  int64_t L, ldy, T, out_bstride, seg;
  int n_oct, n_filt, n_seg;
  float scale;
  int row0[CF_MAXOCT];
};

__device__ __forceinline__ int cf_slot(int c, int sh) { return sh < 31 ? c + (c >> sh) : c; }

// one sample into (hi, mid, lo)
__device__ __forceinline__ void cf_split1(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r = x - (float)h;
  m = (__bf16)r;
  l = (__bf16)(r - (float)m);
}

// Level LV from level LV - 1: lane `tid` produces OPL consecutive samples of the step's cf_new(LV).
template <int LV>
__device__ __forceinline__ void cf_decimate(float* lds, int tid, const float (&hp)[10], float hc, float scale, int64_t n_first,
                                            int64_t len) {
  constexpr int NEW = cf_new(LV);
  constexpr int OPL = NEW >= 8 * CF_NT ? 8 : NEW >= 4 * CF_NT ? 4 : NEW >= 2 * CF_NT ? 2 : 1;
  if (tid * OPL >= NEW) return;
  const float* E = lds + CfC<LV - 1>::OFF_F32;
  const float* O = E + CfC<LV - 1>::NP;
  const int i0 = tid * OPL;                      // first output of this lane; its centre pair is m = 10 + i0
  float ow[OPL + 20], ev[OPL];
  if (OPL == 8) {
    // (level 1 reads level 0: skewed chunks, cf_sk0)
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int f = i0 + 4 * q;
      const float4 v = *reinterpret_cast<const float4*>(O + (LV == 1 ? f + ((f >> 6) << 2) : f));
      ow[4 * q] = v.x; ow[4 * q + 1] = v.y; ow[4 * q + 2] = v.z; ow[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = 10 + i0 + 2 * q;
      const float2 v = *reinterpret_cast<const float2*>(E + (LV == 1 ? f + ((f >> 6) << 2) : f));
      ev[2 * q] = v.x; ev[2 * q + 1] = v.y;
    }
  } else if (OPL == 4) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(O + i0 + 4 * q);
      ow[4 * q] = v.x; ow[4 * q + 1] = v.y; ow[4 * q + 2] = v.z; ow[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float2 v = *reinterpret_cast<const float2*>(E + 10 + i0 + 2 * q);
      ev[2 * q] = v.x; ev[2 * q + 1] = v.y;
    }
  } else if (OPL == 2) {
#pragma unroll
    for (int q = 0; q < 11; ++q) {
      const float2 v = *reinterpret_cast<const float2*>(O + i0 + 2 * q);
      ow[2 * q] = v.x; ow[2 * q + 1] = v.y;
    }
    const float2 v = *reinterpret_cast<const float2*>(E + 10 + i0);
    ev[0] = v.x; ev[1] = v.y;
  } else {
#pragma unroll
    for (int q = 0; q < 20; ++q) ow[q] = O[i0 + q];
    ev[0] = E[10 + i0];
  }
  float yv[OPL];
#pragma unroll
  for (int q = 0; q < OPL; ++q) {
    float acc = hc * ev[q];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc = fmaf(hp[i], ow[q + 10 + i] + ow[q + 9 - i], acc);
    yv[q] = acc * scale;
  }
  // samples outside the level's length are zero (a lane whose run lies inside -- all but the lanes at a signal end -- skips this)
  if (n_first + i0 < 0 || n_first + i0 + OPL > len) {
#pragma unroll
    for (int q = 0; q < OPL; ++q) {
      const int64_t n = n_first + i0 + q;
      yv[q] = (n >= 0 && n < len) ? yv[q] : 0.f;
    }
  }
  // float32 copy (the next level's input), even / odd samples apart
  if (LV < CF_MAXOCT) {
    float* E2 = lds + CfC<LV>::OFF_F32;
    float* O2 = E2 + CfC<LV>::NP;
    constexpr int H2 = cf_hist(LV) / 2;
    if (OPL >= 2) {
#pragma unroll
      for (int q = 0; q < OPL / 2; q += (OPL >= 4 ? 2 : 1)) {
        if (OPL >= 4) {
          *reinterpret_cast<float2*>(E2 + H2 + i0 / 2 + q) = make_float2(yv[2 * q], yv[2 * q + 2]);
          *reinterpret_cast<float2*>(O2 + H2 + i0 / 2 + q) = make_float2(yv[2 * q + 1], yv[2 * q + 3]);
        } else {
          E2[H2 + i0 / 2] = yv[0];
          O2[H2 + i0 / 2] = yv[1];
        }
      }
    } else {
      ((i0 & 1) ? O2 : E2)[H2 + (i0 >> 1)] = yv[0];
    }
  }
  // the three bfloat16 planes of the level's frame window
  {
    constexpr int SH = cf_sh(LV), NS = CfC<LV>::SLOTS;
    char* pl = reinterpret_cast<char*>(lds + CfC<LV>::OFF_PL);
    const int p0 = cf_kept(LV) + i0;
    const int off = cf_slot(p0 >> 3, SH) * 16 + 2 * (p0 & 7);
    __bf16 h[OPL], m[OPL], l[OPL];
#pragma unroll
    for (int q = 0; q < OPL; ++q) cf_split1(yv[q], h[q], m[q], l[q]);
    if (OPL == 8) {
      cf_v8bf vh, vm, vl;
#pragma unroll
      for (int q = 0; q < 8; ++q) { vh[q] = h[q]; vm[q] = m[q]; vl[q] = l[q]; }
      *reinterpret_cast<cf_v8bf*>(pl + off) = vh;
      *reinterpret_cast<cf_v8bf*>(pl + NS * 16 + off) = vm;
      *reinterpret_cast<cf_v8bf*>(pl + 2 * NS * 16 + off) = vl;
    } else if (OPL == 4) {
      cf_v4bf vh, vm, vl;
#pragma unroll
      for (int q = 0; q < 4; ++q) { vh[q] = h[q]; vm[q] = m[q]; vl[q] = l[q]; }
      *reinterpret_cast<cf_v4bf*>(pl + off) = vh;
      *reinterpret_cast<cf_v4bf*>(pl + NS * 16 + off) = vm;
      *reinterpret_cast<cf_v4bf*>(pl + 2 * NS * 16 + off) = vl;
    } else if (OPL == 2) {
      cf_v2bf vh, vm, vl;
      vh[0] = h[0]; vh[1] = h[1]; vm[0] = m[0]; vm[1] = m[1]; vl[0] = l[0]; vl[1] = l[1];
      *reinterpret_cast<cf_v2bf*>(pl + off) = vh;
      *reinterpret_cast<cf_v2bf*>(pl + NS * 16 + off) = vm;
      *reinterpret_cast<cf_v2bf*>(pl + 2 * NS * 16 + off) = vl;
    } else {
      *reinterpret_cast<__bf16*>(pl + off) = h[0];
      *reinterpret_cast<__bf16*>(pl + NS * 16 + off) = m[0];
      *reinterpret_cast<__bf16*>(pl + 2 * NS * 16 + off) = l[0];
    }
  }
}

// Row tile r of the sixteen complete frames of octave LV - 1 (one wave): OUT[16 rows, 16 frames] = A[16, 256] x FRAMES[256, 16];
// the even and the odd k-steps run as two independent accumulation chains.
template <int LV>
__device__ __forceinline__ cf_v4f cf_product(const float* lds, int lane, const cf_v8bf (&areg)[3][8]) {
  constexpr int SH = cf_sh(LV), NS = CfC<LV>::SLOTS, HOP = cf_hop(LV);
  const char* pl = reinterpret_cast<const char*>(lds + CfC<LV>::OFF_PL);
  int lq = lane;
  asm volatile("" : "+v"(lq));                   // (the operand addresses are formed per block: hoisted out of the step loop
                                                 //  for seven levels they do not fit the registers left beside the table)
  const int n = lq & 15, kk = lq >> 4;
  cf_v4f acc[2];
  acc[0] = cf_v4f{0.f, 0.f, 0.f, 0.f};
  acc[1] = cf_v4f{0.f, 0.f, 0.f, 0.f};
  auto operand = [&](int s, cf_v8bf& bh, cf_v8bf& bm, cf_v8bf& bl) {
    if (HOP >= 8) {
      const int c = n * (HOP / 8) + 4 * s + kk;
      const int off = cf_slot(c, SH) * 16;
      bh = *reinterpret_cast<const cf_v8bf*>(pl + off);
      bl = *reinterpret_cast<const cf_v8bf*>(pl + 2 * NS * 16 + off);
      bm = *reinterpret_cast<const cf_v8bf*>(pl + NS * 16 + off);
    } else {                                     // hop 4: frames start on 8-byte boundaries
      const int off = 2 * (HOP * n + 32 * s + 8 * kk);
      cf_v4bf x0 = *reinterpret_cast<const cf_v4bf*>(pl + off), x1 = *reinterpret_cast<const cf_v4bf*>(pl + off + 8);
      cf_v4bf m0 = *reinterpret_cast<const cf_v4bf*>(pl + NS * 16 + off), m1 = *reinterpret_cast<const cf_v4bf*>(pl + NS * 16 + off + 8);
      cf_v4bf l0 = *reinterpret_cast<const cf_v4bf*>(pl + 2 * NS * 16 + off), l1 = *reinterpret_cast<const cf_v4bf*>(pl + 2 * NS * 16 + off + 8);
#pragma unroll
      for (int q = 0; q < 4; ++q) { bh[q] = x0[q]; bh[4 + q] = x1[q]; bm[q] = m0[q]; bm[4 + q] = m1[q]; bl[q] = l0[q]; bl[4 + q] = l1[q]; }
    }
  };
  // One k-step at a time: its three operand reads are issued a k-step ahead (two register sets of 12), its six matrix
  // instructions alternate between the two accumulators (a wave's consecutive instructions are independent; per
  // accumulator the small terms still come first within a step).  The multiplying wave is alone with its job on its SIMD's
  // matrix pipe: what it must not do is wait for every read in front of its use.
  cf_v8bf bh[2], bm[2], bl[2];
  operand(0, bh[0], bm[0], bl[0]);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int q = s & 1;
    if (s + 1 < 8) operand(s + 1, bh[q ^ 1], bm[q ^ 1], bl[q ^ 1]);
    __builtin_amdgcn_sched_barrier(0);
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(areg[2][s], bh[q], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(areg[0][s], bl[q], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(areg[1][s], bm[q], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(areg[1][s], bh[q], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(areg[0][s], bm[q], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(areg[0][s], bh[q], acc[1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  return acc[0] + acc[1];
}

// slide one array of 16-byte slots: slots [shift, shift + n) -> [0, n), n <= 64 (one wave)
__device__ __forceinline__ void cf_slide16(float* base, int lane, int n, int shift) {
  uint4* p = reinterpret_cast<uint4*>(base);
  uint4 v = make_uint4(0, 0, 0, 0);
  if (lane < n) v = p[lane + shift];
  wave_lds_sync();
  if (lane < n) p[lane] = v;
}

__global__ __launch_bounds__(CF_BLOCK) void cqt_fused_kernel(const float* __restrict__ y, const float* __restrict__ taps,
                                                          const uint4* __restrict__ gsplit, float2* __restrict__ out, CfParams P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t b = blockIdx.x / P.n_seg;
  const int64_t sg = blockIdx.x - b * P.n_seg;
  const int64_t s0 = sg * P.seg;
  const float* yb = y + b * P.ldy;
  // frames this workgroup stores
  const int64_t tlo = s0 / 512;
  int64_t thi = (s0 + P.seg) / 512;
  if (thi > P.T) thi = P.T;
  if (tlo >= thi) return;
  // level lengths
  int64_t len[CF_MAXOCT + 1];
  len[0] = P.L;
#pragma unroll
  for (int l = 1; l <= CF_MAXOCT; ++l) len[l] = (len[l - 1] + 1) >> 1;
  // the decimator: centre and the ten odd-offset taps on one side (the filter is symmetric)
  const float hc = taps[20];
  float hp[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) hp[i] = taps[21 + 2 * i];
  // Roles: waves 0 .. 7 decimate (and slide the windows), waves 8 .. 11 multiply -- one per SIMD, so that the matrix pipe
  // works beside the two decimating waves of its SIMD (a wave alone issues one vector instruction per four cycles: two
  // keep the vector pipe busy, the third brings the matrix instructions).  A multiplying wave holds one 16-row tile of the
  // operand table in registers for the whole launch (tile mw & 1, three terms: 96 VGPRs) and takes the octaves of group
  // mw >> 1, at most one per phase.
  const bool dec = w < 8;
  const int mw = w - 8, rt = mw & 1, grp = mw >> 1;
  for (int i = tid; i < CF_LDS_FLOATS; i += CF_BLOCK) lds[i] = 0.f;

  // raw samples of a step: lane tid owns samples 16 tid .. 16 tid + 15.  The request runs a whole step ahead, so it must not
  // end in a join of two code paths (the compiler would wait for the data there): every lane always issues the four 16-byte
  // loads (dword-aligned: any row stride) -- from its own run where that lies inside the signal, from the tap table
  // otherwise (41 floats that are always there) -- and a lane whose run crosses an end replaces them element by element
  // when the step begins (raw_fix).
  typedef float cf_f4u __attribute__((ext_vector_type(4), aligned(4)));
  cf_f4u raw[4];
  auto run_inside = [&](int64_t X) {
    const int64_t g = X - CF_STEP + 16 * tid;
    return g >= 0 && g + 16 <= P.L;
  };
  auto fetch = [&](int64_t X) {                  // the step ending at input sample X
    const int64_t g = X - CF_STEP + 16 * tid;
    const float* src = run_inside(X) ? yb + g : taps;
#pragma unroll
    for (int q = 0; q < 4; ++q) raw[q] = *reinterpret_cast<const cf_f4u*>(src + 4 * q);
  };
  auto raw_fix = [&](int64_t X) {
    if (run_inside(X)) return;
    const int64_t g = X - CF_STEP + 16 * tid;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int64_t i = g + 4 * q + j; raw[q][j] = (i >= 0 && i < P.L) ? yb[i] : 0.f; }
    }
  };
  // slides: the part of an array the next step still needs moves to its front (one wave per array, no workgroup barrier:
  // an array is slid in a phase in which nothing else touches it)
  auto slide_plane = [&](int lv, int p) {
    int off = 0, ns = 0, kept = 0, nw = 0, sh = 31;
    switch (lv) {
#define CF_CASE(LV) case LV: { constexpr int a_ = CfC<LV>::OFF_PL, b_ = CfC<LV>::SLOTS, c_ = cf_kept(LV), d_ = cf_new(LV), e_ = cf_sh(LV); off = a_; ns = b_; kept = c_; nw = d_; sh = e_; } break;
      CF_CASE(1) CF_CASE(2) CF_CASE(3) CF_CASE(4) CF_CASE(5) CF_CASE(6) default: CF_CASE(7)
#undef CF_CASE
    }
    const int nk = cf_slot(kept / 8 - 1, sh) + 1;              // kept slots (<= 64)
    const int shift = cf_slot(nw / 8, sh);                      // (new / 8 is a multiple of 2^sh: the skew shifts by a constant)
    int lq = lane;
    asm volatile("" : "+v"(lq));                 // (addresses formed per call: see store_block)
    cf_slide16(lds + off + p * ns * 4, lq, nk, shift);
  };
  auto slide_f32 = [&](int lv, int odd) {
    int off = 0, np = 0, h2 = 0, nw2 = 0;
    switch (lv) {
#define CF_CASE(LV) case LV: { constexpr int a_ = CfC<LV>::OFF_F32, b_ = CfC<LV>::NP, c_ = cf_hist(LV) / 2, d_ = cf_new(LV) / 2; off = a_; np = b_; h2 = c_; nw2 = d_; } break;
      CF_CASE(0) CF_CASE(1) CF_CASE(2) CF_CASE(3) CF_CASE(4) CF_CASE(5) default: CF_CASE(6)
#undef CF_CASE
    }
    float* a = lds + off + odd * np;
    int lq = lane;
    asm volatile("" : "+v"(lq));
    float v = 0.f;
    const int src = lq + nw2, sks = lv == 0 ? src + ((src >> 6) << 2) : src, skd = lv == 0 ? lq + ((lq >> 6) << 2) : lq;   // cf_sk0
    if (lq < h2) v = a[sks];
    wave_lds_sync();
    if (lq < h2) a[skd] = v;
  };
  auto store_block = [&](int o, cf_v4f acc, int64_t t0) {       // row tile rt of octave o
    int lq = lane;
    asm volatile("" : "+v"(lq));                 // (the rows' addresses are formed here, per block: hoisted out of the step
                                                 //  loop, seven 64-bit lane constants went to scratch and came back behind a
                                                 //  wait for every outstanding load)
    const int64_t t = t0 + (lq & 15);
    if (t >= tlo && t < thi) {
      const int f = 2 * (lq >> 4) + 8 * rt;
      float2* op = out + b * P.out_bstride + (int64_t)(P.row0[o] + f) * P.T + t;
      if (f < P.n_filt) op[0] = make_float2(acc[0], acc[1]);
      if (f + 1 < P.n_filt) op[P.T] = make_float2(acc[2], acc[3]);
    }
  };
  const int64_t kfirst = s0 == 0 ? 0 : -CF_LEAD;
  // the last step: the deepest octave's block (one step behind the others) reaches thi; one more, because the products of
  // levels 3 and 7 run in the first phase of the NEXT step
  int64_t klast = (thi * 512 - s0 + cf_lag(CF_MAXOCT) + CF_STEP - 1) / CF_STEP + 1;
  if (klast < kfirst) klast = kfirst;
  if (dec) fetch(s0 + CF_STEP * (kfirst + 1));
  __syncthreads();
#ifdef SYG_CQF_STAMP
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
#endif
  // (a block none of whose sixteen frames this workgroup stores -- the lead-in and the way out -- is not multiplied)
  auto stored = [&](int o, int64_t t0) { return o < P.n_oct && t0 + 16 > tlo && t0 < thi; };
  // Four phases (barriers) per step.  The decimating waves run two level chains side by side -- 1 -> 2 -> 3 on this step's
  // samples, 4 -> 5 -> 6 -> 7 on level 3 as the PREVIOUS step left it (levels 4 .. 7 and their octaves are one step
  // behind: xd) -- and slide an array in a phase behind its last reader and ahead of its next writer; the multiplying waves
  // take a level's octave in the phase after the one that completed its window.  Two loops, one per role (in one loop the
  // operand table of the one role and the sample registers of the other are live everywhere: 91 spills at 168 registers);
  // both pass the same four barriers per step.
  if (dec) {
    for (int64_t k = kfirst; k <= klast; ++k) {
      const int64_t X = s0 + CF_STEP * (k + 1);
      const int64_t xb = X - CF_STEP, xd = xb - CF_STEP;
      // phase 0: samples + level 4; the planes of levels 2 and 6 (multiplied in the previous phase) and their float32 copies
      // (read by levels 3 and 7 there) slide
      raw_fix(X);
      {
        float* E = lds + CfC<0>::OFF_F32;
        float* O = E + CfC<0>::NP;
        const int m = cf_hist(0) / 2 + 8 * tid;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f = m + 2 * q, fs = f + ((f >> 6) << 2);           // cf_sk0
          *reinterpret_cast<float2*>(E + fs) = make_float2(raw[q][0], raw[q][2]);
          *reinterpret_cast<float2*>(O + fs) = make_float2(raw[q][1], raw[q][3]);
        }
      }
      fetch(X + CF_STEP);
      if (P.n_oct >= 4) cf_decimate<4>(lds, tid, hp, hc, P.scale, (xd >> 4) - 32, len[4]);
      for (int j = w; j < 10; j += 8) {
        if (j < 6) slide_plane(j < 3 ? 2 : 6, j % 3); else slide_f32(j < 8 ? 2 : 6, j & 1);
      }
      __syncthreads();
      CQF_T(0);
      // phase 1: levels 1 and 5; planes of levels 3 and 7, float32 level 3
      cf_decimate<1>(lds, tid, hp, hc, P.scale, (xb >> 1) - 32, len[1]);
      if (P.n_oct >= 5) cf_decimate<5>(lds, tid, hp, hc, P.scale, (xd >> 5) - 32, len[5]);
      if (w < 6) slide_plane(w < 3 ? 3 : 7, w % 3); else slide_f32(3, w & 1);
      __syncthreads();
      CQF_T(1);
      // phase 2: levels 2 and 6; planes of level 4, float32 levels 0 and 4
      if (P.n_oct >= 2) cf_decimate<2>(lds, tid, hp, hc, P.scale, (xb >> 2) - 32, len[2]);
      if (P.n_oct >= 6) cf_decimate<6>(lds, tid, hp, hc, P.scale, (xd >> 6) - 32, len[6]);
      if (w < 3) slide_plane(4, w); else if (w < 7) slide_f32(w < 5 ? 0 : 4, (w - 3) & 1);
      __syncthreads();
      CQF_T(2);
      // phase 3: levels 3 and 7; planes and float32 copies of levels 1 and 5
      if (P.n_oct >= 3) cf_decimate<3>(lds, tid, hp, hc, P.scale, (xb >> 3) - 32, len[3]);
      if (P.n_oct >= 7) cf_decimate<7>(lds, tid, hp, hc, P.scale, (xd >> 7) - 32, len[7]);
      for (int j = w; j < 10; j += 8) {
        if (j < 6) slide_plane(j < 3 ? 1 : 5, j % 3); else slide_f32(j < 8 ? 1 : 5, j & 1);
      }
      __syncthreads();
      CQF_T(3);
    }
  } else {
    // this wave's half of the operand table: row tile mw & 1, all k-steps, three terms (96 VGPRs for the whole launch)
    cf_v8bf areg[3][8];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int ss = 0; ss < 8; ++ss) {
        const uint4 v = gsplit[((p * 2 + rt) * 8 + ss) * 64 + lane];
        areg[p][ss] = __builtin_bit_cast(cf_v8bf, v);
      }
    for (int64_t k = kfirst; k <= klast; ++k) {
      const int64_t X = s0 + CF_STEP * (k + 1);
      const int64_t xb = X - CF_STEP, xd = xb - CF_STEP;
      // phase 0: octaves 2 and 6 of the PREVIOUS step (levels 3 and 7 were completed in its last phase)
      {
        const int64_t t0 = grp == 0 ? (xb - CF_STEP - cf_lag(3)) / 512 : (xd - CF_STEP - cf_lag(7)) / 512;
        if (grp == 0) { if (stored(2, t0)) store_block(2, cf_product<3>(lds, lane, areg), t0); }
        else { if (stored(6, t0)) store_block(6, cf_product<7>(lds, lane, areg), t0); }
      }
      __syncthreads();
      // phase 1: octave 3 (level 4 of phase 0)
      if (grp == 0) {
        const int64_t t0 = (xd - cf_lag(4)) / 512;
        if (stored(3, t0)) store_block(3, cf_product<4>(lds, lane, areg), t0);
      }
      __syncthreads();
      // phase 2: octaves 0 and 4 (levels 1 and 5 of phase 1)
      {
        const int64_t t0 = grp == 0 ? (xb - cf_lag(1)) / 512 : (xd - cf_lag(5)) / 512;
        if (grp == 0) { if (stored(0, t0)) store_block(0, cf_product<1>(lds, lane, areg), t0); }
        else { if (stored(4, t0)) store_block(4, cf_product<5>(lds, lane, areg), t0); }
      }
      __syncthreads();
      // phase 3: octaves 1 and 5 (levels 2 and 6 of phase 2)
      {
        const int64_t t0 = grp == 0 ? (xb - cf_lag(2)) / 512 : (xd - cf_lag(6)) / 512;
        if (grp == 0) { if (stored(1, t0)) store_block(1, cf_product<2>(lds, lane, areg), t0); }
        else { if (stored(5, t0)) store_block(5, cf_product<6>(lds, lane, areg), t0); }
      }
      __syncthreads();
    }
  }
#ifdef SYG_CQF_STAMP
  if (tid == 0 && blockIdx.x < 256)
    for (int i = 0; i < 8; ++i) cqf_stamp[blockIdx.x * 8 + i] = tsum[i];
#endif
}

}  // namespace
}  // namespace syg

using namespace syg;

// y [B, L] -> out [B, n_bins, T] complex (interleaved float pairs; octave o fills rows row0[o] .. row0[o] + n_filt - 1),
// the transform cqt.hip's kernels compute level by level -- one early decimation, then octave o = 0 .. n_oct - 1 at frame
// length 256 and hop 256 >> o on the signal decimated o + 1 times, every octave with the operand table `gsplit`
// (sygnals_amd.ops.cqt_pack_bf16x3: [3][2][8][64] 16-byte entries) -- in one launch.  taps: the 41-tap half-band
// decimator (zero at the even offsets from its centre, symmetric), scale: sqrt(2).  T: at most the smallest centred frame count over the octaves.
extern "C" int syg_cqt_fused_f32(const float* y, int64_t B, int64_t L, int64_t ldy, const float* taps, int ntaps, float scale,
                                 const void* gsplit, int n_filt, int n_oct, const int32_t* row0_host, int64_t T, float* out,
                                 int64_t out_bstride, void* stream) {
  SYG_REQUIRE(y && taps && gsplit && row0_host && out, "cqt_fused: null pointer argument");
  SYG_REQUIRE(ntaps == 41, "cqt_fused: the decimator must have 41 taps (got %d)", ntaps);
  SYG_REQUIRE(B >= 1 && L >= 1 && ldy >= L, "cqt_fused: bad B/L/ldy");
  SYG_REQUIRE(n_oct >= 1 && n_oct <= CF_MAXOCT && n_filt >= 1 && n_filt <= 16, "cqt_fused: n_oct must be in [1, 7], n_filt in [1, 16]");
  {
    // frames: the smallest centred frame count over the octaves (a level length that rounds up can add one to 1 + L / 512)
    int64_t ll = L, tmax = -1;
    for (int o = 0; o < n_oct; ++o) {
      ll = (ll + 1) >> 1;
      const int64_t to = 1 + ll / (256 >> o);
      tmax = (tmax < 0 || to < tmax) ? to : tmax;
    }
    SYG_REQUIRE(T >= 1 && T <= tmax, "cqt_fused: T=%lld exceeds the centred frame count %lld", (long long)T, (long long)tmax);
  }
  SYG_REQUIRE(((uintptr_t)gsplit) % 16 == 0, "cqt_fused: operand table must be 16-byte aligned");
  CfParams P;
  memset(&P, 0, sizeof(P));
  for (int o = 0; o < n_oct; ++o) {
    P.row0[o] = row0_host[o];
    SYG_REQUIRE(P.row0[o] >= 0 && out_bstride >= (int64_t)(P.row0[o] + n_filt) * T, "cqt_fused: output rows out of range");
  }
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  // segments: whole steps, at least eight (the lead-in costs six), about one per CU
  const int64_t steps = (T * 512 + CF_STEP - 1) / CF_STEP;
  int64_t per = (steps * B + cus - 1) / cus;
  if (per < 8) per = 8;
  P.seg = per * CF_STEP;
  P.n_seg = (int)((T * 512 + P.seg - 1) / P.seg);
  SYG_REQUIRE((int64_t)P.n_seg * B < ((int64_t)1 << 31), "cqt_fused: too many segments");
  P.L = L; P.ldy = ldy; P.T = T; P.out_bstride = out_bstride; P.n_oct = n_oct; P.n_filt = n_filt; P.scale = scale;
  const size_t lds = (size_t)CF_LDS_FLOATS * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)cqt_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    set_error("cqt_fused: cannot reserve %zu B LDS: %s", lds, hipGetErrorString(e));
    return SYG_E_LAUNCH;
  }
  hipLaunchKernelGGL(cqt_fused_kernel, dim3((unsigned)(P.n_seg * B)), dim3(CF_BLOCK), lds, (hipStream_t)stream, y, taps,
                     (const uint4*)gsplit, (float2*)out, P);
  SYG_CHECK_LAUNCH("cqt_fused");
  return SYG_OK;
}

#ifdef SYG_CQF_STAMP
extern "C" int syg_debug_cqf_stamps(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(syg::cqf_stamp), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
