// Kernel template of the fp32 MFMA implicit-GEMM convolution; see conv_f32.hip for the design notes.
// Included by conv_f32_{fwd,dgrad,wgrad}.hip, each of which instantiates one MODE (parallel compilation).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace acgconv {


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };

// f(0), f(1), ... f(N-1) while f returns true; false as soon as one call does
template <int I, int N, class F>
__device__ __forceinline__ bool run_unrolled(F&& f) {
  if constexpr (I < N) {
    if (!f(std::integral_constant<int, I>{})) return false;
    return run_unrolled<I + 1, N>(f);
  } else {
    return true;
  }
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
constexpr int BK = 32;  // k per K-step = 8 quads
constexpr int kMaxTaps = 64;

struct ConvArgs {
  const float* gsrc;   // gathered tensor: x (FWD, WGRAD) or dy (DGRAD)
  const float* dense;  // w (FWD, DGRAD) or dy (WGRAD)
  float* out;          // final tensor (splits == 1) or `splits` slabs of out_numel floats
  long long out_numel;
  unsigned g_bytes, d_bytes;   // sizes of gsrc / dense for the buffer descriptors (hardware range check)
  float accumulate;    // WGRAD with splits == 1: out = accumulate * out + value
  int batch, H, W, C, OH, OW, K, KH, KW, sh, sw, pt, pl;
  int Cx;              // channel pitch of x / dx in memory (>= C)
  int Ky;              // channel pitch of y / dy in memory (>= K)
  int splits;
  unsigned mg_ow, mg_oh, mg_cp;   // magic multipliers: r / OW = (r * mg_ow) >> sh_ow for 0 <= r < 2^31 (host: fast_div)
  int sh_ow, sh_oh, sh_cp;        // _cp: division by the gathered channel count padded to a multiple of 4
  // FWD / DGRAD with splits == 1, optional: the layer's BatchNorm statistics out of the epilogue.  Each block leaves the
  // per-channel SUM of ITS tile rows (of the values as stored) and their sum of squared deviations from the tile's own
  // mean (M2) at stats[((g * stats_nblk + b) * 2 + {0: sum, 1: M2}) * N + n]; bn.hip merges the tiles (tile_stats below);
  // FWD: g = tile_row / stats_tpg, b = tile_row % stats_tpg; DGRAD (one group): b = class * stats_tpg + tile_row.
  float* stats;
  int stats_nblk, stats_tpg;
  // FWD / DGRAD slabs (splits > 1) handed to the layer's BatchNorm: 0 = each slab [rows][pitch] like the tensor; > 0 = the
  // ACG_SLABS_QUADS layout [channels / 4][slab_rows][4] (slab_rows = all rows of the tensor), element (m, n) at
  // ((n >> 2) * slab_rows + m) * 4 + (n & 3)
  int slab_rows;
  // FWD launched as the merged input gradient of a stride-2 layer with few input channels (conv_f32.hip run_merged): the N =
  // 4 * shuf_c output columns of pixel (b, p, q) are the shuf_c channels of the 2 x 2 output pixels (2p + ph, 2q + pw) of a
  // tensor that is shuf_w pixels wide at the channel pitch shuf_pitch; column n = (2 ph + pw) * shuf_c + c.  0 = off.
  int shuf_c, shuf_w, shuf_pitch;
  // Small maps (DGRAD; conv_f32.hip prepare): rows are ordered pixel-major - m = pixel * batch + b, so a tile covers one or
  // two pixels over the batch - and the tile walks only the taps ANY of its rows can see: on a 4 x 4 map half the taps of a
  // 5 x 5 filter fall into the padding for every row of such a tile, and their K-steps would multiply zeros.
  int compact;
  int Nv;              // FWD / DGRAD: only the first Nv output columns are computed (acg_conv_desc dgrad_c / adj_dgrad_c); 0 = all
  int tap_classes;     // FWD, stride > 1: 1 = the taps are walked class by class (tap_class_pos below); 0 = row-major
  int korder;          // bf16 kernels, FWD / DGRAD, gathered channels a multiple of 64: 1 = K-steps walk the taps of one 64-channel chunk
                       // before the next chunk (0: all channels of a tap before the next tap) - conv_bf16_kernel.h
  int out_f32;         // bf16 kernels, FWD / DGRAD: the result is stored as float32 (at the bf16 tensor's pitch, round8): a head layer
  // EPI kernel variants only (acg_deconv2d_fwd_bias_act): out = act(acc + bias[n]), stored as float32 at the pitch Cx
  const float* bias;
  int act;
  float leak;
};

// Division by a launch-constant through multiply-high (Granlund-Montgomery, N = 31): a runtime integer division is
// ~40 VALU instructions on this ISA, and WGRAD derives 32 gather rows per K-step.
struct FastDiv { unsigned magic; int shift; };
static inline FastDiv fast_div(int d) {
  int s = 0;
  while ((1ll << s) < d) ++s;
  const unsigned long long num = 1ull << (31 + s);
  return FastDiv{(unsigned)((num + (unsigned long long)d - 1) / (unsigned long long)d), 31 + s};
}
__device__ __forceinline__ int div_fast(int r, unsigned magic, int shift) {
  return (int)(((unsigned long long)(unsigned)r * magic) >> shift);
}

// Weight-gradient blocks: block l of gx * splits (x fastest) -> (tile, split).  The hardware deals workgroups round-robin
// over the 8 XCDs in dispatch order, so blocks l, l+8, l+16 ... share an XCD; all gx tiles of one split read the same
// pixel range of both operands (each at its own filter taps): give them to ONE XCD, so that range comes out of HBM / the
// Infinity Cache once instead of once per XCD (measured: g/tconv4's pair at 128x128 fetched 878 MB per launch, 6x its
// operands).  Splits that are not a multiple of 8 keep the plain order.  Placement only: results do not change.
__device__ __forceinline__ void wgrad_xcd_map(int l, int gx, int splits, int& tile, int& split) {
  if ((splits & 7) == 0) {
    const int x8 = l & 7, q = l >> 3, grp = q / gx;
    tile = q - grp * gx;
    split = grp * 8 + x8;
  } else {
    split = l / gx;
    tile = l - split * gx;
  }
}

struct alignas(16) RowInfo {
  int base;            // element offset of the row's (tap 0, channel 0) source element (may be virtual)
  unsigned mask_lo, mask_hi;  // bit t set <=> filter tap t reads inside the tensor
  int out_off;         // DGRAD: element offset of the output pixel; unused otherwise
};

__device__ __forceinline__ unsigned long long tap_mask(int lo_a, int hi_a, int lo_b, int hi_b, int nb) {
  // bits (a * nb + b) for a in [lo_a, hi_a), b in [lo_b, hi_b)
  if (hi_a <= lo_a || hi_b <= lo_b) return 0ull;
  const unsigned long long row = ((1ull << (hi_b - lo_b)) - 1ull) << lo_b;
  unsigned long long m = 0ull;
  for (int a = lo_a; a < hi_a; ++a) m |= row << (a * nb);
  return m;
}

// Class-major tap order of a STRIDED forward convolution (ConvArgs::tap_classes).  Tap (i, j) of a stride-(sh, sw) conv reads
// the input lattice (y = sh * p + i - pt, x = sw * q + j - pl): taps with equal (i % sh, j % sw) read the SAME sub-lattice of the
// input, shifted by whole output pixels, and taps of different classes read DISJOINT sub-lattices.  In row-major order the taps
// that share a cache line are up to two filter rows (10 K-steps x channel chunks x every tile running on the XCD) apart: the
// vertical re-use misses the 4 MB L2 and every input byte is fetched from the memory side ~3.5 times (TCC_MISS 36 % of the
// K-loop's requests, profiles/r4/h_bf16_kloop_load_path.txt).  Walking one class after the other keeps a tile on one sub-lattice - its
// window / (sh * sw), 43 KB for a 256-row tile of 64 channels - for all of the class's taps; it is a permutation of the K index
// only (tap tables and row masks are built in that order), so results differ by summation order alone.
// Built for stride 2 x 2 (every strided layer of the reference's models): class (a, b) = (i & 1, j & 1), na(a) x nb(b) taps each.
__device__ __forceinline__ int tap_class_pos(int i, int j, int KH, int KW) {
  const int a = i & 1, b = j & 1;
  const int na0 = (KH + 1) >> 1, nb0 = (KW + 1) >> 1;
  const int na = a ? KH >> 1 : na0, nb = b ? KW >> 1 : nb0;
  return (a ? na0 * KW : 0) + (b ? na * nb0 : 0) + (i >> 1) * nb + (j >> 1);
}
__device__ __forceinline__ unsigned long long tap_mask_classes(int lo_i, int hi_i, int lo_j, int hi_j, int KH, int KW) {
  // the rectangle [lo_i, hi_i) x [lo_j, hi_j) of taps is a rectangle of (ti, tj) in each of the four classes
  unsigned long long m = 0ull;
  const int na0 = (KH + 1) >> 1, nb0 = (KW + 1) >> 1;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int na = a ? KH >> 1 : na0, nb = b ? KW >> 1 : nb0;
      const int base = (a ? na0 * KW : 0) + (b ? na * nb0 : 0);
      m |= tap_mask(max(0, (lo_i - a + 1) >> 1), max(0, (hi_i - a + 1) >> 1), max(0, (lo_j - b + 1) >> 1), max(0, (hi_j - b + 1) >> 1), nb) << base;
    }
  return m;
}

// 32-bit operations only: 64-bit shifts and compares are quarter-rate on the VALU, and a wave's VALU work does NOT
// hide behind its own MFMAs (tools/conv_kloop.py experiments in profiles/), so every instruction in the K-loop counts
__device__ __forceinline__ bool tap_ok(const RowInfo& ri, int t) {
  const unsigned w = t < 32 ? ri.mask_lo : ri.mask_hi;
  return (w >> (t & 31)) & 1u;
}

// Branch-free guarded loads through buffer descriptors.  A divergent `if (ok) load` makes hipcc wait for every
// load inside its branch (vmcnt(0) per load: the loads of a K-step serialise), and selecting the loaded VALUE
// drags the wait up to the load as well.  Instead each operand tensor is addressed through a raw buffer
// resource (base + 32-bit byte offset, hardware range check): a masked lane gets an offset beyond the buffer
// and the load returns zeros without touching memory.  One v_cndmask on a 32-bit offset per load, no 64-bit
// pointer arithmetic, and every load of a stage issues back to back; the first wait is NST-1 K-steps later.
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOob = 0xFFFFFF00u;   // >= any num_records (tensors are < 2^30 elements)

__device__ __forceinline__ f4 guarded_quad(__amdgpu_buffer_rsrc_t rs, int elem_off, bool ok) {
  const unsigned off = ok ? (unsigned)elem_off * 4u : kOob;
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}
// Ragged quad (channel count not a multiple of 4): only the first nvalid (1..3) elements exist.
__device__ __forceinline__ f4 guarded_ragged(__amdgpu_buffer_rsrc_t rs, int elem_off, bool ok, int nvalid) {
  f4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned off = (ok && e < nvalid) ? (unsigned)(elem_off + e) * 4u : kOob;
    r[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
  }
  return r;
}
__device__ __forceinline__ float guarded_scalar(__amdgpu_buffer_rsrc_t rs, int elem_off, bool ok) {
  const unsigned off = ok ? (unsigned)elem_off * 4u : kOob;
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
}

// ---- BatchNorm statistics of a tile, out of the accumulators -----------------------------------------------------------
// E[x^2] - E[x]^2 from plain float32 sums loses the variance once |mean| >> std (late GAN training: d/conv layers with a
// drifted mean).  So nothing here ever squares an uncentred value: a thread centres its rows of a column on the first of
// them, turns the result into (count, sum, M2 about its own mean), and partial results are merged pairwise with the
// parallel-variance formula (Chan, Golub, LeVeque): M2 = M2a + M2b + (mean_b - mean_a)^2 * na * nb / (na + nb).
struct TileStat { float n, sum, m2; };
__device__ __forceinline__ TileStat stat_merge(const TileStat a, const TileStat b) {
  const float n = a.n + b.n;
  const float ma = a.sum / fmaxf(a.n, 1.f), mb = b.sum / fmaxf(b.n, 1.f), d = mb - ma;
  return TileStat{n, a.sum + b.sum, a.m2 + b.m2 + d * d * (a.n * b.n / fmaxf(n, 1.f))};
}
// equal, non-zero counts (full tiles): no divisions by the counts
__device__ __forceinline__ TileStat stat_merge_equal(const TileStat a, const TileStat b) {
  const float d = (b.sum - a.sum) / a.n;
  return TileStat{a.n + b.n, a.sum + b.sum, a.m2 + b.m2 + d * d * (0.5f * a.n)};
}
// The epilogue step shared by the fp32 and the bf16 kernel.  `value(a, b, r)` = accumulator element as it will be stored;
// `red`: >= 3 * WM * BN floats of LDS that nothing reads any more (the A tiles after the K loop's last barrier).
template <int BM, int BN, int WM, int TA, int TB, class V>
__device__ __forceinline__ void tile_stats_epilogue(V&& value, float* red, float* out_sum, int N, int M, int m0, int n0, int wm0,
                                                    int wn0, int wr, int lrow, int lk, int tid) {
  const bool rows_full = m0 + BM <= M;                      // block-uniform: full tiles skip the per-element row test
#pragma unroll
  for (int b = 0; b < TB; ++b) {
    TileStat t;
    if (rows_full) {
      const float c = value(0, b, 0);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = value(a, b, r) - c; s1 += d; s2 += d * d; }
      constexpr float n = 16.f * TA;
      t = TileStat{n, n * c + s1, fmaxf(s2 - s1 * s1 * (1.f / n), 0.f)};
      const TileStat o{n, __shfl_xor(t.sum, 32, 64), __shfl_xor(t.m2, 32, 64)};
      t = stat_merge_equal(t, o);
    } else {
      float c = 0.f, n = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lk;
          if (m0 + row < M) {
            const float v = value(a, b, r);
            if (n == 0.f) c = v;
            const float d = v - c;
            n += 1.f; s1 += d; s2 += d * d;
          }
        }
      t = TileStat{n, n * c + s1, fmaxf(s2 - s1 * s1 / fmaxf(n, 1.f), 0.f)};
      const TileStat o{__shfl_xor(t.n, 32, 64), __shfl_xor(t.sum, 32, 64), __shfl_xor(t.m2, 32, 64)};
      t = stat_merge(t, o);
    }
    if (lk == 0) {
      float* const q = red + (wr * BN + wn0 + 32 * b + lrow) * 3;
      q[0] = t.n; q[1] = t.sum; q[2] = t.m2;
    }
  }
  __syncthreads();
  if (tid < BN && n0 + tid < N) {
    TileStat t{red[tid * 3], red[tid * 3 + 1], red[tid * 3 + 2]};
#pragma unroll
    for (int w = 1; w < WM; ++w) {
      const float* const q = red + (w * BN + tid) * 3;
      t = stat_merge(t, TileStat{q[0], q[1], q[2]});
    }
    out_sum[n0 + tid] = t.sum; out_sum[N + n0 + tid] = t.m2;
  }
}

// RAGGED: gathered channel count not a multiple of 4; NVEC: dense operand rows are float4-loadable (N % 4 == 0).
// Both are compile-time so that the slow variants never share registers (and waits) with the fast one.
// (bf16 tensors take their own kernel: conv_bf16_kernel.h.)
//
// Pipeline (one barrier per K-step): LDS holds TWO K-steps of both tiles.  In iteration ks a wave (1) drains the
// register stage of step ks+1 into the other LDS buffer, (2) runs the MFMAs of step ks out of the current buffer and
// (3) issues the global loads of step ks+NST into the stage that was drained one iteration earlier.
// What the measurements behind this shape say (tools/conv_kloop.py, tools/micro/*.hip, profiles/r1):
//  * v_mfma_f32_32x32x2_f32 runs on the SIMD's fp32 vector ALUs (MI355X quotes the same 157.3 TFLOP/s for fp32
//    vector and fp32 matrix): VALU instructions of ANY wave on the SIMD do not overlap it - SIMD time per K-step is
//    the SUM of the MFMA passes and every VALU instruction issued, whatever the occupancy; LDS instructions overlap
//    only partly (tools/micro/mfma_overlap.hip with 1, 2 and 4 waves per SIMD).  More resident blocks hide latency,
//    not issue cycles.  Hence a loader that costs as few VALU instructions as possible: running offsets instead of
//    divisions and multiplies, 32-bit mask tests, LDS lookups fetched once per iteration, row infos in registers;
//  * the loop body must be ONE basic block with a single predecessor per unrolled copy, every load predicated by an
//    out-of-range offset instead of a branch: only then does hipcc count outstanding loads exactly (vmcnt(N), N > 0)
//    and the loads really run NST-1 K-steps ahead;
//  * a chain of dependent MFMAs through one accumulator already issues every 68-74 cycles (no extra sets needed).
// The block program.  (bx, by, bz) / gx stand in for blockIdx / gridDim.x so that conv_pair_f32 below can run the blocks
// of two contractions out of one 1-D grid.
// LDS of one block program: two K-steps of both tiles, the row infos, the tap tables.
template <int MODE, int BM, int BN>
constexpr int conv_lds_bytes() {
  return 2 * 8 * (BM + BN) * (int)sizeof(f4) + ((MODE == MODE_WGRAD) ? 2 * 256 : BM) * (int)sizeof(RowInfo) +
         2 * kMaxTaps * (int)sizeof(int);
}

// LIN (FWD, !RAGGED, NVEC): the gathered channel count is a multiple of 4 (Cs == Cp), so filter row k of W[(tap, c)][n] is
// row k of the [Kdim][N] matrix - the dense operand needs no (tap, channel) state (conv_body: run_boff).
template <int MODE, int BM, int BN, int WM, int WN, bool RAGGED, bool NVEC, bool EPI = false, bool LIN = false>
__device__ __forceinline__ void conv_body(const ConvArgs& p, const int bx, const int by, const int bz, const int gx, char* smem) {
  static_assert(WM * WN == 4, "4 waves per block");
  constexpr int TA = BM / (32 * WM), TB = BN / (32 * WN);
  constexpr int QA = BM / 32, QB = BN / 32;  // quads per thread per K-step
  constexpr int NROW = (MODE == MODE_WGRAD) ? 2 * 256 : BM;  // WGRAD: row infos of 2 x 8 K-steps (chunk parity)
#ifndef ACG_NST
#define ACG_NST 3      // whole-step sweep: 2 -> 339.5, 3 -> 349, 4 -> 344, 5 -> 340 steps/s
#endif
  constexpr int NST = ACG_NST;           // register stages: global loads run NST K-steps ahead of their MFMAs

  constexpr int KSLOTS = 8;              // 16-byte k-slots (quads) per tile column
  constexpr int ASZ = KSLOTS * BM, BSZ = KSLOTS * BN;
  static_assert(conv_lds_bytes<MODE, BM, BN>() == (2 * ASZ + 2 * BSZ) * (int)sizeof(f4) + NROW * (int)sizeof(RowInfo) + 2 * kMaxTaps * (int)sizeof(int), "LDS layout");
  f4* const As_all = reinterpret_cast<f4*>(smem);                         // [2 * ASZ]
  f4* const Bs_all = As_all + 2 * ASZ;                                    // [2 * BSZ]
  RowInfo* const rows = reinterpret_cast<RowInfo*>(Bs_all + 2 * BSZ);     // [NROW]
  int* const tapA = reinterpret_cast<int*>(rows + NROW);                  // [kMaxTaps]
  int* const tapB = tapA + kMaxTaps;                                      // [kMaxTaps]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gsrc), 0, p.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dense), 0, p.d_bytes, 0x00020000);

  // ---- problem geometry (wave-uniform) -------------------------------------------------------
  // Cs: channels of the gathered tensor; Cp: Cs padded to a multiple of 4 (quad granularity per tap).
  int M, N, Kdim, Cs, Cp, ntaps;
  int ph = 0, pw = 0, i0 = 0, j0 = 0, nti = 1, ntj = 1, dp0 = 0, dq0 = 0, Hc = 0, Wc = 0;
  if constexpr (MODE == MODE_FWD) {
    Cs = p.C; Cp = (Cs + 3) & ~3; ntaps = p.KH * p.KW;
    M = p.batch * p.OH * p.OW; N = p.K; Kdim = ntaps * Cp;
  } else if constexpr (MODE == MODE_DGRAD) {
    const int cls = by;
    ph = cls / p.sw; pw = cls - ph * p.sw;
    Hc = ph < p.H ? (p.H - ph + p.sh - 1) / p.sh : 0;
    Wc = pw < p.W ? (p.W - pw + p.sw - 1) / p.sw : 0;
    Cs = p.K; Cp = (Cs + 3) & ~3;
    M = p.batch * Hc * Wc; N = p.C;
    i0 = (ph + p.pt) % p.sh; j0 = (pw + p.pl) % p.sw;
    nti = i0 < p.KH ? (p.KH - i0 + p.sh - 1) / p.sh : 0;
    ntj = j0 < p.KW ? (p.KW - j0 + p.sw - 1) / p.sw : 0;
    dp0 = (ph + p.pt - i0) / p.sh; dq0 = (pw + p.pl - j0) / p.sw;
    ntaps = nti * ntj; Kdim = ntaps * Cp;
  } else {
    Cs = p.C; Cp = (Cs + 3) & ~3; ntaps = p.KH * p.KW;
    M = ntaps * Cp; N = p.K; Kdim = p.batch * p.OH * p.OW;
  }
  // NS: floats between two rows of the dense operand (FWD: filter rows W[(tap, c)][0..K)); N: the columns computed
  const int NS = MODE == MODE_FWD ? p.K : N;
  if (MODE != MODE_WGRAD && p.Nv > 0) N = p.Nv;
  const int tiles_n = (N + BN - 1) / BN;
  // XCD-aware order: consecutive workgroup ids are dealt round-robin over the 8 XCDs (each with its own L2), so
  // give every XCD a CONTIGUOUS run of tiles - neighbours in the run share A rows / filter columns in that L2.
  // Bijective for any grid size; placement affects speed only.
  int bid = bx;
  if constexpr (MODE != MODE_WGRAD) {      // (weight gradients are placed by wgrad_xcd_map in the kernel wrappers)
    const int nwg = gx, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  if (m0 >= M) return;  // block-uniform: DGRAD classes smaller than class 0

  int nk = (Kdim + BK - 1) / BK;
  int per = (nk + p.splits - 1) / p.splits;
  int ks_begin = bz * per;
  int ks_end = min(nk, ks_begin + per);

  // ---- tap tables ------------------------------------------------------------------------------
  if (tid < kMaxTaps && tid < ntaps) {
    const int t = tid;
    if constexpr (MODE == MODE_DGRAD) {
      const int ti = t / ntj, tj = t - ti * ntj;
      tapA[t] = -(ti * p.OW + tj) * p.Ky;
      tapB[t] = ((i0 + p.sh * ti) * p.KW + (j0 + p.sw * tj)) * p.C * p.K;
    } else {
      const int i = t / p.KW, j = t - i * p.KW;
      const int u = (MODE == MODE_FWD && p.tap_classes) ? tap_class_pos(i, j, p.KH, p.KW) : t;      // position in the K order
      tapA[u] = (i * p.W + j) * p.Cx;
      tapB[u] = MODE == MODE_FWD ? t * p.C * p.K : 0;      // FWD: first filter row of the tap (the table-driven dense rows below)
    }
  }

  // ---- per-row gather info -------------------------------------------------------------------
  auto fill_row_fwd = [&](int r /* global row (b,p,q) */, int limit) -> RowInfo {
    RowInfo ri; ri.base = 0; ri.mask_lo = 0; ri.mask_hi = 0; ri.out_off = 0;
    if (r < limit) {
      int q, b, pp;
      if (MODE == MODE_FWD && p.compact) { b = r % p.batch; const int px = r / p.batch; q = px % p.OW; pp = px / p.OW; }    // pixel-major rows
      else { const int t2 = div_fast(r, p.mg_ow, p.sh_ow); q = r - t2 * p.OW; b = div_fast(t2, p.mg_oh, p.sh_oh); pp = t2 - b * p.OH; }
      const int y0 = pp * p.sh - p.pt, x0 = q * p.sw - p.pl;
      ri.base = ((b * p.H + y0) * p.W + x0) * p.Cx;
      if (p.shuf_c) ri.out_off = ((b * (2 * p.OH) + 2 * pp) * p.shuf_w + 2 * q) * p.shuf_pitch;
      else if (MODE == MODE_FWD && p.compact) ri.out_off = ((b * p.OH + pp) * p.OW + q) * (p.slab_rows ? 4 : p.Ky);
      const unsigned long long m = (MODE == MODE_FWD && p.tap_classes) ? tap_mask_classes(max(0, -y0), min(p.KH, p.H - y0), max(0, -x0), min(p.KW, p.W - x0), p.KH, p.KW)
                                                                      : tap_mask(max(0, -y0), min(p.KH, p.H - y0), max(0, -x0), min(p.KW, p.W - x0), p.KW);
      ri.mask_lo = (unsigned)m; ri.mask_hi = (unsigned)(m >> 32);
    }
    return ri;
  };
  if constexpr (MODE == MODE_FWD) {
    for (int r = tid; r < BM; r += 256) rows[r] = fill_row_fwd(m0 + r, M);
  } else if constexpr (MODE == MODE_DGRAD) {
    for (int r = tid; r < BM; r += 256) {
      RowInfo ri; ri.base = 0; ri.mask_lo = 0; ri.mask_hi = 0; ri.out_off = 0;
      const int m = m0 + r;
      if (m < M) {
        int w2, h2, b;
        if (p.compact) { b = m % p.batch; const int pq = m / p.batch; w2 = pq % Wc; h2 = pq / Wc; }     // pixel-major rows
        else { w2 = m % Wc; const int t2 = m / Wc; h2 = t2 % Hc; b = t2 / Hc; }
        const int y0 = h2 + dp0, x0 = w2 + dq0;  // dY coordinates of tap (0,0)
        ri.base = ((b * p.OH + y0) * p.OW + x0) * p.Ky;
        ri.out_off = ((b * p.H + h2 * p.sh + ph) * p.W + (w2 * p.sw + pw)) * (p.slab_rows ? 4 : p.Cx);
        // tap (ti,tj) reads dY[y0 - ti][x0 - tj]
        const unsigned long long mk = tap_mask(max(0, y0 - p.OH + 1), min(nti, y0 + 1), max(0, x0 - p.OW + 1), min(ntj, x0 + 1), ntj);
        ri.mask_lo = (unsigned)mk; ri.mask_hi = (unsigned)(mk >> 32);
      }
      rows[r] = ri;
    }
  }
  __syncthreads();
  if constexpr (MODE != MODE_WGRAD) {
    if (p.compact) {      // block-uniform.  Keep only the taps a row of this tile can see: tap tables and row masks in compact order.
      __shared__ unsigned umask[2];
      if (tid < 2) umask[tid] = 0u;
      __syncthreads();
      if (tid < BM) {
        const RowInfo ri = rows[tid];
        if (ri.mask_lo) atomicOr(&umask[0], ri.mask_lo);
        if (ri.mask_hi) atomicOr(&umask[1], ri.mask_hi);
      }
      __syncthreads();
      const unsigned long long U = (unsigned long long)umask[0] | ((unsigned long long)umask[1] << 32);
      int ta = 0, tb = 0;
      const bool mine = tid < ntaps && ((U >> tid) & 1ull);
      if (mine) { ta = tapA[tid]; tb = tapB[tid]; }
      unsigned long long rm = 0ull;        // this row's mask over the compact taps
      if (tid < BM) {
        const RowInfo ri = rows[tid];
        const unsigned long long old = (unsigned long long)ri.mask_lo | ((unsigned long long)ri.mask_hi << 32);
        int pos = 0;
        for (int t = 0; t < ntaps; ++t)
          if ((U >> t) & 1ull) { rm |= ((old >> t) & 1ull) << pos; ++pos; }
      }
      __syncthreads();
      if (mine) { const int pos = __popcll(U & ((1ull << tid) - 1ull)); tapA[pos] = ta; tapB[pos] = tb; }
      if (tid < BM) { rows[tid].mask_lo = (unsigned)rm; rows[tid].mask_hi = (unsigned)(rm >> 32); }
      ntaps = __popcll(U);
      Kdim = ntaps * Cp;
      nk = (Kdim + BK - 1) / BK;
      per = (nk + p.splits - 1) / p.splits;
      ks_begin = bz * per;
      ks_end = min(nk, ks_begin + per);
      __syncthreads();
    }
  }

  // ---- loaders: NST register stages ---------------------------------------------------------------
  // k-fast operands (A of FWD/DGRAD, B of DGRAD) are gathered one quad per (row, k/4); the others
  // (B of FWD, A and B of WGRAD) are contiguous along the tile column, so a thread loads a 4x4 block
  // (4 k-rows x float4 of columns) and transposes it in registers into 4 quads.
  // Transposed loaders: the 32 x (BN/4) row-quads of a K-step are dealt evenly to the 256 threads - LPB = BN/32
  // consecutive k-rows (4, 2 or 1) of one column quad each, i.e. a whole, half or quarter 4x4 block - so every
  // thread runs the same instruction stream (no idle waves, no divergent branch in the loop) and the transposed
  // pieces are written as 16-, 8- or 4-byte parts of the k-quads in LDS.
  constexpr int LPA = BM / 32, LPB = BN / 32;
  static_assert(!LIN || (MODE == MODE_FWD && NVEC && !RAGGED), "LIN: float4-able forward contraction");
  // (Filter rows as column quads - a thread owns one column and fetches the four k-rows of a quad with four dword loads off
  // one offset register, so the quad arrives in LDS order without a register transpose - was measured on top of LIN: 31.5
  // instead of 36.8 VALU instructions per K-step but 10 instead of 4 VMEM instructions, 404.4 vs 406.0 steps/s: dropped.)
  constexpr int RA = (MODE == MODE_WGRAD) ? LPA : QA;
  constexpr int RB = (MODE == MODE_DGRAD) ? QB : (NVEC ? LPB : QB);
  f4 ra[NST][RA], rb[NST][RB];

  // (tap, channel) of a k index by multiply-high: the loaders are straight-line code (no running state, no loops)
  auto tap_of = [&](int kp, int& t, int& c) { t = div_fast(kp, p.mg_cp, p.sh_cp); c = kp - t * Cp; };
  // transposed loaders: thread -> (column quad jn, k quad kq); active while kq < 8
  const int jb = tid % (BN / 4), r0b = (tid / (BN / 4)) * LPB, kqb = r0b >> 2, subb = r0b & 3;
  constexpr bool nvec = NVEC;         // dense rows are 16-byte aligned and quads never straddle N
  const int ja = tid % (BM / 4), r0a = (tid / (BM / 4)) * LPA, kqa = r0a >> 2, suba = r0a & 3;
  int wg_t = 0, wg_off = 0, wg_nvalid = 0;   // WGRAD: this thread's fixed (padded) output-row quad -> (tap, channel)
  if constexpr (MODE == MODE_WGRAD) {
    const int mp = m0 + 4 * ja;
    if (mp < M) {
      wg_t = mp / Cp;
      const int c = mp - wg_t * Cp;
      wg_nvalid = Cs - c;            // >= 1
      wg_off = tapA[wg_t] + c;
    }
  }


  constexpr bool ragged = RAGGED;     // quads at the end of a tap are partial

  // `live` false (a step beyond this block's K range): every lane addresses out of range, the loads return zeros
  // without touching memory - the loop body stays branch-free, so hipcc counts outstanding loads exactly
  // (s_waitcnt vmcnt(N), N > 0) instead of draining them at every control-flow join.
  // The loads of a K-step are NLD independent "pieces" (A pieces first, then B) so that the K-loop can deal them out
  // between MFMAs; shared address arithmetic is written per piece and merged by the compiler.
  constexpr int NLD = RA + RB;
  // Everything a K-step's loads need from LDS (tap table entry, WGRAD row infos) is fetched by load_prep at the top
  // of an iteration, so no piece waits on an LDS round trip between two MFMAs; the row infos of the k-fast gathers
  // never change and live in registers.
  RowInfo myrow[MODE == MODE_WGRAD ? 1 : QA];
  if constexpr (MODE != MODE_WGRAD) {
#pragma unroll
    for (int u = 0; u < QA; ++u) myrow[u] = rows[(tid >> 3) + 32 * u];
  }
  // Running (tap, channel) state of this thread's k-fast quad and dense rows, advanced by one K-step (32 k) per
  // load_prep call with adds and ONE conditional wrap (32 = step_t * Cp + step_c with step_c < Cp): the K-loop has
  // no division, no multiply and no 64-bit arithmetic.  load_prep must therefore be called for consecutive ks,
  // starting at ks_begin - which is how the pipeline below issues its loads.
  const int step_t = BK / Cp, step_c = BK - step_t * Cp;
  int run_kt = 0, run_kc = 0, run_bt = 0, run_bc = 0, run_boff = 0;
  if constexpr (MODE != MODE_WGRAD) tap_of(ks_begin * BK + 4 * (tid & 7), run_kt, run_kc);
  const int nB = n0 + 4 * jb;                       // first column of the dense-row quads (nvec)
  // LIN: the filter rows are linear in k, one running offset, valid while k < Kdim.
  int run_bk = 0;                                   // LIN: k of this thread's first dense row
  if constexpr (LIN) {
    run_bk = ks_begin * BK + r0b;
    run_boff = run_bk * NS + nB;
  } else if constexpr (MODE == MODE_FWD && nvec) {
    tap_of(ks_begin * BK + 4 * kqb, run_bt, run_bc);
    run_boff = (run_bt * Cs + run_bc + subb) * NS + nB;
  }
  if constexpr (MODE == MODE_WGRAD && nvec) run_boff = (ks_begin * BK + r0b) * p.Ky + nB;
  int nK[MODE == MODE_DGRAD ? QB : 1];                                     // DGRAD: n * K of this thread's filter rows
  bool nOk[MODE == MODE_DGRAD ? QB : 1];
  if constexpr (MODE == MODE_DGRAD) {
#pragma unroll
    for (int u = 0; u < QB; ++u) { const int n = n0 + (tid >> 3) + 32 * u; nK[u] = n * p.K; nOk[u] = n < N; }
  }
#ifdef ACG_NORM_PROBE
  // COST PROBE (tools only, `make probe`; never in the shipped library): what normalise-on-load would add to the loaders -
  // act((x - mean) * rstd + beta) on every GATHERED element on its way into LDS, zero padding applied after the activation.
  // Parameters that leave the values unchanged (scale 1, shift 0, floor -inf) come from LDS so that nothing folds away;
  // the in-bounds flag of every gathered quad is carried from its load to its store, as the real thing would have to.
  __shared__ f4 probe_par[3][16];
  if (tid < 16) { probe_par[0][tid] = f4{1.f, 1.f, 1.f, 1.f}; probe_par[1][tid] = f4{0.f, 0.f, 0.f, 0.f}; probe_par[2][tid] = f4{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f}; }
  bool probe_ok[NST][RA];
  f4 probe_s = {1.f, 1.f, 1.f, 1.f}, probe_t = {0.f, 0.f, 0.f, 0.f}, probe_lo = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
  __syncthreads();
  if constexpr (MODE == MODE_WGRAD) { probe_s = probe_par[0][tid & 15]; probe_t = probe_par[1][tid & 15]; probe_lo = probe_par[2][tid & 15]; }   // a thread's channel quad is fixed
  auto probe_apply = [&](f4 q, const f4& ps, const f4& pt, const f4& pl, bool ok) {
    f4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float z = fmaf(q[e], ps[e], pt[e]); r[e] = ok ? fmaxf(z, pl[e]) : 0.f; }
    return r;
  };
#endif
  struct Prep {
    int t, kc, aoff, tb, bt, bc, boff, brow;
    bool kv;
    RowInfo wri[MODE == MODE_WGRAD ? LPA : 1];
  };
  auto load_prep = [&](int ks, bool live) {
    Prep q{};
    if constexpr (MODE == MODE_WGRAD) {
#pragma unroll
      for (int e = 0; e < LPA; ++e) q.wri[e] = rows[((ks - ks_begin) & 15) * BK + r0a + e];
    } else {
      q.kc = run_kc;
      q.kv = live && run_kt < ntaps;
      q.t = q.kv ? run_kt : 0;
      q.aoff = tapA[q.t] + run_kc;
      if constexpr (MODE == MODE_DGRAD) q.tb = tapB[q.t] + run_kc;
      run_kt += step_t; run_kc += step_c;
      const bool wrap = run_kc >= Cp;
      run_kc -= wrap ? Cp : 0; run_kt += wrap ? 1 : 0;
    }
    if constexpr (LIN) {
      q.boff = run_boff; q.brow = run_bk;
      run_boff += BK * NS; run_bk += BK;
    } else if constexpr (MODE == MODE_FWD && nvec) {
      // (tap, channel) state + the tap's first filter row from the table: the taps of a compacted tile are not consecutive
      q.bt = run_bt; q.bc = run_bc + subb; q.boff = tapB[min(run_bt, ntaps - 1)] + (run_bc + subb) * NS + nB;
      run_bt += step_t; run_bc += step_c;
      const bool wrap = run_bc >= Cp;
      run_bc -= wrap ? Cp : 0; run_bt += wrap ? 1 : 0;
    }
    if constexpr (MODE == MODE_WGRAD && nvec) {
      q.boff = run_boff; q.brow = ks * BK + r0b;
      run_boff += BK * p.Ky;
    }
    return q;
  };
  auto load_piece = [&](auto stage, int ks, bool live, const Prep& q, auto pc) {
    constexpr int ST = decltype(stage)::value;
    constexpr int I = decltype(pc)::value;
    if constexpr (I < RA) {
      if constexpr (MODE == MODE_WGRAD) {   // raw rows now; the transpose happens at store time
        const RowInfo ri = q.wri[I];
        const bool ok = live && wg_nvalid > 0 && tap_ok(ri, wg_t);
#ifdef ACG_NORM_PROBE
        probe_ok[ST][I] = ok;
#endif
        if constexpr (!ragged) ra[ST][I] = guarded_quad(rs_g, ri.base + wg_off, ok);
        else ra[ST][I] = guarded_ragged(rs_g, ri.base + wg_off, ok, wg_nvalid);
      } else {
        // k-fast gather: 8 consecutive lanes walk 8 quads (128 contiguous bytes) of one gathered row
        const RowInfo ri = myrow[I];
#ifdef ACG_NORM_PROBE
        probe_ok[ST][I] = q.kv && tap_ok(ri, q.t);
#endif
        if constexpr (!ragged) ra[ST][I] = guarded_quad(rs_g, ri.base + q.aoff, q.kv && tap_ok(ri, q.t));
        else ra[ST][I] = guarded_ragged(rs_g, ri.base + q.aoff, q.kv && tap_ok(ri, q.t), Cs - q.kc);
      }
    } else {
      constexpr int U = I - RA;
      if constexpr (MODE == MODE_DGRAD) {  // B[k=(tap,o)][n=c] = W[tap][c][o], contiguous along o
        if constexpr (!ragged) rb[ST][U] = guarded_quad(rs_d, q.tb + nK[U], q.kv && nOk[U]);
        else rb[ST][U] = guarded_ragged(rs_d, q.tb + nK[U], q.kv && nOk[U], Cs - q.kc);
      } else if constexpr (nvec) {
        // dense operand: FWD W[(tap,c)][n] rows, WGRAD dY[(b,p,q)][n] rows; raw row, transposed at store time
        bool ok;
        if constexpr (LIN) ok = q.brow < Kdim;          // the LPB rows of a thread lie in one k-quad; Kdim % 4 == 0
        else if constexpr (MODE == MODE_FWD) ok = q.bt < ntaps && q.bc + U < Cs;
        else ok = q.brow + U < Kdim;
        rb[ST][U] = guarded_quad(rs_d, q.boff + U * (MODE == MODE_WGRAD ? p.Ky : NS), live && ok && nB < N);
      } else {  // ragged N (25, 5, 3, 1 ...): 4 k-rows of one column per quad, lanes along n
        constexpr int stepB = 256 / BN;
        const int n = n0 + (tid % BN), kq = tid / BN + stepB * U;
        f4 v;
        if constexpr (MODE == MODE_FWD) {
          int t2, c2;
          tap_of(ks * BK + 4 * kq, t2, c2);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = guarded_scalar(rs_d, tapB[min(t2, ntaps - 1)] + (c2 + e) * NS + n, live && t2 < ntaps && c2 + e < Cs && n < N);
        } else {
          const int r = ks * BK + 4 * kq;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = guarded_scalar(rs_d, (r + e) * p.Ky + n, live && r + e < Kdim && n < N);
        }
        rb[ST][U] = v;
      }
    }
  };
  auto load_tiles = [&](auto stage, int ks, bool live) {
    const Prep q = load_prep(ks, live);
    static_for<0, NLD>([&](auto pc) { load_piece(stage, ks, live, q, pc); });
  };

  // LDS column permutation: physical = P(col) ^ kq with P(32q + 4j + e) = 32q + 8e + (j ^ 4(e>>1)).
  // Conflict-free for (i) k-fast stores (8 lanes: one column, kq = 0..7), (ii) transposed stores (8 lanes:
  // columns 4j+e for 8 consecutive j) and (iii) the MFMA operand reads (32 consecutive columns, one kq).
  auto pcol = [](int col, int kq) {
    const int j = (col >> 2) & 7, e = col & 3;
    return ((col & ~31) | (e << 3) | (j ^ ((e >> 1) << 2))) ^ kq;
  };

  // write quad `kq` of tile column `col`
  auto put = [&](f4* tile, int width, int kq, int col, const f4& q) { tile[kq * width + pcol(col, kq)] = q; };

  // write k-rows sub..sub+L-1 of quad `kq` of tile column `col` (L = 4, 2, 1: a whole, half or quarter quad)
  auto put_part = [&](f4* tile, int width, int kq, int sub, int col, auto lc, const float* v) {
    constexpr int L = decltype(lc)::value;
    char* dst = reinterpret_cast<char*>(tile + kq * width + pcol(col, kq)) + sub * 4;
    if constexpr (L == 4) *reinterpret_cast<f4*>(dst) = f4{v[0], v[1], v[2], v[3]};
    else if constexpr (L == 2) *reinterpret_cast<f2*>(dst) = f2{v[0], v[1]};
    else *reinterpret_cast<float*>(dst) = v[0];
  };

  auto store_tiles = [&](auto stage, int buf) {
    constexpr int ST = decltype(stage)::value;
    f4* const As = As_all + buf * ASZ;
    f4* const Bs = Bs_all + buf * BSZ;
    if constexpr (MODE == MODE_WGRAD) {
#ifdef ACG_NORM_PROBE
#pragma unroll
      for (int e = 0; e < LPA; ++e) ra[ST][e] = probe_apply(ra[ST][e], probe_s, probe_t, probe_lo, probe_ok[ST][e]);
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v[LPA];
#pragma unroll
        for (int e = 0; e < LPA; ++e) v[e] = ra[ST][e][i];
        put_part(As, BM, kqa, suba, 4 * ja + i, std::integral_constant<int, LPA>{}, v);
      }
    } else {
      const int kq = tid & 7, rg = tid >> 3;
#ifdef ACG_NORM_PROBE
      {   // the channel quad of this K-step's gathers: one parameter lookup per K-step (all QA quads of a thread share it)
        const int c4 = (kq + ST) & 15;
        const f4 ps = probe_par[0][c4], pt = probe_par[1][c4], pl = probe_par[2][c4];
#pragma unroll
        for (int u = 0; u < QA; ++u) ra[ST][u] = probe_apply(ra[ST][u], ps, pt, pl, probe_ok[ST][u]);
      }
#endif
#pragma unroll
      for (int u = 0; u < QA; ++u) put(As, BM, kq, rg + 32 * u, ra[ST][u]);
      if constexpr (MODE == MODE_DGRAD) {
#pragma unroll
        for (int u = 0; u < QB; ++u) put(Bs, BN, kq, rg + 32 * u, rb[ST][u]);
      }
    }
    if constexpr (MODE != MODE_DGRAD) {
      if constexpr (nvec) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v[LPB];
#pragma unroll
          for (int e = 0; e < LPB; ++e) v[e] = rb[ST][e][i];
          put_part(Bs, BN, kqb, subb, 4 * jb + i, std::integral_constant<int, LPB>{}, v);
        }
      } else {
        constexpr int stepB = 256 / BN;
        const int nb = tid % BN, kq0 = tid / BN;
#pragma unroll
        for (int u = 0; u < QB; ++u) put(Bs, BN, kq0 + stepB * u, nb, rb[ST][u]);
      }
    }
  };

  // ---- main loop ----------------------------------------------------------------------------------
  // Accumulator sets rotated by k within a K-step (summed in the epilogue).  Measured (tools/micro/mfma_rate.hip):
  // a chain of dependent v_mfma_f32_32x32x2_f32 through ONE accumulator already issues every 68-74 cycles, so one
  // set is enough; the knob stays for experiments.
#ifndef ACG_NSET
#define ACG_NSET 1
#endif
  constexpr int NSET = (TA * TB >= 4) ? 1 : ACG_NSET;
  f32x16 accs[NSET][TA][TB];
  auto& acc = accs[0];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int z = 0; z < NSET; ++z) accs[z][a][b][r] = 0.f;
      }

  const int wr = wave / WN, wc = wave - wr * WN;
  const int wm0 = wr * (BM / WM), wn0 = wc * (BN / WN);
  const int lrow = lane & 31, lk = lane >> 5;

  // MFMA work of one K-step as NM single instructions in NT groups (a group = one k-slot per lane half): group t
  // reads quad 2t+h of BOTH tiles for lane half h - the MFMA contracts slot (lane>>5) of A with the same slot of B,
  // so the k order inside a K-step is free - and feeds it to GM MFMAs.
  constexpr int NT = 4, GM = 4 * TA * TB, NM = NT * GM;
  f4 av[2][TA], bv[2][TB];        // operand fragments of group t live in av[t & 1]
  auto frag_read = [&](int buf, auto tc) {
    constexpr int T = decltype(tc)::value;
    const f4* const As = As_all + buf * ASZ;
    const f4* const Bs = Bs_all + buf * BSZ;
    const int kq = 2 * T + lk;
#pragma unroll
    for (int a = 0; a < TA; ++a) av[T & 1][a] = As[kq * BM + pcol(wm0 + 32 * a + lrow, kq)];
#pragma unroll
    for (int b = 0; b < TB; ++b) bv[T & 1][b] = Bs[kq * BN + pcol(wn0 + 32 * b + lrow, kq)];
  };
  auto mfma = [&](auto ic) {
    constexpr int I = decltype(ic)::value;
    constexpr int T = I / GM, R = I % GM, AB = R % (TA * TB), A = AB / TB, B = AB % TB;
    constexpr int E = R / (TA * TB);
    accs[E % NSET][A][B] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[T & 1][A][E], bv[T & 1][B][E], accs[E % NSET][A][B], 0, 0, 0);
  };

  // WGRAD gathers along the reduction: the row infos change every K-step.  They are derived 8 K-steps (256 rows,
  // one per thread) at a time into the half of `rows` selected by the chunk parity, one barrier ahead of their
  // first use, so the K-loop itself carries no divergent "first 32 threads" section.
  auto wgrad_rows = [&](int chunk) {
    if constexpr (MODE == MODE_WGRAD) rows[(chunk & 1) * 256 + tid] = fill_row_fwd((ks_begin + 8 * chunk) * BK + tid, Kdim);
  };

  // prologue: stages 0..NST-1 <- steps ks_begin..+NST-1; step ks_begin -> LDS buffer 0
  if (ks_begin < ks_end) {
    if constexpr (MODE == MODE_WGRAD) { wgrad_rows(0); __syncthreads(); }
    static_for<0, NST>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      load_tiles(jc, ks_begin + J, ks_begin + J < ks_end);
    });
    store_tiles(std::integral_constant<int, 0>{}, 0);
    __syncthreads();
  }
  // iteration ks: LDS[buf] holds step ks; stage J (its source) is free; stage J+1 holds step ks+1.
  // The loads of the step NST ahead are dealt out between the MFMAs and the order is pinned with sched_barrier: it
  // keeps the fragment reads of group t+1 in front of the MFMAs of group t and spreads the VMEM issue over the K-step.
  auto iterate = [&](auto jc, int ks, int buf) {
    constexpr int J = decltype(jc)::value;
    using SN = std::integral_constant<int, (J + 1) % NST>;
    const bool live = ks + NST < ks_end;
    frag_read(buf, std::integral_constant<int, 0>{});
    const Prep q = load_prep(ks + NST, live);
    store_tiles(SN{}, buf ^ 1);       // overlaps the LDS latency of the first fragments and of the lookups
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NM>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      mfma(ic);
      if constexpr (I % GM == 0 && I / GM + 1 < NT) frag_read(buf, std::integral_constant<int, I / GM + 1>{});
      static_for<0, NLD>([&](auto pc) {
        constexpr int P = decltype(pc)::value;
        if constexpr ((P * NM) / NLD == I) load_piece(jc, ks + NST, live, q, pc);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (MODE == MODE_WGRAD) {
      const int rel = ks - ks_begin + NST + 1;      // first step whose loads are issued in the NEXT iteration
      if ((rel & 7) == 0 && ks + NST + 1 < ks_end) wgrad_rows(rel >> 3);
    }
    __syncthreads();
  };
  // (stage, LDS buffer) repeat with period UNR = lcm(NST, 2); every unrolled iteration has exactly ONE predecessor
  // besides the loop entry, so the compiler's outstanding-load bookkeeping stays exact (chained `if (ks+j < end)`
  // do not: each skipped iteration is a second path into the next one and the merge drains vmcnt to 0)
  if (ks_begin < ks_end) {
    constexpr int UNR = (NST % 2 == 0) ? NST : 2 * NST;
    int ks = ks_begin;
    while (run_unrolled<0, UNR>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      iterate(std::integral_constant<int, I % NST>{}, ks, I & 1);
      return ++ks < ks_end;
    })) {
    }
  }

  if constexpr (NSET > 1) {
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        if constexpr (NSET == 4) acc[a][b] = (accs[0][a][b] + accs[1][a][b]) + (accs[2][a][b] + accs[3][a][b]);
        else acc[a][b] = accs[0][a][b] + accs[NSET - 1][a][b];
      }
  }
  // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) --------------
  if constexpr (MODE != MODE_WGRAD) {
    if (p.stats != nullptr) {     // BatchNorm statistics of this tile (ConvArgs::stats): thread -> lane-half pair -> waves of a column
      int g = 0, blk;
      if constexpr (MODE == MODE_DGRAD) { blk = by * p.stats_tpg + tm; } else { g = tm / p.stats_tpg; blk = tm - g * p.stats_tpg; }
      tile_stats_epilogue<BM, BN, WM, TA, TB>([&](int a, int b, int r) { return acc[a][b][r]; }, reinterpret_cast<float*>(smem),
                                              p.stats + ((long long)g * p.stats_nblk + blk) * 2 * N, N, M, m0, n0, wm0, wn0, wr, lrow, lk, tid);
    }
  }
  if constexpr (EPI) {          // bias + activation of a layer built with normalizer_fn = None (models.py:20-21: tanh(deconv + b))
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      const int n = n0 + wn0 + 32 * b + lrow;
      const float bv = n < N ? p.bias[n] : 0.f;
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = acg::act_apply(p.act, acc[a][b][r] + bv, p.leak);
    }
  }
  float* outp = p.out + (p.splits > 1 ? (long long)bz * p.out_numel : 0ll);
  // Full tiles take a straight-line path - one base per 32 x 32 sub-tile, sixteen stores at row strides - instead of a
  // bounds test, a branch and a 64-bit index multiply per element (conv_bf16_kernel.h has the measurement).
  // (rows full; a ragged last column tile - 3, 6, 25 or 138 output channels - only masks lanes, once per 32-column sub-tile)
  const bool full = m0 + BM <= M && (MODE != MODE_WGRAD || (Cp == Cs && !(p.splits == 1 && p.accumulate != 0.f)));
  // column term of an element's offset: n, or in the quad slab layout (ConvArgs::slab_rows) the quad's run of rows + n & 3
  const bool quads = MODE != MODE_WGRAD && p.slab_rows > 0;
  // FWD rows addressed by their out_off: the merged input gradient (columns = 2 x 2 pixel, channel) and pixel-major small maps
  const bool shuf = MODE == MODE_FWD && (p.shuf_c > 0 || p.compact);
  auto col_off = [&](int n) -> long long {
    if (MODE == MODE_FWD && p.shuf_c > 0) { const int cls = n / p.shuf_c; return (long long)((cls >> 1) * p.shuf_w + (cls & 1)) * p.shuf_pitch + (n - cls * p.shuf_c); }
    return quads ? (long long)(n >> 2) * p.slab_rows * 4 + (n & 3) : (long long)n;
  };
  const long long row_pitch = MODE == MODE_WGRAD ? N : (quads ? 4 : p.Ky);
  if (full) {
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      if (n0 + wn0 + 32 * b + lrow < N) {
        const long long co = col_off(n0 + wn0 + 32 * b + lrow);
        if (MODE == MODE_DGRAD || shuf) {
#pragma unroll
          for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              outp[(long long)rows[wm0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lk].out_off + co] = acc[a][b][r];
        } else {
#pragma unroll
          for (int a = 0; a < TA; ++a) {
            float* const o = outp + (long long)(m0 + wm0 + 32 * a + 4 * lk) * row_pitch + co;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(long long)((r & 3) + 8 * (r >> 2)) * row_pitch] = acc[a][b][r];
          }
        }
      }
    }
    return;
  }
  // partial row tiles: per ROW one validity test and one base index (the weight gradient's division by the padded channel count
  // included), per element only the column mask
  const bool acc_out = MODE == MODE_WGRAD && p.splits == 1 && p.accumulate != 0.f;
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lk;
      const int m = m0 + row;
      if (m >= M) continue;
      long long base;
      if (MODE == MODE_DGRAD || shuf) {
        base = rows[row].out_off;
      } else if constexpr (MODE == MODE_WGRAD) {
        const int t = div_fast(m, p.mg_cp, p.sh_cp), c = m - t * Cp;      // (Cp == Cs: c < Cs always)
        if (c >= Cs) continue;
        base = ((long long)t * Cs + c) * N;
      } else {
        base = (long long)m * row_pitch;
      }
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        const int n = n0 + wn0 + 32 * b + lrow;
        if (n < N) {
          const long long o = base + col_off(n);
          float v = acc[a][b][r];
          if (acc_out) v += p.accumulate * outp[o];
          outp[o] = v;
        }
      }
    }
}

template <int MODE, int BM, int BN, int WM, int WN, bool RAGGED, bool NVEC, bool EPI = false, bool LIN = false>
__global__ __launch_bounds__(256) void conv_mfma_f32(const ConvArgs p) {
  __shared__ __align__(16) char smem[conv_lds_bytes<MODE, BM, BN>()];
  if constexpr (MODE == MODE_WGRAD) {      // grid (tiles, 1, splits)
    int tile, split;
    wgrad_xcd_map((int)(blockIdx.x + gridDim.x * blockIdx.z), (int)gridDim.x, (int)gridDim.z, tile, split);
    conv_body<MODE, BM, BN, WM, WN, RAGGED, NVEC>(p, tile, 0, split, (int)gridDim.x, smem);
  } else {
    conv_body<MODE, BM, BN, WM, WN, RAGGED, NVEC, EPI, LIN>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, smem);
  }
}


// ---- two contractions in one launch -------------------------------------------------------------------------------
// A layer's input gradient (DGRAD, or FWD on the adjoint descriptor for a transposed layer) and its weight gradient
// both consume dy and are independent of each other.  Launched one after the other, each runs with about one block
// per CU and pays its own launch gap, prologue and tail; out of ONE grid the blocks of the second contraction become
// the second resident block of every CU (0.74 -> 0.53 us per K-step) and fill the first one's tail.  Two streams
// would do the same, but a cross-stream edge costs ~18 us on this stack (profiles/r1/w_stream_edges.txt).
// Blocks [0, nA) run contraction A with the grid (gxA, gyA, *), the rest run the weight gradient with (gxB, 1, *).
struct PairGeom { int nA, gxA, gyA, gxB; };

template <int MODE_A, int BMA, int BNA, int WMA, int WNA, int BMB, int BNB, int WMB, int WNB, bool RAGGED, bool LIN_A = false>
__global__ __launch_bounds__(256) void conv_pair_f32(const ConvArgs a, const ConvArgs b, const PairGeom g) {
  constexpr int LA = conv_lds_bytes<MODE_A, BMA, BNA>(), LB = conv_lds_bytes<MODE_WGRAD, BMB, BNB>();
  __shared__ __align__(16) char smem[LA > LB ? LA : LB];      // one block runs one of the two programs
  const int L = (int)blockIdx.x;
  if (L < g.nA) {
    const int t = L / g.gxA;
    conv_body<MODE_A, BMA, BNA, WMA, WNA, RAGGED, true, false, LIN_A>(a, L - t * g.gxA, t % g.gyA, t / g.gyA, g.gxA, smem);
  } else {
    int tile, split;
    wgrad_xcd_map(L - g.nA, g.gxB, b.splits, tile, split);
    conv_body<MODE_WGRAD, BMB, BNB, WMB, WNB, RAGGED, true>(b, tile, 0, split, g.gxB, smem);
  }
}

struct Plan {
  int cfg;       // 2: 128x32, 3: 64x64 (0 and 1 were the retired 128x128 / 128x64 tiles)
  int bm, bn;
  long long M, N;  // per class (class 0 = largest) GEMM extents (M padded per tap for WGRAD)
  int classes, nk, splits;
  bool ragged, nvec, bf16;
  bool direct;     // a few-MFLOP float32 contraction with at most 8 output channels: the direct kernels of conv_direct.hip, one launch, no slabs
  long long tiles, out_numel;
};

// conv_direct.hip
int launch_direct(int which, const ConvArgs& a, hipStream_t st);
int launch_direct_pair(const ConvArgs& a, const ConvArgs& b, hipStream_t st);

template <int MODE>
int launch_mode(const Plan& pl, const ConvArgs& a, hipStream_t st);

// conv_f32_pair.hip: A (modeA = MODE_FWD or MODE_DGRAD) and a weight gradient B in one launch; fp32, float4-able dense
// operands, both gathers ragged or neither (pair_supported) - the caller launches the two separately otherwise.
bool pair_supported(int modeA, const Plan& pa, const Plan& pb);
// conv_f32_dgrad.hip / conv_bf16_dgrad.hip: the one EPI instantiation each (a transposed layer's forward on the 128x32 tile)
int launch_deconv_fwd_epi(const Plan& pl, const ConvArgs& a, hipStream_t st);
int launch_deconv_fwd_epi16(const Plan& pl, const ConvArgs& a, hipStream_t st);
int launch_pair(int modeA, const Plan& pa, const ConvArgs& a, const Plan& pb, const ConvArgs& b, hipStream_t st);

template <int MODE, bool RAGGED, bool NVEC>
static inline void launch_cfg(const Plan& pl, const ConvArgs& a, hipStream_t st) {
  const dim3 grid((unsigned)(acg::ceil_div(pl.M, pl.bm) * acg::ceil_div(pl.N, pl.bn)), (unsigned)pl.classes, (unsigned)pl.splits);
  // two tile shapes: 128x32 for narrow N, 64x64 otherwise.  128x128 / 128x64 variants existed through v4; with the
  // one-barrier pipeline they lost every layer of the tuning sweep (profiles/r1) and were dropped.
  if constexpr (MODE == MODE_FWD && !RAGGED && NVEC) {
    if ((a.C & 3) == 0 && !a.compact && !a.tap_classes) {     // gathered channels a multiple of 4, taps in natural order: the LIN variant (filter rows linear in k)
      if (pl.cfg == 2) ACG_LAUNCH((conv_mfma_f32<MODE, 128, 32, 4, 1, RAGGED, NVEC, false, true>), grid, dim3(256), 0, st, a);
      else ACG_LAUNCH((conv_mfma_f32<MODE, 64, 64, 2, 2, RAGGED, NVEC, false, true>), grid, dim3(256), 0, st, a);
      return;
    }
  }
  if (pl.cfg == 2) ACG_LAUNCH((conv_mfma_f32<MODE, 128, 32, 4, 1, RAGGED, NVEC>), grid, dim3(256), 0, st, a);
  else ACG_LAUNCH((conv_mfma_f32<MODE, 64, 64, 2, 2, RAGGED, NVEC>), grid, dim3(256), 0, st, a);
}
template <int MODE>
static inline void launch_variant(const Plan& pl, const ConvArgs& a, hipStream_t st) {
  if (pl.ragged) {
    if (pl.nvec) launch_cfg<MODE, true, true>(pl, a, st);
    else launch_cfg<MODE, true, false>(pl, a, st);
  } else {
    if (pl.nvec) launch_cfg<MODE, false, true>(pl, a, st);
    else launch_cfg<MODE, false, false>(pl, a, st);
  }
}

#define ACG_DEFINE_CONV_LAUNCH(MODE)                                                    \
  template <>                                                                           \
  int launch_mode<MODE>(const Plan& pl, const ConvArgs& a, hipStream_t st) {            \
    launch_variant<MODE>(pl, a, st);                                                    \
    return acg::check_launch("conv_mfma_f32");                                          \
  }

}  // namespace acgconv
