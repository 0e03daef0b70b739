// NHWC implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), gfx950.
//
// One templated kernel serves the three contractions of a conv layer and, through the adjoint
// descriptor, of a TF conv2d_transpose layer (include/acgan_hip.h):
//   FWD    y [M=(b,p,q)][N=o]    = sum_{k=(tap,c)}   G[m][k]      * W[k][n]
//   DGRAD  dx[M=(b,h2,w2)][N=c]  = sum_{k=(tap,o)}   dY~[m][k]    * W^T[k][n]   per stride-parity class
//   WGRAD  dw[M=(tap,c)][N=o]    = sum_{k=(b,p,q)}   G[k][m]      * dY[k][n]
// G is the never-materialised im2col matrix: each block keeps, per gathered row, a base offset and a
// 64-bit mask of in-bounds filter taps in LDS, so the inner gather is one shift/and + one 16-byte load.
// DGRAD runs as stride_h*stride_w parity classes (blockIdx.y), each a dense stride-1 correlation over
// dY with its own tap subset - no multiplies by structural zeros (5x5/s2: 9+6+6+4 = 25 taps in total).
//
// K (or, for WGRAD, M) is indexed as (tap, channel) with the channel count padded per tap to a multiple
// of 4, so a "quad" of 4 consecutive k never straddles a tap and is one dwordx4 gather along NHWC
// channels (any Cin: 3, 6, 138, 266 included - the ragged last quad of a tap is loaded element-wise).
// LDS holds the A and B tiles as quads, [k/4][P(m) ^ k/4] float4 (column permutation P in the kernel): the
// 8 lanes of every ds_write_b128 group and the 16 lanes of every ds_read_b128 group hit distinct 16-byte
// slots, with no padding.  Operands that are contiguous along the tile column instead of along k (the
// filter in FWD, both operands in WGRAD) are loaded as 4x4 blocks and transposed in registers.  Because the MFMA contracts k-slot (lane>>5) of A with the same slot of B, the k order
// inside a K-step is free: lane half h reads quad 2t+h of both tiles and feeds its four floats to four
// consecutive MFMAs - one ds_read_b128 per 32x32 tile per 8 k.
//
// Tiling: 256 threads = 4 waves, block tile BM x BN x 32, wave tile (BM/WM) x (BN/WN) built from 32x32
// MFMA tiles.  Global loads run TWO K-steps ahead of the MFMAs in registers (two named stages), so a
// block hides memory latency on its own even when the grid is too small for many blocks per CU.  Small
// grids are filled by split-K over blockIdx.z into workspace slabs that a second kernel sums in fixed
// order (deterministic; no float atomics).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned 16-byte global load

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };
constexpr int BK = 32;  // k per K-step = 8 quads
constexpr int kMaxTaps = 64;

struct ConvArgs {
  const float* gsrc;   // gathered tensor: x (FWD, WGRAD) or dy (DGRAD)
  const float* dense;  // w (FWD, DGRAD) or dy (WGRAD)
  float* out;          // final tensor (splits == 1) or `splits` slabs of out_numel floats
  long long out_numel;
  float accumulate;    // WGRAD with splits == 1: out = accumulate * out + value
  int batch, H, W, C, OH, OW, K, KH, KW, sh, sw, pt, pl;
  int splits;
};

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

__device__ __forceinline__ bool tap_ok(const RowInfo& ri, int t) {
  const unsigned long long mk = ((unsigned long long)ri.mask_hi << 32) | ri.mask_lo;
  return (mk >> t) & 1ull;
}

// 4 consecutive floats at p (dword aligned); only the first nvalid (1..4) may be touched.
__device__ __forceinline__ f4 load_quad(const float* p, int nvalid) {
  if (nvalid >= 4) {
    const f4u v = *reinterpret_cast<const f4u*>(p);
    return f4{v.x, v.y, v.z, v.w};
  }
  f4 r = {0.f, 0.f, 0.f, 0.f};
  r.x = p[0];
  if (nvalid > 1) r.y = p[1];
  if (nvalid > 2) r.z = p[2];
  return r;
}

template <int MODE, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_mfma_f32(const ConvArgs p) {
  static_assert(WM * WN == 4, "4 waves per block");
  constexpr int TA = BM / (32 * WM), TB = BN / (32 * WN);
  constexpr int QA = BM / 32, QB = BN / 32;  // quads per thread per K-step
  constexpr int NROW = (MODE == MODE_WGRAD) ? BK : BM;

  __shared__ f4 As[8 * BM];
  __shared__ f4 Bs[8 * BN];
  __shared__ RowInfo rows[NROW];
  __shared__ int tapA[kMaxTaps];
  __shared__ int tapB[kMaxTaps];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- problem geometry (wave-uniform) -------------------------------------------------------
  // Cs: channels of the gathered tensor; Cp: Cs padded to a multiple of 4 (quad granularity per tap).
  int M, N, Kdim, Cs, Cp, ntaps;
  int ph = 0, pw = 0, i0 = 0, j0 = 0, nti = 1, ntj = 1, dp0 = 0, dq0 = 0, Hc = 0, Wc = 0;
  if constexpr (MODE == MODE_FWD) {
    Cs = p.C; Cp = (Cs + 3) & ~3; ntaps = p.KH * p.KW;
    M = p.batch * p.OH * p.OW; N = p.K; Kdim = ntaps * Cp;
  } else if constexpr (MODE == MODE_DGRAD) {
    const int cls = blockIdx.y;
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
  const int tiles_n = (N + BN - 1) / BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  if (m0 >= M) return;  // block-uniform: DGRAD classes smaller than class 0

  const int nk = (Kdim + BK - 1) / BK;
  const int per = (nk + p.splits - 1) / p.splits;
  const int ks_begin = blockIdx.z * per;
  const int ks_end = min(nk, ks_begin + per);

  // ---- tap tables ------------------------------------------------------------------------------
  if (tid < kMaxTaps && tid < ntaps) {
    const int t = tid;
    if constexpr (MODE == MODE_DGRAD) {
      const int ti = t / ntj, tj = t - ti * ntj;
      tapA[t] = -(ti * p.OW + tj) * p.K;
      tapB[t] = ((i0 + p.sh * ti) * p.KW + (j0 + p.sw * tj)) * p.C * p.K;
    } else {
      const int i = t / p.KW, j = t - i * p.KW;
      tapA[t] = (i * p.W + j) * p.C;
      tapB[t] = 0;
    }
  }

  // ---- per-row gather info -------------------------------------------------------------------
  auto fill_row_fwd = [&](int r /* global row (b,p,q) */, int limit) -> RowInfo {
    RowInfo ri; ri.base = 0; ri.mask_lo = 0; ri.mask_hi = 0; ri.out_off = 0;
    if (r < limit) {
      const int q = r % p.OW; const int t2 = r / p.OW; const int pp = t2 % p.OH; const int b = t2 / p.OH;
      const int y0 = pp * p.sh - p.pt, x0 = q * p.sw - p.pl;
      ri.base = ((b * p.H + y0) * p.W + x0) * p.C;
      const unsigned long long m = tap_mask(max(0, -y0), min(p.KH, p.H - y0), max(0, -x0), min(p.KW, p.W - x0), p.KW);
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
        const int w2 = m % Wc; const int t2 = m / Wc; const int h2 = t2 % Hc; const int b = t2 / Hc;
        const int y0 = h2 + dp0, x0 = w2 + dq0;  // dY coordinates of tap (0,0)
        ri.base = ((b * p.OH + y0) * p.OW + x0) * p.K;
        ri.out_off = ((b * p.H + h2 * p.sh + ph) * p.W + (w2 * p.sw + pw)) * p.C;
        // tap (ti,tj) reads dY[y0 - ti][x0 - tj]
        const unsigned long long mk = tap_mask(max(0, y0 - p.OH + 1), min(nti, y0 + 1), max(0, x0 - p.OW + 1), min(ntj, x0 + 1), ntj);
        ri.mask_lo = (unsigned)mk; ri.mask_hi = (unsigned)(mk >> 32);
      }
      rows[r] = ri;
    }
  }
  __syncthreads();

  // ---- loaders: two register stages ---------------------------------------------------------------
  // k-fast operands (A of FWD/DGRAD, B of DGRAD) are gathered one quad per (row, k/4); the others
  // (B of FWD, A and B of WGRAD) are contiguous along the tile column, so a thread loads a 4x4 block
  // (4 k-rows x float4 of columns) and transposes it in registers into 4 quads.
  constexpr int RA = (MODE == MODE_WGRAD) ? 4 : QA;
  constexpr int RB = (MODE == MODE_DGRAD) ? QB : 4;
  f4 ra[2][RA], rb[2][RB];
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // running (tap, channel) of this thread's next k-fast quad: kp = ks*32 + 4*(tid&7); no divisions in the loop
  int kt = 0, kc = 0;
  if constexpr (MODE != MODE_WGRAD) {
    const int kp = ks_begin * BK + 4 * (tid & 7);
    kt = kp / Cp; kc = kp - kt * Cp;
  }
  // transposed loaders: thread -> (column quad jn, k quad kq); active while kq < 8
  const int jb = tid % (BN / 4), kqb = tid / (BN / 4);
  const bool actb = kqb < 8;
  const bool nvec = (N & 3) == 0;     // dense rows are 16-byte aligned and quads never straddle N
  int bt = 0, bc = 0;                 // FWD: running (tap, channel) of the B rows kp = ks*32 + 4*kqb
  if constexpr (MODE == MODE_FWD) {
    const int kp = ks_begin * BK + 4 * (nvec ? kqb : 0);
    bt = kp / Cp; bc = kp - bt * Cp;
  }
  const int ja = tid % (BM / 4), kqa = tid / (BM / 4);
  const bool acta = kqa < 8;
  int wg_t = 0, wg_off = 0, wg_nvalid = 0;   // WGRAD: this thread's fixed (padded) output-row quad -> (tap, channel)
  if constexpr (MODE == MODE_WGRAD) {
    const int mp = m0 + 4 * ja;
    if (acta && mp < M) {
      wg_t = mp / Cp;
      const int c = mp - wg_t * Cp;
      wg_nvalid = Cs - c;            // >= 1
      wg_off = tapA[wg_t] + c;
    }
  }

  auto transpose_into = [&](f4 (&dst)[4], const f4 (&l)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = f4{l[0][i], l[1][i], l[2][i], l[3][i]};
  };

  auto load_tiles = [&](auto stage, int ks) {
    constexpr int ST = decltype(stage)::value;
    if constexpr (MODE == MODE_WGRAD) {
      if (acta) {
        f4 l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const RowInfo ri = rows[4 * kqa + e];
          l[e] = (wg_nvalid > 0 && tap_ok(ri, wg_t)) ? load_quad(p.gsrc + (ri.base + wg_off), wg_nvalid) : zero4;
        }
        transpose_into(ra[ST], l);
      }
    } else {
      // k-fast gather: 8 consecutive lanes walk 8 quads (128 contiguous bytes) of one gathered row
      const int rg = tid >> 3;
      const bool kv = kt < ntaps;
      const int t = kv ? kt : 0;
      const int nvalid = Cs - kc;  // >= 1
      const int aoff = tapA[t] + kc;
#pragma unroll
      for (int u = 0; u < QA; ++u) {
        const RowInfo ri = rows[rg + 32 * u];
        ra[ST][u] = (kv && tap_ok(ri, t)) ? load_quad(p.gsrc + (ri.base + aoff), nvalid) : zero4;
      }
      if constexpr (MODE == MODE_DGRAD) {  // B[k=(tap,o)][n=c] = W[tap][c][o], contiguous along o
        const int boff = tapB[t] + kc;
#pragma unroll
        for (int u = 0; u < QB; ++u) {
          const int n = n0 + rg + 32 * u;
          rb[ST][u] = (kv && n < N) ? load_quad(p.dense + (boff + (long long)n * p.K), nvalid) : zero4;
        }
      }
      kc += BK;
      while (kc >= Cp) { kc -= Cp; ++kt; }
    }
    if constexpr (MODE != MODE_DGRAD) {
      // dense operand: FWD W[(tap,c)][n] rows, WGRAD dY[(b,p,q)][n] rows
      if (nvec) {
        if (actb) {
          const int n = n0 + 4 * jb;
          f4 l[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            long long row;
            bool ok;
            if constexpr (MODE == MODE_FWD) { row = (long long)bt * Cs + bc + e; ok = bt < ntaps && bc + e < Cs; }
            else { row = (long long)ks * BK + 4 * kqb + e; ok = row < Kdim; }
            l[e] = (ok && n < N) ? *reinterpret_cast<const f4*>(p.dense + row * N + n) : zero4;
          }
          transpose_into(rb[ST], l);
        }
      } else {  // ragged N (25, 5, 3, 1 ...): 4 k-rows of one column per quad, lanes along n
        constexpr int stepB = 256 / BN;
        const int n = n0 + (tid % BN), kq0 = tid / BN;
#pragma unroll
        for (int u = 0; u < QB; ++u) {
          const int kq = kq0 + stepB * u;
          f4 v = zero4;
          if constexpr (MODE == MODE_FWD) {
            int t2 = bt, c2 = bc + 4 * kq;     // bt/bc track kp = ks*32 here (kqb term is 0 when !nvec)
            while (c2 >= Cp) { c2 -= Cp; ++t2; }
#pragma unroll
            for (int e = 0; e < 4; ++e)
              v[e] = (t2 < ntaps && c2 + e < Cs && n < N) ? p.dense[((long long)t2 * Cs + c2 + e) * N + n] : 0.f;
          } else {
            const long long r = (long long)ks * BK + 4 * kq;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (r + e < Kdim && n < N) ? p.dense[(r + e) * N + n] : 0.f;
          }
          rb[ST][u] = v;
        }
      }
      if constexpr (MODE == MODE_FWD) {
        bc += BK;
        while (bc >= Cp) { bc -= Cp; ++bt; }
      }
    }
  };

  // LDS column permutation: physical = P(col) ^ kq with P(32q + 4j + e) = 32q + 8e + (j ^ 4(e>>1)).
  // Conflict-free for (i) k-fast stores (8 lanes: one column, kq = 0..7), (ii) transposed stores (8 lanes:
  // columns 4j+e for 8 consecutive j) and (iii) the MFMA operand reads (32 consecutive columns, one kq).
  auto pcol = [](int col, int kq) {
    const int j = (col >> 2) & 7, e = col & 3;
    return ((col & ~31) | (e << 3) | (j ^ ((e >> 1) << 2))) ^ kq;
  };

  auto store_tiles = [&](auto stage) {
    constexpr int ST = decltype(stage)::value;
    if constexpr (MODE == MODE_WGRAD) {
      if (acta) {
#pragma unroll
        for (int i = 0; i < 4; ++i) As[kqa * BM + pcol(4 * ja + i, kqa)] = ra[ST][i];
      }
    } else {
      const int kq = tid & 7, rg = tid >> 3;
#pragma unroll
      for (int u = 0; u < QA; ++u) As[kq * BM + pcol(rg + 32 * u, kq)] = ra[ST][u];
      if constexpr (MODE == MODE_DGRAD) {
#pragma unroll
        for (int u = 0; u < QB; ++u) Bs[kq * BN + pcol(rg + 32 * u, kq)] = rb[ST][u];
      }
    }
    if constexpr (MODE != MODE_DGRAD) {
      if (nvec) {
        if (actb) {
#pragma unroll
          for (int i = 0; i < 4; ++i) Bs[kqb * BN + pcol(4 * jb + i, kqb)] = rb[ST][i];
        }
      } else {
        constexpr int stepB = 256 / BN;
        const int nb = tid % BN, kq0 = tid / BN;
#pragma unroll
        for (int u = 0; u < QB; ++u) { const int kq = kq0 + stepB * u; Bs[kq * BN + pcol(nb, kq)] = rb[ST][u]; }
      }
    }
  };

  // ---- main loop ----------------------------------------------------------------------------------
  f32x16 acc[TA][TB];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int wr = wave / WN, wc = wave - wr * WN;
  const int wm0 = wr * (BM / WM), wn0 = wc * (BN / WN);
  const int lrow = lane & 31, lk = lane >> 5;

  auto compute = [&]() {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int kq = 2 * t + lk;  // lane half h consumes quad 2t+h of BOTH tiles: same k on both sides
      f4 av[TA], bv[TB];
#pragma unroll
      for (int a = 0; a < TA; ++a) av[a] = As[kq * BM + pcol(wm0 + 32 * a + lrow, kq)];
#pragma unroll
      for (int b = 0; b < TB; ++b) bv[b] = Bs[kq * BN + pcol(wn0 + 32 * b + lrow, kq)];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
          for (int b = 0; b < TB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][e], bv[b][e], acc[a][b], 0, 0, 0);
    }
  };

  // WGRAD gathers along the reduction: its 32 row infos change every K-step and must be in LDS
  // (behind a barrier) before the loads of that step are issued.
  auto wgrad_rows = [&](int ks) {
    if constexpr (MODE == MODE_WGRAD) {
      if (tid < BK) rows[tid] = fill_row_fwd(ks * BK + tid, Kdim);
    }
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  if (ks_begin < ks_end) {
    if constexpr (MODE == MODE_WGRAD) { wgrad_rows(ks_begin); __syncthreads(); }
    load_tiles(S0{}, ks_begin);
    if (ks_begin + 1 < ks_end) {
      if constexpr (MODE == MODE_WGRAD) { __syncthreads(); wgrad_rows(ks_begin + 1); __syncthreads(); }
      load_tiles(S1{}, ks_begin + 1);
    }
    if constexpr (MODE == MODE_WGRAD) __syncthreads();
  }
  auto iterate = [&](auto stage, int ks) {
    store_tiles(stage);                  // K-step ks: registers -> LDS
    if (ks + 2 < ks_end) wgrad_rows(ks + 2);
    __syncthreads();
    if (ks + 2 < ks_end) load_tiles(stage, ks + 2);   // runs two K-steps ahead of the MFMAs
    compute();
    __syncthreads();
  };
  for (int ks = ks_begin; ks < ks_end; ks += 2) {
    iterate(S0{}, ks);
    if (ks + 1 < ks_end) iterate(S1{}, ks + 1);
  }

  // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) --------------
  float* outp = p.out + (p.splits > 1 ? (long long)blockIdx.z * p.out_numel : 0ll);
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int n = n0 + wn0 + 32 * b + lrow;
        const int m = m0 + row;
        if (m < M && n < N) {
          long long idx;
          bool ok = true;
          if constexpr (MODE == MODE_DGRAD) {
            idx = (long long)rows[row].out_off + n;
          } else if constexpr (MODE == MODE_WGRAD) {
            if (Cp == Cs) {
              idx = (long long)m * N + n;
            } else {  // drop the per-tap padding rows
              const int t = m / Cp, c = m - t * Cp;
              ok = c < Cs;
              idx = ((long long)t * Cs + c) * N + n;
            }
          } else {
            idx = (long long)m * N + n;
          }
          if (ok) {
            float v = acc[a][b][r];
            if constexpr (MODE == MODE_WGRAD) {
              if (p.splits == 1 && p.accumulate != 0.f) v += p.accumulate * outp[idx];
            }
            outp[idx] = v;
          }
        }
      }
}

// out[i] = accumulate * out[i] + sum_z slabs[z][i]   (fixed summation order)
__global__ __launch_bounds__(256) void splitk_reduce(const float* __restrict__ slabs, float* __restrict__ out,
                                                     long long numel, int splits, float accumulate) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const bool al = ((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 && (numel & 3) == 0;
  const long long n4 = al ? numel / 4 : 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f4 s = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < splits; ++z) s += reinterpret_cast<const f4*>(slabs + (long long)z * numel)[i];
    f4* o = reinterpret_cast<f4*>(out) + i;
    if (accumulate != 0.f) s += accumulate * *o;
    *o = s;
  }
  for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += slabs[(long long)z * numel + i];
    out[i] = (accumulate != 0.f ? accumulate * out[i] : 0.f) + s;
  }
}

// ---- host-side planning ------------------------------------------------------------------------------
struct Plan {
  int cfg;       // 0: 128x128, 1: 128x64, 2: 128x32, 3: 64x64
  int bm, bn;
  long long M, N;  // per class (class 0 = largest) GEMM extents (M padded per tap for WGRAD)
  int classes, nk, splits;
  long long tiles, out_numel;
};

int validate(const acg_conv_desc* d, const char* who) {
  ACG_REQUIRE(d != nullptr, ACG_ERR_INVALID_ARG, "%s: null descriptor", who);
  ACG_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->in_c > 0 && d->out_h > 0 && d->out_w > 0 && d->out_c > 0 &&
                  d->kh > 0 && d->kw > 0 && d->stride_h > 0 && d->stride_w > 0 && d->pad_top >= 0 && d->pad_left >= 0,
              ACG_ERR_INVALID_ARG, "%s: non-positive dimension in descriptor", who);
  ACG_REQUIRE(d->kh * d->kw <= kMaxTaps, ACG_ERR_UNSUPPORTED, "%s: %dx%d filter exceeds %d taps", who, d->kh, d->kw, kMaxTaps);
  ACG_REQUIRE(d->pad_top < d->kh && d->pad_left < d->kw, ACG_ERR_INVALID_ARG, "%s: padding not smaller than the filter", who);
  ACG_REQUIRE((d->out_h - 1) * d->stride_h - d->pad_top < d->in_h && (d->out_w - 1) * d->stride_w - d->pad_left < d->in_w,
              ACG_ERR_INVALID_ARG, "%s: output extent reads entirely outside the input", who);
  const long long lim = 2147483647ll;
  const long long nx = (long long)d->batch * d->in_h * d->in_w * d->in_c;
  const long long ny = (long long)d->batch * d->out_h * d->out_w * d->out_c;
  const long long nw = (long long)d->kh * d->kw * ((d->in_c + 3) & ~3) * d->out_c;
  ACG_REQUIRE(nx < lim && ny < lim && nw < lim, ACG_ERR_UNSUPPORTED, "%s: tensor exceeds 2^31 elements", who);
  return ACG_OK;
}

// Tuning hook (acg_debug_conv_plan): force a tile configuration / split-K factor; -1 = heuristic.
int g_force_cfg = -1, g_force_splits = -1;

Plan make_plan(const acg_conv_desc& d, int which) {
  Plan pl{};
  long long K;
  const long long cin_p = (d.in_c + 3) & ~3, cout_p = (d.out_c + 3) & ~3;
  if (which == ACG_CONV_FWD) {
    pl.M = (long long)d.batch * d.out_h * d.out_w; pl.N = d.out_c; K = (long long)d.kh * d.kw * cin_p; pl.classes = 1;
    pl.out_numel = pl.M * pl.N;
  } else if (which == ACG_CONV_DGRAD) {
    const int hc = (d.in_h + d.stride_h - 1) / d.stride_h, wc = (d.in_w + d.stride_w - 1) / d.stride_w;
    pl.M = (long long)d.batch * hc * wc; pl.N = d.in_c;
    K = (long long)((d.kh + d.stride_h - 1) / d.stride_h) * ((d.kw + d.stride_w - 1) / d.stride_w) * cout_p;
    pl.classes = d.stride_h * d.stride_w;
    pl.out_numel = (long long)d.batch * d.in_h * d.in_w * d.in_c;
  } else {
    pl.M = (long long)d.kh * d.kw * cin_p; pl.N = d.out_c; K = (long long)d.batch * d.out_h * d.out_w; pl.classes = 1;
    pl.out_numel = (long long)d.kh * d.kw * d.in_c * d.out_c;
  }
  pl.nk = (int)((K + BK - 1) / BK);
  if (pl.nk < 1) pl.nk = 1;
  auto tiles_for = [&](int bm, int bn) { return acg::ceil_div(pl.M, bm) * acg::ceil_div(pl.N, bn) * pl.classes; };
  static const int kBM[4] = {128, 128, 128, 64}, kBN[4] = {128, 64, 32, 64};
  if (pl.N <= 32) pl.cfg = 2;
  else if (pl.N <= 64) pl.cfg = 1;
  else if (tiles_for(128, 128) >= 192) pl.cfg = 0;
  else pl.cfg = 3;
  if (g_force_cfg >= 0 && g_force_cfg < 4) pl.cfg = g_force_cfg;
  pl.bm = kBM[pl.cfg]; pl.bn = kBN[pl.cfg];
  pl.tiles = tiles_for(pl.bm, pl.bn);
  long long s = acg::ceil_div(512, pl.tiles);
  s = std::min<long long>(s, std::max(1, pl.nk / 4));
  s = std::min<long long>(s, 64);
  if (g_force_splits >= 1) s = std::min<long long>(g_force_splits, pl.nk);
  pl.splits = (int)std::max<long long>(s, 1);
  return pl;
}

template <int MODE>
int launch(const Plan& pl, const ConvArgs& a, hipStream_t st) {
  const dim3 grid((unsigned)(acg::ceil_div(pl.M, pl.bm) * acg::ceil_div(pl.N, pl.bn)), (unsigned)pl.classes, (unsigned)pl.splits);
  switch (pl.cfg) {
    case 0: hipLaunchKernelGGL((conv_mfma_f32<MODE, 128, 128, 2, 2>), grid, dim3(256), 0, st, a); break;
    case 1: hipLaunchKernelGGL((conv_mfma_f32<MODE, 128, 64, 2, 2>), grid, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((conv_mfma_f32<MODE, 128, 32, 4, 1>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((conv_mfma_f32<MODE, 64, 64, 2, 2>), grid, dim3(256), 0, st, a); break;
  }
  return acg::check_launch("conv_mfma_f32");
}

int run(int which, const float* gsrc, const float* dense, float* out, float accumulate, const acg_conv_desc* d,
        void* ws, size_t ws_bytes, acg_stream_t stream, const char* who) {
  if (int rc = validate(d, who)) return rc;
  ACG_REQUIRE(gsrc && dense && out, ACG_ERR_INVALID_ARG, "%s: null tensor pointer", who);
  const Plan pl = make_plan(*d, which);
  const size_t need = pl.splits > 1 ? (size_t)pl.splits * (size_t)pl.out_numel * sizeof(float) : 0;
  ACG_REQUIRE(ws_bytes >= need && (need == 0 || ws != nullptr), ACG_ERR_WORKSPACE, "%s: workspace %zu bytes < required %zu", who, ws_bytes, need);
  ConvArgs a{};
  a.gsrc = gsrc; a.dense = dense; a.out = pl.splits > 1 ? (float*)ws : out; a.out_numel = pl.out_numel;
  a.accumulate = accumulate;
  a.batch = d->batch; a.H = d->in_h; a.W = d->in_w; a.C = d->in_c; a.OH = d->out_h; a.OW = d->out_w; a.K = d->out_c;
  a.KH = d->kh; a.KW = d->kw; a.sh = d->stride_h; a.sw = d->stride_w; a.pt = d->pad_top; a.pl = d->pad_left;
  a.splits = pl.splits;
  hipStream_t st = acg::to_stream(stream);
  int rc;
  if (which == ACG_CONV_FWD) rc = launch<MODE_FWD>(pl, a, st);
  else if (which == ACG_CONV_DGRAD) rc = launch<MODE_DGRAD>(pl, a, st);
  else rc = launch<MODE_WGRAD>(pl, a, st);
  if (rc) return rc;
  if (pl.splits > 1) {
    const int blocks = (int)std::min<long long>(acg::ceil_div(pl.out_numel, 1024), 2048);
    hipLaunchKernelGGL(splitk_reduce, dim3(std::max(blocks, 1)), dim3(256), 0, st, (const float*)ws, out, pl.out_numel, pl.splits,
                       which == ACG_CONV_WGRAD ? accumulate : 0.f);
    return acg::check_launch("splitk_reduce");
  }
  return ACG_OK;
}

}  // namespace

extern "C" {

int32_t acg_conv_desc_init(acg_conv_desc* d, int32_t batch, int32_t in_h, int32_t in_w, int32_t in_c, int32_t kh,
                           int32_t kw, int32_t out_c, int32_t stride, int32_t same) {
  ACG_REQUIRE(d && batch > 0 && in_h > 0 && in_w > 0 && in_c > 0 && kh > 0 && kw > 0 && out_c > 0 && stride > 0,
              ACG_ERR_INVALID_ARG, "conv_desc_init: non-positive dimension");
  d->batch = batch; d->in_h = in_h; d->in_w = in_w; d->in_c = in_c; d->out_c = out_c;
  d->kh = kh; d->kw = kw; d->stride_h = d->stride_w = stride;
  if (same) {  // TF 'SAME' (SURVEY A.1): out = ceil(in/s), pad_before = total // 2
    d->out_h = (in_h + stride - 1) / stride; d->out_w = (in_w + stride - 1) / stride;
    const int th = std::max((d->out_h - 1) * stride + kh - in_h, 0), tw = std::max((d->out_w - 1) * stride + kw - in_w, 0);
    d->pad_top = th / 2; d->pad_left = tw / 2;
  } else {
    ACG_REQUIRE(in_h >= kh && in_w >= kw, ACG_ERR_INVALID_ARG, "conv_desc_init: VALID kernel larger than input");
    d->out_h = (in_h - kh) / stride + 1; d->out_w = (in_w - kw) / stride + 1;
    d->pad_top = d->pad_left = 0;
  }
  return ACG_OK;
}

int32_t acg_debug_conv_plan(int32_t cfg, int32_t splits) {
  g_force_cfg = cfg; g_force_splits = splits;
  return ACG_OK;
}

size_t acg_conv2d_workspace_bytes(const acg_conv_desc* d, int32_t which, int32_t dtype) {
  (void)dtype;
  if (!d || validate(d, "conv2d_workspace_bytes") != ACG_OK || which < 0 || which > 2) return 0;
  const Plan pl = make_plan(*d, which);
  return pl.splits > 1 ? (size_t)pl.splits * (size_t)pl.out_numel * sizeof(float) : 0;
}

int32_t acg_conv2d_fwd(const void* x, const void* w, void* y, const acg_conv_desc* d, int32_t dtype, void* ws,
                       size_t wsb, acg_stream_t s) {
  ACG_REQUIRE_F32(dtype);
  return run(ACG_CONV_FWD, (const float*)x, (const float*)w, (float*)y, 0.f, d, ws, wsb, s, "conv2d_fwd");
}
int32_t acg_conv2d_dgrad(const void* dy, const void* w, void* dx, const acg_conv_desc* d, int32_t dtype, void* ws,
                         size_t wsb, acg_stream_t s) {
  ACG_REQUIRE_F32(dtype);
  return run(ACG_CONV_DGRAD, (const float*)dy, (const float*)w, (float*)dx, 0.f, d, ws, wsb, s, "conv2d_dgrad");
}
int32_t acg_conv2d_wgrad(const void* x, const void* dy, float* dw, float accumulate, const acg_conv_desc* d,
                         int32_t dtype, void* ws, size_t wsb, acg_stream_t s) {
  ACG_REQUIRE_F32(dtype);
  return run(ACG_CONV_WGRAD, (const float*)x, (const float*)dy, dw, accumulate, d, ws, wsb, s, "conv2d_wgrad");
}
int32_t acg_deconv2d_fwd(const void* x, const void* w, void* y, const acg_conv_desc* adj, int32_t dtype, void* ws,
                         size_t wsb, acg_stream_t s) {
  ACG_REQUIRE_F32(dtype);
  return run(ACG_CONV_DGRAD, (const float*)x, (const float*)w, (float*)y, 0.f, adj, ws, wsb, s, "deconv2d_fwd");
}
int32_t acg_deconv2d_dgrad(const void* dy, const void* w, void* dx, const acg_conv_desc* adj, int32_t dtype, void* ws,
                           size_t wsb, acg_stream_t s) {
  ACG_REQUIRE_F32(dtype);
  return run(ACG_CONV_FWD, (const float*)dy, (const float*)w, (float*)dx, 0.f, adj, ws, wsb, s, "deconv2d_dgrad");
}
int32_t acg_deconv2d_wgrad(const void* x, const void* dy, float* dw, float accumulate, const acg_conv_desc* adj,
                           int32_t dtype, void* ws, size_t wsb, acg_stream_t s) {
  ACG_REQUIRE_F32(dtype);
  // roles exchanged: the deconv's output gradient is the adjoint conv's input
  return run(ACG_CONV_WGRAD, (const float*)dy, (const float*)x, dw, accumulate, adj, ws, wsb, s, "deconv2d_wgrad");
}

}  // extern "C"
