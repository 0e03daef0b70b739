// NHWC implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), gfx950.
//
// One templated kernel serves the three contractions of a conv layer and, through the adjoint
// descriptor, of a TF conv2d_transpose layer (include/acgan_hip.h):
//   FWD    y [M=(b,p,q)][N=o]    = sum_{k=(tap,c)}   G[m][k]      * W[k][n]
//   DGRAD  dx[M=(b,h2,w2)][N=c]  = sum_{k=(tap,o)}   dY~[m][k]    * W^T[k][n]   per stride-parity class
//   WGRAD  dw[M=(tap,c)][N=o]    = sum_{k=(b,p,q)}   G[k][m]      * dY[k][n]
// G is the never-materialised im2col matrix: each block keeps, per gathered row, a base offset and a
// 64-bit mask of in-bounds filter taps in LDS, so the inner gather is one shift/and + one load.
// DGRAD runs as stride_h*stride_w parity classes (blockIdx.y), each a dense stride-1 correlation over
// dY with its own tap subset - no multiplies by structural zeros (5x5/s2: 9+6+6+4 = 25 taps in total).
//
// Tiling: 256 threads = 4 waves, block tile BM x BN x 32, wave tile (BM/WM) x (BN/WN) built from 32x32
// MFMA tiles; A and B tiles live in LDS k-major ([k][m], row pitch = tile + 1 words) so every
// ds_read_b32 / ds_write_b32 of a wave hits 32 distinct banks.  Global loads for K-step s+1 are issued
// into registers before the MFMAs of step s.  Small grids are filled by split-K over blockIdx.z into
// workspace slabs that a second kernel sums in fixed order (deterministic; no float atomics).
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };
constexpr int BK = 32;
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

template <int MODE, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_mfma_f32(const ConvArgs p) {
  static_assert(WM * WN == 4, "4 waves per block");
  constexpr int LDA = BM + 1, LDB = BN + 1;
  constexpr int TA = BM / (32 * WM), TB = BN / (32 * WN);
  constexpr int EA = BM * BK / 256, EB = BN * BK / 256;
  constexpr int NROW = (MODE == MODE_WGRAD) ? 2 * BK : BM;

  __shared__ float As[BK * LDA];
  __shared__ float Bs[BK * LDB];
  __shared__ RowInfo rows[NROW];
  __shared__ int tapA[kMaxTaps];
  __shared__ int tapB[kMaxTaps];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- problem geometry (wave-uniform) -------------------------------------------------------
  int M, N, Kdim, Cs, ntaps;
  int ph = 0, pw = 0, i0 = 0, j0 = 0, nti = 1, ntj = 1, dp0 = 0, dq0 = 0, Hc = 0, Wc = 0;
  if constexpr (MODE == MODE_FWD) {
    M = p.batch * p.OH * p.OW; N = p.K; Cs = p.C; ntaps = p.KH * p.KW; Kdim = ntaps * p.C;
  } else if constexpr (MODE == MODE_DGRAD) {
    const int cls = blockIdx.y;
    ph = cls / p.sw; pw = cls - ph * p.sw;
    Hc = ph < p.H ? (p.H - ph + p.sh - 1) / p.sh : 0;
    Wc = pw < p.W ? (p.W - pw + p.sw - 1) / p.sw : 0;
    M = p.batch * Hc * Wc; N = p.C; Cs = p.K;
    i0 = (ph + p.pt) % p.sh; j0 = (pw + p.pl) % p.sw;
    nti = i0 < p.KH ? (p.KH - i0 + p.sh - 1) / p.sh : 0;
    ntj = j0 < p.KW ? (p.KW - j0 + p.sw - 1) / p.sw : 0;
    dp0 = (ph + p.pt - i0) / p.sh; dq0 = (pw + p.pl - j0) / p.sw;
    ntaps = nti * ntj; Kdim = ntaps * p.K;
  } else {
    M = p.KH * p.KW * p.C; N = p.K; Cs = p.C; ntaps = p.KH * p.KW; Kdim = p.batch * p.OH * p.OW;
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
  } else {
    if (tid < BK && ks_begin < ks_end) rows[(ks_begin & 1) * BK + tid] = fill_row_fwd(ks_begin * BK + tid, Kdim);
  }
  __syncthreads();

  // ---- loaders ------------------------------------------------------------------------------------
  float ra[EA], rb[EB];
  // WGRAD: this thread's fixed output row m -> (tap, channel)
  int wg_t = 0, wg_off = 0; bool wg_valid = false;
  if constexpr (MODE == MODE_WGRAD) {
    const int m = m0 + (tid % BM);
    wg_valid = m < M;
    if (wg_valid) { wg_t = m / Cs; wg_off = tapA[wg_t] + (m - wg_t * Cs); }
  }

  auto load_tiles = [&](int ks) {
    if constexpr (MODE == MODE_WGRAD) {
      const RowInfo* rbuf = rows + (ks & 1) * BK;
      constexpr int stepA = 256 / BM;
      const int kk0 = tid / BM;
#pragma unroll
      for (int u = 0; u < EA; ++u) {
        const RowInfo ri = rbuf[kk0 + stepA * u];
        const unsigned long long mk = ((unsigned long long)ri.mask_hi << 32) | ri.mask_lo;
        const bool v = wg_valid && ((mk >> wg_t) & 1ull);
        ra[u] = v ? p.gsrc[ri.base + wg_off] : 0.f;
      }
      constexpr int stepB = 256 / BN;
      const int n = n0 + (tid % BN), kb0 = tid / BN;
#pragma unroll
      for (int u = 0; u < EB; ++u) {
        const int r = ks * BK + kb0 + stepB * u;
        rb[u] = (r < Kdim && n < N) ? p.dense[(long long)r * N + n] : 0.f;
      }
    } else {
      const int k = ks * BK + (tid & 31);
      const bool kv = k < Kdim;
      const int t = kv ? k / Cs : 0;
      const int c = k - t * Cs;
      const int aoff = tapA[t] + c;
      const int r0 = tid >> 5;
#pragma unroll
      for (int u = 0; u < EA; ++u) {
        const RowInfo ri = rows[r0 + 8 * u];
        const unsigned long long mk = ((unsigned long long)ri.mask_hi << 32) | ri.mask_lo;
        const bool v = kv && ((mk >> t) & 1ull);
        ra[u] = v ? p.gsrc[ri.base + aoff] : 0.f;
      }
      if constexpr (MODE == MODE_FWD) {
        constexpr int stepB = 256 / BN;
        const int n = n0 + (tid % BN), kb0 = tid / BN;
#pragma unroll
        for (int u = 0; u < EB; ++u) {
          const int kb = ks * BK + kb0 + stepB * u;
          rb[u] = (kb < Kdim && n < N) ? p.dense[(long long)kb * N + n] : 0.f;
        }
      } else {  // DGRAD: B[k=(tap,o)][n=c] = W[tap][c][o]
        const int boff = tapB[t] + c;  // c here is the dY channel o
#pragma unroll
        for (int u = 0; u < EB; ++u) {
          const int n = n0 + r0 + 8 * u;
          rb[u] = (kv && n < N) ? p.dense[boff + n * p.K] : 0.f;
        }
      }
    }
  };

  auto store_tiles = [&]() {
    if constexpr (MODE == MODE_WGRAD) {
      constexpr int stepA = 256 / BM, stepB = 256 / BN;
      const int mm = tid % BM, kk0 = tid / BM;
#pragma unroll
      for (int u = 0; u < EA; ++u) As[(kk0 + stepA * u) * LDA + mm] = ra[u];
      const int nn = tid % BN, kb0 = tid / BN;
#pragma unroll
      for (int u = 0; u < EB; ++u) Bs[(kb0 + stepB * u) * LDB + nn] = rb[u];
    } else {
      const int kk = tid & 31, r0 = tid >> 5;
#pragma unroll
      for (int u = 0; u < EA; ++u) As[kk * LDA + r0 + 8 * u] = ra[u];
      if constexpr (MODE == MODE_FWD) {
        constexpr int stepB = 256 / BN;
        const int nn = tid % BN, kb0 = tid / BN;
#pragma unroll
        for (int u = 0; u < EB; ++u) Bs[(kb0 + stepB * u) * LDB + nn] = rb[u];
      } else {
#pragma unroll
        for (int u = 0; u < EB; ++u) Bs[kk * LDB + r0 + 8 * u] = rb[u];
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

  if (ks_begin < ks_end) load_tiles(ks_begin);
  for (int ks = ks_begin; ks < ks_end; ++ks) {
    store_tiles();
    if constexpr (MODE == MODE_WGRAD) {
      if (tid < BK && ks + 1 < ks_end) rows[((ks + 1) & 1) * BK + tid] = fill_row_fwd((ks + 1) * BK + tid, Kdim);
    }
    __syncthreads();
    if (ks + 1 < ks_end) load_tiles(ks + 1);
#pragma unroll
    for (int k2 = 0; k2 < BK / 2; ++k2) {
      const int k = 2 * k2 + lk;
      float av[TA], bv[TB];
#pragma unroll
      for (int a = 0; a < TA; ++a) av[a] = As[k * LDA + wm0 + 32 * a + lrow];
#pragma unroll
      for (int b = 0; b < TB; ++b) bv[b] = Bs[k * LDB + wn0 + 32 * b + lrow];
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
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
          if constexpr (MODE == MODE_DGRAD) idx = (long long)rows[row].out_off + n;
          else idx = (long long)m * N + n;
          float v = acc[a][b][r];
          if constexpr (MODE == MODE_WGRAD) {
            if (p.splits == 1 && p.accumulate != 0.f) v += p.accumulate * outp[idx];
          }
          outp[idx] = v;
        }
      }
}

// out[i] = accumulate * out[i] + sum_z slabs[z][i]   (fixed summation order)
__global__ __launch_bounds__(256) void splitk_reduce(const float* __restrict__ slabs, float* __restrict__ out,
                                                     long long numel, int splits, float accumulate) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += slabs[(long long)z * numel + i];
    out[i] = (accumulate != 0.f ? accumulate * out[i] : 0.f) + s;
  }
}

// ---- host-side planning ------------------------------------------------------------------------------
struct Plan {
  int cfg;       // 0: 128x128, 1: 128x64, 2: 128x32, 3: 64x64
  int bm, bn;
  long long M, N;  // per class (class 0 = largest) GEMM extents
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
  const long long nw = (long long)d->kh * d->kw * d->in_c * d->out_c;
  ACG_REQUIRE(nx < lim && ny < lim && nw < lim, ACG_ERR_UNSUPPORTED, "%s: tensor exceeds 2^31 elements", who);
  return ACG_OK;
}

Plan make_plan(const acg_conv_desc& d, int which) {
  Plan pl{};
  long long K;
  if (which == ACG_CONV_FWD) {
    pl.M = (long long)d.batch * d.out_h * d.out_w; pl.N = d.out_c; K = (long long)d.kh * d.kw * d.in_c; pl.classes = 1;
    pl.out_numel = pl.M * pl.N;
  } else if (which == ACG_CONV_DGRAD) {
    const int hc = (d.in_h + d.stride_h - 1) / d.stride_h, wc = (d.in_w + d.stride_w - 1) / d.stride_w;
    pl.M = (long long)d.batch * hc * wc; pl.N = d.in_c;
    K = (long long)((d.kh + d.stride_h - 1) / d.stride_h) * ((d.kw + d.stride_w - 1) / d.stride_w) * d.out_c;
    pl.classes = d.stride_h * d.stride_w;
    pl.out_numel = (long long)d.batch * d.in_h * d.in_w * d.in_c;
  } else {
    pl.M = (long long)d.kh * d.kw * d.in_c; pl.N = d.out_c; K = (long long)d.batch * d.out_h * d.out_w; pl.classes = 1;
    pl.out_numel = pl.M * pl.N;
  }
  pl.nk = (int)((K + BK - 1) / BK);
  if (pl.nk < 1) pl.nk = 1;
  auto tiles_for = [&](int bm, int bn) { return acg::ceil_div(pl.M, bm) * acg::ceil_div(pl.N, bn) * pl.classes; };
  if (pl.N <= 32) { pl.cfg = 2; pl.bm = 128; pl.bn = 32; }
  else if (pl.N <= 64) { pl.cfg = 1; pl.bm = 128; pl.bn = 64; }
  else if (tiles_for(128, 128) >= 192) { pl.cfg = 0; pl.bm = 128; pl.bn = 128; }
  else { pl.cfg = 3; pl.bm = 64; pl.bn = 64; }
  pl.tiles = tiles_for(pl.bm, pl.bn);
  long long s = acg::ceil_div(512, pl.tiles);
  s = std::min<long long>(s, std::max(1, pl.nk / 4));
  s = std::min<long long>(s, 64);
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
    const int blocks = (int)std::min<long long>(acg::ceil_div(pl.out_numel, 256), 2048);
    hipLaunchKernelGGL(splitk_reduce, dim3(blocks), dim3(256), 0, st, (const float*)ws, out, pl.out_numel, pl.splits,
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
