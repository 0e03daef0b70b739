// Direct (non-MFMA) kernels for the few-MFLOP convolutions of the models with at most 8 output channels: d/conv6 (2x2x512
// -> 1) and g/sconv5 (4x4x16 -> 5, VALID) - models.py:44-51,87-88 (the planner decides: make_plan in conv_f32.hip).  On the tiled MFMA path these are 1-8 output tiles: the planner
// splits them 2-16 ways over K to occupy the chip and a second launch sums the slabs - 8-12 us per contraction for ~0 FLOP
// (profiles/r2/h_conv_table_f32_config2.txt: 72 us of the 2.5 ms step).  Here every contraction is ONE launch with no
// workspace:
//   FWD    a wave per output pixel: lanes split the flattened (tap, channel) reduction, 64 independent gathers per pass,
//          N <= 16 accumulators per lane, one shuffle reduction per output channel;
//   DGRAD  a thread per input element (pixel, channel): the taps that reach it, times the N output channels;
//   WGRAD  a wave per (tap, channel) filter row: lanes split the pixels (dy rows contiguous across the wave), N
//          accumulators per lane, one shuffle reduction per output channel.
// (First cut, measured: lanes over channels inside a serial tap loop and 32 filter rows x 8 pixel groups per block left
// g/sconv4's weight gradient on 9 blocks walking 64 dependent-latency iterations each: 72 us.  Parallelism first.)
// Round 1's direct kernel (one output per 8-lane group, a serial chain of dependent loads: 73-129 us) was the wrong shape,
// not the wrong idea.  float32 only: the bf16 tiled path does not split these layers and already runs them in ~5 us, and
// with 16 output channels (g/sconv4) a lane's strided loads make the launch latency-bound (profiles/r3/d_direct_small_convs.txt).
#include "conv_bf16_kernel.h"

namespace acgconv {

namespace {


// y[b,p,q,n] = sum_{tap in bounds, c} x[b, p*s - pt + i, q*s - pl + j, c] * W[tap][c][n]
// A BLOCK per output pixel: its 256 threads split the flattened (tap, channel) reduction, U steps at a time with all
// loads of the U steps issued before the first multiply - these layers leave one or two waves per CU, so nothing but the
// wave's own independent loads hides a memory round trip (a wave per pixel walking 32 dependent passes: 9 us for d/conv6).
template <int NB>
__global__ __launch_bounds__(256) void direct_fwd(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                  const ConvArgs p) {
  constexpr int U = NB == 1 ? 8 : (NB == 8 ? 4 : 2);
  __shared__ float red[4 * NB];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m = blockIdx.x;
  const int xp = p.Cx, yp = p.Ky;
  const int q = m % p.OW, t2 = m / p.OW, pp = t2 % p.OH, b = t2 / p.OH;
  const int y0 = pp * p.sh - p.pt, x0 = q * p.sw - p.pl;
  float acc[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) acc[n] = 0.f;
  const int Ktot = p.KH * p.KW * p.C;
  const float* const xb = x + (long long)b * p.H * p.W * xp;
  for (int k0 = tid; k0 < Ktot; k0 += 256 * U) {
    float xv[U], wv[U][NB];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 256 * u;
      const bool live = k < Ktot;
      const int kk = live ? k : 0;
      const int tap = kk / p.C, c = kk - tap * p.C, i = tap / p.KW, j = tap - i * p.KW;
      const int yy = y0 + i, xx = x0 + j;
      const bool in = live && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
      xv[u] = in ? xb[(yy * p.W + xx) * xp + c] : 0.f;
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        wv[u][n] = (n < p.K) ? w[(long long)kk * p.K + n] : 0.f;          // W[tap][c][n] with (tap * C + c) = k
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[n] = fmaf(xv[u], wv[u][n], acc[n]);
  }
#pragma unroll
  for (int n = 0; n < NB; ++n) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[n] += __shfl_xor(acc[n], off, 64);
  }
  if (lane == 0) {
#pragma unroll
    for (int n = 0; n < NB; ++n) red[wave * NB + n] = acc[n];
  }
  __syncthreads();
  if (tid < NB && tid < p.K) {
    const float v = red[tid] + red[NB + tid] + red[2 * NB + tid] + red[3 * NB + tid];
    y[(long long)m * yp + tid] = v;
  }
}

// dx[b,y,x,c] = sum_{taps (i,j) with y + pt - i = p*s, x + pl - j = q*s, (p,q) in range} sum_n dy[b,p,q,n] * W[tap][c][n]
template <int NB>
__device__ __forceinline__ void direct_dgrad_body(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, const ConvArgs& p,
                                                  int block, int nblocks) {
  const int xp = p.Cx, yp = p.Ky;
  const long long total = (long long)p.batch * p.H * p.W * p.C;
  for (long long e = (long long)block * 256 + threadIdx.x; e < total; e += (long long)nblocks * 256) {
    const int c = (int)(e % p.C);
    const long long pix = e / p.C;
    const int xx = (int)(pix % p.W), yy = (int)((pix / p.W) % p.H), b = (int)(pix / ((long long)p.W * p.H));
    float acc = 0.f;
    // the taps that reach (yy, xx): i = i0 + sh * t with i0 = (yy + pt) mod sh, output row pp = pp0 - t (same in x) - two
    // divisions per element instead of four per tap
    const int i0 = (yy + p.pt) % p.sh, pp0 = (yy + p.pt - i0) / p.sh, j0 = (xx + p.pl) % p.sw, q0 = (xx + p.pl - j0) / p.sw;
    for (int i = i0, pp = pp0; i < p.KH && pp >= 0; i += p.sh, --pp) {
      if (pp >= p.OH) continue;
      for (int j = j0, q = q0; j < p.KW && q >= 0; j += p.sw, --q) {
        if (q >= p.OW) continue;
        const float* dr = dy + ((long long)(b * p.OH + pp) * p.OW + q) * yp;
        const int tap = i * p.KW + j;
        float dv[NB], wv[NB];                       // both rows requested before the first multiply
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          dv[n] = n < p.K ? dr[n] : 0.f;
          wv[n] = n < p.K ? w[((long long)tap * p.C + c) * p.K + n] : 0.f;
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) acc = fmaf(dv[n], wv[n], acc);
      }
    }
    dx[pix * xp + c] = acc;
  }
}

// dw[tap][c][n] = accumulate * dw + sum_{b,p,q in bounds} x[b, p*s - pt + i, q*s - pl + j, c] * dy[b,p,q,n]
// a wave per filter row (tap, c); `block` = 4 rows
template <int NB>
__device__ __forceinline__ void direct_wgrad_body(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, const ConvArgs& p,
                                                  int block) {
  const int xp = p.Cx, yp = p.Ky;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = block * 4 + wave, nrows = p.KH * p.KW * p.C;
  if (row >= nrows) return;
  const int M = p.batch * p.OH * p.OW;
  const int tap = row / p.C, c = row - tap * p.C, i = tap / p.KW, j = tap - i * p.KW;
  float acc[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n) acc[n] = 0.f;
  constexpr int U = NB == 1 ? 8 : (NB == 8 ? 4 : 2);
  for (int m0 = lane; m0 < M; m0 += 64 * U) {
    float xv[U], dv[U][NB];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = m0 + 64 * u;
      const bool live = m < M;
      const int mm = live ? m : 0;
      const int q = mm % p.OW, t2 = mm / p.OW, pp = t2 % p.OH, b = t2 / p.OH;
      const int yy = pp * p.sh - p.pt + i, xx = q * p.sw - p.pl + j;
      const bool in = live && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
      xv[u] = in ? x[((long long)(b * p.H + yy) * p.W + xx) * xp + c] : 0.f;
      const float* dr = dy + (long long)mm * yp;
#pragma unroll
      for (int n = 0; n < NB; ++n) dv[u][n] = n < p.K ? dr[n] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[n] = fmaf(xv[u], dv[u][n], acc[n]);
  }
#pragma unroll
  for (int n = 0; n < NB; ++n) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[n] += __shfl_xor(acc[n], off, 64);
  }
  if (lane == 0) {
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      if (n < p.K) {
        float* o = dw + (long long)row * p.K + n;
        *o = (p.accumulate != 0.f ? p.accumulate * *o : 0.f) + acc[n];
      }
    }
  }
}

template <int NB>
__global__ __launch_bounds__(256) void direct_dgrad(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, const ConvArgs p) {
  direct_dgrad_body<NB>(dy, w, dx, p, (int)blockIdx.x, (int)gridDim.x);
}
template <int NB>
__global__ __launch_bounds__(256) void direct_wgrad(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, const ConvArgs p) {
  direct_wgrad_body<NB>(x, dy, dw, p, (int)blockIdx.x);
}
// a layer's input gradient and weight gradient out of one grid: blocks [0, nA) run the input gradient
template <int NB>
__global__ __launch_bounds__(256) void direct_pair(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, const ConvArgs a,
                                                   const float* __restrict__ x, float* __restrict__ dw, const ConvArgs b, int nA) {
  if ((int)blockIdx.x < nA) direct_dgrad_body<NB>(dy, w, dx, a, (int)blockIdx.x, nA);
  else direct_wgrad_body<NB>(x, dy, dw, b, (int)blockIdx.x - nA);
}

int dgrad_blocks(const ConvArgs& a) {
  const long long total = (long long)a.batch * a.H * a.W * a.C;
  return (int)std::max<long long>(1, std::min<long long>(acg::ceil_div(total, 256), 2048));
}
int wgrad_blocks(const ConvArgs& a) { return (int)acg::ceil_div((long long)a.KH * a.KW * a.C, 4); }

}  // namespace

// One contraction on the direct kernels.  `a` as prepared for the tiled path (gsrc / dense / out as that mode defines them).
int launch_direct(int which, const ConvArgs& a, hipStream_t st) {
  const bool one = a.K <= 1;
  if (which == ACG_CONV_FWD) {
    const dim3 grid((unsigned)((long long)a.batch * a.OH * a.OW));
    if (one) ACG_LAUNCH((direct_fwd<1>), grid, dim3(256), 0, st, a.gsrc, a.dense, a.out, a);
    else ACG_LAUNCH((direct_fwd<8>), grid, dim3(256), 0, st, a.gsrc, a.dense, a.out, a);
    return acg::check_launch("direct_fwd");
  }
  if (which == ACG_CONV_DGRAD) {
    const dim3 grid((unsigned)dgrad_blocks(a));
    if (one) ACG_LAUNCH((direct_dgrad<1>), grid, dim3(256), 0, st, a.gsrc, a.dense, a.out, a);
    else ACG_LAUNCH((direct_dgrad<8>), grid, dim3(256), 0, st, a.gsrc, a.dense, a.out, a);
    return acg::check_launch("direct_dgrad");
  }
  const dim3 grid((unsigned)wgrad_blocks(a));
  if (one) ACG_LAUNCH((direct_wgrad<1>), grid, dim3(256), 0, st, a.gsrc, a.dense, a.out, a);
  else ACG_LAUNCH((direct_wgrad<8>), grid, dim3(256), 0, st, a.gsrc, a.dense, a.out, a);
  return acg::check_launch("direct_wgrad");
}

// input gradient (a: DGRAD) + weight gradient (b: WGRAD) of one conv layer in one launch
int launch_direct_pair(const ConvArgs& a, const ConvArgs& b, hipStream_t st) {
  const int nA = dgrad_blocks(a);
  const dim3 grid((unsigned)(nA + wgrad_blocks(b)));
  if (a.K <= 1) ACG_LAUNCH((direct_pair<1>), grid, dim3(256), 0, st, a.gsrc, a.dense, a.out, a, b.gsrc, b.out, b, nA);
  else ACG_LAUNCH((direct_pair<8>), grid, dim3(256), 0, st, a.gsrc, a.dense, a.out, a, b.gsrc, b.out, b, nA);
  return acg::check_launch("direct_pair");
}

}  // namespace acgconv
