// Loss heads of train.py:72-85 / ops.py:19-50,100-120.  Each kernel produces the scalar value AND the
// gradient w.r.t. its input in one pass over the data (the gradients of these losses do not depend on
// anything upstream), so the backward pass of the networks starts from ready-made buffers.
// Reductions are two-stage (per-block partials in the workspace, summed in fixed order in fp64 by a
// one-block finalize) - deterministic, no float atomics.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

constexpr int kMaxBlocks = 2048;
constexpr size_t kPartialBytes = (size_t)kMaxBlocks * 2 * sizeof(float);

int blocks_for(long long n) {
  long long b = acg::ceil_div(n, 256);
  if (b > kMaxBlocks) b = kMaxBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

// L1 sum + gradient-difference loss and d(w_l1*L1 + w_gdl*GDL)/dgen.
//   gdx(e) = gen[x+1]-gen[x] (0 beyond the right edge), gdy(e) = gen[y]-gen[y+1] (0 beyond the bottom edge)
//   GDL = sum | |tdx|-|gdx| | + | |tdy|-|gdy| | ;   h(t,g) = -sgn(|t|-|g|)*sgn(g) = d| |t|-|g| |/dg
//   dGDL/dgen[y,x] = -hx[y,x] + hx[y,x-1] + hy[y,x] - hy[y-1,x]
__device__ __forceinline__ float hfun(float t, float g) { return -acg::sgnf(fabsf(t) - fabsf(g)) * acg::sgnf(g); }

// One thread per pixel (all C <= 4 channels): the pixel coordinates cost two shifts / masks for power-of-two frames (64,
// 128) instead of three integer divisions per ELEMENT, and the ten neighbour loads of a channel sit at compile-time
// offsets.  SUMS = false: gradient only - the training step fetches no loss value, so the block reductions, the partial
// stores and the finalize launch are skipped (round 2: 9.4 us for this 4.7 MB pass, two launches).
template <int CC, bool SUMS>
__global__ __launch_bounds__(256) void frame_loss_k(const float* __restrict__ gen, const float* __restrict__ gt,
                                                    float* __restrict__ part, float* __restrict__ dgen, long long npix,
                                                    int H, int W, int Crt, int lw, int lh, float w_l1, float w_gdl) {
  __shared__ float scratch[16];
  const int C = CC ? CC : Crt;
  const long long stride = (long long)gridDim.x * 256;
  const int rowp = W * C;
  float l1 = 0.f, gdl = 0.f;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += stride) {
    int x, y;
    if (lw >= 0) { x = (int)(p & (W - 1)); y = (int)((p >> lw) & (H - 1)); }
    else { x = (int)(p % W); y = (int)((p / W) % H); }
    const long long i0 = p * C;
    const bool xr = x + 1 < W, yd = y + 1 < H, xl = x > 0, yu = y > 0;
#pragma unroll
    for (int c = 0; c < (CC ? CC : 4); ++c) {
      if (c < C) {
        const long long i = i0 + c;
        const float g = gen[i], t = gt[i];
        const float gr = xr ? gen[i + C] : 0.f, tr = xr ? gt[i + C] : 0.f;
        const float gd = yd ? gen[i + rowp] : 0.f, td = yd ? gt[i + rowp] : 0.f;
        const float e = g - t;
        const float gdx = gr - g, tdx = tr - t, gdy = g - gd, tdy = t - td;
        if constexpr (SUMS) {
          l1 += fabsf(e);
          gdl += fabsf(fabsf(tdx) - fabsf(gdx)) + fabsf(fabsf(tdy) - fabsf(gdy));
        }
        if (dgen) {
          float d = -hfun(tdx, gdx) + hfun(tdy, gdy);
          if (xl) d += hfun(t - gt[i - C], g - gen[i - C]);
          if (yu) d -= hfun(gt[i - rowp] - t, gen[i - rowp] - g);
          dgen[i] = w_l1 * acg::sgnf(e) + w_gdl * d;
        }
      }
    }
  }
  if constexpr (SUMS) {
    const float s1 = acg::block_sum(l1, scratch);
    const float s2 = acg::block_sum(gdl, scratch);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = s1; part[2 * blockIdx.x + 1] = s2; }
  }
}

// out[j] = sum_b part[b*width + j]  (optionally mapped through the PSNR formula)
__global__ __launch_bounds__(256) void finalize_k(const float* __restrict__ part, float* __restrict__ out, int nblk,
                                                  int width, int psnr, double count) {
  __shared__ double scratch[16];
  for (int j = 0; j < width; ++j) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) s += part[b * width + j];
    s = acg::block_sum(s, scratch);
    if (threadIdx.x == 0) out[j] = psnr ? (float)(10.0 * log(1.0 / (s / count)) / log(10.0)) : (float)s;
  }
}

__global__ __launch_bounds__(256) void sqdiff_partial_k(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ part, long long n) {
  __shared__ float scratch[16];
  const long long stride = (long long)gridDim.x * 256;
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { const float e = a[i] - b[i]; s += e * e; }
  s = acg::block_sum(s, scratch);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// ---- small heads: one 1024-thread block, n <= 65536 ---------------------------------------------------
// gss: sum of squares of the GLOBAL batch (data parallel, all-reduced by the caller) to take the norm from instead of
// this rank's own; sumsq_only: out[0] = this rank's sum of squares, nothing else
__global__ __launch_bounds__(1024) void l2norm_loss_k(const float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ out, float* __restrict__ dp, int n, float scale,
                                                      const float* __restrict__ gss, int sumsq_only) {
  __shared__ double scratch[16];
  __shared__ double s_norm;
  double ss = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) { const double e = (double)p[i] - (double)g[i]; ss += e * e; }
  ss = acg::block_sum(ss, scratch);
  if (sumsq_only) {
    if (threadIdx.x == 0) out[0] = (float)ss;
    return;
  }
  if (threadIdx.x == 0) { s_norm = sqrt(gss ? (double)gss[0] : ss); out[0] = (float)s_norm; }
  __syncthreads();
  if (dp) {
    const double nrm = s_norm;
    for (int i = threadIdx.x; i < n; i += 1024)
      dp[i] = nrm > 0.0 ? (float)((double)scale * ((double)p[i] - (double)g[i]) / nrm) : 0.f;
  }
}

__global__ __launch_bounds__(1024) void sigmoid_ce_loss_k(const float* __restrict__ x, float label, float* __restrict__ out,
                                                          float* __restrict__ dx, int n, float scale) {
  __shared__ double scratch[16];
  double s = 0.0;
  const float gs = scale / (float)n;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float v = x[i];
    const float ex = expf(-fabsf(v));
    s += (double)(fmaxf(v, 0.f) - v * label + log1pf(ex));
    if (dx) {
      const float sig = v >= 0.f ? 1.f / (1.f + ex) : ex / (1.f + ex);
      dx[i] = gs * (sig - label);
    }
  }
  s = acg::block_sum(s, scratch);
  if (threadIdx.x == 0) out[0] = (float)(s / (double)n);
}

__global__ __launch_bounds__(1024) void mean_loss_k(const float* __restrict__ x, float* __restrict__ out,
                                                    float* __restrict__ dx, int n, float scale) {
  __shared__ double scratch[16];
  double s = 0.0;
  const float gs = scale / (float)n;
  for (int i = threadIdx.x; i < n; i += 1024) { s += (double)x[i]; if (dx) dx[i] = gs; }
  s = acg::block_sum(s, scratch);
  if (threadIdx.x == 0) out[0] = (float)(s / (double)n);
}

__global__ void scalar_combine_k(float* out, const float* i0, float w0, const float* i1, float w1, const float* i2,
                                 float w2, const float* i3, float w3) {
  double v = 0.0;
  if (i0) v += (double)w0 * (double)i0[0];
  if (i1) v += (double)w1 * (double)i1[0];
  if (i2) v += (double)w2 * (double)i2[0];
  if (i3) v += (double)w3 * (double)i3[0];
  out[0] = (float)v;
}

}  // namespace

extern "C" {

size_t acg_frame_loss_workspace_bytes(int64_t n) { (void)n; return kPartialBytes; }

int32_t acg_frame_loss(const void* gen, const void* gt, float* out2, void* dgen, int32_t B, int32_t H, int32_t W, int32_t C,
                       float w_l1, float w_gdl, int32_t dtype, void* ws, size_t wsb, acg_stream_t stream) {
  ACG_REQUIRE_F32(dtype);
  ACG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C <= 4, ACG_ERR_INVALID_ARG, "frame_loss: non-positive size / more than 4 channels");
  ACG_REQUIRE(gen && gt && (out2 || dgen), ACG_ERR_INVALID_ARG, "frame_loss: null pointer");
  ACG_REQUIRE(!out2 || (ws && wsb >= kPartialBytes), ACG_ERR_WORKSPACE, "frame_loss: workspace too small");
  const long long npix = (long long)B * H * W;
  const int nblk = blocks_for(npix);
  const bool pow2 = (W & (W - 1)) == 0 && (H & (H - 1)) == 0;
  int lw = -1, lh = -1;
  if (pow2) { lw = 0; while ((1 << lw) < W) ++lw; lh = 0; while ((1 << lh) < H) ++lh; }
  hipStream_t st = acg::to_stream(stream);
#define ACG_FRAME_LOSS(CV, SV) ACG_LAUNCH((frame_loss_k<CV, SV>), dim3(nblk), dim3(256), 0, st, (const float*)gen, (const float*)gt, (float*)ws, (float*)dgen, npix, H, W, C, lw, lh, w_l1, w_gdl)
  if (out2) { if (C == 3) ACG_FRAME_LOSS(3, true); else ACG_FRAME_LOSS(0, true); }
  else { if (C == 3) ACG_FRAME_LOSS(3, false); else ACG_FRAME_LOSS(0, false); }
#undef ACG_FRAME_LOSS
  if (int rc = acg::check_launch("frame_loss")) return rc;
  if (!out2) return ACG_OK;          // gradient only: no values, no finalize launch
  ACG_LAUNCH(finalize_k, dim3(1), dim3(256), 0, st, (const float*)ws, out2, nblk, 2, 0, 1.0);
  return acg::check_launch("frame_loss finalize");
}

int32_t acg_l2norm_loss(const float* pred, const float* gt, float* out, float* dpred, int64_t n, float scale, acg_stream_t stream) {
  ACG_REQUIRE(n > 0 && n <= 65536, ACG_ERR_INVALID_ARG, "l2norm_loss: n outside 1..65536");
  ACG_REQUIRE(pred && gt && out, ACG_ERR_INVALID_ARG, "l2norm_loss: null pointer");
  ACG_LAUNCH(l2norm_loss_k, dim3(1), dim3(1024), 0, acg::to_stream(stream), pred, gt, out, dpred, (int)n, scale, (const float*)nullptr, 0);
  return acg::check_launch("l2norm_loss");
}

int32_t acg_sumsq_diff(const float* pred, const float* gt, float* out, int64_t n, acg_stream_t stream) {
  ACG_REQUIRE(n > 0 && n <= 65536, ACG_ERR_INVALID_ARG, "sumsq_diff: n outside 1..65536");
  ACG_REQUIRE(pred && gt && out, ACG_ERR_INVALID_ARG, "sumsq_diff: null pointer");
  ACG_LAUNCH(l2norm_loss_k, dim3(1), dim3(1024), 0, acg::to_stream(stream), pred, gt, out, (float*)nullptr, (int)n, 0.f, (const float*)nullptr, 1);
  return acg::check_launch("sumsq_diff");
}

int32_t acg_l2norm_loss_global(const float* pred, const float* gt, const float* global_sumsq, float* out, float* dpred, int64_t n,
                               float scale, acg_stream_t stream) {
  ACG_REQUIRE(n > 0 && n <= 65536, ACG_ERR_INVALID_ARG, "l2norm_loss_global: n outside 1..65536");
  ACG_REQUIRE(pred && gt && global_sumsq && out, ACG_ERR_INVALID_ARG, "l2norm_loss_global: null pointer");
  ACG_LAUNCH(l2norm_loss_k, dim3(1), dim3(1024), 0, acg::to_stream(stream), pred, gt, out, dpred, (int)n, scale, global_sumsq, 0);
  return acg::check_launch("l2norm_loss_global");
}

int32_t acg_sigmoid_ce_loss(const float* logits, float label, float* out, float* dlogits, int64_t n, float scale, acg_stream_t stream) {
  ACG_REQUIRE(n > 0 && n <= 65536, ACG_ERR_INVALID_ARG, "sigmoid_ce_loss: n outside 1..65536");
  ACG_REQUIRE(logits && out, ACG_ERR_INVALID_ARG, "sigmoid_ce_loss: null pointer");
  ACG_LAUNCH(sigmoid_ce_loss_k, dim3(1), dim3(1024), 0, acg::to_stream(stream), logits, label, out, dlogits, (int)n, scale);
  return acg::check_launch("sigmoid_ce_loss");
}

int32_t acg_mean_loss(const float* x, float* out, float* dx, int64_t n, float scale, acg_stream_t stream) {
  ACG_REQUIRE(n > 0 && n <= 65536, ACG_ERR_INVALID_ARG, "mean_loss: n outside 1..65536");
  ACG_REQUIRE(x && out, ACG_ERR_INVALID_ARG, "mean_loss: null pointer");
  ACG_LAUNCH(mean_loss_k, dim3(1), dim3(1024), 0, acg::to_stream(stream), x, out, dx, (int)n, scale);
  return acg::check_launch("mean_loss");
}

int32_t acg_psnr(const void* a, const void* b, float* out, int64_t n, int32_t dtype, void* ws, size_t wsb, acg_stream_t stream) {
  ACG_REQUIRE_F32(dtype);
  ACG_REQUIRE(n > 0 && a && b && out, ACG_ERR_INVALID_ARG, "psnr: bad argument");
  ACG_REQUIRE(ws && wsb >= kPartialBytes, ACG_ERR_WORKSPACE, "psnr: workspace too small");
  const int nblk = blocks_for(n);
  hipStream_t st = acg::to_stream(stream);
  ACG_LAUNCH(sqdiff_partial_k, dim3(nblk), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)ws, (long long)n);
  if (int rc = acg::check_launch("psnr partial")) return rc;
  ACG_LAUNCH(finalize_k, dim3(1), dim3(256), 0, st, (const float*)ws, out, nblk, 1, 1, (double)n);
  return acg::check_launch("psnr finalize");
}

int32_t acg_scalar_combine(float* out, const float* i0, float w0, const float* i1, float w1, const float* i2, float w2,
                           const float* i3, float w3, acg_stream_t stream) {
  ACG_REQUIRE(out, ACG_ERR_INVALID_ARG, "scalar_combine: null output");
  ACG_LAUNCH(scalar_combine_k, dim3(1), dim3(1), 0, acg::to_stream(stream), out, i0, w0, i1, w1, i2, w2, i3, w3);
  return acg::check_launch("scalar_combine");
}

}  // extern "C"
