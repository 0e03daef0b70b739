// Fused BatchNorm(train, no gamma) + activation, and bias + activation, for [rows, C] NHWC views.
//
// HBM-bound; every pass walks memory in address order: a 256-thread block is laid out as
// (256 / Cb) rows x Cb channels (Cb = min(C, 256)), so consecutive lanes touch consecutive floats and
// a wave never needs a per-element modulo.  Per-channel statistics: pass 1 leaves per-block partial
// sums (shifted by the group's first row, so E[x^2]-E[x]^2 cannot cancel catastrophically) in the
// workspace; pass 2 re-derives mean / rstd from the partials in fp64 inside every block (a few KB from
// L2) and applies normalise + activation - two launches, no atomics, deterministic.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

constexpr int kMaxPartialBlocks = 128;
constexpr int kMaxC = 1024;

struct ColMap {
  int Cb, RPP, cl, rsub, nchunk;
  bool active;
};
__device__ __forceinline__ ColMap col_map(int C) {
  ColMap m;
  m.Cb = C < 256 ? C : 256;
  m.RPP = 256 / m.Cb;
  m.cl = threadIdx.x % m.Cb;
  m.rsub = threadIdx.x / m.Cb;
  m.active = m.rsub < m.RPP;
  m.nchunk = (C + m.Cb - 1) / m.Cb;
  return m;
}

// Reduce two per-thread values over the rsub dimension; result valid where rsub == 0.
__device__ __forceinline__ void reduce_rsub(const ColMap& m, float& a, float& b, float* sh /* 512 floats */) {
  __syncthreads();
  sh[threadIdx.x] = a;
  sh[256 + threadIdx.x] = b;
  __syncthreads();
  if (m.active && m.rsub == 0) {
    float sa = 0.f, sb = 0.f;
    for (int r = 0; r < m.RPP; ++r) { sa += sh[r * m.Cb + m.cl]; sb += sh[256 + r * m.Cb + m.cl]; }
    a = sa; b = sb;
  }
}

// ---- BN forward ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_stats_partial(const float* __restrict__ x, float* __restrict__ part,
                                                        long long R, int C, int nblk) {
  __shared__ float sh[512];
  const ColMap m = col_map(C);
  const int g = blockIdx.y, b = blockIdx.x;
  const float* xg = x + (long long)g * R * C;
  const long long rpb = (R + nblk - 1) / nblk;
  const long long r0 = (long long)b * rpb, r1 = min(R, r0 + rpb);
  for (int ch = 0; ch < m.nchunk; ++ch) {
    const int c = ch * m.Cb + m.cl;
    float s1 = 0.f, s2 = 0.f;
    if (m.active && c < C) {
      const float pivot = xg[c];
      for (long long r = r0 + m.rsub; r < r1; r += m.RPP) {
        const float d = xg[r * C + c] - pivot;
        s1 += d; s2 += d * d;
      }
    }
    reduce_rsub(m, s1, s2, sh);
    if (m.active && m.rsub == 0 && c < C) {
      float* o = part + (((long long)g * nblk + b) * C + c) * 2;
      o[0] = s1; o[1] = s2;
    }
  }
}

__global__ __launch_bounds__(256) void bn_apply_fwd(const float* __restrict__ x, const float* __restrict__ beta,
                                                    const float* __restrict__ part, float* __restrict__ y,
                                                    float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                    long long R, int C, int nblk, float eps, int act, float leak) {
  __shared__ float smean[kMaxC], srstd[kMaxC];
  const int g = blockIdx.y;
  const float* xg = x + (long long)g * R * C;
  float* yg = y + (long long)g * R * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < nblk; ++b) {
      const float* o = part + (((long long)g * nblk + b) * C + c) * 2;
      s1 += o[0]; s2 += o[1];
    }
    const double inv = 1.0 / (double)R;
    const double dm = s1 * inv;
    double var = s2 * inv - dm * dm;
    var = var > 0.0 ? var : 0.0;
    const float mean = (float)((double)xg[c] + dm);
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    smean[c] = mean; srstd[c] = rstd;
    if (blockIdx.x == 0) { save_mean[g * C + c] = mean; save_rstd[g * C + c] = rstd; }
  }
  __syncthreads();
  const ColMap m = col_map(C);
  if (!m.active) return;
  for (int ch = 0; ch < m.nchunk; ++ch) {
    const int c = ch * m.Cb + m.cl;
    if (c >= C) continue;
    const float mean = smean[c], rstd = srstd[c], bt = beta[c];
    for (long long r = (long long)blockIdx.x * m.RPP + m.rsub; r < R; r += (long long)gridDim.x * m.RPP)
      yg[r * C + c] = acg::act_apply(act, (xg[r * C + c] - mean) * rstd + bt, leak);
  }
}

// ---- BN backward ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_bwd_partial(const float* __restrict__ x, const float* __restrict__ dy,
                                                      const float* __restrict__ beta, const float* __restrict__ save_mean,
                                                      const float* __restrict__ save_rstd, float* __restrict__ part,
                                                      long long R, int C, int nblk, int act, float leak) {
  __shared__ float sh[512];
  const ColMap m = col_map(C);
  const int g = blockIdx.y, b = blockIdx.x;
  const float* xg = x + (long long)g * R * C;
  const float* dyg = dy + (long long)g * R * C;
  const long long rpb = (R + nblk - 1) / nblk;
  const long long r0 = (long long)b * rpb, r1 = min(R, r0 + rpb);
  for (int ch = 0; ch < m.nchunk; ++ch) {
    const int c = ch * m.Cb + m.cl;
    float s1 = 0.f, s2 = 0.f;
    if (m.active && c < C) {
      const float mean = save_mean[g * C + c], rstd = save_rstd[g * C + c], bt = beta[c];
      for (long long r = r0 + m.rsub; r < r1; r += m.RPP) {
        const float xh = (xg[r * C + c] - mean) * rstd;
        const float dp = dyg[r * C + c] * acg::act_deriv_pre(act, xh + bt, leak);
        s1 += dp; s2 += dp * xh;
      }
    }
    reduce_rsub(m, s1, s2, sh);
    if (m.active && m.rsub == 0 && c < C) {
      float* o = part + (((long long)g * nblk + b) * C + c) * 2;
      o[0] = s1; o[1] = s2;
    }
  }
}

__global__ __launch_bounds__(256) void bn_apply_bwd(const float* __restrict__ x, const float* __restrict__ dy,
                                                    const float* __restrict__ beta, const float* __restrict__ save_mean,
                                                    const float* __restrict__ save_rstd, const float* __restrict__ part,
                                                    float* __restrict__ dx, float* __restrict__ dbeta, float dbeta_acc,
                                                    long long R, int C, int groups, int nblk, int act, float leak) {
  __shared__ float sm1[kMaxC], sm2[kMaxC];  // s1/R, s2/R of this block's group
  const int g = blockIdx.y;
  const bool writer = blockIdx.x == 0 && g == 0;
  for (int c = threadIdx.x; c < C; c += 256) {
    double total = 0.0;
    for (int gg = 0; gg < groups; ++gg) {
      if (gg != g && !writer) continue;
      double s1 = 0.0, s2 = 0.0;
      for (int b = 0; b < nblk; ++b) {
        const float* o = part + (((long long)gg * nblk + b) * C + c) * 2;
        s1 += o[0]; s2 += o[1];
      }
      total += s1;
      if (gg == g) { sm1[c] = (float)(s1 / (double)R); sm2[c] = (float)(s2 / (double)R); }
    }
    if (writer) dbeta[c] = (dbeta_acc != 0.f ? dbeta_acc * dbeta[c] : 0.f) + (float)total;
  }
  __syncthreads();
  const ColMap m = col_map(C);
  if (!m.active) return;
  const float* xg = x + (long long)g * R * C;
  const float* dyg = dy + (long long)g * R * C;
  float* dxg = dx + (long long)g * R * C;
  for (int ch = 0; ch < m.nchunk; ++ch) {
    const int c = ch * m.Cb + m.cl;
    if (c >= C) continue;
    const float mean = save_mean[g * C + c], rstd = save_rstd[g * C + c], bt = beta[c];
    const float m1 = sm1[c], m2 = sm2[c];
    for (long long r = (long long)blockIdx.x * m.RPP + m.rsub; r < R; r += (long long)gridDim.x * m.RPP) {
      const float xh = (xg[r * C + c] - mean) * rstd;
      const float dp = dyg[r * C + c] * acg::act_deriv_pre(act, xh + bt, leak);
      dxg[r * C + c] = rstd * (dp - m1 - xh * m2);
    }
  }
}

// ---- bias + activation --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bias_act_fwd_k(const float* __restrict__ x, const float* __restrict__ bias,
                                                      float* __restrict__ y, long long R, int C, int act, float leak) {
  const ColMap m = col_map(C);
  if (!m.active) return;
  for (int ch = 0; ch < m.nchunk; ++ch) {
    const int c = ch * m.Cb + m.cl;
    if (c >= C) continue;
    const float bt = bias ? bias[c] : 0.f;
    for (long long r = (long long)blockIdx.x * m.RPP + m.rsub; r < R; r += (long long)gridDim.x * m.RPP)
      y[r * C + c] = acg::act_apply(act, x[r * C + c] + bt, leak);
  }
}

// dx = dy * act'(y) and per-block column sums of it -> part[nblk][C]
__global__ __launch_bounds__(256) void bias_act_bwd_partial(const float* __restrict__ y, const float* __restrict__ dy,
                                                            float* __restrict__ dx, float* __restrict__ part,
                                                            long long R, int C, int nblk, int act, float leak) {
  __shared__ float sh[512];
  const ColMap m = col_map(C);
  const int b = blockIdx.x;
  const long long rpb = (R + nblk - 1) / nblk;
  const long long r0 = (long long)b * rpb, r1 = min(R, r0 + rpb);
  for (int ch = 0; ch < m.nchunk; ++ch) {
    const int c = ch * m.Cb + m.cl;
    float s1 = 0.f, s2 = 0.f;
    if (m.active && c < C) {
      for (long long r = r0 + m.rsub; r < r1; r += m.RPP) {
        const float gq = dy[r * C + c] * acg::act_deriv_out(act, y[r * C + c], leak);
        if (dx) dx[r * C + c] = gq;
        s1 += gq;
      }
    }
    reduce_rsub(m, s1, s2, sh);
    if (part && m.active && m.rsub == 0 && c < C) part[(long long)b * C + c] = s1;
  }
}

__global__ __launch_bounds__(256) void colsum_finalize(const float* __restrict__ part, float* __restrict__ out,
                                                       float out_acc, int C, int nblk) {
  for (int c = blockIdx.x * 256 + threadIdx.x; c < C; c += gridDim.x * 256) {
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += part[(long long)b * C + c];
    out[c] = (out_acc != 0.f ? out_acc * out[c] : 0.f) + (float)s;
  }
}

int partial_blocks(long long R, int C) {
  const int Cb = C < 256 ? C : 256, RPP = 256 / Cb;
  long long n = R / ((long long)RPP * 8);
  if (n < 1) n = 1;
  if (n > kMaxPartialBlocks) n = kMaxPartialBlocks;
  return (int)n;
}
int apply_blocks(long long R, int C) {
  const int Cb = C < 256 ? C : 256, RPP = 256 / Cb;
  long long n = acg::ceil_div(R, (long long)RPP * 4);
  if (n < 1) n = 1;
  if (n > 1024) n = 1024;
  return (int)n;
}

int check_bn(const char* who, long long rows, int C, int groups) {
  ACG_REQUIRE(rows > 0 && C > 0 && groups > 0, ACG_ERR_INVALID_ARG, "%s: non-positive size", who);
  ACG_REQUIRE(rows % groups == 0, ACG_ERR_INVALID_ARG, "%s: rows (%lld) not divisible by groups (%d)", who, rows, groups);
  ACG_REQUIRE(C <= kMaxC, ACG_ERR_UNSUPPORTED, "%s: more than %d channels", who, kMaxC);
  ACG_REQUIRE(groups <= 65535, ACG_ERR_UNSUPPORTED, "%s: too many groups", who);
  return ACG_OK;
}

}  // namespace

extern "C" {

size_t acg_bn_workspace_bytes(int64_t rows, int32_t channels, int32_t groups) {
  (void)rows;
  if (channels <= 0 || groups <= 0) return 0;
  return (size_t)groups * kMaxPartialBlocks * (size_t)channels * 2 * sizeof(float);
}

int32_t acg_bn_act_fwd(const void* x, const float* beta, void* y, float* save_mean, float* save_rstd, int64_t rows,
                       int32_t C, int32_t groups, float eps, int32_t act, float leak, int32_t dtype, void* ws,
                       size_t wsb, acg_stream_t stream) {
  ACG_REQUIRE_F32(dtype);
  if (int rc = check_bn("bn_act_fwd", rows, C, groups)) return rc;
  ACG_REQUIRE(x && beta && y && save_mean && save_rstd, ACG_ERR_INVALID_ARG, "bn_act_fwd: null pointer");
  ACG_REQUIRE(act == ACG_ACT_NONE || act == ACG_ACT_RELU || act == ACG_ACT_LRELU, ACG_ERR_UNSUPPORTED, "bn_act_fwd: activation %d", act);
  ACG_REQUIRE(ws && wsb >= acg_bn_workspace_bytes(rows, C, groups), ACG_ERR_WORKSPACE, "bn_act_fwd: workspace too small");
  const long long R = rows / groups;
  const int nblk = partial_blocks(R, C);
  hipStream_t st = acg::to_stream(stream);
  hipLaunchKernelGGL(bn_stats_partial, dim3(nblk, groups), dim3(256), 0, st, (const float*)x, (float*)ws, R, C, nblk);
  if (int rc = acg::check_launch("bn_stats_partial")) return rc;
  hipLaunchKernelGGL(bn_apply_fwd, dim3(apply_blocks(R, C), groups), dim3(256), 0, st, (const float*)x, beta,
                     (const float*)ws, (float*)y, save_mean, save_rstd, R, C, nblk, eps, act, leak);
  return acg::check_launch("bn_apply_fwd");
}

int32_t acg_bn_act_bwd(const void* x, const void* dy, const float* beta, const float* save_mean, const float* save_rstd,
                       void* dx, float* dbeta, float dbeta_acc, int64_t rows, int32_t C, int32_t groups, int32_t act,
                       float leak, int32_t dtype, void* ws, size_t wsb, acg_stream_t stream) {
  ACG_REQUIRE_F32(dtype);
  if (int rc = check_bn("bn_act_bwd", rows, C, groups)) return rc;
  ACG_REQUIRE(x && dy && beta && save_mean && save_rstd && dx && dbeta, ACG_ERR_INVALID_ARG, "bn_act_bwd: null pointer");
  ACG_REQUIRE(act == ACG_ACT_NONE || act == ACG_ACT_RELU || act == ACG_ACT_LRELU, ACG_ERR_UNSUPPORTED, "bn_act_bwd: activation %d", act);
  ACG_REQUIRE(ws && wsb >= acg_bn_workspace_bytes(rows, C, groups), ACG_ERR_WORKSPACE, "bn_act_bwd: workspace too small");
  const long long R = rows / groups;
  const int nblk = partial_blocks(R, C);
  hipStream_t st = acg::to_stream(stream);
  hipLaunchKernelGGL(bn_bwd_partial, dim3(nblk, groups), dim3(256), 0, st, (const float*)x, (const float*)dy, beta,
                     save_mean, save_rstd, (float*)ws, R, C, nblk, act, leak);
  if (int rc = acg::check_launch("bn_bwd_partial")) return rc;
  hipLaunchKernelGGL(bn_apply_bwd, dim3(apply_blocks(R, C), groups), dim3(256), 0, st, (const float*)x, (const float*)dy,
                     beta, save_mean, save_rstd, (const float*)ws, (float*)dx, dbeta, dbeta_acc, R, C, groups, nblk, act, leak);
  return acg::check_launch("bn_apply_bwd");
}

size_t acg_bias_workspace_bytes(int64_t rows, int32_t channels) {
  (void)rows;
  return channels > 0 ? (size_t)kMaxPartialBlocks * (size_t)channels * sizeof(float) : 0;
}

int32_t acg_bias_act_fwd(const void* x, const float* bias, void* y, int64_t rows, int32_t C, int32_t act, float leak,
                         int32_t dtype, acg_stream_t stream) {
  ACG_REQUIRE_F32(dtype);
  ACG_REQUIRE(rows > 0 && C > 0, ACG_ERR_INVALID_ARG, "bias_act_fwd: non-positive size");
  ACG_REQUIRE(x && y, ACG_ERR_INVALID_ARG, "bias_act_fwd: null pointer");
  ACG_REQUIRE(act >= ACG_ACT_NONE && act <= ACG_ACT_TANH, ACG_ERR_INVALID_ARG, "bias_act_fwd: activation %d", act);
  hipLaunchKernelGGL(bias_act_fwd_k, dim3(apply_blocks(rows, C)), dim3(256), 0, acg::to_stream(stream), (const float*)x,
                     bias, (float*)y, (long long)rows, C, act, leak);
  return acg::check_launch("bias_act_fwd");
}

int32_t acg_bias_act_bwd(const void* y, const void* dy, void* dx, float* dbias, float dbias_acc, int64_t rows, int32_t C,
                         int32_t act, float leak, int32_t dtype, void* ws, size_t wsb, acg_stream_t stream) {
  ACG_REQUIRE_F32(dtype);
  ACG_REQUIRE(rows > 0 && C > 0, ACG_ERR_INVALID_ARG, "bias_act_bwd: non-positive size");
  ACG_REQUIRE(y && dy && (dbias || dx), ACG_ERR_INVALID_ARG, "bias_act_bwd: null pointer");
  ACG_REQUIRE(act >= ACG_ACT_NONE && act <= ACG_ACT_TANH, ACG_ERR_INVALID_ARG, "bias_act_bwd: activation %d", act);
  ACG_REQUIRE(dx || act == ACG_ACT_NONE, ACG_ERR_INVALID_ARG, "bias_act_bwd: dx NULL requires ACG_ACT_NONE");
  ACG_REQUIRE(!dbias || (ws && wsb >= acg_bias_workspace_bytes(rows, C)), ACG_ERR_WORKSPACE, "bias_act_bwd: workspace too small");
  const int nblk = partial_blocks(rows, C);
  hipStream_t st = acg::to_stream(stream);
  hipLaunchKernelGGL(bias_act_bwd_partial, dim3(nblk), dim3(256), 0, st, (const float*)y, (const float*)dy, (float*)dx,
                     dbias ? (float*)ws : (float*)nullptr, (long long)rows, C, nblk, act, leak);
  if (int rc = acg::check_launch("bias_act_bwd_partial")) return rc;
  if (!dbias) return ACG_OK;
  hipLaunchKernelGGL(colsum_finalize, dim3((C + 255) / 256), dim3(256), 0, st, (const float*)ws, dbias, dbias_acc, C, nblk);
  return acg::check_launch("colsum_finalize");
}

}  // extern "C"
