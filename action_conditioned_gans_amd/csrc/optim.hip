// TensorFlow-1.0 Adam / RMSProp over FLAT float32 buffers (train.py:91-102) with the discriminator
// weight clip of train.py:89,140-143 fused after the update (defect D6 resolved as update -> clip).
// All variables of one scope are views into one flat buffer, so a whole network updates in ONE launch:
// pure streaming, 16 B per lane, read p/g/slots + write p/slots once.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

int grid_for(long long n4) {
  long long b = acg::ceil_div(n4, 256);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

__device__ __forceinline__ float clipf(float v, int use, float lo, float hi) { return use ? fminf(fmaxf(v, lo), hi) : v; }

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float lr_t, float b1, float b2, float eps,
                                      float gs, int use_clip, float lo, float hi) {
  g *= gs;
  m = b1 * m + (1.f - b1) * g;
  v = b2 * v + (1.f - b2) * g * g;
  p = clipf(p - lr_t * m / (sqrtf(v) + eps), use_clip, lo, hi);
}

__global__ __launch_bounds__(256) void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, const int* __restrict__ step, long long n, float lr,
                                              float b1, float b2, float eps, float gs, int use_clip, float lo, float hi) {
  // lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)   (SURVEY A.6); fp64 pow keeps the tiny 1-b^t differences exact.  ONE thread per
  // block evaluates it: two double-precision pow calls are several hundred instructions, more than the whole update of
  // the one or two float4 a thread owns.
  __shared__ float s_lr_t;
  if (threadIdx.x == 0) {
    const int t = *step;
    s_lr_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t)));
  }
  __syncthreads();
  const float lr_t = s_lr_t;
  const long long stride = (long long)gridDim.x * 256;
  const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                    reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  const long long n4 = al ? n / 4 : 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    adam1(pp.x, gg.x, mm.x, vv.x, lr_t, b1, b2, eps, gs, use_clip, lo, hi);
    adam1(pp.y, gg.y, mm.y, vv.y, lr_t, b1, b2, eps, gs, use_clip, lo, hi);
    adam1(pp.z, gg.z, mm.z, vv.z, lr_t, b1, b2, eps, gs, use_clip, lo, hi);
    adam1(pp.w, gg.w, mm.w, vv.w, lr_t, b1, b2, eps, gs, use_clip, lo, hi);
    reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
  }
  for (long long i = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    adam1(p[i], g[i], m[i], v[i], lr_t, b1, b2, eps, gs, use_clip, lo, hi);
}

__device__ __forceinline__ void rms1(float& p, float g, float& ms, float lr, float decay, float eps, float gs, int use_clip,
                                     float lo, float hi) {
  g *= gs;
  ms = decay * ms + (1.f - decay) * g * g;
  p = clipf(p - lr * g / sqrtf(ms + eps), use_clip, lo, hi);
}

__global__ __launch_bounds__(256) void rmsprop_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ ms,
                                                 long long n, float lr, float decay, float eps, float gs, int use_clip,
                                                 float lo, float hi) {
  const long long stride = (long long)gridDim.x * 256;
  const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(ms)) & 15) == 0;
  const long long n4 = al ? n / 4 : 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i], ss = reinterpret_cast<float4*>(ms)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    rms1(pp.x, gg.x, ss.x, lr, decay, eps, gs, use_clip, lo, hi);
    rms1(pp.y, gg.y, ss.y, lr, decay, eps, gs, use_clip, lo, hi);
    rms1(pp.z, gg.z, ss.z, lr, decay, eps, gs, use_clip, lo, hi);
    rms1(pp.w, gg.w, ss.w, lr, decay, eps, gs, use_clip, lo, hi);
    reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(ms)[i] = ss;
  }
  for (long long i = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    rms1(p[i], g[i], ms[i], lr, decay, eps, gs, use_clip, lo, hi);
}

__global__ __launch_bounds__(256) void clip_k(float* __restrict__ p, long long n, float lo, float hi) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = fminf(fmaxf(p[i], lo), hi);
}

__global__ void step_inc_k(int* step) { *step += 1; }

}  // namespace

extern "C" {

int32_t acg_adam_step(float* param, const float* grad, float* m, float* v, const int32_t* step_dev, int64_t n, float lr,
                      float beta1, float beta2, float eps, float grad_scale, int32_t use_clip, float clip_lo, float clip_hi,
                      acg_stream_t stream) {
  ACG_REQUIRE(n > 0 && param && grad && m && v && step_dev, ACG_ERR_INVALID_ARG, "adam_step: bad argument");
  ACG_LAUNCH(adam_k, dim3(grid_for(n / 4 + 1)), dim3(256), 0, acg::to_stream(stream), param, grad, m, v, step_dev,
                     (long long)n, lr, beta1, beta2, eps, grad_scale, use_clip, clip_lo, clip_hi);
  return acg::check_launch("adam_step");
}

int32_t acg_rmsprop_step(float* param, const float* grad, float* ms, int64_t n, float lr, float decay, float eps,
                         float grad_scale, int32_t use_clip, float clip_lo, float clip_hi, acg_stream_t stream) {
  ACG_REQUIRE(n > 0 && param && grad && ms, ACG_ERR_INVALID_ARG, "rmsprop_step: bad argument");
  ACG_LAUNCH(rmsprop_k, dim3(grid_for(n / 4 + 1)), dim3(256), 0, acg::to_stream(stream), param, grad, ms,
                     (long long)n, lr, decay, eps, grad_scale, use_clip, clip_lo, clip_hi);
  return acg::check_launch("rmsprop_step");
}

int32_t acg_clip(float* param, int64_t n, float lo, float hi, acg_stream_t stream) {
  ACG_REQUIRE(n > 0 && param, ACG_ERR_INVALID_ARG, "clip: bad argument");
  ACG_LAUNCH(clip_k, dim3(grid_for(n)), dim3(256), 0, acg::to_stream(stream), param, (long long)n, lo, hi);
  return acg::check_launch("clip");
}

int32_t acg_step_inc(int32_t* step_dev, acg_stream_t stream) {
  ACG_REQUIRE(step_dev, ACG_ERR_INVALID_ARG, "step_inc: null counter");
  ACG_LAUNCH(step_inc_k, dim3(1), dim3(1), 0, acg::to_stream(stream), step_dev);
  return acg::check_launch("step_inc");
}

}  // extern "C"
