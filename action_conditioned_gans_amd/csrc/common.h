// Shared host/device helpers for libacgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/acgan_hip.h"

namespace acg {

// Thread-local error message behind acg_last_error(); returns `code`.
int fail(int code, const char* fmt, ...);
// Checks the launch that was just enqueued (no synchronisation; safe under graph capture).
int check_launch(const char* what);

static inline hipStream_t to_stream(acg_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Every launch first drops whatever error an unrelated HIP call left behind on this thread (torch and RCCL make
// calls whose benign failures stay "last error" until somebody reads them, e.g. peer access already enabled), so that
// check_launch reports the launch it follows and nothing else.
#define ACG_LAUNCH(...)                \
  do {                                 \
    (void)hipGetLastError();           \
    hipLaunchKernelGGL(__VA_ARGS__);   \
  } while (0)

constexpr int kWave = 64;  // CDNA wavefront

// ---- device-side reductions -----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Sum over a block of up to 1024 threads; result valid in thread 0.  `scratch` holds >= 16 Ts.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();  // scratch may still be read from a previous call
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  T r = T(0);
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) r += scratch[i];
  return r;
}

__device__ __forceinline__ float sgnf(float v) { return (v > 0.f) - (v < 0.f); }

// Activation value / derivative w.r.t. the pre-activation u (lrelu is ops.py:22-26: f1*u + f2*|u|).
__device__ __forceinline__ float act_apply(int act, float u, float leak) {
  switch (act) {
    case ACG_ACT_RELU: return u > 0.f ? u : 0.f;
    case ACG_ACT_LRELU: return 0.5f * (1.f + leak) * u + 0.5f * (1.f - leak) * fabsf(u);
    case ACG_ACT_TANH: return tanhf(u);
    default: return u;
  }
}
__device__ __forceinline__ float act_deriv_pre(int act, float u, float leak) {
  switch (act) {
    case ACG_ACT_RELU: return u > 0.f ? 1.f : 0.f;
    case ACG_ACT_LRELU: return 0.5f * (1.f + leak) + 0.5f * (1.f - leak) * sgnf(u);
    default: return 1.f;
  }
}
// Derivative expressed through the OUTPUT y (sign(y) == sign(u) for relu/lrelu; tanh' = 1 - y^2).
__device__ __forceinline__ float act_deriv_out(int act, float y, float leak) {
  switch (act) {
    case ACG_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case ACG_ACT_LRELU: return 0.5f * (1.f + leak) + 0.5f * (1.f - leak) * sgnf(y);
    case ACG_ACT_TANH: return 1.f - y * y;
    default: return 1.f;
  }
}

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- storage types ---------------------------------------------------------------------------------
// A `dtype` argument names the storage type of a call's activation-class tensors: ACG_F32 / ACG_BF16 for all of
// them, or ACG_DTYPE2(first, second) where two differ (include/acgan_hip.h).
static inline bool dt_valid(int dt) {
  if (dt == ACG_F32 || dt == ACG_BF16) return true;
  return (dt & ~0xFF) == 0x100 && (dt & 0xF) <= ACG_BF16 && ((dt >> 4) & 0xF) <= ACG_BF16;
}
static inline int dt_first(int dt) { return (dt & 0x100) ? (dt & 0xF) : dt; }
static inline int dt_second(int dt) { return (dt & 0x100) ? ((dt >> 4) & 0xF) : dt; }
static inline int dt_size(int t) { return t == ACG_BF16 ? 2 : 4; }

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ float ldf(const T* p) { return (float)*p; }
template <typename T>
__device__ __forceinline__ void stf(T* p, float v) { *p = (T)v; }
// V consecutive elements (V = 4: one 16-byte / 8-byte access, p aligned to it; V = 1: scalar)
template <int V, typename T>
__device__ __forceinline__ void ldv(const T* p, float (&v)[V]) {
  if constexpr (V == 4 && sizeof(T) == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  else if constexpr (V == 4) { const bf16x4 t = *reinterpret_cast<const bf16x4*>(p); v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3]; }
  else v[0] = (float)*p;
}
template <int V, typename T>
__device__ __forceinline__ void stv(T* p, const float (&v)[V]) {
  if constexpr (V == 4 && sizeof(T) == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  else if constexpr (V == 4) *reinterpret_cast<bf16x4*>(p) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  else *p = (T)v[0];
}

}  // namespace acg

#define ACG_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) return acg::fail(code, __VA_ARGS__); \
  } while (0)
#define ACG_REQUIRE_F32(dtype) ACG_REQUIRE((dtype) == ACG_F32, ACG_ERR_UNSUPPORTED, "%s: only ACG_F32 is implemented for this entry point", __func__)

// Run BODY with TA / TB bound to the storage types named by a dtype argument: (f32, f32), (bf16, bf16) or the mixed
// pair (bf16, f32) - a bf16 network whose loss-facing tensor stays float32.
#define ACG_WITH_TYPES(dt, who, BODY)                                                                         \
  do {                                                                                                        \
    const int acg_t1 = acg::dt_valid(dt) ? acg::dt_first(dt) : -1, acg_t2 = acg::dt_valid(dt) ? acg::dt_second(dt) : -1; \
    if (acg_t1 == ACG_F32 && acg_t2 == ACG_F32) { using TA = float; using TB = float; BODY; }                  \
    else if (acg_t1 == ACG_BF16 && acg_t2 == ACG_BF16) { using TA = __bf16; using TB = __bf16; BODY; }         \
    else if (acg_t1 == ACG_BF16 && acg_t2 == ACG_F32) { using TA = __bf16; using TB = float; BODY; }           \
    else return acg::fail(ACG_ERR_UNSUPPORTED, "%s: dtype %d (storage types must be f32, bf16 or bf16 -> f32)", who, (int)(dt)); \
  } while (0)
// the same with the fourth pair (f32 -> bf16): float32 sources written into a bf16 network's tensors
#define ACG_WITH_TYPES_ANY(dt, who, BODY)                                                                     \
  do {                                                                                                        \
    const int acg_t1 = acg::dt_valid(dt) ? acg::dt_first(dt) : -1, acg_t2 = acg::dt_valid(dt) ? acg::dt_second(dt) : -1; \
    if (acg_t1 == ACG_F32 && acg_t2 == ACG_F32) { using TA = float; using TB = float; BODY; }                  \
    else if (acg_t1 == ACG_BF16 && acg_t2 == ACG_BF16) { using TA = __bf16; using TB = __bf16; BODY; }         \
    else if (acg_t1 == ACG_BF16 && acg_t2 == ACG_F32) { using TA = __bf16; using TB = float; BODY; }           \
    else if (acg_t1 == ACG_F32 && acg_t2 == ACG_BF16) { using TA = float; using TB = __bf16; BODY; }           \
    else return acg::fail(ACG_ERR_UNSUPPORTED, "%s: dtype %d", who, (int)(dt));                                \
  } while (0)
