// TensorFlow-1.0 Adam / RMSProp over FLAT float32 buffers (train.py:91-102) with the discriminator
// weight clip of train.py:89,140-143 fused after the update (defect D6 resolved as update -> clip).
// All variables of one scope are views into one flat buffer, so a whole network updates in ONE launch:
// pure streaming, 16 B per lane, read p/g/slots + write p/slots once.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"

namespace {

int grid_for(long long n4) {
  long long b = acg::ceil_div(n4, 256);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

__device__ __forceinline__ float clipf(float v, int use, float lo, float hi) { return use ? fminf(fmaxf(v, lo), hi) : v; }

// (explicitly rounded operations: the same sequence whichever kernel inlines it - hipcc otherwise contracts multiplies and adds
// into FMAs differently in different contexts, and the flat kernels and opt_prepare_k below must agree bit for bit)
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float lr_t, float b1, float b2, float eps,
                                      float gs, int use_clip, float lo, float hi) {
  g = __fmul_rn(g, gs);
  m = __fmaf_rn(b1, m, __fmul_rn(1.f - b1, g));
  v = __fmaf_rn(b2, v, __fmul_rn(__fmul_rn(1.f - b2, g), g));
  p = clipf(__fsub_rn(p, __fdiv_rn(__fmul_rn(lr_t, m), __fadd_rn(sqrtf(v), eps))), use_clip, lo, hi);
}

__global__ __launch_bounds__(256) void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, const int* __restrict__ step, long long n, float lr,
                                              float b1, float b2, float eps, float gs, int use_clip, float lo, float hi) {
  // lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)   (SURVEY A.6); fp64 pow keeps the tiny 1-b^t differences exact.  ONE thread per
  // block evaluates it: two double-precision pow calls are several hundred instructions, more than the whole update of
  // the one or two float4 a thread owns.
  __shared__ float s_lr_t;
  const long long stride = (long long)gridDim.x * 256;
  const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                    reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  const long long n4 = al ? n / 4 : 0;
  // Round 5: the first float4 of every thread (the grid is sized so that it is nearly the only one) is loaded BEFORE the
  // learning-rate prologue: that prologue is a dependent read of the step counter plus ~1 us of double-precision pow on one
  // thread, during which the whole chip used to wait with nothing in flight (19.4 us for 133 MB in situ, 3.2 TB/s).
  const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool have = i0 < n4;
  float4 pp, mm, vv, gg;
  if (have) {
    pp = reinterpret_cast<float4*>(p)[i0]; mm = reinterpret_cast<float4*>(m)[i0]; vv = reinterpret_cast<float4*>(v)[i0];
    gg = reinterpret_cast<const float4*>(g)[i0];
  }
  if (threadIdx.x == 0) {
    const int t = *step;
    s_lr_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t)));
  }
  __syncthreads();
  const float lr_t = s_lr_t;
  if (have) {
    adam1(pp.x, gg.x, mm.x, vv.x, lr_t, b1, b2, eps, gs, use_clip, lo, hi);
    adam1(pp.y, gg.y, mm.y, vv.y, lr_t, b1, b2, eps, gs, use_clip, lo, hi);
    adam1(pp.z, gg.z, mm.z, vv.z, lr_t, b1, b2, eps, gs, use_clip, lo, hi);
    adam1(pp.w, gg.w, mm.w, vv.w, lr_t, b1, b2, eps, gs, use_clip, lo, hi);
    reinterpret_cast<float4*>(p)[i0] = pp; reinterpret_cast<float4*>(m)[i0] = mm; reinterpret_cast<float4*>(v)[i0] = vv;
  }
  for (long long i = i0 + stride; i < n4; i += stride) {
    pp = reinterpret_cast<float4*>(p)[i]; mm = reinterpret_cast<float4*>(m)[i]; vv = reinterpret_cast<float4*>(v)[i];
    gg = reinterpret_cast<const float4*>(g)[i];
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
  g = __fmul_rn(g, gs);
  ms = __fmaf_rn(decay, ms, __fmul_rn(__fmul_rn(1.f - decay, g), g));
  p = clipf(__fsub_rn(p, __fdiv_rn(__fmul_rn(lr, g), sqrtf(__fadd_rn(ms, eps)))), use_clip, lo, hi);
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

// ---- bf16 pipeline: optimizer update + refresh of the bf16 filter copies in ONE launch (acg_opt_step_prepare_bf16) ------------
// The conv kernels of a bf16 network read two bf16 copies of every filter [taps][A][B] - rm [taps][A][round8(B)] and the
// transpose tr [taps][B][round8(A)] (acg_weights_prepare_bf16) - which have to follow the float32 master weights after every
// update: a second launch over the whole scope right behind the optimizer's (round 3: 14 of the 37.5 us of D's update).
// Here a block owns one 32 (a) x 32 (b) tile of one tap of one filter: it updates the tile's elements (the same adam1 / rms1
// as the flat kernels: bit-identical parameters and slots), stores the rm run as it goes and the tr run through an LDS
// transpose; what lies between the filters in the flat buffer (beta, biases, alignment gaps) goes to one block per gap.
struct OptScalars {
  float lr, b1, b2, eps, gs, lo, hi;
  int use_clip;
};
struct OptPrepList {
  long long off[ACG_PREP_MAX];            // element offset of filter e in the flat buffers
  __bf16* rm[ACG_PREP_MAX];
  __bf16* tr[ACG_PREP_MAX];
  int taps[ACG_PREP_MAX], A[ACG_PREP_MAX], B[ACG_PREP_MAX];
  int first_block[ACG_PREP_MAX + 1];      // tile blocks of entry e: [first_block[e], first_block[e + 1])
  long long gap_lo[ACG_PREP_MAX + 1], gap_len[ACG_PREP_MAX + 1];      // gap j is block first_block[count] + j
};

template <int KIND>      // 0: Adam, 1: RMSProp
__global__ __launch_bounds__(256) void opt_prepare_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ s1,
                                                     float* __restrict__ s2, const int* __restrict__ step, const OptScalars a,
                                                     const OptPrepList l, int count) {
  __shared__ float tile[32][33];
  __shared__ float s_lr_t;
  float lr_t = a.lr;
  // Adam's bias-corrected learning rate: one thread derives it (a dependent read of the step counter + double-precision pow,
  // ~1-2 us); the block picks it up with `sync_lr()` - in the float4 path AFTER its loads are in flight (round 5)
  if (KIND == 0 && threadIdx.x == 0) {
    const int t = *step;
    s_lr_t = (float)((double)a.lr * sqrt(1.0 - pow((double)a.b2, (double)t)) / (1.0 - pow((double)a.b1, (double)t)));
  }
  auto sync_lr = [&]() { if (KIND == 0) { __syncthreads(); lr_t = s_lr_t; } };     // block-uniform call sites only
  auto upd = [&](long long i) -> float {
    float pp = p[i], m = s1[i];
    if (KIND == 0) { float v = s2[i]; adam1(pp, g[i], m, v, lr_t, a.b1, a.b2, a.eps, a.gs, a.use_clip, a.lo, a.hi); s2[i] = v; }
    else rms1(pp, g[i], m, a.lr, a.b1, a.eps, a.gs, a.use_clip, a.lo, a.hi);
    p[i] = pp; s1[i] = m;
    return pp;
  };
  const int blk = (int)blockIdx.x;
  if (blk >= l.first_block[count]) {             // a run between two filters: beta / bias vectors, alignment gaps
    const int j = blk - l.first_block[count];
    sync_lr();
    for (long long i = threadIdx.x; i < l.gap_len[j]; i += 256) upd(l.gap_lo[j] + i);
    return;
  }
  int e = 0;
  while (e + 1 < count && blk >= l.first_block[e + 1]) ++e;
  const int A = l.A[e], B = l.B[e], A8 = (A + 7) & ~7, B8 = (B + 7) & ~7;
  const int ta_n = (A8 + 31) >> 5, tb_n = (B8 + 31) >> 5;
  int t = blk - l.first_block[e];
  const int tb = t % tb_n; t /= tb_n;
  const int ta = t % ta_n, tap = t / ta_n;
  const int a0 = ta * 32, b0 = tb * 32;
  const int r = threadIdx.x >> 3, q = threadIdx.x & 7;
  if ((B & 3) == 0) {
    // rows are 16-byte aligned: thread -> (a = a0 + r, b run 4q .. 4q + 3), float4 accesses, the rm run as one 8-byte store
    const int ar = a0 + r, bq = b0 + 4 * q;
    float nv[4] = {0.f, 0.f, 0.f, 0.f};
    const bool have = ar < A && bq < B;
    const long long i0 = l.off[e] + ((long long)tap * A + ar) * B + bq;
    float4 pp, mm, gg, vv;
    if (have) {
      pp = *reinterpret_cast<float4*>(p + i0); mm = *reinterpret_cast<float4*>(s1 + i0);
      gg = *reinterpret_cast<const float4*>(g + i0);
      if (KIND == 0) vv = *reinterpret_cast<float4*>(s2 + i0);
    }
    sync_lr();
    if (have) {
      if (KIND == 0) {
        adam1(pp.x, gg.x, mm.x, vv.x, lr_t, a.b1, a.b2, a.eps, a.gs, a.use_clip, a.lo, a.hi);
        adam1(pp.y, gg.y, mm.y, vv.y, lr_t, a.b1, a.b2, a.eps, a.gs, a.use_clip, a.lo, a.hi);
        adam1(pp.z, gg.z, mm.z, vv.z, lr_t, a.b1, a.b2, a.eps, a.gs, a.use_clip, a.lo, a.hi);
        adam1(pp.w, gg.w, mm.w, vv.w, lr_t, a.b1, a.b2, a.eps, a.gs, a.use_clip, a.lo, a.hi);
        *reinterpret_cast<float4*>(s2 + i0) = vv;
      } else {
        rms1(pp.x, gg.x, mm.x, a.lr, a.b1, a.eps, a.gs, a.use_clip, a.lo, a.hi);
        rms1(pp.y, gg.y, mm.y, a.lr, a.b1, a.eps, a.gs, a.use_clip, a.lo, a.hi);
        rms1(pp.z, gg.z, mm.z, a.lr, a.b1, a.eps, a.gs, a.use_clip, a.lo, a.hi);
        rms1(pp.w, gg.w, mm.w, a.lr, a.b1, a.eps, a.gs, a.use_clip, a.lo, a.hi);
      }
      *reinterpret_cast<float4*>(p + i0) = pp; *reinterpret_cast<float4*>(s1 + i0) = mm;
      nv[0] = pp.x; nv[1] = pp.y; nv[2] = pp.z; nv[3] = pp.w;
    }
    // rm [taps][A][B8]: the row's run of four (pad columns b in [B, B8) are zeros)
    if (ar < A && bq < B8)
      *reinterpret_cast<acg::bf16x4*>(l.rm[e] + ((long long)tap * A + ar) * B8 + bq) = acg::bf16x4{(__bf16)nv[0], (__bf16)nv[1], (__bf16)nv[2], (__bf16)nv[3]};
#pragma unroll
    for (int u = 0; u < 4; ++u) tile[r][4 * q + u] = nv[u];
  } else {
    // rows at any 4-byte offset (266 gathered channels, 1 or 5 output channels): thread -> (a = a0 + t / 32 + 8u, b = b0 + t % 32),
    // a wave instruction = two rows of 32 consecutive floats
    const int bl = threadIdx.x & 31, bb = b0 + bl;
    sync_lr();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int al = (threadIdx.x >> 5) + 8 * u, aa = a0 + al;
      float v = 0.f;
      if (aa < A && bb < B) v = upd(l.off[e] + ((long long)tap * A + aa) * B + bb);
      if (aa < A && bb < B8) l.rm[e][((long long)tap * A + aa) * B8 + bb] = (__bf16)v;
      tile[al][bl] = v;
    }
  }
  __syncthreads();
  // tr [taps][B][A8]: thread -> (b = b0 + r, a run 4q .. 4q + 3); pad rows a in [A, A8) come out as the zeros of invalid elements
  const int bt = b0 + r, at = a0 + 4 * q;
  if (bt < B && at < A8)
    *reinterpret_cast<acg::bf16x4*>(l.tr[e] + ((long long)tap * B + bt) * A8 + at) =
        acg::bf16x4{(__bf16)tile[4 * q][r], (__bf16)tile[4 * q + 1][r], (__bf16)tile[4 * q + 2][r], (__bf16)tile[4 * q + 3][r]};
}

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

int32_t acg_opt_step_prepare_bf16(float* param, const float* grad, float* slot1, float* slot2, const int32_t* step_dev, int64_t n,
                                  const acg_opt_args* args, const acg_prep_list* list, int32_t count, acg_stream_t stream) {
  ACG_REQUIRE(n > 0 && param && grad && slot1 && args && list, ACG_ERR_INVALID_ARG, "opt_step_prepare_bf16: null pointer / n <= 0");
  ACG_REQUIRE(count >= 1 && count <= ACG_PREP_MAX, ACG_ERR_INVALID_ARG, "opt_step_prepare_bf16: 1..%d filters", ACG_PREP_MAX);
  ACG_REQUIRE(args->kind == 0 || args->kind == 1, ACG_ERR_INVALID_ARG, "opt_step_prepare_bf16: kind %d (0 = Adam, 1 = RMSProp)", args->kind);
  ACG_REQUIRE(args->kind == 1 || (slot2 && step_dev), ACG_ERR_INVALID_ARG, "opt_step_prepare_bf16: Adam needs slot2 and the step counter");
  ACG_REQUIRE((reinterpret_cast<uintptr_t>(param) & 15) == 0 && (reinterpret_cast<uintptr_t>(grad) & 15) == 0 && (reinterpret_cast<uintptr_t>(slot1) & 15) == 0 &&
              (reinterpret_cast<uintptr_t>(slot2) & 15) == 0, ACG_ERR_INVALID_ARG, "opt_step_prepare_bf16: flat buffers must be 16-byte aligned");
  // the filters, in the order they lie in the flat buffer; what is left between them are the gaps
  int order[ACG_PREP_MAX];
  for (int i = 0; i < count; ++i) order[i] = i;
  std::sort(order, order + count, [&](int x, int y) { return list->src[x] < list->src[y]; });
  OptPrepList l;
  long long cursor = 0, blocks = 0;
  int ngaps = 0;
  for (int k = 0; k < count; ++k) {
    const int i = order[k];
    ACG_REQUIRE(list->src[i] && list->rm[i] && list->tr[i] && list->taps[i] > 0 && list->a[i] > 0 && list->b[i] > 0, ACG_ERR_INVALID_ARG,
                "opt_step_prepare_bf16: bad filter entry %d", i);
    const long long off = (const float*)list->src[i] - param, numel = (long long)list->taps[i] * list->a[i] * list->b[i];
    ACG_REQUIRE(off >= cursor && off + numel <= n && (off & 3) == 0, ACG_ERR_INVALID_ARG,
                "opt_step_prepare_bf16: filter %d does not lie inside the flat buffer (16-byte aligned, no overlap)", i);
    if (off > cursor) { l.gap_lo[ngaps] = cursor; l.gap_len[ngaps] = off - cursor; ++ngaps; }
    cursor = off + numel;
    l.off[k] = off; l.rm[k] = (__bf16*)list->rm[i]; l.tr[k] = (__bf16*)list->tr[i];
    l.taps[k] = list->taps[i]; l.A[k] = list->a[i]; l.B[k] = list->b[i];
    l.first_block[k] = (int)blocks;
    const int A8 = (list->a[i] + 7) & ~7, B8 = (list->b[i] + 7) & ~7;
    blocks += (long long)list->taps[i] * ((A8 + 31) / 32) * ((B8 + 31) / 32);
    ACG_REQUIRE(blocks < (1ll << 30), ACG_ERR_UNSUPPORTED, "opt_step_prepare_bf16: too many tiles");
  }
  if (n > cursor) { l.gap_lo[ngaps] = cursor; l.gap_len[ngaps] = n - cursor; ++ngaps; }
  l.first_block[count] = (int)blocks;
  const OptScalars a{args->lr, args->beta1_or_decay, args->beta2, args->eps, args->grad_scale, args->clip_lo, args->clip_hi, args->use_clip};
  hipStream_t st = acg::to_stream(stream);
  if (args->kind == 0) ACG_LAUNCH((opt_prepare_k<0>), dim3((unsigned)(blocks + ngaps)), dim3(256), 0, st, param, grad, slot1, slot2, (const int*)step_dev, a, l, (int)count);
  else ACG_LAUNCH((opt_prepare_k<1>), dim3((unsigned)(blocks + ngaps)), dim3(256), 0, st, param, grad, slot1, slot2, (const int*)step_dev, a, l, (int)count);
  return acg::check_launch("opt_step_prepare_bf16");
}

int32_t acg_step_inc(int32_t* step_dev, acg_stream_t stream) {
  ACG_REQUIRE(step_dev, ACG_ERR_INVALID_ARG, "step_inc: null counter");
  ACG_LAUNCH(step_inc_k, dim3(1), dim3(1), 0, acg::to_stream(stream), step_dev);
  return acg::check_launch("step_inc");
}

}  // extern "C"
