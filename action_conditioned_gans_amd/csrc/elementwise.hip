// Channel plumbing (tf.tile / tf.concat of train.py:48-50,64,68 and models.py:16,38,84), the gradient
// fan-in add, and the library's error / version entry points.  All kernels are pure streaming copies:
// one thread per output float, grid-stride, consecutive lanes on consecutive addresses.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <algorithm>

#include "common.h"

namespace acg {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ACG_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return ACG_OK;
}

}  // namespace acg

namespace {

int grid_for(long long n) {
  long long b = acg::ceil_div(n, 256);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

// y[r, 0:ca] = a[r, :]; y[r, ca:ca+cb] = b[r / rows_per_b, :]  (rows_per_b = 1: plain concat; = hw: tiled actions);
// y rows are `pitch` floats apart, pad channels are not written
template <typename TI, typename TS, typename TO>
__global__ __launch_bounds__(256) void concat_k(const TI* __restrict__ a, const TS* __restrict__ b,
                                                TO* __restrict__ y, long long rows, int ca, int cb, int rows_per_b,
                                                int pitch) {
  const int cy = ca + cb;
  const long long n = rows * cy, stride = (long long)gridDim.x * 256;
  if (n < (1ll << 31)) {   // 32-bit index arithmetic: a 64-bit division per element costs more than the copy
    const unsigned n32 = (unsigned)n, st32 = (unsigned)stride, ucy = (unsigned)cy, urpb = (unsigned)rows_per_b;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n32; i += st32) {
      const unsigned r = i / ucy, c = i - r * ucy;
      if (c >= (unsigned)ca) acg::stf(y + (size_t)r * pitch + c, acg::ldf(b + (size_t)(r / urpb) * cb + (c - ca)));
      else if (a != nullptr) acg::stf(y + (size_t)r * pitch + c, acg::ldf(a + (size_t)r * ca + c));
    }
    return;
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const long long r = i / cy;
    const int c = (int)(i - r * cy);
    if (c >= ca) acg::stf(y + r * pitch + c, acg::ldf(b + (r / rows_per_b) * cb + (c - ca)));
    else if (a != nullptr) acg::stf(y + r * pitch + c, acg::ldf(a + r * ca + c));
  }
}

// y[r, ca : ca + cb] = b[r / rows_per_b, :] only: the features y[r, 0:ca] were written in place by their producer
template <typename TO>
__global__ __launch_bounds__(256) void tile_actions_k(const float* __restrict__ b, TO* __restrict__ y, unsigned rows, int ca, int cb,
                                                      unsigned rows_per_b, int pitch) {
  const unsigned n = rows * (unsigned)cb;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const unsigned r = i / (unsigned)cb, j = i - r * (unsigned)cb;
    acg::stf(y + (size_t)r * pitch + ca + j, b[(size_t)(r / rows_per_b) * cb + j]);
  }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void slice_k(const TI* __restrict__ src, TO* __restrict__ dst, float acc,
                                               long long rows, int c_src, int c_off, int c_dst) {
  const long long n = rows * c_dst, stride = (long long)gridDim.x * 256;
  if (n < (1ll << 31) && rows * c_src < (1ll << 31)) {
    const unsigned n32 = (unsigned)n, st32 = (unsigned)stride, ud = (unsigned)c_dst;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n32; i += st32) {
      const unsigned r = i / ud, c = i - r * ud;
      const float v = acg::ldf(src + r * (unsigned)c_src + c_off + c);
      acg::stf(dst + i, acc != 0.f ? acc * acg::ldf(dst + i) + v : v);
    }
    return;
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const long long r = i / c_dst;
    const int c = (int)(i - r * c_dst);
    const float v = acg::ldf(src + r * c_src + c_off + c);
    acg::stf(dst + i, acc != 0.f ? acc * acg::ldf(dst + i) + v : v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void add_k(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  const unsigned al = 4 * sizeof(T) - 1;
  const long long n4 = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(y)) & al) == 0 ? n / 4 : 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float u[4], v[4];
    acg::ldv<4>(a + 4 * i, u); acg::ldv<4>(b + 4 * i, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) u[e] += v[e];
    acg::stv<4>(y + 4 * i, u);
  }
  for (long long i = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) acg::stf(y + i, acg::ldf(a + i) + acg::ldf(b + i));
}

struct CopyList {
  const float* src[ACG_COPY_MAX];
  void* dst[ACG_COPY_MAX];
  long long rows[ACG_COPY_MAX];
  int cols[ACG_COPY_MAX], pitch[ACG_COPY_MAX], half[ACG_COPY_MAX];
  int div[ACG_COPY_MAX], mod[ACG_COPY_MAX];      // source row of destination row r: (r / div) % mod (mod 0: no wrap)
};

// blockIdx.y = segment; float4 when the segment is dense and aligned, else one float per thread with 32-bit row math
__global__ __launch_bounds__(256) void copy_many_k(const CopyList l) {
  const int sgm = blockIdx.y;
  const float* __restrict__ src = l.src[sgm];
  const long long rows = l.rows[sgm];
  const int cols = l.cols[sgm], pitch = l.pitch[sgm];
  const long long n = rows * cols, stride = (long long)gridDim.x * 256;
  const int dv = l.div[sgm], md = l.mod[sgm];
  if (dv > 1 || md > 0) {          // tiled source rows (the action vector over a feature map): small, 32-bit row math
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
      const unsigned r = (unsigned)(i / cols), c = (unsigned)(i - (long long)r * cols);
      unsigned sr = r / (unsigned)dv;
      if (md > 0) sr %= (unsigned)md;
      const float v = src[(size_t)sr * cols + c];
      if (l.half[sgm]) reinterpret_cast<__bf16*>(l.dst[sgm])[(size_t)r * pitch + c] = (__bf16)v;
      else reinterpret_cast<float*>(l.dst[sgm])[(size_t)r * pitch + c] = v;
    }
    return;
  }
  if (cols <= 4 && pitch != cols && rows * (long long)pitch < (1ll << 31)) {
    // a few channels at a wider pitch (fed frames: 3 channels into pitch 4 / 8 tensors - most of a step's feed bytes): a thread
    // per ROW, no division (round 4 divided a 64-bit element index by `cols` per element: 8 us per feed launch in situ)
    const unsigned nr = (unsigned)rows, st = (unsigned)stride;
    for (unsigned r = blockIdx.x * 256 + threadIdx.x; r < nr; r += st) {
      float v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = c < cols ? src[r * (unsigned)cols + c] : 0.f;
      if (l.half[sgm]) {
        __bf16* dh = reinterpret_cast<__bf16*>(l.dst[sgm]) + r * (unsigned)pitch;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < cols) dh[c] = (__bf16)v[c];
      } else {
        float* df = reinterpret_cast<float*>(l.dst[sgm]) + r * (unsigned)pitch;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < cols) df[c] = v[c];
      }
    }
    return;
  }
  if (l.half[sgm]) {   // float32 source -> bf16 destination
    __bf16* __restrict__ dh = reinterpret_cast<__bf16*>(l.dst[sgm]);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
      const long long r = i / cols;
      dh[r * pitch + (i - r * cols)] = (__bf16)src[i];
    }
    return;
  }
  float* __restrict__ dst = reinterpret_cast<float*>(l.dst[sgm]);
  if (pitch == cols && (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += stride) d4[i] = s4[i];
    return;
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const long long r = i / cols;
    dst[r * pitch + (i - r * cols)] = src[i];
  }
}

}  // namespace

extern "C" {

int32_t acg_version(void) { return ACG_ABI_VERSION; }
const char* acg_build_info(void) { return "hip gfx950 (fp32 MFMA 32x32x2; bf16 MFMA 32x32x16)"; }
const char* acg_last_error(void) { return acg::g_err; }

int32_t acg_stream_edge_create(acg_edge_t* edge) {
  ACG_REQUIRE(edge != nullptr, ACG_ERR_INVALID_ARG, "stream_edge_create: null output");
  hipEvent_t ev = nullptr;
  const hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence);
  ACG_REQUIRE(e == hipSuccess, ACG_ERR_LAUNCH, "stream_edge_create: %s", hipGetErrorString(e));
  *edge = (acg_edge_t)ev;
  return ACG_OK;
}
int32_t acg_stream_edge_destroy(acg_edge_t edge) {
  ACG_REQUIRE(edge != nullptr, ACG_ERR_INVALID_ARG, "stream_edge_destroy: null edge");
  const hipError_t e = hipEventDestroy((hipEvent_t)edge);
  ACG_REQUIRE(e == hipSuccess, ACG_ERR_LAUNCH, "stream_edge_destroy: %s", hipGetErrorString(e));
  return ACG_OK;
}
int32_t acg_stream_edge(acg_edge_t edge, acg_stream_t from, acg_stream_t to) {
  ACG_REQUIRE(edge != nullptr, ACG_ERR_INVALID_ARG, "stream_edge: null edge");
  hipError_t e = hipEventRecord((hipEvent_t)edge, acg::to_stream(from));
  ACG_REQUIRE(e == hipSuccess, ACG_ERR_LAUNCH, "stream_edge (record): %s", hipGetErrorString(e));
  e = hipStreamWaitEvent(acg::to_stream(to), (hipEvent_t)edge, 0);
  ACG_REQUIRE(e == hipSuccess, ACG_ERR_LAUNCH, "stream_edge (wait): %s", hipGetErrorString(e));
  return ACG_OK;
}

int32_t acg_concat_actions_fwd(const void* x, const float* actions, void* y, int32_t B, int32_t hw, int32_t c, int32_t a,
                               int32_t y_pitch, int32_t dtype, acg_stream_t stream) {
  ACG_REQUIRE(dtype == ACG_F32 || dtype == ACG_BF16, ACG_ERR_UNSUPPORTED, "concat_actions_fwd: dtype %d", dtype);
  ACG_REQUIRE(B > 0 && hw > 0 && c > 0 && a > 0, ACG_ERR_INVALID_ARG, "concat_actions_fwd: non-positive size");
  ACG_REQUIRE(actions && y, ACG_ERR_INVALID_ARG, "concat_actions_fwd: null pointer");
  const int pitch = y_pitch > 0 ? y_pitch : c + a;
  ACG_REQUIRE(pitch >= c + a, ACG_ERR_INVALID_ARG, "concat_actions_fwd: pitch smaller than the row");
  const long long rows = (long long)B * hw;
  if (!x) {      // the features are already in y (their producer wrote them at this pitch): only the tiled actions
    ACG_REQUIRE(rows * a < (1ll << 31), ACG_ERR_UNSUPPORTED, "concat_actions_fwd: tensor too large");
    if (dtype == ACG_BF16) ACG_LAUNCH(tile_actions_k<__bf16>, dim3(grid_for(rows * a)), dim3(256), 0, acg::to_stream(stream), actions, (__bf16*)y, (unsigned)rows, c, a, (unsigned)hw, pitch);
    else ACG_LAUNCH(tile_actions_k<float>, dim3(grid_for(rows * a)), dim3(256), 0, acg::to_stream(stream), actions, (float*)y, (unsigned)rows, c, a, (unsigned)hw, pitch);
    return acg::check_launch("concat_actions_fwd");
  }
  if (dtype == ACG_BF16)
    ACG_LAUNCH((concat_k<__bf16, float, __bf16>), dim3(grid_for(rows * (c + a))), dim3(256), 0, acg::to_stream(stream), (const __bf16*)x,
               actions, (__bf16*)y, rows, c, a, hw, pitch);
  else
    ACG_LAUNCH((concat_k<float, float, float>), dim3(grid_for(rows * (c + a))), dim3(256), 0, acg::to_stream(stream), (const float*)x,
               actions, (float*)y, rows, c, a, hw, pitch);
  return acg::check_launch("concat_actions_fwd");
}

int32_t acg_concat_channels_fwd(const void* a, const void* b, void* y, int64_t rows, int32_t ca, int32_t cb, int32_t y_pitch,
                                int32_t dtype, acg_stream_t stream) {
  ACG_REQUIRE(rows > 0 && ca > 0 && cb >= 0, ACG_ERR_INVALID_ARG, "concat_channels_fwd: non-positive size");
  ACG_REQUIRE((a || cb > 0) && (b || cb == 0) && y, ACG_ERR_INVALID_ARG, "concat_channels_fwd: null pointer");
  const int pitch = y_pitch > 0 ? y_pitch : ca + cb;
  ACG_REQUIRE(pitch >= ca + cb, ACG_ERR_INVALID_ARG, "concat_channels_fwd: pitch smaller than the row");
  ACG_WITH_TYPES_ANY(dtype, "concat_channels_fwd",
                     ACG_LAUNCH((concat_k<TA, TA, TB>), dim3(grid_for(rows * (ca + cb))), dim3(256), 0, acg::to_stream(stream), (const TA*)a,
                                (const TA*)b, (TB*)y, (long long)rows, ca, cb, 1, pitch));
  return acg::check_launch("concat_channels_fwd");
}

int32_t acg_slice_channels(const void* src, void* dst, float accumulate, int64_t rows, int32_t c_src, int32_t c_off,
                           int32_t c_dst, int32_t dtype, acg_stream_t stream) {
  ACG_REQUIRE(rows > 0 && c_src > 0 && c_dst > 0, ACG_ERR_INVALID_ARG, "slice_channels: non-positive size");
  ACG_REQUIRE(c_off >= 0 && c_off + c_dst <= c_src, ACG_ERR_INVALID_ARG, "slice_channels: range outside source");
  ACG_REQUIRE(src && dst, ACG_ERR_INVALID_ARG, "slice_channels: null pointer");
  ACG_WITH_TYPES(dtype, "slice_channels",
                 ACG_LAUNCH((slice_k<TA, TB>), dim3(grid_for(rows * c_dst)), dim3(256), 0, acg::to_stream(stream), (const TA*)src, (TB*)dst,
                            accumulate, (long long)rows, c_src, c_off, c_dst));
  return acg::check_launch("slice_channels");
}

int32_t acg_copy_many(const acg_copy_list* list, int32_t count, int32_t dtype, acg_stream_t stream) {
  ACG_REQUIRE_F32(dtype);     /* the sources; destinations per segment (dst_dtype) */
  ACG_REQUIRE(list && count >= 1 && count <= ACG_COPY_MAX, ACG_ERR_INVALID_ARG, "copy_many: 1..%d segments", ACG_COPY_MAX);
  CopyList l{};
  long long most = 0;
  for (int i = 0; i < count; ++i) {
    ACG_REQUIRE(list->src[i] && list->dst[i] && list->rows[i] > 0 && list->cols[i] > 0, ACG_ERR_INVALID_ARG, "copy_many: bad segment %d", i);
    const int pitch = list->dst_pitch[i] > 0 ? list->dst_pitch[i] : list->cols[i];
    ACG_REQUIRE(pitch >= list->cols[i], ACG_ERR_INVALID_ARG, "copy_many: segment %d pitch smaller than its row", i);
    ACG_REQUIRE(list->dst_dtype[i] == ACG_F32 || list->dst_dtype[i] == ACG_BF16, ACG_ERR_UNSUPPORTED, "copy_many: segment %d dtype", i);
    l.src[i] = (const float*)list->src[i]; l.dst[i] = list->dst[i]; l.half[i] = list->dst_dtype[i] == ACG_BF16;
    l.rows[i] = list->rows[i]; l.cols[i] = list->cols[i]; l.pitch[i] = pitch;
    ACG_REQUIRE(list->src_div[i] >= 0 && list->src_mod[i] >= 0 && list->rows[i] < (1ll << 31), ACG_ERR_INVALID_ARG, "copy_many: segment %d tiling", i);
    l.div[i] = list->src_div[i] > 0 ? list->src_div[i] : 1; l.mod[i] = list->src_mod[i];
    most = std::max<long long>(most, list->rows[i] * list->cols[i]);
  }
  ACG_LAUNCH(copy_many_k, dim3(grid_for(most / 4 + 1), count), dim3(256), 0, acg::to_stream(stream), l);
  return acg::check_launch("copy_many");
}

int32_t acg_add(const void* a, const void* b, void* y, int64_t n, int32_t dtype, acg_stream_t stream) {
  ACG_REQUIRE(dtype == ACG_F32 || dtype == ACG_BF16, ACG_ERR_UNSUPPORTED, "add: dtype %d", dtype);
  ACG_REQUIRE(n > 0 && a && b && y, ACG_ERR_INVALID_ARG, "add: bad argument");
  if (dtype == ACG_BF16)
    ACG_LAUNCH(add_k<__bf16>, dim3(grid_for(n / 4 + 1)), dim3(256), 0, acg::to_stream(stream), (const __bf16*)a, (const __bf16*)b, (__bf16*)y, (long long)n);
  else
    ACG_LAUNCH(add_k<float>, dim3(grid_for(n / 4 + 1)), dim3(256), 0, acg::to_stream(stream), (const float*)a, (const float*)b, (float*)y, (long long)n);
  return acg::check_launch("add");
}

}  // extern "C"
