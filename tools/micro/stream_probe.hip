// What can a per-channel elementwise pass (BatchNorm apply: y = relu((x - m[c]) * r[c] + b[c])) and a per-channel column-sum pass
// reach on MI355X at the tensor sizes of a batch-32 step (4 - 67 MB), launched the way the step launches them (N launches
// captured in a HIP graph)?  Variants of thread mapping, block size, loads in flight and grid size; prints us and GB/s.
//   hipcc -O3 --offload-arch=gfx950 -o stream_probe tools/micro/stream_probe.hip && ./stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// ---- elementwise ------------------------------------------------------------------------------------------------
// flat: thread t of block b owns float4 number (b * U + u) * NT + t, u < U; its 4 channels are fixed when NT * 4 % C == 0
template <int NT, int U, bool PARAMS>
__global__ __launch_bounds__(NT) void ew_flat(const float4* __restrict__ x, float4* __restrict__ y, const float* __restrict__ m,
                                              const float* __restrict__ r, const float* __restrict__ b, long long n4, int C) {
  const long long base = (long long)blockIdx.x * U * NT + threadIdx.x;
  float4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = x[min(base + (long long)u * NT, n4 - 1)];
  float4 mm = make_float4(0, 0, 0, 0), rr = make_float4(1, 1, 1, 1), bb = mm;
  if (PARAMS) {
    const int c = (int)((threadIdx.x * 4) % C);
    mm = *(const float4*)(m + c); rr = *(const float4*)(r + c); bb = *(const float4*)(b + c);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (base + (long long)u * NT < n4) {
      float4 t = v[u];
      t.x = fmaxf((t.x - mm.x) * rr.x + bb.x, 0.f); t.y = fmaxf((t.y - mm.y) * rr.y + bb.y, 0.f);
      t.z = fmaxf((t.z - mm.z) * rr.z + bb.z, 0.f); t.w = fmaxf((t.w - mm.w) * rr.w + bb.w, 0.f);
      y[base + (long long)u * NT] = t;
    }
  }
}

// tiled (the shape bn_apply_fwd has today): 8 channel lanes x NT / 8 row lanes, block (x = row chunk, y = 32-channel chunk)
template <int NT, int U>
__global__ __launch_bounds__(NT) void ew_tiled(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ m,
                                               const float* __restrict__ r, const float* __restrict__ b, long long R, int C) {
  const int cq = threadIdx.x & 7, rl = threadIdx.x >> 3, c = (blockIdx.y * 8 + cq) * 4;
  const long long rstep = (long long)gridDim.x * (NT / 8), r0 = (long long)blockIdx.x * (NT / 8) + rl;
  float4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = *(const float4*)(x + min(r0 + u * rstep, R - 1) * C + c);
  const float4 mm = *(const float4*)(m + c), rr = *(const float4*)(r + c), bb = *(const float4*)(b + c);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (r0 + u * rstep < R) {
      float4 t = v[u];
      t.x = fmaxf((t.x - mm.x) * rr.x + bb.x, 0.f); t.y = fmaxf((t.y - mm.y) * rr.y + bb.y, 0.f);
      t.z = fmaxf((t.z - mm.z) * rr.z + bb.z, 0.f); t.w = fmaxf((t.w - mm.w) * rr.w + bb.w, 0.f);
      *(float4*)(y + (r0 + u * rstep) * C + c) = t;
    }
  }
}

// ---- column sums (BatchNorm backward's first pass: two tensors in, per-block partial sums out) ----------------------------
// flat mapping: a block walks U consecutive chunks of NT float4; threads with equal (t * 4) % C hold the same channels:
// reduced through LDS at the end; partials [nblk][2][C]
template <int NT, int U>
__global__ __launch_bounds__(NT) void cs_flat(const float4* __restrict__ x, const float4* __restrict__ dy, float* __restrict__ part,
                                              long long n4, int C, int iters) {
  __shared__ float sh[2 * 4 * NT];
  float4 s1 = make_float4(0, 0, 0, 0), s2 = s1;
  for (int it = 0; it < iters; ++it) {
    const long long base = ((long long)blockIdx.x * iters + it) * U * NT + threadIdx.x;
    float4 a[U], d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const long long i = min(base + (long long)u * NT, n4 - 1); a[u] = x[i]; d[u] = dy[i]; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float w = base + (long long)u * NT < n4 ? 1.f : 0.f;
      s1.x += w * d[u].x; s1.y += w * d[u].y; s1.z += w * d[u].z; s1.w += w * d[u].w;
      s2.x += w * d[u].x * a[u].x; s2.y += w * d[u].y * a[u].y; s2.z += w * d[u].z * a[u].z; s2.w += w * d[u].w * a[u].w;
    }
  }
  // threads t and t' hold the same channels when (t - t') * 4 % C == 0: period P = C / 4 threads
  const int P = C / 4, t = threadIdx.x;
  float* s = sh;
  s[t * 8 + 0] = s1.x; s[t * 8 + 1] = s1.y; s[t * 8 + 2] = s1.z; s[t * 8 + 3] = s1.w;
  s[t * 8 + 4] = s2.x; s[t * 8 + 5] = s2.y; s[t * 8 + 6] = s2.z; s[t * 8 + 7] = s2.w;
  __syncthreads();
  if (t < P * 8) {            // P * 8 outputs, each the sum of NT / P entries
    const int q = t / 8, j = t % 8;
    float acc = 0.f;
    for (int k = q; k < NT; k += P) acc += s[k * 8 + j];
    part[(long long)blockIdx.x * 2 * C + (j >> 2) * C + q * 4 + (j & 3)] = acc;
  }
}

struct Timer {
  hipStream_t st;
  hipEvent_t e0, e1;
  Timer() { CK(hipStreamCreate(&st)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); }
  template <typename F>
  float us(F launch, int n_graph = 20, int reps = 5) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n_graph; ++i) launch(st);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1e3f / (n_graph * reps);
  }
};

int main() {
  Timer T;
  const long long Rs[] = {8192, 32768, 131072};
  const int C = 128;
  float *m, *r, *b;
  CK(hipMalloc(&m, 4096)); CK(hipMalloc(&r, 4096)); CK(hipMalloc(&b, 4096));
  CK(hipMemset(m, 0, 4096)); CK(hipMemset(r, 0, 4096)); CK(hipMemset(b, 0, 4096));
  for (long long R : Rs) {
    const long long n = R * C, n4 = n / 4;
    float *x, *dy, *y, *part;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&dy, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&part, 64 << 20));
    CK(hipMemset(x, 0x3c, n * 4)); CK(hipMemset(dy, 0x3c, n * 4));
    const double mb2 = 2.0 * n * 4 / 1e6;
    printf("== [%lld x %d] fp32: elementwise moves %.1f MB, column sums read %.1f MB\n", R, C, mb2, mb2);
#define EW_FLAT(NT, U, PARAMS) { const int grid = (int)((n4 + (long long)NT * U - 1) / ((long long)NT * U)); \
    float us = T.us([&](hipStream_t s) { hipLaunchKernelGGL((ew_flat<NT, U, PARAMS>), dim3(grid), dim3(NT), 0, s, (const float4*)x, (float4*)y, m, r, b, n4, C); }); \
    printf("  ew_flat  NT %4d U %d params %d grid %6d : %7.2f us %7.0f GB/s\n", NT, U, (int)PARAMS, grid, us, mb2 / us * 1e3); }
    EW_FLAT(256, 1, false) EW_FLAT(256, 2, false) EW_FLAT(256, 4, false) EW_FLAT(256, 8, false)
    EW_FLAT(512, 2, false) EW_FLAT(512, 4, false) EW_FLAT(1024, 2, false) EW_FLAT(1024, 4, false)
    EW_FLAT(256, 2, true) EW_FLAT(256, 4, true) EW_FLAT(256, 8, true) EW_FLAT(512, 4, true) EW_FLAT(1024, 4, true) EW_FLAT(1024, 8, true)
#define EW_TILED(NT, U, PASSES) { const dim3 grid((unsigned)((R + (NT / 8) * PASSES - 1) / ((NT / 8) * PASSES)), C / 32); \
    float us = T.us([&](hipStream_t s) { hipLaunchKernelGGL((ew_tiled<NT, U>), grid, dim3(NT), 0, s, (const float*)x, y, m, r, b, R, C); }); \
    printf("  ew_tiled NT %4d U %d grid %5d x %d : %7.2f us %7.0f GB/s\n", NT, U, grid.x, grid.y, us, mb2 / us * 1e3); }
    EW_TILED(256, 4, 4) EW_TILED(256, 8, 8) EW_TILED(1024, 4, 4) EW_TILED(512, 4, 4)
#define CS_FLAT(NT, U, ITERS) { const int grid = (int)((n4 + (long long)NT * U * ITERS - 1) / ((long long)NT * U * ITERS)); \
    float us = T.us([&](hipStream_t s) { hipLaunchKernelGGL((cs_flat<NT, U>), dim3(grid), dim3(NT), 0, s, (const float4*)x, (const float4*)dy, part, n4, C, ITERS); }); \
    printf("  cs_flat  NT %4d U %d iters %d grid %6d : %7.2f us %7.0f GB/s\n", NT, U, ITERS, grid, us, mb2 / us * 1e3); }
    CS_FLAT(256, 2, 1) CS_FLAT(256, 4, 1) CS_FLAT(256, 4, 2) CS_FLAT(256, 4, 4) CS_FLAT(512, 4, 1) CS_FLAT(512, 4, 2) CS_FLAT(1024, 4, 1) CS_FLAT(1024, 2, 2)
    CK(hipFree(x)); CK(hipFree(dy)); CK(hipFree(y)); CK(hipFree(part));
  }
  // the boundary itself: an empty kernel in the same graph
  return 0;
}
