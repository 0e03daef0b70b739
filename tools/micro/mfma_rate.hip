// Issue-rate probe for v_mfma_f32_32x32x2_f32 on gfx950: cycles per MFMA for one wave per SIMD vs several, with
// 1, 2 or 4 independent accumulators.  Evidence for the conv K-loop design (profiles/).
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/micro/mfma_rate.hip -o gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(1024) void probe(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % NACC], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int threads, int blocks, float* d) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(threads), 0, 0, d, 10);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_wave = 16.0 * iters;
  const double waves_per_simd = (double)threads / 64 / 4 * ((double)blocks / 256);
  const double ns_per_mfma = ms * 1e6 / mfma_per_wave / waves_per_simd;   // pipe time per MFMA on one SIMD
  const double tflops = (double)blocks * threads / 64 * mfma_per_wave * 4096.0 / (ms * 1e-3) / 1e12;
  printf("acc sets %d  threads/block %4d  blocks %4d  waves/SIMD %.0f : %.3f ms  %.1f ns per MFMA per SIMD (%.0f cycles @2.4GHz)  %.1f TFLOP/s\n",
         NACC, threads, blocks, waves_per_simd, ms, ns_per_mfma, ns_per_mfma * 2.4, tflops);
}

int main() {
  float* d;
  hipMalloc(&d, 1024 * 2048 * sizeof(float));
  for (int threads : {256, 512, 1024}) {
    run<1>(threads, 256, d);
    run<2>(threads, 256, d);
    run<4>(threads, 256, d);
  }
  run<1>(256, 512, d);
  run<4>(256, 512, d);
  run<4>(256, 2048, d);
  hipFree(d);
  return 0;
}
