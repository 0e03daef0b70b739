// Does a wave's own VALU / LDS work hide behind its MFMAs on gfx950?  One wave per SIMD runs 16-MFMA groups
// (v_mfma_f32_32x32x2_f32, one accumulator) with NV integer VALU ops and NL ds_read_b128 placed after every MFMA.
// Evidence for the conv K-loop design (profiles/).
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NV, int NL, bool ZERO, int NACC = 1>
__global__ __launch_bounds__(1024) void probe(float* out, const float* in, int iters) {
  __shared__ f4 lds[1024];
  lds[threadIdx.x & 1023] = f4{1.f, 2.f, 3.f, 4.f};
  lds[(threadIdx.x + 256) & 1023] = f4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  f32x16 accs[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) accs[i][r] = 0.f;
  float a = ZERO ? 0.f : in[threadIdx.x & 255], b = ZERO ? 0.f : in[(threadIdx.x & 255) + 256];
  unsigned x = threadIdx.x * 2654435761u, y = blockIdx.x + 12345u;
  f4 l = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      accs[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, accs[u % NACC], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) { x = x * 3u + y; y ^= x >> 3; }   // 2 dependent chains of cheap ops (mul by 3 = lshl_add)
#pragma unroll
      for (int q = 0; q < NL; ++q) l += lds[(x + q * 64 + threadIdx.x) & 1023];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = l[0] + l[1] + l[2] + l[3] + __builtin_bit_cast(float, (x ^ y) & 0x3fffffu);
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += accs[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int NL, bool ZERO, int NACC = 1>
void run(float* d, const float* in, int threads = 256) {
  const int iters = 2000, blocks = 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<NV, NL, ZERO, NACC>), dim3(blocks), dim3(threads), 0, 0, d, in, 10);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<NV, NL, ZERO, NACC>), dim3(blocks), dim3(threads), 0, 0, d, in, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("waves/SIMD %d  acc sets %d  VALU pairs/MFMA %2d  ds_read_b128/MFMA %d  data %s : %.1f ns of SIMD time per MFMA slot (%.0f cycles @2.4GHz)\n", threads / 256, NACC, NV, NL, ZERO ? "zero" : "rand",
         ms * 1e6 / (16.0 * iters) / (threads / 256), ms * 1e6 / (16.0 * iters) * 2.4 / (threads / 256));
}

int main() {
  float *d, *in;
  (void)hipMalloc(&d, 256 * 1024 * sizeof(float));
  (void)hipMalloc(&in, 512 * sizeof(float));
  float h[512];
  for (int i = 0; i < 512; ++i) h[i] = (float)((i * 7919) % 1000) / 500.f - 1.f;
  (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  run<0, 0, true>(d, in);
  run<0, 0, false>(d, in);
  run<2, 0, false>(d, in);
  run<4, 0, false>(d, in);
  run<6, 0, false>(d, in);
  run<8, 0, false>(d, in);
  run<12, 0, false>(d, in);
  run<0, 1, false>(d, in);
  run<0, 2, false>(d, in);
  run<4, 1, false>(d, in);
  run<0, 0, false, 4>(d, in);
  run<4, 0, false, 4>(d, in);
  run<8, 0, false, 4>(d, in);
  run<0, 2, false, 4>(d, in);
  run<4, 0, false, 2>(d, in);
  for (int th : {512, 1024}) {
    run<0, 0, false>(d, in, th);
    run<4, 0, false>(d, in, th);
    run<8, 0, false>(d, in, th);
    run<0, 2, false>(d, in, th);
    run<4, 1, false>(d, in, th);
  }
  return 0;
}
