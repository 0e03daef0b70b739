// What does ONE CU's load path deliver for the conv K-loop's access pattern - 8 gathered rows x 128 contiguous bytes per
// wave-instruction, 16 B per lane, L2-resident source - (a) as LDS-DMA pieces (buffer_load_dwordx4 ... lds), (b) as ordinary
// buffer_load_dwordx4 into VGPRs, (c) both kinds mixed half / half, each with and without MFMAs running beside them?
// One 512-thread block per CU (8 waves, 150 KB of LDS so that no second block fits), every wave issues PIECES loads per
// iteration and keeps one iteration in flight (counted vmcnt), exactly as conv_bf16_glds.h does.  Prints GB/s per CU and
// bytes per clock at 2.4 GHz.
//   hipcc -O3 --offload-arch=gfx950 -o load_path_probe tools/micro/load_path_probe.hip && ./load_path_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;

constexpr int NT = 512, PIECES = 6, LDS_BYTES = 150 * 1024;

// MODE 0: all PIECES as LDS-DMA; 1: all as register loads.  MFMA: 16 v_mfma_f32_32x32x16_bf16 per iteration; DSR: 16 ds_read_b128 per
// wave and iteration (the fragment reads of a 64 x 64 wave tile); BAR: one block barrier per iteration
template <int MODE, bool MFMA, bool DSR, bool BAR>
__global__ __launch_bounds__(NT) void probe(const char* __restrict__ src, long long src_bytes, int foot_rows, int iters, float* __restrict__ sink) {
  extern __shared__ __align__(16) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)src_bytes, 0x00020000);
  // this block's footprint: foot_rows rows of 128 B starting at block * foot_rows (wrapping inside the buffer)
  const long long base_row = (long long)blockIdx.x * foot_rows;
  const long long total_rows = src_bytes / 128;
  unsigned seed = 2654435761u * (unsigned)(blockIdx.x * 8 + wave + 1);
  f32x16 acc0 = {0}, acc1 = {0};
  f4 keep = {0, 0, 0, 0};
  bf8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane - j)); }
  constexpr int NDMA = MODE == 0 ? PIECES : 0, NREG = PIECES - NDMA;
  f4 r0[NREG > 0 ? NREG : 1], r1[NREG > 0 ? NREG : 1];
  // register loads are inline asm and every wait is written by hand: beside an LDS-DMA in flight hipcc would wait vmcnt(0)
  // for any ordinary load (cdna_hip_programming.md section 5, trap 4b) and the pipeline would drain every iteration
  auto issue = [&](auto bufc, f4 (&r)[NREG > 0 ? NREG : 1]) {
    constexpr int slotbuf = decltype(bufc)::value;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      seed = seed * 1664525u + 1013904223u;
      // 8 rows per wave-instruction: rows (rnd + lane / 8 * 2) - neighbouring output pixels of a stride-2 conv are 2 pixels apart
      const long long row = (base_row + ((seed >> 8) & (unsigned)(foot_rows - 1)) + 2 * (lane >> 3)) & (total_rows - 1);     // both powers of two
      unsigned off = (unsigned)(row * 128 + (lane & 7) * 16);
      asm volatile("" : "+v"(off));
      if (i < NDMA) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + ((slotbuf * PIECES + i) * 8 + wave) * 1024), 16, off, 0, 0, 0);
      } else {
        r[i - NDMA] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
      }
    }
  };
  auto consume = [&](f4 (&r)[NREG > 0 ? NREG : 1]) {
    if constexpr (MFMA) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
      }
    }
    if constexpr (DSR) {
      const f4* cells = reinterpret_cast<const f4*>(smem);
#pragma unroll
      for (int q = 0; q < 16; ++q) { f4 v = cells[(lane & 31) * 8 + ((q + (lane >> 5) * 4) ^ ((lane >> 1) & 7)) % 8 + 256 * (q & 3) + 2048 * wave]; asm volatile("" :: "v"(v)); }
    }
    if constexpr (NREG > 0) {
#pragma unroll
      for (int i = 0; i < NREG; ++i) keep += r[i];        // hipcc's own counted wait (no LDS-DMA in flight in this mode)
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES) : "memory");      // the previous iteration's loads are back
    }
    if constexpr (BAR) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
  };
  issue(std::integral_constant<int, 0>{}, r0);
  for (int it = 0; it < iters; it += 2) {
    issue(std::integral_constant<int, 1>{}, r1);
    consume(r0);
    issue(std::integral_constant<int, 0>{}, r0);
    consume(r1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float s = keep.x + keep.y + keep.z + keep.w + acc0[0] + acc1[5] + reinterpret_cast<float*>(smem)[tid];
  if (s == 123.456f) sink[0] = s;
}

template <typename F>
float time_us(F launch, hipStream_t st, int reps = 5) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipStream_t st; CK(hipStreamCreate(&st));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const long long src_bytes = 1ll << 30;
  char* src; float* sink;
  CK(hipMalloc(&src, src_bytes)); CK(hipMemset(src, 0x3c, src_bytes)); CK(hipMalloc(&sink, 64));
  const int iters = 2000;
#define RUN(MODE, MFMA, DSR, BAR, FOOT) { \
    CK(hipFuncSetAttribute((const void*)probe<MODE, MFMA, DSR, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)); \
    float us = time_us([&]() { hipLaunchKernelGGL((probe<MODE, MFMA, DSR, BAR>), dim3(ncu), dim3(NT), LDS_BYTES, st, (const char*)src, src_bytes, FOOT, iters, sink); }, st); \
    const double bytes_cu = (double)iters * PIECES * 8 * 1024; \
    printf("  %s mfma %d ds_read %d barrier %d footprint %6d KB/CU : %8.1f us  %6.1f GB/s per CU  %5.1f B/clk @2.4GHz  (%.2f TB/s chip)  %5.0f clk/iteration\n", \
           MODE == 0 ? "LDS-DMA  " : "registers", (int)MFMA, (int)DSR, (int)BAR, FOOT / 8, us, bytes_cu / us / 1e3, bytes_cu / us / 1e3 / 2.4, bytes_cu * ncu / us / 1e6, us * 2400.0 / iters); }
  printf("%d CUs, %d pieces (1 KB each) per wave and iteration, 8 waves per CU, one iteration in flight\n", ncu, PIECES);
  const int foots[] = {512, 1024, 8192, 262144};      // rows of 128 B per CU: 64 KB, 128 KB (about one tile's window), 1 MB, 32 MB (beyond L2 and MALL)
  for (int f : foots) {
    RUN(0, false, false, false, f) RUN(1, false, false, false, f)
    RUN(0, true, false, false, f) RUN(1, true, false, false, f)
    RUN(0, true, true, false, f) RUN(1, true, true, false, f)
    RUN(0, true, true, true, f) RUN(1, true, true, true, f)
  }
  return 0;
}
