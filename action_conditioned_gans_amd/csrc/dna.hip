// Dynamic Neural Advection tail (models.py:60-72) as a fused softmax + LDS-windowed k x k stencil.
//
// The reference materialises three [B,H,W,k*k,3] tensors; here a block owns a TY x 64 pixel tile of one
// image: the k*k logits of its pixels (one contiguous span per tile row) are streamed ONCE, coalesced,
// into LDS ([pixel][k*k | 1] - odd pitch, so a wave's per-pixel reads hit distinct banks), the image
// window (tile + halo, zero outside the frame = TF SAME zero padding) is staged once, and every thread
// then runs its pixel's max / exp / weighted gather out of LDS.  Algorithmic HBM traffic:
// (k*k + 2*C) floats per pixel forward, (2*k*k + 2*C) backward.
// The backward pass overwrites the staged logits in place with dlogits and streams them back coalesced.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "common.h"

namespace {

constexpr int TX = 64;

// CC: compile-time channel count (3 = RGB fast path) or 0 = runtime C.
template <int K, int TY, bool BWD, int CC>
__global__ __launch_bounds__(TX* TY) void dna_kernel(const float* __restrict__ logits, const float* __restrict__ img,
                                                     const float* __restrict__ dout, float* __restrict__ out,
                                                     int H, int W, int Crt) {
  constexpr int KK = K * K, S = KK | 1, NT = TX * TY, P = (K - 1) / 2;
  constexpr int WW = TX + K - 1, WH = TY + K - 1;
  __shared__ __attribute__((aligned(16))) float lg[NT * S];
  __shared__ float win[WH * WW * 4];
  const int C = CC ? CC : Crt;

  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY, b = blockIdx.z;
  const int txv = min(TX, W - x0);

  // image window (zero outside the frame): each window row is one contiguous span of the image row
  for (int wy = tid / 64; wy < WH; wy += NT / 64) {
    const int y = y0 - P + wy;
    const bool yok = y >= 0 && y < H;
    const long long rowbase = (long long)(b * H + (yok ? y : 0)) * W * C;
    for (int f = tid % 64; f < WW * C; f += 64) {
      const int gx = (x0 - P) * C + f;          // element index inside the image row
      win[wy * WW * C + f] = (yok && gx >= 0 && gx < W * C) ? img[rowbase + gx] : 0.f;
    }
  }
  // logits: TY contiguous spans of txv*KK floats; float4 when the span is 16-byte aligned and LDS is linear
  const long long pix0 = (long long)(b * H + y0) * W + x0;
  const bool vec = (S == KK) && ((pix0 * KK) % 4 == 0) && (((long long)W * KK) % 4 == 0) && ((txv * KK) % 4 == 0) &&
                   ((reinterpret_cast<uintptr_t>(logits) & 15) == 0);
  if (vec) {
    const int span4 = txv * KK / 4;
    for (int ty = 0; ty < TY; ++ty) {
      if (y0 + ty >= H) break;
      const float4* src = reinterpret_cast<const float4*>(logits + (pix0 + (long long)ty * W) * KK);
      float4* dst = reinterpret_cast<float4*>(lg + ty * TX * S);
      for (int i = tid; i < span4; i += NT) dst[i] = src[i];
    }
  } else {
    for (int idx = tid; idx < TY * TX * KK; idx += NT) {
      const int ty = idx / (TX * KK), e = idx - ty * (TX * KK);
      if (y0 + ty < H && e < txv * KK) lg[(ty * TX + e / KK) * S + e % KK] = logits[(pix0 + (long long)ty * W) * KK + e];
    }
  }
  __syncthreads();

  const int ty = tid / TX, tx = tid - ty * TX;
  const int y = y0 + ty, x = x0 + tx;
  const bool valid = y < H && x < W;
  float* l = lg + tid * S;
  const float* wbase = win + (ty * WW + tx) * C;

  if (valid) {
    float mx = l[0];
#pragma unroll
    for (int t = 1; t < KK; ++t) mx = fmaxf(mx, l[t]);
    const long long pix = (long long)(b * H + y) * W + x;
    if constexpr (!BWD) {
      float den = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const float e = __expf(l[i * K + j] - mx);
          const float* wp = wbase + (i * WW + j) * C;
          den += e;
          a0 += e * wp[0];
          if (C > 1) a1 += e * wp[1];
          if (C > 2) a2 += e * wp[2];
          if (C > 3) a3 += e * wp[3];
        }
      const float inv = 1.f / den;
      float* o = out + pix * C;
      o[0] = a0 * inv;
      if (C > 1) o[1] = a1 * inv;
      if (C > 2) o[2] = a2 * inv;
      if (C > 3) o[3] = a3 * inv;
    } else {
      const float* dO = dout + pix * C;
      const float d0 = dO[0], d1 = C > 1 ? dO[1] : 0.f, d2 = C > 2 ? dO[2] : 0.f, d3 = C > 3 ? dO[3] : 0.f;
      auto gfun = [&](int i, int j) {
        const float* wp = wbase + (i * WW + j) * C;
        float g = d0 * wp[0];
        if (C > 1) g += d1 * wp[1];
        if (C > 2) g += d2 * wp[2];
        if (C > 3) g += d3 * wp[3];
        return g;
      };
      float den = 0.f, dotn = 0.f;
#pragma unroll
      for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const float e = __expf(l[i * K + j] - mx);
          den += e;
          dotn += e * gfun(i, j);
        }
      const float inv = 1.f / den, dot = dotn * inv;
#pragma unroll
      for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const float mt = __expf(l[i * K + j] - mx) * inv;
          l[i * K + j] = mt * (gfun(i, j) - dot);
        }
    }
  }
  if constexpr (BWD) {
    __syncthreads();
    if (vec) {
      const int span4 = txv * KK / 4;
      for (int ty2 = 0; ty2 < TY; ++ty2) {
        if (y0 + ty2 >= H) break;
        float4* dstg = reinterpret_cast<float4*>(out + (pix0 + (long long)ty2 * W) * KK);
        const float4* srcl = reinterpret_cast<const float4*>(lg + ty2 * TX * S);
        for (int i = tid; i < span4; i += NT) dstg[i] = srcl[i];
      }
    } else {
      for (int idx = tid; idx < TY * TX * KK; idx += NT) {
        const int ty2 = idx / (TX * KK), e = idx - ty2 * (TX * KK);
        if (y0 + ty2 < H && e < txv * KK) out[(pix0 + (long long)ty2 * W) * KK + e] = lg[(ty2 * TX + e / KK) * S + e % KK];
      }
    }
  }
}

// Larger kernels (K >= 6; config 5 uses 11x11 = 121 taps, 484 B of logits per pixel): a thread per pixel would need the
// whole tile's logits in LDS (31 KB per 64 pixels: a handful of waves per CU, measured 12 % of the HBM peak).  Here a
// 16-lane ROW of a wave owns one pixel and the lanes split its taps (t = lane16 + 16 r): logits are read straight
// from global memory in 64-byte runs, stay in registers for both softmax passes, and the per-pixel maximum, sum and
// channel sums are 4-step butterflies inside the 16-lane row.  A block (4 waves) walks 64 consecutive pixels of one
// image row, 16 at a time; only the image window (11 x 74 x C floats) goes through LDS.
// CC: compile-time channel count (3 = RGB) or 0 = runtime C (kept small: its channel loops stay rolled).
template <int K, bool BWD, int CC>
__global__ __launch_bounds__(256) void dna_rows_kernel(const float* __restrict__ logits, const float* __restrict__ img,
                                                       const float* __restrict__ dout, float* __restrict__ out,
                                                       int H, int W, int Crt) {
  constexpr int KK = K * K, P = (K - 1) / 2, R = (KK + 15) / 16, WW = 64 + K - 1;
  const int C = CC ? CC : Crt;
  __shared__ float win[K * WW * 4];
  const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;        // 16 pixel groups per block
  const int x0 = blockIdx.x * 64, y = blockIdx.y, b = blockIdx.z;
  for (int i = tid; i < K * WW * C; i += 256) {
    const int c = i % C, p = i / C, wx = p % WW, wy = p / WW;
    const int gy = y - P + wy, gx = x0 - P + wx;
    win[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? img[(((long long)b * H + gy) * W + gx) * C + c] : 0.f;
  }
  __syncthreads();
  // all-reduce inside a 16-lane DPP row: xor 1, xor 2 (quad permutes), then the two mirror steps - four VALU
  // instructions with a DPP operand instead of four LDS-permute round trips
  auto dpp = [](float v, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, false));
  };
  using QP1 = std::integral_constant<int, 0xB1>;      // quad_perm [1,0,3,2]
  using QP2 = std::integral_constant<int, 0x4E>;      // quad_perm [2,3,0,1]
  using RHM = std::integral_constant<int, 0x141>;     // row_half_mirror
  using RM = std::integral_constant<int, 0x140>;      // row_mirror
  auto row_max = [&](float v) {
    v = fmaxf(v, dpp(v, QP1{})); v = fmaxf(v, dpp(v, QP2{})); v = fmaxf(v, dpp(v, RHM{})); v = fmaxf(v, dpp(v, RM{}));
    return v;
  };
  auto row_sum = [&](float v) {
    v += dpp(v, QP1{}); v += dpp(v, QP2{}); v += dpp(v, RHM{}); v += dpp(v, RM{});
    return v;
  };
  int woff[R];                                                        // window offset of this lane's taps (pixel 0 of the tile)
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int t = l16 + 16 * r, i = t / K, j = t - i * K;
    woff[r] = (i * WW + j) * C;
  }
  // the logits of all four pixels of this 16-lane row are requested up front (4 x R loads in flight per lane): the
  // per-pixel work below is a dependent chain (max -> exp -> sums) that would otherwise sit behind each round trip
  float lall[4][R];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int x = x0 + it * 16 + grp;
    const float* lp = logits + (((long long)b * H + y) * W + min(x, W - 1)) * KK;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int t = l16 + 16 * r;
      lall[it][r] = t < KK ? lp[t] : -3.0e38f;
    }
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int px = it * 16 + grp, x = x0 + px;
    if (x >= W) continue;                                             // uniform per 16-lane row: the DPP steps stay inside it
    const long long pix = ((long long)b * H + y) * W + x;
    float l[R];
    float mx = -3.0e38f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      l[r] = lall[it][r];
      mx = fmaxf(mx, l[r]);
    }
    mx = row_max(mx);
    float den = 0.f, a[4] = {0.f, 0.f, 0.f, 0.f}, d[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BWD) {
#pragma unroll
      for (int c = 0; c < 4; ++c) d[c] = c < C ? dout[pix * C + c] : 0.f;
    }
    float g[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int t = l16 + 16 * r;
      const float e = t < KK ? __expf(l[r] - mx) : 0.f;
      const float* wp = win + woff[r] + px * C;
      den += e;
      if constexpr (!BWD) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < C) a[c] += e * (t < KK ? wp[c] : 0.f);
      } else {
        float gg = 0.f;
        if (t < KK) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (c < C) gg += d[c] * wp[c];
        }
        g[r] = gg;
        a[0] += e * gg;                                               // numerator of dot = sum_t m_t g_t
      }
      l[r] = e;
    }
    den = row_sum(den);
    const float inv = 1.f / den;
    if constexpr (!BWD) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c < C) {
          const float v = row_sum(a[c]);
          if (l16 == c) out[pix * C + c] = v * inv;
        }
      }
    } else {
      const float dot = row_sum(a[0]) * inv;
      float* op = out + pix * KK;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int t = l16 + 16 * r;
        if (t < KK) op[t] = l[r] * inv * (g[r] - dot);
      }
    }
  }
}

template <int K, bool BWD>
int launch_k(const float* logits, const float* img, const float* dout, float* out, int B, int H, int W, int C, hipStream_t st) {
#ifndef ACG_DNA_ROWS_MIN
#define ACG_DNA_ROWS_MIN 6      // measured (profiles/r1/c_dna_microbench.txt): the row kernel wins from k = 6 (~2x), loses at 5
#endif
  if constexpr (K >= ACG_DNA_ROWS_MIN) {
    const dim3 grid((W + 63) / 64, H, B);
    if (C == 3) ACG_LAUNCH((dna_rows_kernel<K, BWD, 3>), grid, dim3(256), 0, st, logits, img, dout, out, H, W, C);
    else ACG_LAUNCH((dna_rows_kernel<K, BWD, 0>), grid, dim3(256), 0, st, logits, img, dout, out, H, W, C);
    return acg::check_launch(BWD ? "dna_bwd" : "dna_fwd");
  }
  constexpr int TY = K <= 6 ? 4 : 1;
  const dim3 grid((W + TX - 1) / TX, (H + TY - 1) / TY, B);
  if (C == 3) ACG_LAUNCH((dna_kernel<K, TY, BWD, 3>), grid, dim3(TX * TY), 0, st, logits, img, dout, out, H, W, C);
  else ACG_LAUNCH((dna_kernel<K, TY, BWD, 0>), grid, dim3(TX * TY), 0, st, logits, img, dout, out, H, W, C);
  return acg::check_launch(BWD ? "dna_bwd" : "dna_fwd");
}

template <bool BWD>
int dispatch(int k, const float* logits, const float* img, const float* dout, float* out, int B, int H, int W, int C, hipStream_t st) {
  switch (k) {
#define ACG_DNA_CASE(KV) case KV: return launch_k<KV, BWD>(logits, img, dout, out, B, H, W, C, st);
    ACG_DNA_CASE(1) ACG_DNA_CASE(2) ACG_DNA_CASE(3) ACG_DNA_CASE(4) ACG_DNA_CASE(5) ACG_DNA_CASE(6)
    ACG_DNA_CASE(7) ACG_DNA_CASE(8) ACG_DNA_CASE(9) ACG_DNA_CASE(10) ACG_DNA_CASE(11)
#undef ACG_DNA_CASE
    default: return acg::fail(ACG_ERR_UNSUPPORTED, "dna: ksize %d outside 1..11", k);
  }
}

int check(const char* who, int B, int H, int W, int C, int k) {
  ACG_REQUIRE(B > 0 && H > 0 && W > 0, ACG_ERR_INVALID_ARG, "%s: non-positive size", who);
  ACG_REQUIRE(C >= 1 && C <= 4, ACG_ERR_INVALID_ARG, "%s: channels %d outside 1..4", who, C);
  ACG_REQUIRE(B <= 65535 && H <= 65535, ACG_ERR_UNSUPPORTED, "%s: grid too large", who);
  ACG_REQUIRE((long long)B * H * W * k * k < 2147483647ll, ACG_ERR_UNSUPPORTED, "%s: tensor exceeds 2^31 elements", who);
  return ACG_OK;
}

}  // namespace

extern "C" {

int32_t acg_dna_fwd(const void* logits, const void* image, void* out, int32_t B, int32_t H, int32_t W, int32_t C,
                    int32_t k, int32_t dtype, acg_stream_t stream) {
  ACG_REQUIRE_F32(dtype);
  if (int rc = check("dna_fwd", B, H, W, C, k)) return rc;
  ACG_REQUIRE(logits && image && out, ACG_ERR_INVALID_ARG, "dna_fwd: null pointer");
  return dispatch<false>(k, (const float*)logits, (const float*)image, nullptr, (float*)out, B, H, W, C, acg::to_stream(stream));
}

int32_t acg_dna_bwd(const void* logits, const void* image, const void* dout, void* dlogits, int32_t B, int32_t H,
                    int32_t W, int32_t C, int32_t k, int32_t dtype, acg_stream_t stream) {
  ACG_REQUIRE_F32(dtype);
  if (int rc = check("dna_bwd", B, H, W, C, k)) return rc;
  ACG_REQUIRE(logits && image && dout && dlogits, ACG_ERR_INVALID_ARG, "dna_bwd: null pointer");
  return dispatch<true>(k, (const float*)logits, (const float*)image, (const float*)dout, (float*)dlogits, B, H, W, C,
                        acg::to_stream(stream));
}

}  // extern "C"
