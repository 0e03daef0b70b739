// Dynamic Neural Advection tail (models.py:60-72) as a fused softmax + LDS-windowed k x k stencil.
//
// The reference materialises three [B,H,W,k*k,3] tensors; here a block owns a TY x 64 pixel tile of one
// image: the k*k logits of its pixels (one contiguous span per tile row) are streamed ONCE, coalesced,
// into LDS ([pixel][k*k | 1] - odd pitch, so a wave's per-pixel reads hit distinct banks), the image
// window (tile + halo, zero outside the frame = TF SAME zero padding) is staged once, and every thread
// then runs its pixel's max / exp / weighted gather out of LDS.  Algorithmic HBM traffic:
// (k*k + 2*C) floats per pixel forward, (2*k*k + 2*C) backward.
// The backward pass overwrites the staged logits in place with dlogits and streams them back coalesced.
// The bias of the producing layer (g/tconv4, models.py:54-59: no BatchNorm, no activation) is folded in: the kernels
// take softmax(logits + bias), and backward leaves per-block column sums of dlogits from which ONE small launch forms
// dbias - the separate bias pass over the 13 MB logits tensor and its backward are gone.
// TL: storage type of logits / dlogits (float: dense k*k per pixel; __bf16: pitch round8(k*k), pad taps untouched).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "common.h"

namespace {

constexpr int TX = 64;

// The frame's second home (train.py:63-66: D(fake) reads concat(current frame, generated frame)): forward also writes the
// frame into channels [off, off + C) of a pitched tensor of the conv storage type - the discriminator's input - and backward
// adds the gradient that arrives through those channels to dout; ptr == nullptr: none.  When the tensor is exactly
// concat(image, frame) at a pitch of 8 (off == C == 3), forward writes the WHOLE pixel - image channels and zero pad included.
struct Second {
  void* ptr;
  int pitch, off, half;
};
__device__ __forceinline__ void second_store(const Second& s2, long long pix, int c, float v) {
  if (s2.half) reinterpret_cast<__bf16*>(s2.ptr)[pix * s2.pitch + s2.off + c] = (__bf16)v;
  else reinterpret_cast<float*>(s2.ptr)[pix * s2.pitch + s2.off + c] = v;
}
// concat(image, frame) as ONE vector store per pixel when the second home is exactly that (3 + 3 channels at a pitch of 8:
// the discriminator input of train.py:63-66, the image channels being the pixel's own window centre) - three scalar stores
// at a 32-byte stride cost the forward kernel a quarter of its time; other layouts take the scalar path
__device__ __forceinline__ void second_store_pixel(const Second& s2, long long pix, int C, const float* img_c, const float (&f)[4]) {
  if (C == 3 && s2.off == 3 && s2.pitch == 8) {
    if (s2.half) {
      typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
      const bf8v v = {(__bf16)img_c[0], (__bf16)img_c[1], (__bf16)img_c[2], (__bf16)f[0], (__bf16)f[1], (__bf16)f[2], (__bf16)0.f, (__bf16)0.f};
      *reinterpret_cast<bf8v*>(reinterpret_cast<__bf16*>(s2.ptr) + pix * 8) = v;
    } else {
      float4* o = reinterpret_cast<float4*>(reinterpret_cast<float*>(s2.ptr) + pix * 8);
      o[0] = make_float4(img_c[0], img_c[1], img_c[2], f[0]);
      o[1] = make_float4(f[1], f[2], 0.f, 0.f);
    }
    return;
  }
  for (int c = 0; c < C; ++c) second_store(s2, pix, c, f[c]);
}
__device__ __forceinline__ float second_load(const Second& s2, long long pix, int c) {
  return s2.half ? (float)reinterpret_cast<const __bf16*>(s2.ptr)[pix * s2.pitch + s2.off + c]
                 : reinterpret_cast<const float*>(s2.ptr)[pix * s2.pitch + s2.off + c];
}

template <typename TL>
__host__ __device__ constexpr int logit_pitch(int kk) { return sizeof(TL) == 2 ? (kk + 7) & ~7 : kk; }

// CC: compile-time channel count (3 = RGB fast path) or 0 = runtime C.
template <int K, int TY, bool BWD, int CC, typename TL>
__global__ __launch_bounds__(TX* TY) void dna_kernel(const TL* __restrict__ logits, const float* __restrict__ bias,
                                                     const float* __restrict__ img, const float* __restrict__ dout,
                                                     void* __restrict__ outv, float* __restrict__ bpart, int H, int W, int Crt,
                                                     const Second s2) {
  constexpr int KK = K * K, S = KK | 1, NT = TX * TY, P = (K - 1) / 2, LP = logit_pitch<TL>(KK);
  constexpr int WW = TX + K - 1, WH = TY + K - 1;
  constexpr bool F32 = sizeof(TL) == 4;
  __shared__ __attribute__((aligned(16))) float lg[NT * S];
  __shared__ float win[WH * WW * 4];
  __shared__ float bs[KK];
  const int C = CC ? CC : Crt;
  float* const out = reinterpret_cast<float*>(outv);     // forward: the frame (float32)
  TL* const dlog = reinterpret_cast<TL*>(outv);          // backward: dlogits

  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY, b = blockIdx.z;
  const int txv = min(TX, W - x0);
  if (tid < KK) bs[tid] = bias ? bias[tid] : 0.f;

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
  const bool vec = F32 && (S == KK) && ((pix0 * KK) % 4 == 0) && (((long long)W * KK) % 4 == 0) && ((txv * KK) % 4 == 0) &&
                   ((reinterpret_cast<uintptr_t>(logits) & 15) == 0);
  if constexpr (F32) {
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
        if (y0 + ty < H && e < txv * KK) lg[(ty * TX + e / KK) * S + e % KK] = acg::ldf(logits + (pix0 + (long long)ty * W) * KK + e);
      }
    }
  } else {
    // bf16: a tile row is one contiguous span of txv * LP elements; 8-byte pieces (4 taps of one pixel, LP % 4 == 0)
    constexpr int Q = LP / 4;
    for (int idx = tid; idx < TY * TX * Q; idx += NT) {
      const int ty = idx / (TX * Q), e = idx - ty * (TX * Q), px = e / Q, q = e - px * Q;
      if (y0 + ty < H && px < txv && 4 * q < KK) {
        float v[4];
        acg::ldv<4>(logits + (pix0 + (long long)ty * W + px) * LP + 4 * q, v);
        float* d = lg + (ty * TX + px) * S + 4 * q;
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (4 * q + u < KK) d[u] = v[u];
      }
    }
  }
  __syncthreads();

  const int ty = tid / TX, tx = tid - ty * TX;
  const int y = y0 + ty, x = x0 + tx;
  const bool valid = y < H && x < W;
  float* l = lg + tid * S;
  const float* wbase = win + (ty * WW + tx) * C;

  if (valid) {
    // round 5: the pixel's logits live in REGISTERS from here on (one LDS read each; round 4 added the bias in place - a
    // write-back nobody needed in the forward pass - and re-read every logit for the maximum and again for the exponentials;
    // backward also evaluated every tap's window dot product twice)
    float v[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) v[t] = l[t] + bs[t];    // softmax(logits + bias)
    float mx = v[0];
#pragma unroll
    for (int t = 1; t < KK; ++t) mx = fmaxf(mx, v[t]);
    const long long pix = (long long)(b * H + y) * W + x;
    if constexpr (!BWD) {
      float den = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const float e = __expf(v[i * K + j] - mx);
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
      if (s2.ptr) {
        const float fr[4] = {a0 * inv, a1 * inv, a2 * inv, a3 * inv};
        second_store_pixel(s2, pix, C, wbase + (P * WW + P) * C, fr);      // window centre = this pixel of the image
      }
    } else {
      const float* dO = dout + pix * C;
      float d0 = dO[0], d1 = C > 1 ? dO[1] : 0.f, d2 = C > 2 ? dO[2] : 0.f, d3 = C > 3 ? dO[3] : 0.f;
      if (s2.ptr) {
        d0 += second_load(s2, pix, 0);
        if (C > 1) d1 += second_load(s2, pix, 1);
        if (C > 2) d2 += second_load(s2, pix, 2);
        if (C > 3) d3 += second_load(s2, pix, 3);
      }
      float gv[KK];
      float den = 0.f, dotn = 0.f;
#pragma unroll
      for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const float* wp = wbase + (i * WW + j) * C;
          float g = d0 * wp[0];
          if (C > 1) g += d1 * wp[1];
          if (C > 2) g += d2 * wp[2];
          if (C > 3) g += d3 * wp[3];
          const float e = __expf(v[i * K + j] - mx);
          gv[i * K + j] = g;
          v[i * K + j] = e;
          den += e;
          dotn += e * g;
        }
      const float inv = 1.f / den, dot = dotn * inv;
#pragma unroll
      for (int t = 0; t < KK; ++t) l[t] = (v[t] * inv) * (gv[t] - dot);
    }
  } else if constexpr (BWD) {
#pragma unroll
    for (int t = 0; t < KK; ++t) l[t] = 0.f;             // pixels outside the frame: nothing to add to dbias
  }
  if constexpr (BWD) {
    __syncthreads();
    if constexpr (F32) {
      if (vec) {
        const int span4 = txv * KK / 4;
        for (int ty2 = 0; ty2 < TY; ++ty2) {
          if (y0 + ty2 >= H) break;
          float4* dstg = reinterpret_cast<float4*>(dlog + (pix0 + (long long)ty2 * W) * KK);
          const float4* srcl = reinterpret_cast<const float4*>(lg + ty2 * TX * S);
          for (int i = tid; i < span4; i += NT) dstg[i] = srcl[i];
        }
      } else {
        for (int idx = tid; idx < TY * TX * KK; idx += NT) {
          const int ty2 = idx / (TX * KK), e = idx - ty2 * (TX * KK);
          if (y0 + ty2 < H && e < txv * KK) acg::stf(dlog + (pix0 + (long long)ty2 * W) * KK + e, lg[(ty2 * TX + e / KK) * S + e % KK]);
        }
      }
    } else {
      constexpr int Q = LP / 4;
      for (int idx = tid; idx < TY * TX * Q; idx += NT) {
        const int ty2 = idx / (TX * Q), e = idx - ty2 * (TX * Q), px = e / Q, q = e - px * Q;
        if (y0 + ty2 < H && px < txv && 4 * q < KK) {
          const float* sl = lg + (ty2 * TX + px) * S + 4 * q;
          float v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = 4 * q + u < KK ? sl[u] : 0.f;
          acg::stv<4>(dlog + (pix0 + (long long)ty2 * W + px) * LP + 4 * q, v);
        }
      }
    }
    if (bpart) {   // column sums of this block's dlogits: thread (seg, t) sums 32 pixels, 8 (NT / 32) segments through LDS
      const int t = tid & 31, seg = tid >> 5;
      float sacc = 0.f;
      if (t < KK)
        for (int px = seg * 32; px < seg * 32 + 32; ++px) sacc += lg[px * S + t];
      __syncthreads();
      if (t < KK) win[seg * 32 + t] = sacc;
      __syncthreads();
      if (tid < KK) {
        float tot = 0.f;
        for (int g2 = 0; g2 < NT / 32; ++g2) tot += win[g2 * 32 + tid];
        bpart[(long long)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * KK + tid] = tot;
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
template <int K, bool BWD, int CC, typename TL>
__global__ __launch_bounds__(256) void dna_rows_kernel(const TL* __restrict__ logits, const float* __restrict__ bias,
                                                       const float* __restrict__ img, const float* __restrict__ dout,
                                                       void* __restrict__ outv, float* __restrict__ bpart, int H, int W, int Crt,
                                                       const Second s2) {
  // E taps per lane and load: one float, or two adjacent bf16 (so that a 16-lane row still reads 64-byte runs)
  constexpr int E = sizeof(TL) == 2 ? 2 : 1;
  constexpr int KK = K * K, P = (K - 1) / 2, R = (KK + 16 * E - 1) / (16 * E), WW = 64 + K - 1, LP = logit_pitch<TL>(KK);
  const int C = CC ? CC : Crt;
  __shared__ float win[K * WW * 4];
  __shared__ float bsum[BWD ? 16 * R * 16 * E : 1];
  float* const out = reinterpret_cast<float*>(outv);
  TL* const dlog = reinterpret_cast<TL*>(outv);
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
  // this lane's taps: t(r, e) = E * (l16 + 16 r) + e
  int woff[R][E];
  float bia[R][E], bacc[R][E];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int t = E * (l16 + 16 * r) + e, i = t / K, j = t - i * K;
      woff[r][e] = (i * WW + j) * C;
      bia[r][e] = (bias && t < KK) ? bias[t] : 0.f;
      bacc[r][e] = 0.f;
    }
  // the logits of all four pixels of this 16-lane row are requested up front (4 x R loads in flight per lane): the
  // per-pixel work below is a dependent chain (max -> exp -> sums) that would otherwise sit behind each round trip
  float lall[4][R][E];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int x = x0 + it * 16 + grp;
    const TL* lp = logits + (((long long)b * H + y) * W + min(x, W - 1)) * LP;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int t0 = E * (l16 + 16 * r);
      if constexpr (E == 2) {
        typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
        const bf2v v = t0 < LP ? *reinterpret_cast<const bf2v*>(lp + t0) : bf2v{(__bf16)0.f, (__bf16)0.f};
        lall[it][r][0] = t0 < KK ? (float)v[0] + bia[r][0] : -3.0e38f;
        lall[it][r][1] = t0 + 1 < KK ? (float)v[1] + bia[r][1] : -3.0e38f;
      } else {
        lall[it][r][0] = t0 < KK ? acg::ldf(lp + t0) + bia[r][0] : -3.0e38f;
      }
    }
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int px = it * 16 + grp, x = x0 + px;
    if (x >= W) continue;                                             // uniform per 16-lane row: the DPP steps stay inside it
    const long long pix = ((long long)b * H + y) * W + x;
    float l[R][E];
    float mx = -3.0e38f;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int e = 0; e < E; ++e) {
        l[r][e] = lall[it][r][e];
        mx = fmaxf(mx, l[r][e]);
      }
    mx = row_max(mx);
    float den = 0.f, a[4] = {0.f, 0.f, 0.f, 0.f}, d[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BWD) {
#pragma unroll
      for (int c = 0; c < 4; ++c) d[c] = c < C ? dout[pix * C + c] + (s2.ptr ? second_load(s2, pix, c) : 0.f) : 0.f;
    }
    float g[R][E];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int t = E * (l16 + 16 * r) + e;
        const float ex = t < KK ? __expf(l[r][e] - mx) : 0.f;
        const float* wp = win + woff[r][e] + px * C;
        den += ex;
        if constexpr (!BWD) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (c < C) a[c] += ex * (t < KK ? wp[c] : 0.f);
        } else {
          float gg = 0.f;
          if (t < KK) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (c < C) gg += d[c] * wp[c];
          }
          g[r][e] = gg;
          a[0] += ex * gg;                                            // numerator of dot = sum_t m_t g_t
        }
        l[r][e] = ex;
      }
    den = row_sum(den);
    const float inv = 1.f / den;
    if constexpr (!BWD) {
      float fr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c < C) {
          fr[c] = row_sum(a[c]) * inv;                                     // a butterfly: every lane of the row holds the sum
          if (l16 == c) out[pix * C + c] = fr[c];
        }
      }
      if (s2.ptr && l16 == 4) second_store_pixel(s2, pix, C, win + (P * WW + P + px) * C, fr);
    } else {
      const float dot = row_sum(a[0]) * inv;
      TL* op = dlog + pix * LP;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int t0 = E * (l16 + 16 * r);
        float dv[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
          dv[e] = t0 + e < KK ? l[r][e] * inv * (g[r][e] - dot) : 0.f;
          bacc[r][e] += dv[e];
        }
        if constexpr (E == 2) {
          typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
          if (t0 < KK) *reinterpret_cast<bf2v*>(op + t0) = bf2v{(__bf16)dv[0], (__bf16)dv[1]};   // t0 + 1 <= LP - 1: a zero pad at most
        } else {
          if (t0 < KK) acg::stf(op + t0, dv[0]);
        }
      }
    }
  }
  if constexpr (BWD) {
    if (bpart) {   // dbias partial of this block: the 16 pixel groups' sums of each tap through LDS
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < E; ++e) bsum[grp * (R * 16 * E) + E * (l16 + 16 * r) + e] = bacc[r][e];
      __syncthreads();
      if (tid < KK) {
        float tot = 0.f;
        for (int g2 = 0; g2 < 16; ++g2) tot += bsum[g2 * (R * 16 * E) + tid];
        bpart[(long long)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * KK + tid] = tot;
      }
    }
  }
}

// out[seg][t] = (acc * out[seg][t] +) sum of part[row][t] over this segment's rows; block (x = 32 taps, y = segment) of
// 256 threads = 8 row lanes x 32 taps.  One segment writes dbias directly; large partial counts (config 5: 8192 rows)
// go through kSegs segment sums and a second, single-segment launch.
constexpr int kSegs = 32;
__global__ __launch_bounds__(256) void dna_dbias_sum(const float* __restrict__ part, float* __restrict__ out, float acc, int KK, int nrows) {
  __shared__ double sh[256];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
  const int per = (nrows + gridDim.y - 1) / gridDim.y, r0 = blockIdx.y * per, r1 = min(nrows, r0 + per);
  double s2 = 0.0;
  if (c < KK) {
#pragma unroll 4
    for (int b2 = r0 + rl; b2 < r1; b2 += 8) s2 += part[(long long)b2 * KK + c];
  }
  sh[threadIdx.x] = s2;
  __syncthreads();
  if (rl == 0 && c < KK) {
#pragma unroll
    for (int r = 1; r < 8; ++r) s2 += sh[r * 32 + cl];
    float* o = out + (long long)blockIdx.y * KK + c;
    *o = (acc != 0.f ? acc * *o : 0.f) + (float)s2;
  }
}

#ifndef ACG_DNA_ROWS_MIN
#define ACG_DNA_ROWS_MIN 6      // measured (profiles/r1/c_dna_microbench.txt): the row kernel wins from k = 6 (~2x), loses at 5
#endif

// grid of the kernel that serves ksize k (also the number of dbias partial rows)
#ifndef ACG_DNA_TY
#define ACG_DNA_TY 4            // tile rows of dna_kernel (64 x TY pixels, one thread each)
#endif
dim3 dna_grid(int k, int B, int H, int W) {
  if (k >= ACG_DNA_ROWS_MIN) return dim3((W + 63) / 64, H, B);
  return dim3((W + TX - 1) / TX, (H + ACG_DNA_TY - 1) / ACG_DNA_TY, B);
}

template <int K, bool BWD, typename TL>
int launch_k(const TL* logits, const float* bias, const float* img, const float* dout, void* out, float* bpart, int B, int H, int W,
             int C, hipStream_t st, const Second& s2) {
  const dim3 grid = dna_grid(K, B, H, W);
  if constexpr (K >= ACG_DNA_ROWS_MIN) {
    if (C == 3) ACG_LAUNCH((dna_rows_kernel<K, BWD, 3, TL>), grid, dim3(256), 0, st, logits, bias, img, dout, out, bpart, H, W, C, s2);
    else ACG_LAUNCH((dna_rows_kernel<K, BWD, 0, TL>), grid, dim3(256), 0, st, logits, bias, img, dout, out, bpart, H, W, C, s2);
  } else {
    constexpr int TY = ACG_DNA_TY;
    if (C == 3) ACG_LAUNCH((dna_kernel<K, TY, BWD, 3, TL>), grid, dim3(TX * TY), 0, st, logits, bias, img, dout, out, bpart, H, W, C, s2);
    else ACG_LAUNCH((dna_kernel<K, TY, BWD, 0, TL>), grid, dim3(TX * TY), 0, st, logits, bias, img, dout, out, bpart, H, W, C, s2);
  }
  return acg::check_launch(BWD ? "dna_bwd" : "dna_fwd");
}

template <bool BWD, typename TL>
int dispatch(int k, const void* logits, const float* bias, const float* img, const float* dout, void* out, float* bpart, int B, int H,
             int W, int C, hipStream_t st, const Second& s2) {
  switch (k) {
#define ACG_DNA_CASE(KV) case KV: return launch_k<KV, BWD, TL>((const TL*)logits, bias, img, dout, out, bpart, B, H, W, C, st, s2);
    ACG_DNA_CASE(1) ACG_DNA_CASE(2) ACG_DNA_CASE(3) ACG_DNA_CASE(4) ACG_DNA_CASE(5) ACG_DNA_CASE(6)
    ACG_DNA_CASE(7) ACG_DNA_CASE(8) ACG_DNA_CASE(9) ACG_DNA_CASE(10) ACG_DNA_CASE(11)
#undef ACG_DNA_CASE
    default: return acg::fail(ACG_ERR_UNSUPPORTED, "dna: ksize %d outside 1..11", k);
  }
}

int check(const char* who, int B, int H, int W, int C, int k, int dtype) {
  ACG_REQUIRE(dtype == ACG_F32 || dtype == ACG_BF16, ACG_ERR_UNSUPPORTED, "%s: dtype %d", who, dtype);
  ACG_REQUIRE(B > 0 && H > 0 && W > 0, ACG_ERR_INVALID_ARG, "%s: non-positive size", who);
  ACG_REQUIRE(C >= 1 && C <= 4, ACG_ERR_INVALID_ARG, "%s: channels %d outside 1..4", who, C);
  ACG_REQUIRE(B <= 65535 && H <= 65535, ACG_ERR_UNSUPPORTED, "%s: grid too large", who);
  ACG_REQUIRE((long long)B * H * W * ((k * k + 7) & ~7) < 2147483647ll, ACG_ERR_UNSUPPORTED, "%s: tensor exceeds 2^31 elements", who);
  return ACG_OK;
}

}  // namespace

extern "C" {

size_t acg_dna_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t k) {
  if (B <= 0 || H <= 0 || W <= 0 || k < 1 || k > 11) return 0;
  const dim3 g = dna_grid(k, B, H, W);
  return ((size_t)g.x * g.y * g.z + kSegs) * (size_t)(k * k) * sizeof(float);
}

namespace {
int second_of(const char* who, const void* ptr, int pitch, int off, int dt, int C, Second* s2) {
  s2->ptr = const_cast<void*>(ptr); s2->pitch = pitch; s2->off = off; s2->half = dt == ACG_BF16;
  if (!ptr) return ACG_OK;
  ACG_REQUIRE(dt == ACG_F32 || dt == ACG_BF16, ACG_ERR_UNSUPPORTED, "%s: second tensor dtype %d", who, dt);
  ACG_REQUIRE(off >= 0 && pitch >= off + C, ACG_ERR_INVALID_ARG, "%s: second tensor: %d channels at offset %d do not fit pitch %d", who, C, off, pitch);
  return ACG_OK;
}
}  // namespace

int32_t acg_dna_fwd(const void* logits, const float* bias, const void* image, void* out, void* out2, int32_t out2_pitch,
                    int32_t out2_offset, int32_t out2_dtype, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, int32_t dtype,
                    acg_stream_t stream) {
  if (int rc = check("dna_fwd", B, H, W, C, k, dtype)) return rc;
  ACG_REQUIRE(logits && image && out, ACG_ERR_INVALID_ARG, "dna_fwd: null pointer");
  Second s2;
  if (int rc = second_of("dna_fwd", out2, out2_pitch, out2_offset, out2_dtype, C, &s2)) return rc;
  if (dtype == ACG_BF16) return dispatch<false, __bf16>(k, logits, bias, (const float*)image, nullptr, out, nullptr, B, H, W, C, acg::to_stream(stream), s2);
  return dispatch<false, float>(k, logits, bias, (const float*)image, nullptr, out, nullptr, B, H, W, C, acg::to_stream(stream), s2);
}

int32_t acg_dna_bwd(const void* logits, const float* bias, const void* image, const void* dout, const void* dout2, int32_t dout2_pitch,
                    int32_t dout2_offset, int32_t dout2_dtype, void* dlogits, float* dbias, float dbias_acc, int32_t B, int32_t H,
                    int32_t W, int32_t C, int32_t k, int32_t dtype, void* ws, size_t wsb, acg_stream_t stream) {
  if (int rc = check("dna_bwd", B, H, W, C, k, dtype)) return rc;
  ACG_REQUIRE(logits && image && dout && dlogits, ACG_ERR_INVALID_ARG, "dna_bwd: null pointer");
  Second s2;
  if (int rc = second_of("dna_bwd", dout2, dout2_pitch, dout2_offset, dout2_dtype, C, &s2)) return rc;
  ACG_REQUIRE(!dbias || (ws && wsb >= acg_dna_workspace_bytes(B, H, W, k)), ACG_ERR_WORKSPACE, "dna_bwd: workspace too small for dbias");
  hipStream_t st = acg::to_stream(stream);
  float* bpart = dbias ? (float*)ws : nullptr;
  int rc;
  if (dtype == ACG_BF16) rc = dispatch<true, __bf16>(k, logits, bias, (const float*)image, (const float*)dout, dlogits, bpart, B, H, W, C, st, s2);
  else rc = dispatch<true, float>(k, logits, bias, (const float*)image, (const float*)dout, dlogits, bpart, B, H, W, C, st, s2);
  if (rc || !dbias) return rc;
  const dim3 g = dna_grid(k, B, H, W);
  const int nblk = (int)(g.x * g.y * g.z), KK = k * k, tb = (KK + 31) / 32;
  if (nblk <= 1024) {
    ACG_LAUNCH(dna_dbias_sum, dim3(tb, 1), dim3(256), 0, st, (const float*)bpart, dbias, dbias_acc, KK, nblk);
  } else {
    float* seg = bpart + (size_t)nblk * KK;
    ACG_LAUNCH(dna_dbias_sum, dim3(tb, kSegs), dim3(256), 0, st, (const float*)bpart, seg, 0.f, KK, nblk);
    ACG_LAUNCH(dna_dbias_sum, dim3(tb, 1), dim3(256), 0, st, (const float*)seg, dbias, dbias_acc, KK, kSegs);
  }
  return acg::check_launch("dna_dbias_sum");
}

}  // extern "C"
