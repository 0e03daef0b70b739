// CDNA transformation (reference ops.py:52-98): M per-sample k x k kernels, predicted as a vector per sample, are
// normalised ( k = relu(p - shift) + shift,  n = k / sum_uv k ) and applied as a depthwise SAME correlation to the
// previous image.  The reference then splits the depthwise output [B,H,W,C*M] (channel q = c*M + m) into M pieces of C
// channels along the channel axis; piece j, channel i is q = j*C + i, i.e. colour c = q / M with mask m = q % M.
// That indexing is reproduced as it is written.
//
// HBM-bound stencil, sibling of the DNA gather (dna.hip): a block owns a 16x16 pixel tile of one sample, stages the
// image window (tile + halo, zeros outside = SAME padding) and the sample's normalised kernels in LDS once, and every
// thread produces all C*M outputs of its pixel from registers.  Algorithmic bytes, forward: (1 + M) image-sized
// tensors; backward: (1 + M) read + 1 written, plus the k*k*M kernel gradients per sample.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

constexpr int kTile = 16;       // pixels per tile side (256 threads = one per pixel)
constexpr int kMaxK = 7, kMaxM = 32, kMaxC = 4;

struct Geo {
  int B, H, W, C, M, K, pad;
  int tiles_x, tiles_y;
};

// normalised kernels of sample b into LDS: kn[(u*K+v)*M + m]; also the per-mask sums S[m]
__device__ __forceinline__ void stage_kernels(const float* __restrict__ params, int b, const Geo& g, float shift,
                                              float* kn, float* S) {
  const int kk = g.K * g.K, n = kk * g.M;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float p = params[(long long)b * n + i];
    kn[i] = fmaxf(p - shift, 0.f) + shift;
  }
  __syncthreads();
  if ((int)threadIdx.x < g.M) {
    float s = 0.f;
    for (int t = 0; t < kk; ++t) s += kn[t * g.M + threadIdx.x];
    S[threadIdx.x] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) kn[i] /= S[i % g.M];
  __syncthreads();
}

// tile + halo of a [H,W,Cs] plane set into LDS as win[(y*ww + x)*Cs + c], zeros outside the image
__device__ __forceinline__ void stage_window(const float* __restrict__ src, int Cs, int y0, int x0, const Geo& g,
                                             float* win) {
  const int ww = kTile + g.K - 1, n = ww * ww * Cs;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int c = i % Cs, p = i / Cs, x = p % ww, y = p / ww;
    const int gy = y0 + y - g.pad, gx = x0 + x - g.pad;
    win[i] = (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) ? src[((long long)gy * g.W + gx) * Cs + c] : 0.f;
  }
}

template <int K>
__global__ __launch_bounds__(256) void cdna_fwd_k(const float* __restrict__ params, const float* __restrict__ img,
                                                  float* __restrict__ out, float* __restrict__ kern_norm, Geo g,
                                                  float shift) {
  __shared__ float kn[kMaxK * kMaxK * kMaxM];
  __shared__ float S[kMaxM];
  __shared__ float win[(kTile + kMaxK - 1) * (kTile + kMaxK - 1) * kMaxC];
  const int b = blockIdx.z, ty = blockIdx.y, tx = blockIdx.x;
  constexpr int kk = K * K, ww = kTile + K - 1;
  const int y0 = ty * kTile, x0 = tx * kTile;
  stage_kernels(params, b, g, shift, kn, S);
  if (ty == 0 && tx == 0 && kern_norm)
    for (int i = threadIdx.x; i < kk * g.M; i += blockDim.x) kern_norm[(long long)b * kk * g.M + i] = kn[i];
  stage_window(img + (long long)b * g.H * g.W * g.C, g.C, y0, x0, g, win);
  __syncthreads();
  const int ly = threadIdx.x / kTile, lx = threadIdx.x % kTile;
  const int y = y0 + ly, x = x0 + lx;
  if (y >= g.H || x >= g.W) return;
  const long long plane = (long long)g.B * g.H * g.W * g.C;               // one of the M output pieces
  const long long pix = (((long long)b * g.H + y) * g.W + x) * g.C;
  for (int c = 0; c < g.C; ++c) {
    float w[kk];                                                         // the pixel's window of colour c, in registers
#pragma unroll
    for (int u = 0; u < K; ++u)
#pragma unroll
      for (int v = 0; v < K; ++v) w[u * K + v] = win[((ly + u) * ww + lx + v) * g.C + c];
    for (int m = 0; m < g.M; ++m) {
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < kk; ++t) acc += w[t] * kn[t * g.M + m];       // LDS broadcast reads
      const int q = c * g.M + m;
      out[(q / g.C) * plane + pix + q % g.C] = acc;
    }
  }
}

// d img[b,y,x,c] = sum_{m,u,v} dout_q(c,m)[b, y-u+pad, x-v+pad] * n[b,u,v,m]
// The dout planes of as many masks as fit 48 KB of LDS (all 10 at k=5, C=3) are staged at once as a window with halo
// (K-1-pad before, pad after): one round of loads in flight instead of one memory round trip per mask.
constexpr int kWinFloats = 12288, kDdFloats = 8192;
__global__ __launch_bounds__(256) void cdna_bwd_img_k(const float* __restrict__ kern_norm, const float* __restrict__ dout,
                                                      float* __restrict__ dimg, Geo g) {
  __shared__ float kn[kMaxK * kMaxK * kMaxM];
  __shared__ float win[kWinFloats];                 // [window pixel][mask in chunk][colour]
  const int b = blockIdx.z, y0 = blockIdx.y * kTile, x0 = blockIdx.x * kTile, kk = g.K * g.K;
  const int ww = kTile + g.K - 1, before = g.K - 1 - g.pad;
  const int mc = min(g.M, kWinFloats / (ww * ww * g.C));           // masks per round (>= 1 for every supported shape)
  for (int i = threadIdx.x; i < kk * g.M; i += blockDim.x) kn[i] = kern_norm[(long long)b * kk * g.M + i];
  const int ly = threadIdx.x / kTile, lx = threadIdx.x % kTile;
  const int y = y0 + ly, x = x0 + lx;
  const long long plane = (long long)g.B * g.H * g.W * g.C;
  float acc[kMaxC] = {0.f, 0.f, 0.f, 0.f};
  for (int m0 = 0; m0 < g.M; m0 += mc) {
    const int nm = min(mc, g.M - m0);
    __syncthreads();
    for (int i = threadIdx.x; i < ww * ww * nm * g.C; i += blockDim.x) {
      const int c = i % g.C, r = i / g.C, mm = r % nm, p = r / nm, wx = p % ww, wy = p / ww, q = c * g.M + m0 + mm;
      const int gy = y0 + wy - before, gx = x0 + wx - before;
      win[i] = (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W)
                   ? dout[(q / g.C) * plane + (((long long)b * g.H + gy) * g.W + gx) * g.C + q % g.C] : 0.f;
    }
    __syncthreads();
    for (int u = 0; u < g.K; ++u)
      for (int v = 0; v < g.K; ++v) {
        const float* src = win + ((ly + g.K - 1 - u) * ww + lx + g.K - 1 - v) * nm * g.C;
        for (int mm = 0; mm < nm; ++mm) {
          const float w = kn[(u * g.K + v) * g.M + m0 + mm];
#pragma unroll
          for (int c = 0; c < kMaxC; ++c)
            if (c < g.C) acc[c] += src[mm * g.C + c] * w;
        }
      }
  }
  if (y >= g.H || x >= g.W) return;
#pragma unroll
  for (int c = 0; c < kMaxC; ++c)
    if (c < g.C) dimg[(((long long)b * g.H + y) * g.W + x) * g.C + c] = acc[c];
}

// per-tile partial kernel gradients: part[b][tile][(u*K+v)*M + m] = sum_{pixels in tile, c} dout_q(c,m)[pix] * img[pix+(u,v)-pad, c]
// The tile's dout of as many masks as fit 32 KB is staged at once; 4 pixel groups x 64 tap slots: a thread sums its
// tap over 64 pixels for every staged mask, then the 4 groups are folded through LDS.
__global__ __launch_bounds__(256) void cdna_bwd_kern_partial_k(const float* __restrict__ img, const float* __restrict__ dout,
                                                               float* __restrict__ part, Geo g) {
  __shared__ float win[(kTile + kMaxK - 1) * (kTile + kMaxK - 1) * kMaxC];
  __shared__ float dd[kDdFloats];                   // [pixel][mask in chunk][colour]
  __shared__ float red[256];
  static_assert(kMaxK * kMaxK <= 64, "one wave-slot per tap");
  const int b = blockIdx.z, ty = blockIdx.y, tx = blockIdx.x, y0 = ty * kTile, x0 = tx * kTile;
  const int kk = g.K * g.K, ww = kTile + g.K - 1;
  const int mc = min(g.M, kDdFloats / (kTile * kTile * g.C));
  const long long plane = (long long)g.B * g.H * g.W * g.C;
  stage_window(img + (long long)b * g.H * g.W * g.C, g.C, y0, x0, g, win);
  float* o = part + (((long long)b * g.tiles_y + ty) * g.tiles_x + tx) * kk * g.M;
  const int t = threadIdx.x & 63, pg = threadIdx.x >> 6, u = t / g.K, v = t % g.K;
  for (int m0 = 0; m0 < g.M; m0 += mc) {
    const int nm = min(mc, g.M - m0);
    __syncthreads();
    for (int i = threadIdx.x; i < kTile * kTile * nm * g.C; i += blockDim.x) {
      const int c = i % g.C, r = i / g.C, mm = r % nm, p = r / nm, y = y0 + p / kTile, x = x0 + p % kTile, q = c * g.M + m0 + mm;
      dd[i] = (y < g.H && x < g.W) ? dout[(q / g.C) * plane + (((long long)b * g.H + y) * g.W + x) * g.C + q % g.C] : 0.f;
    }
    __syncthreads();
    for (int mm = 0; mm < nm; ++mm) {
      float acc = 0.f;
      if (t < kk) {
        for (int p = pg * 64; p < pg * 64 + 64; ++p) {
          const float* wsrc = win + ((p / kTile + u) * ww + p % kTile + v) * g.C;
          const float* dsrc = dd + (p * nm + mm) * g.C;
#pragma unroll
          for (int c = 0; c < kMaxC; ++c)
            if (c < g.C) acc += dsrc[c] * wsrc[c];
        }
      }
      __syncthreads();                              // red[] of the previous mask has been consumed
      red[threadIdx.x] = acc;
      __syncthreads();
      if ((int)threadIdx.x < kk)
        o[threadIdx.x * g.M + m0 + mm] = red[threadIdx.x] + red[64 + threadIdx.x] + red[128 + threadIdx.x] + red[192 + threadIdx.x];
    }
  }
}

// dn = sum of the tile partials; through the normalisation: dk = (dn - sum_uv(dn * n)) / S, dp = dk * [p - shift > 0]
__global__ __launch_bounds__(256) void cdna_bwd_kern_final_k(const float* __restrict__ params, const float* __restrict__ kern_norm,
                                                             const float* __restrict__ part, float* __restrict__ dparams,
                                                             Geo g, float shift) {
  __shared__ float dn[kMaxK * kMaxK * kMaxM];
  __shared__ float dot[kMaxM], S[kMaxM];
  const int b = blockIdx.x, kk = g.K * g.K, n = kk * g.M, ntile = g.tiles_x * g.tiles_y;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    double s = 0.0;
    for (int t = 0; t < ntile; ++t) s += part[((long long)b * ntile + t) * n + i];
    dn[i] = (float)s;
  }
  __syncthreads();
  if ((int)threadIdx.x < g.M) {
    const int m = threadIdx.x;
    float d = 0.f, s = 0.f;
    for (int t = 0; t < kk; ++t) {
      d += dn[t * g.M + m] * kern_norm[(long long)b * n + t * g.M + m];
      s += fmaxf(params[(long long)b * n + t * g.M + m] - shift, 0.f) + shift;
    }
    dot[m] = d; S[m] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int m = i % g.M;
    const float dk = (dn[i] - dot[m]) / S[m];
    dparams[(long long)b * n + i] = params[(long long)b * n + i] - shift > 0.f ? dk : 0.f;
  }
}

int make_geo(const char* who, int B, int H, int W, int C, int M, int K, Geo* g) {
  ACG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && M > 0 && K > 0, ACG_ERR_INVALID_ARG, "%s: non-positive size", who);
  ACG_REQUIRE(C <= kMaxC && M <= kMaxM && K <= kMaxK, ACG_ERR_UNSUPPORTED, "%s: supports C <= %d, masks <= %d, k <= %d", who, kMaxC, kMaxM, kMaxK);
  ACG_REQUIRE(B <= 65535, ACG_ERR_UNSUPPORTED, "%s: batch too large", who);
  g->B = B; g->H = H; g->W = W; g->C = C; g->M = M; g->K = K;
  g->pad = (K - 1) / 2;                              // SAME, stride 1: pad_before = (k-1)//2
  g->tiles_x = (W + kTile - 1) / kTile; g->tiles_y = (H + kTile - 1) / kTile;
  return ACG_OK;
}

}  // namespace

extern "C" {

size_t acg_cdna_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t C, int32_t M, int32_t K) {
  Geo g;
  if (make_geo("cdna_workspace_bytes", B, H, W, C, M, K, &g) != ACG_OK) return 0;
  return (size_t)B * g.tiles_x * g.tiles_y * K * K * M * sizeof(float);
}

int32_t acg_cdna_fwd(const void* params, const void* image, void* out, float* kern_norm, int32_t B, int32_t H, int32_t W,
                     int32_t C, int32_t M, int32_t K, float relu_shift, int32_t dtype, acg_stream_t stream) {
  ACG_REQUIRE_F32(dtype);
  Geo g;
  if (int rc = make_geo("cdna_fwd", B, H, W, C, M, K, &g)) return rc;
  ACG_REQUIRE(params && image && out, ACG_ERR_INVALID_ARG, "cdna_fwd: null pointer");
  ACG_REQUIRE(K == 3 || K == 5 || K == 7, ACG_ERR_UNSUPPORTED, "cdna_fwd: kernel size %d (3, 5 or 7)", K);
  const dim3 grid(g.tiles_x, g.tiles_y, B);
  hipStream_t st = acg::to_stream(stream);
  const float *pp = (const float*)params, *im = (const float*)image;
  if (K == 3) ACG_LAUNCH(cdna_fwd_k<3>, grid, dim3(256), 0, st, pp, im, (float*)out, kern_norm, g, relu_shift);
  else if (K == 5) ACG_LAUNCH(cdna_fwd_k<5>, grid, dim3(256), 0, st, pp, im, (float*)out, kern_norm, g, relu_shift);
  else ACG_LAUNCH(cdna_fwd_k<7>, grid, dim3(256), 0, st, pp, im, (float*)out, kern_norm, g, relu_shift);
  return acg::check_launch("cdna_fwd");
}

int32_t acg_cdna_bwd(const void* params, const float* kern_norm, const void* image, const void* dout, void* dparams,
                     void* dimage, int32_t B, int32_t H, int32_t W, int32_t C, int32_t M, int32_t K, float relu_shift,
                     int32_t dtype, void* ws, size_t wsb, acg_stream_t stream) {
  ACG_REQUIRE_F32(dtype);
  Geo g;
  if (int rc = make_geo("cdna_bwd", B, H, W, C, M, K, &g)) return rc;
  ACG_REQUIRE(params && kern_norm && image && dout && dparams, ACG_ERR_INVALID_ARG, "cdna_bwd: null pointer");
  ACG_REQUIRE(ws && wsb >= acg_cdna_workspace_bytes(B, H, W, C, M, K), ACG_ERR_WORKSPACE, "cdna_bwd: workspace too small");
  hipStream_t st = acg::to_stream(stream);
  const dim3 grid(g.tiles_x, g.tiles_y, B);
  if (dimage) {
    ACG_LAUNCH(cdna_bwd_img_k, grid, dim3(256), 0, st, kern_norm, (const float*)dout, (float*)dimage, g);
    if (int rc = acg::check_launch("cdna_bwd_img")) return rc;
  }
  ACG_LAUNCH(cdna_bwd_kern_partial_k, grid, dim3(256), 0, st, (const float*)image, (const float*)dout, (float*)ws, g);
  if (int rc = acg::check_launch("cdna_bwd_kern_partial")) return rc;
  ACG_LAUNCH(cdna_bwd_kern_final_k, dim3(B), dim3(256), 0, st, (const float*)params, kern_norm, (const float*)ws,
                     (float*)dparams, g, relu_shift);
  return acg::check_launch("cdna_bwd_kern_final");
}

}  // extern "C"
