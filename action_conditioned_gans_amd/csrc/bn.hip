// Fused BatchNorm(train, no gamma) + activation, and bias + activation, for [rows, C] NHWC views.
//
// HBM-bound; every pass walks memory in address order: a 256-thread block is laid out as
// (256 / Cb) rows x Cb channels (Cb = min(C, 256)), so consecutive lanes touch consecutive floats and
// a wave never needs a per-element modulo; lanes move float4 when C % 4 == 0.  Per-channel statistics:
// pass 1 leaves per-block partial sums (shifted by the group's first row, so E[x^2]-E[x]^2 cannot
// cancel catastrophically) in the workspace; pass 2 is tiled rows x 32 channels, re-derives the statistics
// of its own channels from the partials (fp64 combine) and applies normalise + activation.  Two launches per
// direction (a launch costs more than the few redundant L2 reads of the partials), no atomics,
// deterministic.
// Tensors of <= 2048 rows per group take ONE launch instead: a block keeps its channels' rows in registers
// (bn_resident_fwd / _bwd below).
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace {

constexpr int kMaxPartialBlocks = 512;
#ifndef ACG_BN_U
#define ACG_BN_U 4
#endif
constexpr int kU = ACG_BN_U;   // row passes whose loads a thread keeps in flight together (two-launch BatchNorm kernels)

#ifdef ACG_TUNING        // tuning builds only (see conv_f32.hip)
int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}
#else
constexpr int env_int(const char*, int dflt) { return dflt; }
#endif
constexpr int kMaxC = 1024;

struct ColMap {
  int Cb, RPP, cl, rsub, nchunk;
  bool active;
};
__device__ __forceinline__ ColMap col_map(int C) {
  ColMap m;
  m.Cb = C < 256 ? C : 256;
  m.RPP = 256 / m.Cb;
  m.cl = threadIdx.x % m.Cb;
  m.rsub = threadIdx.x / m.Cb;
  m.active = m.rsub < m.RPP;
  m.nchunk = (C + m.Cb - 1) / m.Cb;
  return m;
}

// Reduce two per-thread values over the rsub dimension; result valid where rsub == 0.
__device__ __forceinline__ void reduce_rsub(const ColMap& m, float& a, float& b, float* sh /* 512 floats */) {
  __syncthreads();
  sh[threadIdx.x] = a;
  sh[256 + threadIdx.x] = b;
  __syncthreads();
  if (m.active && m.rsub == 0) {
    float sa = 0.f, sb = 0.f;
    for (int r = 0; r < m.RPP; ++r) { sa += sh[r * m.Cb + m.cl]; sb += sh[256 + r * m.Cb + m.cl]; }
    a = sa; b = sb;
  }
}

// ---- vectorised [rows, C] walking ----------------------------------------------------------------------
// V = 4 floats per lane when C % 4 == 0 (every BatchNorm'd layer of the models except d/conv6), else 1.
template <int V>
struct VMap {
  int Cv, Cb, RPP, cl, rsub, nchunk;
  bool active;
};
template <int V>
__device__ __forceinline__ VMap<V> vmap(int C) {
  VMap<V> m;
  m.Cv = C / V;
  m.Cb = m.Cv < 256 ? m.Cv : 256;
  m.RPP = 256 / m.Cb;
  m.cl = threadIdx.x % m.Cb;
  m.rsub = threadIdx.x / m.Cb;
  m.active = m.rsub < m.RPP;
  m.nchunk = (m.Cv + m.Cb - 1) / m.Cb;
  return m;
}
using acg::ldv;
using acg::stv;

// Reduce 2*V per-thread values over the rsub dimension through LDS (sh: 2*V*256 floats); valid where rsub == 0.
template <int V>
__device__ __forceinline__ void reduce_rsub_v(const VMap<V>& m, float (&a)[V], float (&b)[V], float* sh) {
  __syncthreads();
#pragma unroll
  for (int j = 0; j < V; ++j) { sh[j * 256 + threadIdx.x] = a[j]; sh[(V + j) * 256 + threadIdx.x] = b[j]; }
  __syncthreads();
  if (m.active && m.rsub == 0) {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float sa = 0.f, sb = 0.f;
      for (int r = 0; r < m.RPP; ++r) { sa += sh[j * 256 + r * m.Cb + m.cl]; sb += sh[(V + j) * 256 + r * m.Cb + m.cl]; }
      a[j] = sa; b[j] = sb;
    }
  }
}

// Split-K hand-off (acg_bn_act_fwd_slabs / acg_bn_act_bwd_slabs): the producing convolution left `splits` fp32 partial
// slabs; the BatchNorm kernel that would read the tensor sums them itself - in slab order, rounded to the tensor's
// storage type: the very values splitk_reduce would have written - and, where the tensor is needed later (x: BatchNorm
// backward re-reads it), writes it back.  `slabs` == nullptr: the tensor is read as is.
// Two slab layouts (ACG_SLABS_ROWS / ACG_SLABS_QUADS): `qrows` == 0, each slab is laid out like the tensor ([rows][pitch]);
// `qrows` = all rows of the tensor, a slab is [channels / 4][rows][4] - the 16 bytes a register-resident block (4 channels,
// every row) reads per row are then consecutive in memory, where the row layout costs it one cache line per row and slab.
struct Slabs { const float* p; int splits; long long stride; long long qrows; };
template <bool SL, int V, typename T>
__device__ __forceinline__ void ld_or_sum(T* base, long long off, const Slabs& sl, bool write, float (&v)[V], long long qoff = 0) {
  if constexpr (!SL) { ldv<V>(base + off, v); return; }
  const long long so = sl.qrows ? qoff : off;
#pragma unroll
  for (int j = 0; j < V; ++j) v[j] = 0.f;
#pragma unroll 4
  for (int z = 0; z < sl.splits; ++z) {
    float t[V];
    ldv<V>(sl.p + (long long)z * sl.stride + so, t);
#pragma unroll
    for (int j = 0; j < V; ++j) v[j] += t[j];
  }
#pragma unroll
  for (int j = 0; j < V; ++j) v[j] = (float)(T)v[j];
  if (write) stv<V>(base + off, v);
}

// partial layout: part[((g * nblk + b) * 2 + which) * C + c]
// ---- BN forward ---------------------------------------------------------------------------------------
template <int V, typename TX, bool SL = false>
__global__ __launch_bounds__(256) void bn_stats_partial(TX* __restrict__ x, float* __restrict__ part,
                                                        long long R, int C, int nblk, int XP, const Slabs sl) {
  __shared__ float sh[2 * V * 256];
  const VMap<V> m = vmap<V>(C);
  const int g = blockIdx.y, b = blockIdx.x;
  const long long gb = (long long)g * R * XP;
  const long long rpb = (R + nblk - 1) / nblk;
  const long long r0 = (long long)b * rpb, r1 = min(R, r0 + rpb);
  for (int ch = 0; ch < m.nchunk; ++ch) {
    const int cv = ch * m.Cb + m.cl, c = cv * V;
    float s1[V], s2[V], pv[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { s1[j] = 0.f; s2[j] = 0.f; pv[j] = 0.f; }
    if (m.active && cv < m.Cv) {
      ld_or_sum<SL, V>(x, gb + c, sl, false, pv);   // shift by the group's first row: E[d^2]-E[d]^2 cannot cancel catastrophically
      // batches of 4 row passes, all loads of a batch issued before the first use (rows beyond r1 re-read
      // the last valid row with weight 0), so a block's few passes cost ~one memory round trip
      for (long long r = r0 + m.rsub; r < r1; r += kU * m.RPP) {
        float v[kU][V];
#pragma unroll
        for (int u = 0; u < kU; ++u) ld_or_sum<SL, V>(x, gb + min(r + u * m.RPP, r1 - 1) * XP + c, sl, r + u * m.RPP < r1, v[u]);
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const float w = r + u * m.RPP < r1 ? 1.f : 0.f;
#pragma unroll
          for (int j = 0; j < V; ++j) { const float d = (v[u][j] - pv[j]) * w; s1[j] += d; s2[j] += d * d; }
        }
      }
    }
    reduce_rsub_v<V>(m, s1, s2, sh);
    if (m.active && m.rsub == 0 && cv < m.Cv) {
      float* o = part + ((long long)g * nblk + b) * 2 * C + c;
      stv<V>(o, s1);
      stv<V>(o + C, s2);
    }
  }
}

// Tile mapping of the apply kernels: 256 threads = 8 channel lanes (V floats each) x 32 row lanes; a block owns
// the 8*V channels [blockIdx.y*8V, ...) of group blockIdx.z and walks row chunks.  Each block first re-derives
// the statistics of ITS channels from the per-block partials (nblk * 8V * 2 floats, L2-resident: a few loads per
// thread), so no separate finalize launch exists - a launch costs more than this prologue (profiles/r1).
template <int V, int CL = 8>
__device__ __forceinline__ void sum_partials(const float* __restrict__ part, int g, int nblk, int C, int c, bool cvalid,
                                             int rl, float (&s1)[V], float (&s2)[V], float* sh /* 2*V*CL*(blockDim.x/64) floats */) {
#pragma unroll
  for (int j = 0; j < V; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  const int nrl = (int)blockDim.x / CL, nwaves = (int)blockDim.x >> 6;
  if (cvalid) {
#pragma unroll 8
    for (int b = rl; b < nblk; b += nrl) {      // (unrolled: eight partial blocks in flight per thread - this loop is pure latency)
      const float* o = part + ((long long)g * nblk + b) * 2 * C + c;
      float a[V], q[V];
      ldv<V>(o, a); ldv<V>(o + C, q);
#pragma unroll
      for (int j = 0; j < V; ++j) { s1[j] += a[j]; s2[j] += q[j]; }
    }
  }
  // lanes of one channel lane are CL apart: fold the row lanes of the wave, then the waves through LDS
#pragma unroll
  for (int j = 0; j < V; ++j) {
#pragma unroll
    for (int off = CL; off < 64; off <<= 1) { s1[j] += __shfl_xor(s1[j], off, 64); s2[j] += __shfl_xor(s2[j], off, 64); }
  }
  const int cq = threadIdx.x % CL, wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) < CL) {
#pragma unroll
    for (int j = 0; j < V; ++j) { sh[(wave * CL + cq) * 2 * V + j] = s1[j]; sh[(wave * CL + cq) * 2 * V + V + j] = s2[j]; }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < V; ++j) {
    float a = 0.f, q = 0.f;
    for (int w = 0; w < nwaves; ++w) { a += sh[(w * CL + cq) * 2 * V + j]; q += sh[(w * CL + cq) * 2 * V + V + j]; }
    s1[j] = a; s2[j] = q;
  }
}

// Forward statistics of one group from the per-tile (sum, M2) partials a convolution's epilogue left (conv_f32_kernel.h
// tile_stats_epilogue), merged with the parallel-variance formula about a REFERENCE mean - the first tile's - so that
// float32 is enough: with d_b = sum_b - n_b * ref (small: a tile mean deviates from the reference by ~std / sqrt(n_b))
//   mean = ref + sum(d_b) / R,   M2 = sum(M2_b) + sum(d_b^2 / n_b) - sum(d_b)^2 / R
// every term is centred, nothing of size mean^2 is ever formed (round 2 summed plain squares: the variance was gone once
// |mean| >> std; a float64 merge of uncentred terms is robust too, but cost this prologue 4 us per layer).
// Same thread mapping as sum_partials.  n_b from the tile geometry (all tiles full unless run_rows % block_rows).
struct TileGeom { int block_rows, run_rows, blocks_per_run; };
template <int V, int CL = 8>
__device__ __forceinline__ void merge_tile_partials(const float* __restrict__ part, int g, int nblk, int C, int c, bool cvalid, int rl,
                                                    const TileGeom tg, long long R, float eps, float (&mean)[V], float (&rstd)[V],
                                                    float* sh /* (blockDim.x / 64) * CL * 3 * V floats */) {
  float S[V], Q[V], P[V], ref[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { S[j] = 0.f; Q[j] = 0.f; P[j] = 0.f; ref[j] = 0.f; }
  const bool uniform = tg.run_rows % tg.block_rows == 0;      // block-uniform
  const int nrl = (int)blockDim.x / CL, nwaves = (int)blockDim.x >> 6;     // row lanes
  if (cvalid) {
    const float* p0 = part + (long long)g * nblk * 2 * C + c;
    ldv<V>(p0, ref);
    const float inv0 = 1.f / (float)min(tg.block_rows, tg.run_rows);
#pragma unroll
    for (int j = 0; j < V; ++j) ref[j] *= inv0;
    if (uniform) {
      const float nb = (float)tg.block_rows, inv = 1.f / nb;
#pragma unroll 8
      for (int b = rl; b < nblk; b += nrl) {      // (unrolled: the loads of eight partial blocks in flight - this loop is pure latency)
        const float* o = p0 + (long long)b * 2 * C;
        float a[V], q[V];
        ldv<V>(o, a); ldv<V>(o + C, q);
#pragma unroll
        for (int j = 0; j < V; ++j) { const float d = fmaf(-nb, ref[j], a[j]); S[j] += d; P[j] = fmaf(d * inv, d, P[j]); Q[j] += q[j]; }
      }
    } else {
      for (int b = rl; b < nblk; b += nrl) {
        const float* o = p0 + (long long)b * 2 * C;
        float a[V], q[V];
        ldv<V>(o, a); ldv<V>(o + C, q);
        const float nb = (float)min(tg.block_rows, tg.run_rows - (b % tg.blocks_per_run) * tg.block_rows), inv = 1.f / nb;
#pragma unroll
        for (int j = 0; j < V; ++j) { const float d = fmaf(-nb, ref[j], a[j]); S[j] += d; P[j] = fmaf(d * inv, d, P[j]); Q[j] += q[j]; }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) {
#pragma unroll
    for (int off = CL; off < 64; off <<= 1) { S[j] += __shfl_xor(S[j], off, 64); Q[j] += __shfl_xor(Q[j], off, 64); P[j] += __shfl_xor(P[j], off, 64); }
  }
  const int cq = threadIdx.x % CL, wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) < CL) {
#pragma unroll
    for (int j = 0; j < V; ++j) { float* q = sh + ((wave * CL + cq) * 3) * V + j; q[0] = S[j]; q[V] = Q[j]; q[2 * V] = P[j]; }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < V; ++j) {
    float s = 0.f, q = 0.f, pp = 0.f;
    for (int w = 0; w < nwaves; ++w) { const float* z = sh + ((w * CL + cq) * 3) * V + j; s += z[0]; q += z[V]; pp += z[2 * V]; }
    const float inv = 1.f / (float)R, dm = s * inv;
    const float var = fmaxf((q + (pp - s * dm)) * inv, 0.f);
    mean[j] = ref[j] + dm;
    rstd[j] = rsqrtf(var + eps);
  }
}

// mode 0: tile (sum, M2) partials of a convolution epilogue; 1: shifted plain sums of bn_stats_partial; 2: mean / rstd are
// already in save_mean / save_rstd (bn_partials_finalize ran: the convolution left more partial blocks than an apply block
// should re-read)
enum { kPartTiles = 0, kPartShifted = 1, kPartDone = 2 };

// NT threads = CL channel lanes (V elements each: CL * V * sizeof = one 128-byte line per row) x NT / CL row lanes.  Two
// shapes: 256 threads (32 row lanes) for tensors of a few thousand rows, and 1024 threads (128 row lanes) for the large ones -
// a block's prologue re-reads nblk * 2 * CL * V floats of partials whatever its size, so a tensor cut into 1024 small blocks
// moved 128 MB of partials through the L2s for 33 MB of tensor (4 of the 14 us of g/tconv3's BatchNorm at batch 32); 256 blocks
// of 1024 threads read a quarter of that, each thread's share in ONE round trip (8 loads in flight).
template <int V, typename TX, typename TY, int NT = 256, int CL = 8>
__global__ __launch_bounds__(NT) void bn_apply_fwd(const TX* __restrict__ x, const float* __restrict__ beta,
                                                   const float* __restrict__ part, TY* __restrict__ y,
                                                   float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                   long long R, int C, int nblk, float eps, int act, float leak, int XP, int YP,
                                                   int mode, const TileGeom tg) {
  constexpr int RL = NT / CL;
  __shared__ float sh[(NT / 64) * CL * 3 * V];
  const int cq = threadIdx.x % CL, rl = threadIdx.x / CL, g = blockIdx.z;
  const int c = (blockIdx.y * CL + cq) * V;
  const bool cvalid = c < C;
  const TX* xg = x + (long long)g * R * XP;
  TY* yg = y + (long long)g * R * YP;
  // Software-pipelined row loop: a block walks `iters` batches of kU row passes; the loads of batch i + 1 are in flight while
  // batch i is normalised and stored, and the loads of batch 0 - which do not depend on the statistics - are issued BEFORE
  // the prologue.  With one batch per block (round 3) every block of the grid sat in its prologue at the same time and the
  // memory system idled for those 3-4 us: g/tconv3's 33.6 MB pass took 14 us where the same pass without a prologue takes
  // 5.4 (tools/micro/stream_probe.hip); now the grid is sized for ~2 blocks per CU and the prologue is paid once per block,
  // under the first batch's loads.
  const long long rstep = (long long)gridDim.x * RL;
  const long long r0 = (long long)blockIdx.x * RL + rl;
  float va[kU][V], vb[kU][V];
#define ACG_BN_LOAD(buf, rr)                                                                  \
  _Pragma("unroll") for (int u = 0; u < kU; ++u) ldv<V>(xg + min((rr) + u * rstep, R - 1) * XP + c, buf[u])
  if (cvalid && r0 < R) { ACG_BN_LOAD(va, r0); }
  float mean[V], rstd[V], bt[V];
  if (mode == kPartTiles) {
    merge_tile_partials<V, CL>(part, g, nblk, C, c, cvalid, rl, tg, R, eps, mean, rstd, sh);
    if (!cvalid) return;
  } else if (mode == kPartShifted) {       // partials of bn_stats_partial: sums of (x - first row of the group)
    float s1[V], s2[V], pv[V];
    sum_partials<V, CL>(part, g, nblk, C, c, cvalid, rl, s1, s2, sh);
    if (!cvalid) return;
    ldv<V>(xg + c, pv);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const double inv = 1.0 / (double)R, dm = (double)s1[j] * inv;
      double var = (double)s2[j] * inv - dm * dm;
      var = var > 0.0 ? var : 0.0;
      mean[j] = (float)((double)pv[j] + dm);
      rstd[j] = rsqrtf((float)var + eps);        // (the float64 square root and division were most of this kernel's instructions)
    }
  } else {
    if (!cvalid) return;
    ldv<V>(save_mean + g * C + c, mean); ldv<V>(save_rstd + g * C + c, rstd);
  }
  ldv<V>(beta + c, bt);
  if (mode != kPartDone && blockIdx.x == 0 && rl == 0) { stv<V>(save_mean + g * C + c, mean); stv<V>(save_rstd + g * C + c, rstd); }
#define ACG_BN_APPLY(buf, rr)                                                                 \
  _Pragma("unroll") for (int u = 0; u < kU; ++u) {                                            \
    if ((rr) + u * rstep < R) {                                                               \
      _Pragma("unroll") for (int j = 0; j < V; ++j) buf[u][j] = acg::act_apply(act, (buf[u][j] - mean[j]) * rstd[j] + bt[j], leak); \
      stv<V>(yg + ((rr) + u * rstep) * YP + c, buf[u]);                                       \
    }                                                                                         \
  }
  const long long bstep = kU * rstep;
  for (long long r = r0; r < R; r += 2 * bstep) {        // two batches per trip: the buffers alternate with static indices
    if (r + bstep < R) { ACG_BN_LOAD(vb, r + bstep); }
    ACG_BN_APPLY(va, r);
    if (r + bstep < R) {
      if (r + 2 * bstep < R) { ACG_BN_LOAD(va, r + 2 * bstep); }
      ACG_BN_APPLY(vb, r + bstep);
    }
  }
#undef ACG_BN_LOAD
#undef ACG_BN_APPLY
}

// mean / rstd of every (group, channel) from the tile partials, for tensors whose convolution left so many partial blocks
// that each apply block re-deriving them costs more than this launch: config 5's d/conv1 leaves 2048 per group, 512 KB
// per apply block whose own work is 8 KB of the tensor - its BatchNorm ran at 0.8 TB/s (profiles/r2/h_other_ops_bf16_config5.txt)
template <int V, int CL>
__global__ __launch_bounds__(1024) void bn_partials_finalize(const float* __restrict__ part, float* __restrict__ save_mean,
                                                             float* __restrict__ save_rstd, long long R, int C, int nblk, float eps,
                                                             const TileGeom tg) {
  // CL * V channels per block and 1024 / CL lanes over the partial blocks: with CL = 2 a block of 8 channels takes 2048 partial
  // blocks in 8 loads per thread - one round trip - and there are C / 8 blocks instead of C / 32
  __shared__ float sh[16 * CL * 3 * V];
  const int cq = threadIdx.x % CL, rl = threadIdx.x / CL, g = blockIdx.z;
  const int c = (blockIdx.y * CL + cq) * V;
  const bool cvalid = c < C;
  float mean[V], rstd[V];
  merge_tile_partials<V, CL>(part, g, nblk, C, c, cvalid, rl, tg, R, eps, mean, rstd, sh);
  if (cvalid && rl == 0) { stv<V>(save_mean + g * C + c, mean); stv<V>(save_rstd + g * C + c, rstd); }
}

// ---- BN backward ----------------------------------------------------------------------------------------
template <int V, typename TX, typename TY, int U = kU>
__global__ __launch_bounds__(256) void bn_bwd_partial(const TX* __restrict__ x, const TY* __restrict__ dy,
                                                      const float* __restrict__ beta, const float* __restrict__ save_mean,
                                                      const float* __restrict__ save_rstd, float* __restrict__ part,
                                                      long long R, int C, int nblk, int act, float leak, int XP, int YP) {
  __shared__ float sh[2 * V * 256];
  const VMap<V> m = vmap<V>(C);
  const int g = blockIdx.y, b = blockIdx.x;
  const TX* xg = x + (long long)g * R * XP;
  const TY* dyg = dy + (long long)g * R * YP;
  const long long rpb = (R + nblk - 1) / nblk;
  const long long r0 = (long long)b * rpb, r1 = min(R, r0 + rpb);
  for (int ch = 0; ch < m.nchunk; ++ch) {
    const int cv = ch * m.Cb + m.cl, c = cv * V;
    float s1[V], s2[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    if (m.active && cv < m.Cv) {
      float mean[V], rstd[V], bt[V];
      ldv<V>(save_mean + g * C + c, mean); ldv<V>(save_rstd + g * C + c, rstd); ldv<V>(beta + c, bt);
      for (long long r = r0 + m.rsub; r < r1; r += U * m.RPP) {   // batched like bn_stats_partial: U passes of loads in flight
        float xv[U][V], dv[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long long rr = min(r + u * m.RPP, r1 - 1);
          ldv<V>(xg + rr * XP + c, xv[u]); ldv<V>(dyg + rr * YP + c, dv[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float w = r + u * m.RPP < r1 ? 1.f : 0.f;
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const float xh = (xv[u][j] - mean[j]) * rstd[j];
            const float dp = w * dv[u][j] * acg::act_deriv_pre(act, xh + bt[j], leak);
            s1[j] += dp; s2[j] += dp * xh;
          }
        }
      }
    }
    reduce_rsub_v<V>(m, s1, s2, sh);
    if (m.active && m.rsub == 0 && cv < m.Cv) {
      float* o = part + ((long long)g * nblk + b) * 2 * C + c;
      stv<V>(o, s1);
      stv<V>(o + C, s2);
    }
  }
}

template <int V, typename TX, typename TY, typename TD = TX, int NT = 256, int CL = 8>
__global__ __launch_bounds__(NT) void bn_apply_bwd(const TX* __restrict__ x, const TY* __restrict__ dy,
                                                   const float* __restrict__ beta, const float* __restrict__ save_mean,
                                                   const float* __restrict__ save_rstd, const float* __restrict__ part,
                                                   TD* __restrict__ dx, float* __restrict__ dbeta, float dbeta_acc,
                                                   long long R, int C, int groups, int nblk, int act, float leak, int XP, int YP,
                                                   const float* __restrict__ gsums = nullptr, float inv_total = 0.f) {
  constexpr int RL = NT / CL;
  __shared__ float sh[(NT / 64) * CL * 2 * V];
  const int cq = threadIdx.x % CL, rl = threadIdx.x / CL, g = blockIdx.z;
  const int c = (blockIdx.y * CL + cq) * V;
  const bool cvalid = c < C;
  const TX* xg = x + (long long)g * R * XP;
  const TY* dyg = dy + (long long)g * R * YP;
  TD* dxg = dx + (long long)g * R * XP;
  // software-pipelined like bn_apply_fwd: batch 0's loads before the prologue, batch i + 1's under batch i's arithmetic
  const long long rstep = (long long)gridDim.x * RL;
  const long long r0 = (long long)blockIdx.x * RL + rl;
  float xa[kU][V], da[kU][V], xb[kU][V], db[kU][V];
#define ACG_BN_LOAD(bx, bd, rr)                                                               \
  _Pragma("unroll") for (int u = 0; u < kU; ++u) {                                            \
    const long long q = min((rr) + u * rstep, R - 1);                                         \
    ldv<V>(xg + q * XP + c, bx[u]); ldv<V>(dyg + q * YP + c, bd[u]);                          \
  }
  if (cvalid && r0 < R) { ACG_BN_LOAD(xa, da, r0); }
  float s1[V], s2[V];
  if (gsums != nullptr) {            // synchronised BatchNorm: the sums of the GLOBAL batch are given, dbeta is formed elsewhere
#pragma unroll
    for (int j = 0; j < V; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    if (cvalid) { ldv<V>(gsums + (g * 2 + 0) * C + c, s1); ldv<V>(gsums + (g * 2 + 1) * C + c, s2); }
  } else if (blockIdx.x == 0 && g == 0) {   // this block also owns dbeta of its channels: sum over every group
    float tot[V];
#pragma unroll
    for (int j = 0; j < V; ++j) tot[j] = 0.f;
    float k1[V], k2[V];
    for (int gg = groups - 1; gg >= 0; --gg) {   // ends on group 0 = this block's own statistics
      sum_partials<V, CL>(part, gg, nblk, C, c, cvalid, rl, k1, k2, sh);
#pragma unroll
      for (int j = 0; j < V; ++j) tot[j] += k1[j];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) { s1[j] = k1[j]; s2[j] = k2[j]; }
    if (cvalid && rl == 0) {
      float d[V];
      if (dbeta_acc != 0.f) {
        ldv<V>(dbeta + c, d);
#pragma unroll
        for (int j = 0; j < V; ++j) d[j] = dbeta_acc * d[j] + tot[j];
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) d[j] = tot[j];
      }
      stv<V>(dbeta + c, d);
    }
  } else {
    sum_partials<V, CL>(part, g, nblk, C, c, cvalid, rl, s1, s2, sh);
  }
  if (!cvalid) return;
  float mean[V], rstd[V], bt[V], m1[V], m2[V];
  ldv<V>(save_mean + g * C + c, mean); ldv<V>(save_rstd + g * C + c, rstd); ldv<V>(beta + c, bt);
  const float invR = gsums != nullptr ? inv_total : 1.f / (float)R;
#pragma unroll
  for (int j = 0; j < V; ++j) { m1[j] = s1[j] * invR; m2[j] = s2[j] * invR; }
#define ACG_BN_APPLY(bx, bd, rr)                                                              \
  _Pragma("unroll") for (int u = 0; u < kU; ++u) {                                            \
    if ((rr) + u * rstep < R) {                                                               \
      _Pragma("unroll") for (int j = 0; j < V; ++j) {                                         \
        const float xh = (bx[u][j] - mean[j]) * rstd[j];                                      \
        const float dp = bd[u][j] * acg::act_deriv_pre(act, xh + bt[j], leak);                \
        bd[u][j] = rstd[j] * (dp - m1[j] - xh * m2[j]);                                       \
      }                                                                                       \
      stv<V>(dxg + ((rr) + u * rstep) * XP + c, bd[u]);                                       \
    }                                                                                         \
  }
  const long long bstep = kU * rstep;
  for (long long r = r0; r < R; r += 2 * bstep) {
    if (r + bstep < R) { ACG_BN_LOAD(xb, db, r + bstep); }
    ACG_BN_APPLY(xa, da, r);
    if (r + bstep < R) {
      if (r + 2 * bstep < R) { ACG_BN_LOAD(xa, da, r + 2 * bstep); }
      ACG_BN_APPLY(xb, db, r + bstep);
    }
  }
#undef ACG_BN_LOAD
#undef ACG_BN_APPLY
}

// ---- one-launch backward for tensors that fit the chip's registers (round 4) ---------------------------------------------
// The two-launch backward reads x and dy twice (sums, then apply): 84 MB of traffic and two launch boundaries for d/conv1's
// 50 MB at batch 32 (23 us).  Here every block keeps its share of x and dy IN REGISTERS between the two phases: a block of
// 1024 threads = 8 channel lanes x 128 row lanes owns 32 channels x 128 * U rows of one group, loads them once, reduces its
// partial sums (shuffles, then LDS across its 16 waves), and the blocks of a (group, channel chunk) exchange their 64 partial
// sums through memory inside the launch; each block then adds up what the others published and finishes from registers.
//
// The exchange is the data-tagged granule form of cdna_hip_programming.md guideline 16 (R2): every value travels as ONE
// aligned 8-byte {epoch tag, float bits} store at agent scope (write-through) and is read by agent-scope 8-byte loads that are
// repeated until the tag matches - the data is its own flag: no fence, no release / acquire, nothing that depends on block
// placement or dispatch order.  `epoch` is a word of the op's private workspace that the LAST block to finish increments
// (an arrival counter beside it), so every launch - also every replay of a captured graph, whose arguments are frozen - sees
// tags no earlier launch has written; the workspace must be zero before its first use and belong to this op alone.
// All blocks of the grid must be resident together: the host takes this path only for grids of at most one 1024-thread block
// per CU (code-object metadata, gfx950: 68-86 VGPRs forward, 72-116 backward - 16 waves = 4 per SIMD x <= 116 of the 512
// registers a SIMD lane has: ONE such block fits a CU, never two) or at most two 256-thread blocks per CU (70-74 VGPRs: seven
// would fit), and every spin is bounded - a block that gives up sets the workspace's `timeout` word and finishes with what it
// has, so a violated assumption shows up as a wrong result and a flag (which the host turns into an error: Session.run's
// callers check it wherever they synchronise anyway), never as a hung GPU.  What may run BESIDE such a grid: kernels that
// finish on their own (a conv on a second stream, an RCCL collective - its blocks wait for peers on other GPUs, never for
// this grid), which at worst delay the last blocks; what must NOT: another grid-exchange kernel, since two partially resident
// grids can starve each other until both time out - callers whose launches can overlap pass ACG_BN_NO_GRID_EXCHANGE.
struct FusedState { unsigned epoch, done, timeout, pad; };      // first 16 bytes of the workspace; granules behind it
typedef unsigned long long __attribute__((address_space(1))) * gran_ptr;

// The in-launch exchange shared by bn_bwd_fused / bn_fwd_fused.  NT threads = 8 channel lanes x NT / 8 row lanes; a block holds
// NV = 64 partial values (8 lanes x 2 statistics x 4 channels) in red[wave][...] after its own cross-wave step.
template <int NT>
struct FusedExchange {
  static constexpr int CL = 8, V = 4, NW = NT / 64, NV = 2 * V * CL;
  float (*red)[NV];          // [NW][NV] LDS
  unsigned long long* grans;
  FusedState* state;
  unsigned epoch;
  int tid, lane, wave, nrb;
  unsigned spin_limit = 1u << 22;      // polls before a block gives up (seconds); acg_bn_exchange_selftest lowers it

  // the per-thread partials s[0..7] (value index within the lane: {stat 0: 0..3, stat 1: 4..7}) -> this block's 64 sums, published
  __device__ __forceinline__ void publish(float (&s)[2 * V], long long slot /* granule index of this block's value 0 */) {
#pragma unroll
    for (int j = 0; j < 2 * V; ++j) {
#pragma unroll
      for (int off = CL; off < 64; off <<= 1) s[j] += __shfl_xor(s[j], off, 64);
    }
    if (lane < CL) {
#pragma unroll
      for (int j = 0; j < 2 * V; ++j) red[wave][lane * 2 * V + j] = s[j];
    }
    __syncthreads();
    if (tid < NV) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[w][tid];
      __hip_atomic_store((gran_ptr)(grans + slot + tid), ((unsigned long long)epoch << 32) | __float_as_uint(v), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                              // red is reused by gather
  }
  // a grid with ONE row block per (group, chunk) has nobody to exchange with: the block's own 64 sums straight into out[NV] (LDS) -
  // no granule store, no polling round trip (round 5: the 2x2 ... 4x4 layers, whose whole tensor is one row block)
  __device__ __forceinline__ void local(float (&s)[2 * V], float* out) {
#pragma unroll
    for (int j = 0; j < 2 * V; ++j) {
#pragma unroll
      for (int off = CL; off < 64; off <<= 1) s[j] += __shfl_xor(s[j], off, 64);
    }
    if (lane < CL) {
#pragma unroll
      for (int j = 0; j < 2 * V; ++j) red[wave][lane * 2 * V + j] = s[j];
    }
    __syncthreads();
    if (tid < NV) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[w][tid];
      out[tid] = v;
    }
    __syncthreads();
  }
  // sums over the nrb row blocks whose granules start at `base` ([row block][NV]) -> out[NV] (LDS); every thread of the block calls it
  __device__ __forceinline__ void gather(long long base, float* out) {
    const int n = nrb * NV;                       // granule i = row block * NV + value; thread t takes i = t, t + NT, ...: value t % NV
    float acc = 0.f;
    for (int i0 = 0; i0 < n; i0 += 4 * NT) {
      unsigned long long gv[4];
      bool ok = false;
      for (unsigned spins = 0; !ok; ++spins) {
        ok = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k * NT + tid;
          gv[k] = i < n ? __hip_atomic_load((gran_ptr)(grans + base + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)epoch << 32);
          ok = ok && (unsigned)(gv[k] >> 32) == epoch;
        }
        ok = __all(ok);
        if (!ok) {
          if (spins > spin_limit) {                // bounded: give up, flag it, finish with what is there
            if (lane == 0) __hip_atomic_store(&state->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = true;
          } else {
            __builtin_amdgcn_s_sleep(2);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) acc += __uint_as_float((unsigned)gv[k]);
    }
    red[wave][lane] = acc;                        // NT / NV threads per value: thread t holds value t % 64 = its lane
    __syncthreads();
    if (tid < NV) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[w][tid];
      out[tid] = v;
    }
    __syncthreads();
  }
  // the last block of the grid to get here opens the next epoch
  __device__ __forceinline__ void finish() {
    __syncthreads();
    if (tid == 0) {
      const unsigned total = gridDim.x * gridDim.y * gridDim.z;
      const unsigned prev = __hip_atomic_fetch_add(&state->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (prev + 1u == total) {
        __hip_atomic_store(&state->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&state->epoch, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
};

// SL: dy arrives as the `sl.splits` float32 partial slabs of the producing split-K input gradient (ACG_SLABS_ROWS: each laid out like
// the tensor) and is summed while it is loaded - the slab reduction launch of that convolution disappears (acg_bn_act_bwd_slabs)
template <int V, typename TX, typename TY, typename TD, int U, int NT, bool SL = false>
__global__ __launch_bounds__(NT) void bn_bwd_fused(const TX* __restrict__ x, const TY* __restrict__ dy, const float* __restrict__ beta,
                                                   const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
                                                   TD* __restrict__ dx, float* __restrict__ dbeta, float dbeta_acc, long long R, int C,
                                                   int groups, int act, float leak, int XP, int YP, unsigned* __restrict__ ws,
                                                   const Slabs sl = Slabs{nullptr, 0, 0, 0}) {
  using EX = FusedExchange<NT>;
  constexpr int CL = EX::CL, RL = NT / CL, NV = EX::NV;
  static_assert(V == 4, "four channels per lane");
  __shared__ float red[EX::NW][NV];
  __shared__ float tot[2][NV];
  __shared__ unsigned s_epoch;
  const int tid = threadIdx.x;
  const int cq = tid % CL, rl = tid / CL;
  const int rb = blockIdx.x, nrb = gridDim.x, cc = blockIdx.y, g = blockIdx.z;
  const int c = (cc * CL + cq) * V;
  const bool cvalid = c < C;
  const TX* xg = x + (long long)g * R * XP;
  TD* dxg = dx + (long long)g * R * XP;
  FusedState* const state = reinterpret_cast<FusedState*>(ws);
  if (tid == 0) s_epoch = __hip_atomic_load(&state->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;

  // ---- phase 1: this block's rows into registers, dp and x-hat in place, partial sums ------------------------------------------
  const long long r0 = (long long)rb * RL * U + rl;
  float xv[U][V], dv[U][V], mean[V], rstd[V], bt[V];
  if (cvalid) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long q = min(r0 + (long long)u * RL, R - 1);
      ldv<V>(xg + q * XP + c, xv[u]);
      ld_or_sum<SL, V>(const_cast<TY*>(dy), ((long long)g * R + q) * YP + c, sl, false, dv[u]);
    }
    ldv<V>(save_mean + g * C + c, mean); ldv<V>(save_rstd + g * C + c, rstd); ldv<V>(beta + c, bt);
  }
  float s[2 * V];
#pragma unroll
  for (int j = 0; j < 2 * V; ++j) s[j] = 0.f;
  if (cvalid) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float w = r0 + (long long)u * RL < R ? 1.f : 0.f;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float xh = (xv[u][j] - mean[j]) * rstd[j];
        const float dp = w * dv[u][j] * acg::act_deriv_pre(act, xh + bt[j], leak);
        xv[u][j] = xh; dv[u][j] = dp;
        s[j] += dp; s[V + j] += dp * xh;
      }
    }
  }
  __syncthreads();                                // s_epoch
  EX ex{red, reinterpret_cast<unsigned long long*>(ws + 4), state, s_epoch, tid, tid & 63, tid >> 6, nrb};
  auto base_of = [&](int gg) { return (((long long)gg * gridDim.y + cc) * nrb) * NV; };      // granules [group][chunk][row block][NV]
  const bool alone = nrb == 1 && groups == 1;       // (with several groups the dbeta block reads the other groups' granules)
  if (alone) {
    ex.local(s, tot[0]);
  } else {
    ex.publish(s, base_of(g) + (long long)rb * NV);
    // ---- phase 2: every row block's partial sums of this (group, chunk); the dbeta block also takes the other groups' ----------------
    ex.gather(base_of(g), tot[0]);
  }
  if (rb == 0 && g == 0) {                        // dbeta of this chunk's channels: sum of s1 over every group, group 0 last
    float d[V];
#pragma unroll
    for (int j = 0; j < V; ++j) d[j] = 0.f;
    for (int gg = groups - 1; gg >= 1; --gg) {
      ex.gather(base_of(gg), tot[1]);
#pragma unroll
      for (int j = 0; j < V; ++j) d[j] += tot[1][cq * 2 * V + j];
      __syncthreads();
    }
    if (cvalid && rl == 0) {
      float o[V];
      if (dbeta_acc != 0.f) ldv<V>(dbeta + c, o);
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = (dbeta_acc != 0.f ? dbeta_acc * o[j] : 0.f) + (d[j] + tot[0][cq * 2 * V + j]);
      stv<V>(dbeta + c, o);
    }
  }
  // ---- phase 3: dx from the registers ---------------------------------------------------------------------------------------------
  if (cvalid) {
    const float invR = 1.f / (float)R;
    float m1[V], m2[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { m1[j] = tot[0][cq * 2 * V + j] * invR; m2[j] = tot[0][cq * 2 * V + V + j] * invR; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long r = r0 + (long long)u * RL;
      if (r < R) {
#pragma unroll
        for (int j = 0; j < V; ++j) dv[u][j] = rstd[j] * (dv[u][j] - m1[j] - xv[u][j] * m2[j]);
        stv<V>(dxg + r * XP + c, dv[u]);
      }
    }
  }
  ex.finish();
}

// Forward, statistics included (layers whose convolution is split over K leave no epilogue partials): one read of x.  The blocks
// exchange sums of (x - ref) and (x - ref)^2 about the group's first row - what bn_stats_partial leaves, same float64 finish.
// SL: x arrives as the partial slabs of the producing split-K convolution (ACG_SLABS_ROWS), is summed while it is loaded and
// written back (rounded to its storage type: the backward pass reads it) - acg_bn_act_fwd_slabs without a reduction launch
template <int V, typename TX, typename TY, int U, int NT, bool SL = false>
__global__ __launch_bounds__(NT) void bn_fwd_fused(TX* __restrict__ x, const float* __restrict__ beta, TY* __restrict__ y,
                                                   float* __restrict__ save_mean, float* __restrict__ save_rstd, long long R, int C, float eps,
                                                   int act, float leak, int XP, int YP, unsigned* __restrict__ ws,
                                                   const Slabs sl = Slabs{nullptr, 0, 0, 0}) {
  using EX = FusedExchange<NT>;
  constexpr int CL = EX::CL, RL = NT / CL, NV = EX::NV;
  static_assert(V == 4, "four channels per lane");
  __shared__ float red[EX::NW][NV];
  __shared__ float tot[NV];
  __shared__ unsigned s_epoch;
  const int tid = threadIdx.x;
  const int cq = tid % CL, rl = tid / CL;
  const int rb = blockIdx.x, nrb = gridDim.x, cc = blockIdx.y, g = blockIdx.z;
  const int c = (cc * CL + cq) * V;
  const bool cvalid = c < C;
  const long long gb = (long long)g * R * XP;
  TY* yg = y + (long long)g * R * YP;
  FusedState* const state = reinterpret_cast<FusedState*>(ws);
  if (tid == 0) s_epoch = __hip_atomic_load(&state->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const long long r0 = (long long)rb * RL * U + rl;
  float xv[U][V], pv[V], bt[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { pv[j] = 0.f; bt[j] = 0.f; }
  if (cvalid) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long r = r0 + (long long)u * RL;
      ld_or_sum<SL, V>(x, gb + min(r, R - 1) * XP + c, sl, r < R, xv[u]);      // (SL: each row is written back by the one thread that owns it)
    }
    ld_or_sum<SL, V>(x, gb + c, sl, false, pv);   // the group's first row: the common shift
    ldv<V>(beta + c, bt);
  }
  float s[2 * V];
#pragma unroll
  for (int j = 0; j < 2 * V; ++j) s[j] = 0.f;
  if (cvalid) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float w = r0 + (long long)u * RL < R ? 1.f : 0.f;
#pragma unroll
      for (int j = 0; j < V; ++j) { const float d = (xv[u][j] - pv[j]) * w; s[j] += d; s[V + j] += d * d; }
    }
  }
  __syncthreads();                                // s_epoch
  EX ex{red, reinterpret_cast<unsigned long long*>(ws + 4), state, s_epoch, tid, tid & 63, tid >> 6, nrb};
  const long long base = (((long long)g * gridDim.y + cc) * nrb) * NV;
  if (nrb == 1) {
    ex.local(s, tot);
  } else {
    ex.publish(s, base + (long long)rb * NV);
    ex.gather(base, tot);
  }
  if (cvalid) {
    float mean[V], rstd[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const double inv = 1.0 / (double)R, dm = (double)tot[cq * 2 * V + j] * inv;
      double var = (double)tot[cq * 2 * V + V + j] * inv - dm * dm;
      var = var > 0.0 ? var : 0.0;
      mean[j] = (float)((double)pv[j] + dm);
      rstd[j] = rsqrtf((float)var + eps);
    }
    if (rb == 0 && rl == 0) { stv<V>(save_mean + g * C + c, mean); stv<V>(save_rstd + g * C + c, rstd); }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long r = r0 + (long long)u * RL;
      if (r < R) {
#pragma unroll
        for (int j = 0; j < V; ++j) xv[u][j] = acg::act_apply(act, (xv[u][j] - mean[j]) * rstd[j] + bt[j], leak);
        stv<V>(yg + r * YP + c, xv[u]);
      }
    }
  }
  ex.finish();
}

// The exchange on its own (acg_bn_exchange_selftest): block b publishes the value b + 1 in all 64 slots, every block gathers the
// grid's sum n (n + 1) / 2 and leaves it in out[b].  `withhold` >= 0: that block publishes nothing - its peers (and itself) run
// into the spin bound, set the timeout word and finish with a short sum: the failure path of bn_fwd_fused / bn_bwd_fused,
// which share every line of FusedExchange with this kernel, provoked on purpose.
template <int NT>
__global__ __launch_bounds__(NT) void exchange_selftest_k(unsigned* __restrict__ ws, float* __restrict__ out, int withhold, unsigned spin_limit) {
  using EX = FusedExchange<NT>;
  __shared__ float red[EX::NW][EX::NV];
  __shared__ float tot[EX::NV];
  __shared__ unsigned s_epoch;
  const int tid = threadIdx.x;
  FusedState* const state = reinterpret_cast<FusedState*>(ws);
  if (tid == 0) s_epoch = __hip_atomic_load(&state->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  __syncthreads();
  EX ex{red, reinterpret_cast<unsigned long long*>(ws + 4), state, s_epoch, tid, tid & 63, tid >> 6, (int)gridDim.x, spin_limit};
  float s[2 * EX::V];
#pragma unroll
  for (int j = 0; j < 2 * EX::V; ++j) s[j] = tid < EX::CL ? (float)(blockIdx.x + 1) : 0.f;     // row lane 0 carries the block's value
  if ((int)blockIdx.x != withhold) ex.publish(s, (long long)blockIdx.x * EX::NV);
  ex.gather(0, tot);
  if (tid == 0) {
    float lo = tot[0], hi = tot[0];
    for (int v = 1; v < EX::NV; ++v) { lo = fminf(lo, tot[v]); hi = fmaxf(hi, tot[v]); }
    out[blockIdx.x] = lo == hi ? lo : -1.f;       // all 64 sums agree in a healthy exchange
  }
  ex.finish();
}

// Grid of the fused kernels for R rows per group.  What a block sweeps in the exchange grows with the number of row blocks of
// its (group, chunk) - 512 bytes each - so the row blocks are kept FEW: 256-thread blocks (128 rows) only up to 16 of them
// (measured: 512 blocks of 128 rows sweeping 128 KB each ran d/conv1's backward at 19.7 us, 64 blocks of 512 rows at 9.5), else
// 1024-thread blocks of 512 or 1024 rows within ONE block per CU; nt = 0: the tensor does not fit, two launches.
struct FusedShape { int nt, U; dim3 grid; };
// The exchange area of a fused launch: 512 bytes (64 granules) per block behind the 16 state bytes.  kMaxFusedBlocks bounds it
// for acg_bn_workspace_bytes whatever the channel count (a chunk of <= 32 channels costs its 512 bytes per row block: with
// the round-4 bound "one block per CU" alone, 131072 x 16 asked for 131 KB of granules in a 66 KB workspace - ADVICE r4).
constexpr int kMaxFusedBlocks = 512;
constexpr size_t kFusedExchangeBytes = (size_t)kMaxFusedBlocks * 512;
FusedShape fused_shape(long long R, int C, int groups, bool no_grid = false, int splits = 0) {
  static const int ncu = [] { int dev = 0, n = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0; (void)hipGetLastError(); return n; }();
  static const int small_rb = env_int("ACG_BN_FUSED_SMALL_ROWBLOCKS", 16);       // tuning hook
  FusedShape f{0, 0, dim3(1, 1, 1)};
  if (ncu <= 0 || no_grid) return f;
  const long long cch = (C + 31) / 32;
  auto blocks = [&](int rows) { return acg::ceil_div(R, rows) * cch * groups; };
  const long long cap1 = std::min<long long>(ncu, kMaxFusedBlocks), cap2 = std::min<long long>(2ll * ncu, kMaxFusedBlocks);
  // Rows that arrive as `splits` split-K slabs cost a thread `splits` loads per row: the same 128 rows per block as 1024 threads x ONE
  // row each (all of a thread's slab loads in flight at once) instead of 256 threads x four rows (four dependent batches) - round 5,
  // the small split layers' BatchNorm was 7-9 us for tensors of 0.1-2 MB
  static const int wide_slabs = env_int("ACG_BN_SLAB_WIDE", 1);                  // tuning hook
  if (wide_slabs && splits >= 2 && acg::ceil_div(R, 128) <= small_rb && blocks(128) <= cap1) { f.nt = 1024; f.U = 1; f.grid = dim3((unsigned)acg::ceil_div(R, 128), (unsigned)cch, (unsigned)groups); }
  else if (acg::ceil_div(R, 128) <= small_rb && blocks(128) <= cap2) { f.nt = 256; f.U = 4; f.grid = dim3((unsigned)acg::ceil_div(R, 128), (unsigned)cch, (unsigned)groups); }
  else if (blocks(512) <= cap1) { f.nt = 1024; f.U = 4; f.grid = dim3((unsigned)acg::ceil_div(R, 512), (unsigned)cch, (unsigned)groups); }
  else if (blocks(1024) <= cap1) { f.nt = 1024; f.U = 8; f.grid = dim3((unsigned)acg::ceil_div(R, 1024), (unsigned)cch, (unsigned)groups); }
  return f;
}

// ---- register-resident BatchNorm for small tensors (R <= 256 * NR rows per group) ------------------------------
// The two-launch path above costs ~7.5-10 us at ANY size below ~1 MB (two dependent launches, each a memory round
// trip: profiles/r1/q_non_conv_ops.txt); half of the BatchNorm'd tensors of the models are that small.  Here a block owns V
// channels of one group, keeps all R rows of them in registers (NR rows per thread), reduces through LDS and writes
// the result: one launch, every element read once.  Loads are 16 bytes per row (V = 4) - a poor use of cache lines,
// irrelevant at these sizes: the tensor was just written by the conv and sits in L2 / the memory-side cache.
template <int N>
__device__ __forceinline__ void block_sum(float (&v)[N], float* sh /* 4 * N floats */) {
#pragma unroll
  for (int j = 0; j < N; ++j) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[j] += __shfl_xor(v[j], off, 64);
  }
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int j = 0; j < N; ++j) sh[wave * N + j] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < N; ++j) v[j] = sh[j] + sh[N + j] + sh[2 * N + j] + sh[3 * N + j];
}

template <int V, int NR, typename TX, typename TY, bool SL = false>
__global__ __launch_bounds__(256) void bn_resident_fwd(TX* __restrict__ x, const float* __restrict__ beta,
                                                       TY* __restrict__ y, float* __restrict__ save_mean,
                                                       float* __restrict__ save_rstd, int R, int C, float eps, int act, float leak, int XP, int YP,
                                                       const Slabs sl) {
  __shared__ float sh[4 * 2 * V];
  const int c = blockIdx.x * V, g = blockIdx.y;
  const long long gb = (long long)g * R * XP + c;
  TY* yg = y + (long long)g * R * YP + c;
  float v[NR][V], pv[V], bt[V];
  const long long qb = ((long long)blockIdx.x * sl.qrows + (long long)g * R) * 4;       // quad layout (V == 4): this block's run of rows
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int rq = min((int)threadIdx.x + u * 256, R - 1);
    ld_or_sum<SL, V>(x, gb + (long long)rq * XP, sl, (int)threadIdx.x + u * 256 < R, v[u], qb + 4ll * rq);
  }
  ld_or_sum<SL, V>(x, gb, sl, false, pv, qb);      // shift by the group's first row, as in bn_stats_partial
  ldv<V>(beta + c, bt);
  float s[2 * V];
#pragma unroll
  for (int j = 0; j < 2 * V; ++j) s[j] = 0.f;
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const float w = (int)threadIdx.x + u * 256 < R ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < V; ++j) { const float d = (v[u][j] - pv[j]) * w; s[j] += d; s[V + j] += d * d; }
  }
  block_sum<2 * V>(s, sh);
  float mean[V], rstd[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const double inv = 1.0 / (double)R, dm = (double)s[j] * inv;
    double var = (double)s[V + j] * inv - dm * dm;
    var = var > 0.0 ? var : 0.0;
    mean[j] = (float)((double)pv[j] + dm);
    rstd[j] = rsqrtf((float)var + eps);        // (the float64 square root and division were most of this kernel's instructions)
  }
  if (threadIdx.x == 0) { stv<V>(save_mean + g * C + c, mean); stv<V>(save_rstd + g * C + c, rstd); }
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int r = (int)threadIdx.x + u * 256;
    if (r < R) {
#pragma unroll
      for (int j = 0; j < V; ++j) v[u][j] = acg::act_apply(act, (v[u][j] - mean[j]) * rstd[j] + bt[j], leak);
      stv<V>(yg + r * YP, v[u]);
    }
  }
}

// One block per V channels walks the groups in turn (dbeta is the sum over groups of its first reduction).
template <int V, int NR, typename TX, typename TY, bool SL = false, typename TD = TX>
__global__ __launch_bounds__(256) void bn_resident_bwd(const TX* __restrict__ x, const TY* __restrict__ dy,
                                                       const float* __restrict__ beta, const float* __restrict__ save_mean,
                                                       const float* __restrict__ save_rstd, TD* __restrict__ dx,
                                                       float* __restrict__ dbeta, float dbeta_acc, int R, int C, int groups,
                                                       int act, float leak, int XP, int YP, const Slabs sl) {
  __shared__ float sh[4 * 2 * V];
  const int c = blockIdx.x * V;
  float bt[V], tot[V];
  ldv<V>(beta + c, bt);
#pragma unroll
  for (int j = 0; j < V; ++j) tot[j] = 0.f;
  for (int g = 0; g < groups; ++g) {
    const long long base = (long long)g * R * XP + c, ybase = (long long)g * R * YP + c;
    float xv[NR][V], dv[NR][V], mean[V], rstd[V];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int rq = min((int)threadIdx.x + u * 256, R - 1);
      ldv<V>(x + base + rq * XP, xv[u]);
      ld_or_sum<SL, V>(const_cast<TY*>(dy), ybase + (long long)rq * YP, sl, false, dv[u], (((long long)blockIdx.x * sl.qrows + (long long)g * R) + rq) * 4);
    }
    ldv<V>(save_mean + g * C + c, mean); ldv<V>(save_rstd + g * C + c, rstd);
    float s[2 * V];
#pragma unroll
    for (int j = 0; j < 2 * V; ++j) s[j] = 0.f;
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const float w = (int)threadIdx.x + u * 256 < R ? 1.f : 0.f;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float xh = (xv[u][j] - mean[j]) * rstd[j];
        const float dp = w * dv[u][j] * acg::act_deriv_pre(act, xh + bt[j], leak);
        xv[u][j] = xh; dv[u][j] = dp;
        s[j] += dp; s[V + j] += dp * xh;
      }
    }
    block_sum<2 * V>(s, sh);
    const float invR = 1.f / (float)R;
#pragma unroll
    for (int j = 0; j < V; ++j) tot[j] += s[j];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int r = (int)threadIdx.x + u * 256;
      if (r < R) {
#pragma unroll
        for (int j = 0; j < V; ++j) dv[u][j] = rstd[j] * (dv[u][j] - s[j] * invR - xv[u][j] * (s[V + j] * invR));
        stv<V>(dx + base + r * XP, dv[u]);
      }
    }
  }
  if (threadIdx.x == 0) {
    float d[V];
    if (dbeta_acc != 0.f) {
      ldv<V>(dbeta + c, d);
#pragma unroll
      for (int j = 0; j < V; ++j) d[j] = dbeta_acc * d[j] + tot[j];
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) d[j] = tot[j];
    }
    stv<V>(dbeta + c, d);
  }
}

// Rows per thread of the resident kernels, 0 = two-launch path.  `R` = rows a block walks (fwd: one group; bwd: all
// groups in turn).  A block pulls one cache line per row through its CU's L1, ~2R cycles: measured break-even
// against the two-launch path is ~2048 rows (profiles/r1/q_non_conv_ops.txt: 512 rows 3.9 vs 7.5 us, 2048 rows 6.1
// vs 7.7 us, 4096 rows 20.9 vs 9.5 us, 8192 rows 23.5 vs 9.3 us).
int resident_nr(long long R, int max_nr) {
  static const int lim = env_int("ACG_BN_RESIDENT_ROWS", 2048);     // tuning hook; 0 disables the resident kernels
  if (R > lim) return 0;
  for (int nr = 1; nr <= max_nr; nr *= 2)
    if (R <= 256ll * nr) return nr;
  return 0;
}

// ---- bias + activation --------------------------------------------------------------------------------------
// x rows are xp elements apart, y rows yp (>= C; pad channels are neither read nor written)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void bias_act_fwd_k(const TI* __restrict__ x, const float* __restrict__ bias,
                                                      TO* __restrict__ y, long long R, int C, int xp, int yp, int act, float leak) {
  const ColMap m = col_map(C);
  if (!m.active) return;
  for (int ch = 0; ch < m.nchunk; ++ch) {
    const int c = ch * m.Cb + m.cl;
    if (c >= C) continue;
    const float bt = bias ? bias[c] : 0.f;
    for (long long r = (long long)blockIdx.x * m.RPP + m.rsub; r < R; r += (long long)gridDim.x * m.RPP)
      acg::stf(y + r * yp + c, acg::act_apply(act, acg::ldf(x + r * xp + c) + bt, leak));
  }
}

// dx = dy * act'(y) and per-block column sums of it -> part[nblk][C]
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void bias_act_bwd_partial(const TO* __restrict__ y, const TO* __restrict__ dy,
                                                            TI* __restrict__ dx, float* __restrict__ part,
                                                            long long R, int C, int xp, int yp, int nblk, int act, float leak) {
  __shared__ float sh[512];
  const ColMap m = col_map(C);
  const int b = blockIdx.x;
  const long long rpb = (R + nblk - 1) / nblk;
  const long long r0 = (long long)b * rpb, r1 = min(R, r0 + rpb);
  for (int ch = 0; ch < m.nchunk; ++ch) {
    const int c = ch * m.Cb + m.cl;
    float s1 = 0.f, s2 = 0.f;
    if (m.active && c < C) {
      // batches of 4 row passes with all loads in flight; y is not read at all for the identity activation
      for (long long r = r0 + m.rsub; r < r1; r += 4 * m.RPP) {
        float g[4], yv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long rr = min(r + u * m.RPP, r1 - 1);
          g[u] = acg::ldf(dy + rr * yp + c);
          yv[u] = act != ACG_ACT_NONE ? acg::ldf(y + rr * yp + c) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (r + u * m.RPP < r1) {
            const float gq = g[u] * acg::act_deriv_out(act, yv[u], leak);
            if (dx) acg::stf(dx + (r + u * m.RPP) * xp + c, gq);
            s1 += gq;
          }
        }
      }
    }
    reduce_rsub(m, s1, s2, sh);
    if (part && m.active && m.rsub == 0 && c < C) part[(long long)b * C + c] = s1;
  }
}

// out[c] = out_acc * out[c] + sum_b part[b][c]; a block owns 32 channels, 8 row lanes per channel walk the partials
__global__ __launch_bounds__(256) void colsum_finalize(const float* __restrict__ part, float* __restrict__ out,
                                                       float out_acc, int C, int nblk) {
  __shared__ double sh[256];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
  double s = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int b = rl; b < nblk; b += 8) s += part[(long long)b * C + c];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
#pragma unroll
    for (int r = 1; r < 8; ++r) s += sh[r * 32 + cl];
    out[c] = (out_acc != 0.f ? out_acc * out[c] : 0.f) + (float)s;
  }
}

int partial_blocks(long long R, int C) {
  const int Cb = C < 256 ? C : 256, RPP = 256 / Cb;
  long long n = R / ((long long)RPP * 8);   // ~8 row passes per block (two batches of loads): the kernel is latency-bound
  if (n < 1) n = 1;
  if (n > kMaxPartialBlocks) n = kMaxPartialBlocks;
  return (int)n;
}
int apply_blocks(long long R, int C) {
  const int Cb = C < 256 ? C : 256, RPP = 256 / Cb;
  long long n = acg::ceil_div(R, (long long)RPP * 4);
  if (n < 1) n = 1;
  if (n > 1024) n = 1024;
  return (int)n;
}
// vectorised variants: columns are C/V wide
// Rows per partial block: the partial kernels are latency-bound on small tensors (a thread's iterations are serial
// round trips to L2/HBM), so a block takes only `iters` row passes - all of them in flight at once - until the block
// count reaches kMaxPartialBlocks (profiles/r1: bn_bwd_partial took ~12.5 us at ANY size with 16 passes per block).
int vpartial_blocks(long long R, int C, int V, int iters = 16) {
  const int Cv = C / V, Cb = Cv < 256 ? Cv : 256, RPP = 256 / Cb;
  long long n = acg::ceil_div(R, (long long)RPP * iters);
  if (n < 1) n = 1;
  if (n > kMaxPartialBlocks) n = kMaxPartialBlocks;
  return (int)n;
}
int vapply_blocks(long long R, int C, int V) {
  const int Cv = C / V, Cb = Cv < 256 ? Cv : 256, RPP = 256 / Cb;
  long long n = acg::ceil_div(R, (long long)RPP * 4);
  if (n < 1) n = 1;
  if (n > 2048) n = 2048;
  return (int)n;
}

// ---- synchronised BatchNorm (optional data-parallel mode; plain elementwise kernels, nothing tuned) -----------
// moments / sums from the per-block partials: one thread per (group, channel), fp64 combine
template <typename TX>
__global__ __launch_bounds__(256) void bn_moments_finalize(const float* __restrict__ part, const TX* __restrict__ x,
                                                           float* __restrict__ moments, long long R, int C, int XP, int groups, int nblk) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= groups * C) return;
  const int g = i / C, c = i - g * C;
  double s1 = 0.0, s2 = 0.0;
  for (int b = 0; b < nblk; ++b) {
    s1 += part[((long long)g * nblk + b) * 2 * C + c];
    s2 += part[((long long)g * nblk + b) * 2 * C + C + c];
  }
  const double shift = (double)acg::ldf(x + (long long)g * R * XP + c), dm = s1 / (double)R;
  double var = s2 / (double)R - dm * dm;
  moments[(g * 2 + 0) * C + c] = (float)(shift + dm);
  moments[(g * 2 + 1) * C + c] = (float)(var > 0.0 ? var : 0.0);
}
// (mean, var) of the global batch -> the saved mean / rstd the apply kernels read (bn_apply_fwd mode kPartDone)
__global__ __launch_bounds__(256) void bn_moments_to_stats(const float* __restrict__ moments, float* __restrict__ save_mean,
                                                           float* __restrict__ save_rstd, int C, int groups, float eps) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= groups * C) return;
  const int g = i / C, c = i - g * C;
  save_mean[i] = moments[(g * 2 + 0) * C + c];
  save_rstd[i] = 1.0f / sqrtf(moments[(g * 2 + 1) * C + c] + eps);
}
// sums[g][{0,1}][c] = sum over the partial blocks; block (x = 32 channels, y = group) of 256 threads = 8 block lanes x 32
// channels, eight loads in flight per thread (one thread per channel walking 512 partials serially cost 100+ us)
__global__ __launch_bounds__(256) void bn_sums_finalize(const float* __restrict__ part, float* __restrict__ sums, int C,
                                                        int groups, int nblk) {
  __shared__ double sh[2][256];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl, g = blockIdx.y;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
#pragma unroll 8
    for (int b = rl; b < nblk; b += 8) {
      const float* o = part + ((long long)g * nblk + b) * 2 * C + c;
      s1 += (double)o[0]; s2 += (double)o[C];
    }
  }
  sh[0][threadIdx.x] = s1; sh[1][threadIdx.x] = s2;
  __syncthreads();
  if (rl == 0 && c < C) {
#pragma unroll
    for (int r = 1; r < 8; ++r) { s1 += sh[0][r * 32 + cl]; s2 += sh[1][r * 32 + cl]; }
    sums[(g * 2 + 0) * C + c] = (float)s1;
    sums[(g * 2 + 1) * C + c] = (float)s2;
  }
}
__global__ __launch_bounds__(256) void bn_dbeta_local_k(const float* __restrict__ local_sums, float* __restrict__ dbeta,
                                                        float acc, int C, int groups) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int g = 0; g < groups; ++g) s += local_sums[(g * 2 + 0) * C + c];
  dbeta[c] = (acc != 0.f ? acc * dbeta[c] : 0.f) + s;
}

int check_bn(const char* who, long long rows, int C, int groups) {
  ACG_REQUIRE(rows > 0 && C > 0 && groups > 0, ACG_ERR_INVALID_ARG, "%s: non-positive size", who);
  ACG_REQUIRE(rows % groups == 0, ACG_ERR_INVALID_ARG, "%s: rows (%lld) not divisible by groups (%d)", who, rows, groups);
  ACG_REQUIRE(C <= kMaxC, ACG_ERR_UNSUPPORTED, "%s: more than %d channels", who, kMaxC);
  ACG_REQUIRE(groups <= 65535, ACG_ERR_UNSUPPORTED, "%s: too many groups", who);
  return ACG_OK;
}

// Launch shape of the tiled apply kernels (bn_apply_fwd / bn_apply_bwd): NT threads = CL channel lanes x NT / CL row lanes, a
// block walks batches of `kU` passes of its row lanes (software-pipelined).  The grid is sized for about `target` blocks in
// all - two per CU - so that on the large tensors a block runs several batches and its prologue over the partials, paid
// once, hides under the first batch's loads; small tensors get one batch per block as before.
struct ApplyShape { int nt, cl; dim3 grid; };
ApplyShape apply_shape(long long R, int C, int V, int groups, int elem_bytes) {
  ApplyShape a;
  static const int cl16 = env_int("ACG_BN_CL16", 0);                 // tuning hook: 16 channel lanes for 2-byte tensors
  a.cl = (elem_bytes == 2 && V == 4 && cl16) ? 16 : 8;
  a.nt = 256;
  const int cchunks = (C + a.cl * V - 1) / (a.cl * V);
  static const int target = env_int("ACG_BN_APPLY_BLOCKS", 512);     // tuning hook
  const long long batch_rows = (long long)(a.nt / a.cl) * kU;         // rows one block takes per batch
  long long n = acg::ceil_div(R, batch_rows);                         // one batch per block ...
  const long long want = std::max<long long>(1, target / ((long long)cchunks * groups));
  if (n > want) n = want;                                             // ... unless that exceeds the target
  a.grid = dim3((unsigned)n, cchunks, groups);
  return a;
}

template <typename TX, typename TY>
int launch_apply_fwd(bool v4, const TX* x, const float* beta, const float* part, TY* y, float* save_mean, float* save_rstd, long long R,
                     int C, int groups, int nblk, float eps, int act, float leak, int XP, int YP, int mode, const TileGeom tg, hipStream_t st) {
  constexpr bool same = std::is_same<TX, TY>::value;
  if (!same) v4 = false;
  const ApplyShape a = apply_shape(R, C, v4 ? 4 : 1, groups, (int)sizeof(TX));
#define ACG_BN_AF(VV, NN, LL) ACG_LAUNCH((bn_apply_fwd<VV, TX, TY, NN, LL>), a.grid, dim3(NN), 0, st, x, beta, part, y, save_mean, save_rstd, R, C, nblk, eps, act, leak, XP, YP, mode, tg)
  if constexpr (same) {
    if (v4) {
      if (a.cl == 16) ACG_BN_AF(4, 256, 16); else ACG_BN_AF(4, 256, 8);
    } else {
      ACG_BN_AF(1, 256, 8);
    }
  } else {
    ACG_BN_AF(1, 256, 8);
  }
#undef ACG_BN_AF
  return acg::check_launch("bn_apply_fwd");
}

template <typename TX, typename TY, typename TD>
int launch_apply_bwd(bool v4, const TX* x, const TY* dy, const float* beta, const float* save_mean, const float* save_rstd, const float* part,
                     TD* dx, float* dbeta, float dbeta_acc, long long R, int C, int groups, int nblk, int act, float leak, int XP, int YP,
                     const float* gsums, float inv_total, hipStream_t st) {
  constexpr bool same = std::is_same<TX, TY>::value;
  if (!same) v4 = false;
  const ApplyShape a = apply_shape(R, C, v4 ? 4 : 1, groups, (int)sizeof(TX));
#define ACG_BN_AB(VV, NN, LL) ACG_LAUNCH((bn_apply_bwd<VV, TX, TY, TD, NN, LL>), a.grid, dim3(NN), 0, st, x, dy, beta, save_mean, save_rstd, part, dx, dbeta, dbeta_acc, R, C, groups, nblk, act, leak, XP, YP, gsums, inv_total)
  if constexpr (same) {
    if (v4) {
      if (a.cl == 16) ACG_BN_AB(4, 256, 16); else ACG_BN_AB(4, 256, 8);
    } else {
      ACG_BN_AB(1, 256, 8);
    }
  } else {
    ACG_BN_AB(1, 256, 8);
  }
#undef ACG_BN_AB
  return acg::check_launch("bn_apply_bwd");
}

bool vec4_ok(int C, const void* a, const void* b, const void* c) {
  return C % 4 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

// One BatchNorm direction for one pair of storage types.  The mixed pair (bf16 x, float32 y: the loss-facing last
// layer of a bf16 network) exists for the scalar (V = 1) variants only.
template <typename TX, typename TY>
int bn_fwd_typed(const void* x, const float* beta, void* y, float* save_mean, float* save_rstd, long long R, int C, int groups,
                 float eps, int act, float leak, float* part, bool v4, int XP, int YP, hipStream_t st, const Slabs sl = Slabs{nullptr, 0, 0, 0},
                 bool no_grid = false) {
  constexpr bool same = std::is_same<TX, TY>::value;
  TX* xf = (TX*)const_cast<void*>(x);
  TY* yf = (TY*)y;
  if (!same) v4 = false;
  const int V = v4 ? 4 : 1;
  float* const scratch = part + 4;       // partial sums of the two-launch path: behind the 16 state bytes the one-launch kernels own
  // One launch, one read of x, statistics included (bn_fwd_fused) where the grid is resident; small tensors keep the
  // one-block-per-four-channels kernels below `fused_min` rows per group (tuning hook)
  static const int fused_on_f = env_int("ACG_BN_FUSED_FWD", 1), fused_min_f = env_int("ACG_BN_FUSED_MIN_ROWS", 1 << 30);
  if constexpr (same) {
    // (slabs in the rows layout: the fused kernel also where the resident ones would apply - its rows are coalesced, theirs cost a
    // cache line per row and slab)
    const bool rows_slabs = sl.p != nullptr && sl.qrows == 0;
    const FusedShape f = (fused_on_f && v4 && (rows_slabs || (!sl.p && (resident_nr(R, 32) == 0 || R >= fused_min_f)))) ? fused_shape(R, C, groups, no_grid, rows_slabs ? sl.splits : 0) : FusedShape{0, 0, dim3(1, 1, 1)};
    if (f.nt) {
#define ACG_BN_FF(UU, NN) do { if (rows_slabs) ACG_LAUNCH((bn_fwd_fused<4, TX, TY, UU, NN, true>), f.grid, dim3(NN), 0, st, xf, beta, yf, save_mean, save_rstd, R, C, eps, act, leak, XP, YP, (unsigned*)part, sl); \
      else ACG_LAUNCH((bn_fwd_fused<4, TX, TY, UU, NN, false>), f.grid, dim3(NN), 0, st, xf, beta, yf, save_mean, save_rstd, R, C, eps, act, leak, XP, YP, (unsigned*)part, sl); } while (0)
      if (f.nt == 256) ACG_BN_FF(4, 256); else if (f.U == 1) ACG_BN_FF(1, 1024); else if (f.U == 4) ACG_BN_FF(4, 1024); else ACG_BN_FF(8, 1024);
#undef ACG_BN_FF
      return acg::check_launch("bn_fwd_fused");
    }
  }
  if (const int nr = resident_nr(R, 32)) {
    const dim3 rg(C / V, groups);
#define ACG_BN_RES_FWD(VV, NN) do { if (sl.p) ACG_LAUNCH((bn_resident_fwd<VV, NN, TX, TY, true>), rg, dim3(256), 0, st, xf, beta, yf, save_mean, save_rstd, (int)R, C, eps, act, leak, XP, YP, sl); \
    else ACG_LAUNCH((bn_resident_fwd<VV, NN, TX, TY, false>), rg, dim3(256), 0, st, xf, beta, yf, save_mean, save_rstd, (int)R, C, eps, act, leak, XP, YP, sl); } while (0)
#define ACG_BN_RES_FWD_V(NN) do { if constexpr (same) { if (v4) ACG_BN_RES_FWD(4, NN); else ACG_BN_RES_FWD(1, NN); } else ACG_BN_RES_FWD(1, NN); } while (0)
    switch (nr) {
      case 1: ACG_BN_RES_FWD_V(1); break;
      case 2: ACG_BN_RES_FWD_V(2); break;
      case 4: ACG_BN_RES_FWD_V(4); break;
      case 8: ACG_BN_RES_FWD_V(8); break;
      case 16: ACG_BN_RES_FWD_V(16); break;
      default: ACG_BN_RES_FWD_V(32); break;
    }
#undef ACG_BN_RES_FWD_V
#undef ACG_BN_RES_FWD
    return acg::check_launch("bn_resident_fwd");
  }
  static const int stats_iters = env_int("ACG_BN_STATS_ITERS", 8);  // tuning hook
  const int nblk = vpartial_blocks(R, C, V, stats_iters);
  if (sl.p) {
    if (same && v4) ACG_LAUNCH((bn_stats_partial<4, TX, true>), dim3(nblk, groups), dim3(256), 0, st, xf, scratch, R, C, nblk, XP, sl);
    else ACG_LAUNCH((bn_stats_partial<1, TX, true>), dim3(nblk, groups), dim3(256), 0, st, xf, scratch, R, C, nblk, XP, sl);
  } else {
    if (same && v4) ACG_LAUNCH((bn_stats_partial<4, TX, false>), dim3(nblk, groups), dim3(256), 0, st, xf, scratch, R, C, nblk, XP, sl);
    else ACG_LAUNCH((bn_stats_partial<1, TX, false>), dim3(nblk, groups), dim3(256), 0, st, xf, scratch, R, C, nblk, XP, sl);
  }
  if (int rc = acg::check_launch("bn_stats_partial")) return rc;
  return launch_apply_fwd<TX, TY>(v4, xf, beta, (const float*)scratch, yf, save_mean, save_rstd, R, C, groups, nblk, eps, act, leak, XP, YP, (int)kPartShifted, TileGeom{0, 0, 1}, st);
}

template <typename TX, typename TY, typename TD = TX>
int bn_bwd_typed(const void* x, const void* dy, const float* beta, const float* save_mean, const float* save_rstd, void* dx,
                 float* dbeta, float dbeta_acc, long long R, int C, int groups, int act, float leak, float* part, bool v4, int XP, int YP,
                 hipStream_t st, const Slabs sl = Slabs{nullptr, 0, 0, 0}, bool no_grid = false) {
  constexpr bool same = std::is_same<TX, TY>::value;
  const TX* xf = (const TX*)x;
  const TY* dyf = (const TY*)dy;
  TD* dxf = (TD*)dx;
  if (!same) v4 = false;
  const int V = v4 ? 4 : 1;
  float* const scratch = part + 4;       // as in bn_fwd_typed
  // small tensors: ONE block per four channels holds every row (bn_resident_bwd) - unless the fused grid kernel is asked to take
  // them from `fused_min` rows per group on (tuning hook; measured in profiles/r4/d_bn_fused_ab.txt)
  static const int fused_min_b = env_int("ACG_BN_FUSED_MIN_ROWS", 2048);      // backward: 2048 x 128 runs 6.1 us fused, 7.7 resident; below, resident wins
  const bool rows_slabs_b = sl.p != nullptr && sl.qrows == 0;
  const bool prefer_fused_b = same && v4 && (rows_slabs_b || (!sl.p && R >= fused_min_b)) && fused_shape(R, C, groups, no_grid).nt != 0;
  if (const int nr = prefer_fused_b ? 0 : (resident_nr(R * groups, 16) ? resident_nr(R, 16) : 0)) {
    const dim3 rg(C / V);
#define ACG_BN_RES_BWD(VV, NN) do { if (sl.p) ACG_LAUNCH((bn_resident_bwd<VV, NN, TX, TY, true, TD>), rg, dim3(256), 0, st, xf, dyf, beta, save_mean, save_rstd, dxf, dbeta, dbeta_acc, (int)R, C, groups, act, leak, XP, YP, sl); \
    else ACG_LAUNCH((bn_resident_bwd<VV, NN, TX, TY, false, TD>), rg, dim3(256), 0, st, xf, dyf, beta, save_mean, save_rstd, dxf, dbeta, dbeta_acc, (int)R, C, groups, act, leak, XP, YP, sl); } while (0)
#define ACG_BN_RES_BWD_V(NN) do { if constexpr (same) { if (v4) ACG_BN_RES_BWD(4, NN); else ACG_BN_RES_BWD(1, NN); } else ACG_BN_RES_BWD(1, NN); } while (0)
    switch (nr) {
      case 1: ACG_BN_RES_BWD_V(1); break;
      case 2: ACG_BN_RES_BWD_V(2); break;
      case 4: ACG_BN_RES_BWD_V(4); break;
      case 8: ACG_BN_RES_BWD_V(8); break;
      default: ACG_BN_RES_BWD_V(16); break;
    }
#undef ACG_BN_RES_BWD_V
#undef ACG_BN_RES_BWD
    return acg::check_launch("bn_resident_bwd");
  }
  // One launch with the tensor held in registers between the sums and the apply phase (bn_bwd_fused) where the whole grid is
  // resident at one block of 1024 threads per CU
  if constexpr (same) {
    static const int fused_on = env_int("ACG_BN_FUSED_BWD", 1);     // tuning hook
    const FusedShape f = (fused_on && v4 && (!sl.p || rows_slabs_b)) ? fused_shape(R, C, groups, no_grid, rows_slabs_b ? sl.splits : 0) : FusedShape{0, 0, dim3(1, 1, 1)};
    if (f.nt) {
#define ACG_BN_FB(UU, NN) do { if (rows_slabs_b) ACG_LAUNCH((bn_bwd_fused<4, TX, TY, TD, UU, NN, true>), f.grid, dim3(NN), 0, st, xf, dyf, beta, save_mean, save_rstd, dxf, dbeta, dbeta_acc, R, C, groups, act, leak, XP, YP, (unsigned*)part, sl); \
      else ACG_LAUNCH((bn_bwd_fused<4, TX, TY, TD, UU, NN, false>), f.grid, dim3(NN), 0, st, xf, dyf, beta, save_mean, save_rstd, dxf, dbeta, dbeta_acc, R, C, groups, act, leak, XP, YP, (unsigned*)part, sl); } while (0)
      if (f.nt == 256) ACG_BN_FB(4, 256); else if (f.U == 1) ACG_BN_FB(1, 1024); else if (f.U == 4) ACG_BN_FB(4, 1024); else ACG_BN_FB(8, 1024);
#undef ACG_BN_FB
      return acg::check_launch("bn_bwd_fused");
    }
  }
  if (sl.p) return acg::fail(ACG_ERR_UNSUPPORTED, "bn_act_bwd_slabs: %lld rows per group fit neither the one-launch grid kernel nor the register-resident kernels (acg_bn_slabs_layout)", R);
  static const int bwd_iters = env_int("ACG_BN_BWD_ITERS", 4);      // tuning hook
  const int nblk = vpartial_blocks(R, C, V, bwd_iters);
  {
    // passes a block walks: more than one batch of kU -> eight passes of loads in flight at once instead of two serial batches
    const int Cv = C / V, Cb = Cv < 256 ? Cv : 256, RPP = 256 / Cb;
    const bool deep = acg::ceil_div(acg::ceil_div(R, nblk), RPP) > kU;
    if constexpr (same) {
      if (v4 && deep) ACG_LAUNCH((bn_bwd_partial<4, TX, TY, 8>), dim3(nblk, groups), dim3(256), 0, st, xf, dyf, beta, save_mean, save_rstd, scratch, R, C, nblk, act, leak, XP, YP);
      else if (v4) ACG_LAUNCH((bn_bwd_partial<4, TX, TY>), dim3(nblk, groups), dim3(256), 0, st, xf, dyf, beta, save_mean, save_rstd, scratch, R, C, nblk, act, leak, XP, YP);
      else ACG_LAUNCH((bn_bwd_partial<1, TX, TY>), dim3(nblk, groups), dim3(256), 0, st, xf, dyf, beta, save_mean, save_rstd, scratch, R, C, nblk, act, leak, XP, YP);
    } else {
      ACG_LAUNCH((bn_bwd_partial<1, TX, TY>), dim3(nblk, groups), dim3(256), 0, st, xf, dyf, beta, save_mean, save_rstd, scratch, R, C, nblk, act, leak, XP, YP);
    }
  }
  if (int rc = acg::check_launch("bn_bwd_partial")) return rc;
  return launch_apply_bwd<TX, TY, TD>(v4, xf, dyf, beta, save_mean, save_rstd, (const float*)scratch, dxf, dbeta, dbeta_acc, R, C, groups, nblk, act, leak, XP, YP,
                                      (const float*)nullptr, 0.f, st);
}

// BatchNorm + activation whose statistics arrive as per-tile (sum, M2) partials out of the producing convolution's epilogue
// (acg_conv2d_fwd_stats / acg_deconv2d_fwd_stats): the apply pass alone - ONE launch, x is read once; above kFinalizeBlocks
// partial blocks per group a small launch merges them first.
constexpr int kFinalizeBlocks = 512;
template <typename TX, typename TY>
int bn_fwd_partials_typed(const void* x, const float* beta, const float* part, int nblk, const TileGeom tg, void* y, float* save_mean,
                          float* save_rstd, long long R, int C, int groups, float eps, int act, float leak, bool v4, int XP, int YP,
                          hipStream_t st) {
  const TX* xf = (const TX*)x;
  TY* yf = (TY*)y;
  const bool v4p = v4;                     // the partials / statistics side is float32 whatever the tensors are
  int mode = kPartTiles;
  static const int fin = env_int("ACG_BN_FINALIZE_BLOCKS", kFinalizeBlocks);   // tuning hook
  if (nblk > fin) {
    if (v4p) ACG_LAUNCH((bn_partials_finalize<4, 2>), dim3(1, (C + 7) / 8, groups), dim3(1024), 0, st, part, save_mean, save_rstd, R, C, nblk, eps, tg);
    else ACG_LAUNCH((bn_partials_finalize<1, 8>), dim3(1, (C + 7) / 8, groups), dim3(1024), 0, st, part, save_mean, save_rstd, R, C, nblk, eps, tg);
    if (int rc = acg::check_launch("bn_partials_finalize")) return rc;
    mode = kPartDone;
  }
  return launch_apply_fwd<TX, TY>(v4, xf, beta, part, yf, save_mean, save_rstd, R, C, groups, nblk, eps, act, leak, XP, YP, mode, tg, st);
}

}  // namespace

extern "C" {

size_t acg_bn_workspace_bytes(int64_t rows, int32_t channels, int32_t groups) {
  (void)rows;
  if (channels <= 0 || groups <= 0) return 0;
  // 16 bytes of state (FusedState: epoch, done, timeout flag, pad) in front of the partial sums (two-launch path) or the exchange
  // granules (one-launch grid kernels: 512 bytes per block, at most kMaxFusedBlocks blocks - fused_shape) of every path
  return 16 + std::max((size_t)groups * ((size_t)kMaxPartialBlocks * 2 + 2) * (size_t)channels * sizeof(float), kFusedExchangeBytes);
}

int32_t acg_bn_act_fwd(const void* x, const float* beta, void* y, float* save_mean, float* save_rstd, int64_t rows,
                       int32_t C, int32_t x_pitch, int32_t y_pitch, int32_t groups, float eps, int32_t act, float leak, int32_t dtype,
                       int32_t flags, void* ws, size_t wsb, acg_stream_t stream) {
  const int XP = x_pitch > 0 ? x_pitch : C, YP = y_pitch > 0 ? y_pitch : C;
  ACG_REQUIRE(XP >= C && YP >= C, ACG_ERR_INVALID_ARG, "bn_act_fwd: pitch smaller than the row");
  if (int rc = check_bn("bn_act_fwd", rows, C, groups)) return rc;
  ACG_REQUIRE(x && beta && y && save_mean && save_rstd, ACG_ERR_INVALID_ARG, "bn_act_fwd: null pointer");
  ACG_REQUIRE(act == ACG_ACT_NONE || act == ACG_ACT_RELU || act == ACG_ACT_LRELU, ACG_ERR_UNSUPPORTED, "bn_act_fwd: activation %d", act);
  ACG_REQUIRE(ws && wsb >= acg_bn_workspace_bytes(rows, C, groups), ACG_ERR_WORKSPACE, "bn_act_fwd: workspace too small");
  const long long R = rows / groups;
  const bool v4 = vec4_ok(C, x, y, beta) && vec4_ok(C, save_mean, save_rstd, ws) && XP % 4 == 0 && YP % 4 == 0;
  ACG_REQUIRE((flags & ~ACG_BN_NO_GRID_EXCHANGE) == 0, ACG_ERR_INVALID_ARG, "bn_act_fwd: unknown flags 0x%x", flags);
  const bool no_grid = (flags & ACG_BN_NO_GRID_EXCHANGE) != 0;
  ACG_WITH_TYPES(dtype, "bn_act_fwd", return (bn_fwd_typed<TA, TB>(x, beta, y, save_mean, save_rstd, R, C, groups, eps, act, leak, (float*)ws, v4, XP, YP, acg::to_stream(stream), Slabs{nullptr, 0, 0, 0}, no_grid)));
}

int32_t acg_bn_act_fwd_partials(const void* x, const float* beta, const float* partials, int32_t nblk, int32_t block_rows, int32_t run_rows,
                                void* y, float* save_mean, float* save_rstd, int64_t rows, int32_t C, int32_t x_pitch, int32_t y_pitch,
                                int32_t groups, float eps, int32_t act, float leak, int32_t dtype, acg_stream_t stream) {
  const int XP = x_pitch > 0 ? x_pitch : C, YP = y_pitch > 0 ? y_pitch : C;
  ACG_REQUIRE(XP >= C && YP >= C, ACG_ERR_INVALID_ARG, "bn_act_fwd_partials: pitch smaller than the row");
  if (int rc = check_bn("bn_act_fwd_partials", rows, C, groups)) return rc;
  ACG_REQUIRE(x && beta && partials && nblk >= 1 && y && save_mean && save_rstd, ACG_ERR_INVALID_ARG, "bn_act_fwd_partials: null pointer / nblk < 1");
  ACG_REQUIRE(act == ACG_ACT_NONE || act == ACG_ACT_RELU || act == ACG_ACT_LRELU, ACG_ERR_UNSUPPORTED, "bn_act_fwd_partials: activation %d", act);
  const long long R = rows / groups;
  // the blocks of a group tile `runs` runs of run_rows rows each, block_rows at a time (acg_conv2d_stats_layout)
  ACG_REQUIRE(block_rows >= 1 && run_rows >= 1, ACG_ERR_INVALID_ARG, "bn_act_fwd_partials: block_rows / run_rows < 1");
  const int bpr = (int)acg::ceil_div(run_rows, block_rows);
  ACG_REQUIRE(nblk % bpr == 0 && (long long)(nblk / bpr) * run_rows == R, ACG_ERR_INVALID_ARG,
              "bn_act_fwd_partials: %d blocks of %d rows in runs of %d do not cover %lld rows per group", nblk, block_rows, run_rows, R);
  const TileGeom tg{block_rows, run_rows, bpr};
  const bool v4 = vec4_ok(C, x, y, beta) && vec4_ok(C, save_mean, save_rstd, partials) && XP % 4 == 0 && YP % 4 == 0;
  ACG_WITH_TYPES(dtype, "bn_act_fwd_partials", return (bn_fwd_partials_typed<TA, TB>(x, beta, partials, nblk, tg, y, save_mean, save_rstd, R, C, groups, eps, act, leak, v4, XP, YP, acg::to_stream(stream))));
}

int32_t acg_bn_act_bwd(const void* x, const void* dy, const float* beta, const float* save_mean, const float* save_rstd,
                       void* dx, float* dbeta, float dbeta_acc, int64_t rows, int32_t C, int32_t x_pitch, int32_t y_pitch,
                       int32_t groups, int32_t act, float leak, int32_t dtype, int32_t flags, void* ws, size_t wsb, acg_stream_t stream) {
  const int XP = x_pitch > 0 ? x_pitch : C, YP = y_pitch > 0 ? y_pitch : C;
  ACG_REQUIRE(XP >= C && YP >= C, ACG_ERR_INVALID_ARG, "bn_act_bwd: pitch smaller than the row");
  ACG_REQUIRE((flags & ~ACG_BN_NO_GRID_EXCHANGE) == 0, ACG_ERR_INVALID_ARG, "bn_act_bwd: unknown flags 0x%x", flags);
  const bool no_grid = (flags & ACG_BN_NO_GRID_EXCHANGE) != 0;
  const Slabs none{nullptr, 0, 0, 0};
  if (int rc = check_bn("bn_act_bwd", rows, C, groups)) return rc;
  ACG_REQUIRE(x && dy && beta && save_mean && save_rstd && dx && dbeta, ACG_ERR_INVALID_ARG, "bn_act_bwd: null pointer");
  ACG_REQUIRE(act == ACG_ACT_NONE || act == ACG_ACT_RELU || act == ACG_ACT_LRELU, ACG_ERR_UNSUPPORTED, "bn_act_bwd: activation %d", act);
  ACG_REQUIRE(ws && wsb >= acg_bn_workspace_bytes(rows, C, groups), ACG_ERR_WORKSPACE, "bn_act_bwd: workspace too small");
  const long long R = rows / groups;
  const bool v4 = vec4_ok(C, x, dy, dx) && vec4_ok(C, save_mean, save_rstd, ws) && vec4_ok(C, beta, dbeta, ws) && XP % 4 == 0 && YP % 4 == 0;
  // ACG_DTYPE2(ACG_F32, ACG_BF16): a head layer of a bf16 network - x and dy float32, dx bf16 (acgan_hip.h)
  if (dtype == ACG_DTYPE2(ACG_F32, ACG_BF16))
    return bn_bwd_typed<float, float, __bf16>(x, dy, beta, save_mean, save_rstd, dx, dbeta, dbeta_acc, R, C, groups, act, leak, (float*)ws, v4, XP, YP, acg::to_stream(stream), none, no_grid);
  ACG_WITH_TYPES(dtype, "bn_act_bwd", return (bn_bwd_typed<TA, TB>(x, dy, beta, save_mean, save_rstd, dx, dbeta, dbeta_acc, R, C, groups, act, leak, (float*)ws, v4, XP, YP, acg::to_stream(stream), none, no_grid)));
}

int32_t acg_bn_act_fwd_slabs(const float* slabs, int32_t splits, void* x, const float* beta, void* y, float* save_mean, float* save_rstd,
                             int64_t rows, int32_t C, int32_t x_pitch, int32_t y_pitch, int32_t groups, float eps, int32_t act, float leak,
                             int32_t dtype, int32_t layout, int32_t flags, void* ws, size_t wsb, acg_stream_t stream) {
  const int XP = x_pitch > 0 ? x_pitch : C, YP = y_pitch > 0 ? y_pitch : C;
  ACG_REQUIRE(XP >= C && YP >= C, ACG_ERR_INVALID_ARG, "bn_act_fwd_slabs: pitch smaller than the row");
  ACG_REQUIRE((flags & ~ACG_BN_NO_GRID_EXCHANGE) == 0, ACG_ERR_INVALID_ARG, "bn_act_fwd_slabs: unknown flags 0x%x", flags);
  ACG_REQUIRE(layout == ACG_SLABS_ROWS || (layout == ACG_SLABS_QUADS && acg_bn_slabs_layout(rows, C, x_pitch, y_pitch, groups, dtype, 0, flags) == ACG_SLABS_QUADS),
              ACG_ERR_UNSUPPORTED, "bn_act_fwd_slabs: slab layout %d for this tensor (acg_bn_slabs_layout)", layout);
  if (int rc = check_bn("bn_act_fwd_slabs", rows, C, groups)) return rc;
  ACG_REQUIRE(slabs && splits >= 1 && x && beta && y && save_mean && save_rstd, ACG_ERR_INVALID_ARG, "bn_act_fwd_slabs: null pointer / splits < 1");
  ACG_REQUIRE(act == ACG_ACT_NONE || act == ACG_ACT_RELU || act == ACG_ACT_LRELU, ACG_ERR_UNSUPPORTED, "bn_act_fwd_slabs: activation %d", act);
  ACG_REQUIRE(ws && wsb >= acg_bn_workspace_bytes(rows, C, groups), ACG_ERR_WORKSPACE, "bn_act_fwd_slabs: workspace too small");
  const long long R = rows / groups;
  const bool v4 = vec4_ok(C, x, y, beta) && vec4_ok(C, save_mean, save_rstd, ws) && vec4_ok(C, slabs, slabs, slabs) && XP % 4 == 0 && YP % 4 == 0;
  ACG_REQUIRE(layout == ACG_SLABS_ROWS || v4, ACG_ERR_INVALID_ARG, "bn_act_fwd_slabs: the quad layout needs 16-byte aligned pointers");
  const Slabs sl{slabs, splits, (long long)rows * XP, layout == ACG_SLABS_QUADS ? (long long)rows : 0ll};
  ACG_WITH_TYPES(dtype, "bn_act_fwd_slabs", return (bn_fwd_typed<TA, TB>(x, beta, y, save_mean, save_rstd, R, C, groups, eps, act, leak, (float*)ws, v4, XP, YP, acg::to_stream(stream), sl, (flags & ACG_BN_NO_GRID_EXCHANGE) != 0)));
}

int32_t acg_bn_bwd_slabs_ok(int64_t rows, int32_t groups) {
  if (rows <= 0 || groups <= 0 || rows % groups) return 0;
  return (resident_nr(rows, 16) && resident_nr(rows / groups, 16)) ? 1 : 0;
}

int32_t acg_bn_slabs_layout(int64_t rows, int32_t C, int32_t x_pitch, int32_t y_pitch, int32_t groups, int32_t dtype, int32_t backward,
                            int32_t flags) {
  if (rows <= 0 || C <= 0 || groups <= 0 || rows % groups || (flags & ~ACG_BN_NO_GRID_EXCHANGE)) return -1;
  const int XP = x_pitch > 0 ? x_pitch : C, YP = y_pitch > 0 ? y_pitch : C;
  const bool vec = C % 4 == 0 && XP % 4 == 0 && YP % 4 == 0 && acg::dt_valid(dtype) && acg::dt_first(dtype) == acg::dt_second(dtype);
  // the one-launch grid kernels (bn_fwd_fused / bn_bwd_fused) sum slabs laid out like the tensor while they load their rows:
  // coalesced, any tensor whose grid is resident
  static const int fused_slabs = env_int("ACG_BN_SLABS_FUSED", 1);      // tuning hook
  if (fused_slabs && vec && fused_shape(rows / groups, C, groups, (flags & ACG_BN_NO_GRID_EXCHANGE) != 0).nt != 0) return ACG_SLABS_ROWS;
  const bool resident = backward ? acg_bn_bwd_slabs_ok(rows, groups) != 0 : resident_nr(rows / groups, 32) != 0;
  if (backward && !resident) return -1;
  // the register-resident kernels with four channels per block read [channels / 4][rows][4] slabs as consecutive 16-byte rows
  return (resident && vec) ? ACG_SLABS_QUADS : ACG_SLABS_ROWS;
}

int32_t acg_bn_act_bwd_slabs(const void* x, const float* dy_slabs, int32_t splits, const float* beta, const float* save_mean,
                             const float* save_rstd, void* dx, float* dbeta, float dbeta_acc, int64_t rows, int32_t C, int32_t x_pitch,
                             int32_t y_pitch, int32_t groups, int32_t act, float leak, int32_t dtype, int32_t layout, int32_t flags, void* ws, size_t wsb,
                             acg_stream_t stream) {
  const int XP = x_pitch > 0 ? x_pitch : C, YP = y_pitch > 0 ? y_pitch : C;
  ACG_REQUIRE(XP >= C && YP >= C, ACG_ERR_INVALID_ARG, "bn_act_bwd_slabs: pitch smaller than the row");
  ACG_REQUIRE((flags & ~ACG_BN_NO_GRID_EXCHANGE) == 0, ACG_ERR_INVALID_ARG, "bn_act_bwd_slabs: unknown flags 0x%x", flags);
  ACG_REQUIRE(layout == ACG_SLABS_ROWS || (layout == ACG_SLABS_QUADS && acg_bn_slabs_layout(rows, C, x_pitch, y_pitch, groups, dtype, 1, flags) == ACG_SLABS_QUADS),
              ACG_ERR_UNSUPPORTED, "bn_act_bwd_slabs: slab layout %d for this tensor (acg_bn_slabs_layout)", layout);
  if (int rc = check_bn("bn_act_bwd_slabs", rows, C, groups)) return rc;
  ACG_REQUIRE(x && dy_slabs && splits >= 1 && beta && save_mean && save_rstd && dx && dbeta, ACG_ERR_INVALID_ARG, "bn_act_bwd_slabs: null pointer / splits < 1");
  ACG_REQUIRE(act == ACG_ACT_NONE || act == ACG_ACT_RELU || act == ACG_ACT_LRELU, ACG_ERR_UNSUPPORTED, "bn_act_bwd_slabs: activation %d", act);
  ACG_REQUIRE(acg_bn_slabs_layout(rows, C, x_pitch, y_pitch, groups, dtype, 1, flags) >= 0, ACG_ERR_UNSUPPORTED, "bn_act_bwd_slabs: tensor too large for the one-launch kernels (acg_bn_slabs_layout)");
  const long long R = rows / groups;
  const bool v4 = vec4_ok(C, x, dy_slabs, dx) && vec4_ok(C, save_mean, save_rstd, beta) && vec4_ok(C, beta, dbeta, dbeta) && XP % 4 == 0 && YP % 4 == 0;
  ACG_REQUIRE(layout == ACG_SLABS_ROWS || v4, ACG_ERR_INVALID_ARG, "bn_act_bwd_slabs: the quad layout needs 16-byte aligned pointers");
  const Slabs sl{dy_slabs, splits, (long long)rows * YP, layout == ACG_SLABS_QUADS ? (long long)rows : 0ll};
  ACG_WITH_TYPES(dtype, "bn_act_bwd_slabs", return (bn_bwd_typed<TA, TB>(x, nullptr, beta, save_mean, save_rstd, dx, dbeta, dbeta_acc, R, C, groups, act, leak, (float*)ws, v4, XP, YP, acg::to_stream(stream), sl, (flags & ACG_BN_NO_GRID_EXCHANGE) != 0)));
}

int32_t acg_bn_exchange_selftest(void* ws, size_t wsb, float* out, int32_t blocks, int32_t threads, int32_t withhold, uint32_t spin_limit,
                                 acg_stream_t stream) {
  ACG_REQUIRE(ws && out, ACG_ERR_INVALID_ARG, "bn_exchange_selftest: null pointer");
  ACG_REQUIRE(blocks >= 1 && blocks <= kMaxFusedBlocks, ACG_ERR_INVALID_ARG, "bn_exchange_selftest: %d blocks (1..%d)", blocks, kMaxFusedBlocks);
  ACG_REQUIRE(threads == 256 || threads == 1024, ACG_ERR_INVALID_ARG, "bn_exchange_selftest: %d threads per block (256 or 1024)", threads);
  ACG_REQUIRE(withhold < blocks, ACG_ERR_INVALID_ARG, "bn_exchange_selftest: withheld block %d of %d", withhold, blocks);
  ACG_REQUIRE(wsb >= 16 + (size_t)blocks * 512, ACG_ERR_WORKSPACE, "bn_exchange_selftest: workspace too small");
  hipStream_t st = acg::to_stream(stream);
  if (threads == 256) ACG_LAUNCH((exchange_selftest_k<256>), dim3(blocks), dim3(256), 0, st, (unsigned*)ws, out, withhold, spin_limit);
  else ACG_LAUNCH((exchange_selftest_k<1024>), dim3(blocks), dim3(1024), 0, st, (unsigned*)ws, out, withhold, spin_limit);
  return acg::check_launch("exchange_selftest_k");
}

}  // extern "C"

// ---- synchronised BatchNorm entries: any storage-type pair the plain entries take, pitched rows ---------------------------
namespace {
template <typename TX>
int bn_moments_typed(const void* x, float* moments, long long R, int C, int XP, int groups, float* part, bool v4, hipStream_t st) {
  TX* xf = (TX*)const_cast<void*>(x);
  const int V = v4 ? 4 : 1, nblk = vpartial_blocks(R, C, V, 8);
  if (v4) ACG_LAUNCH((bn_stats_partial<4, TX>), dim3(nblk, groups), dim3(256), 0, st, xf, part, R, C, nblk, XP, Slabs{nullptr, 0, 0});
  else ACG_LAUNCH((bn_stats_partial<1, TX>), dim3(nblk, groups), dim3(256), 0, st, xf, part, R, C, nblk, XP, Slabs{nullptr, 0, 0});
  if (int rc = acg::check_launch("bn_stats_partial")) return rc;
  ACG_LAUNCH((bn_moments_finalize<TX>), dim3((groups * C + 255) / 256), dim3(256), 0, st, (const float*)part, (const TX*)x, moments, R, C, XP, groups, nblk);
  return acg::check_launch("bn_moments_finalize");
}
template <typename TX, typename TY>
int bn_fwd_moments_typed(const void* x, const float* beta, void* y, float* save_mean, float* save_rstd, long long R, int C, int groups,
                         float eps, int act, float leak, bool v4, int XP, int YP, hipStream_t st) {
  return launch_apply_fwd<TX, TY>(v4, (const TX*)x, beta, (const float*)nullptr, (TY*)y, save_mean, save_rstd, R, C, groups, 0, eps, act, leak, XP, YP,
                                  (int)kPartDone, TileGeom{0, 0, 1}, st);
}
template <typename TX, typename TY>
int bn_bwd_sums_typed(const void* x, const void* dy, const float* beta, const float* save_mean, const float* save_rstd, float* sums,
                      long long R, int C, int groups, int act, float leak, float* part, bool v4, int XP, int YP, hipStream_t st) {
  constexpr bool same = std::is_same<TX, TY>::value;
  if (!same) v4 = false;
  const int V = v4 ? 4 : 1, nblk = vpartial_blocks(R, C, V, 4);
  if constexpr (same) {
    if (v4) ACG_LAUNCH((bn_bwd_partial<4, TX, TY>), dim3(nblk, groups), dim3(256), 0, st, (const TX*)x, (const TY*)dy, beta, save_mean, save_rstd, part, R, C, nblk, act, leak, XP, YP);
    else ACG_LAUNCH((bn_bwd_partial<1, TX, TY>), dim3(nblk, groups), dim3(256), 0, st, (const TX*)x, (const TY*)dy, beta, save_mean, save_rstd, part, R, C, nblk, act, leak, XP, YP);
  } else {
    ACG_LAUNCH((bn_bwd_partial<1, TX, TY>), dim3(nblk, groups), dim3(256), 0, st, (const TX*)x, (const TY*)dy, beta, save_mean, save_rstd, part, R, C, nblk, act, leak, XP, YP);
  }
  if (int rc = acg::check_launch("bn_bwd_partial")) return rc;
  ACG_LAUNCH(bn_sums_finalize, dim3((C + 31) / 32, groups), dim3(256), 0, st, (const float*)part, sums, C, groups, nblk);
  return acg::check_launch("bn_sums_finalize");
}
template <typename TX, typename TY, typename TD = TX>
int bn_bwd_apply_sums_typed(const void* x, const void* dy, const float* beta, const float* save_mean, const float* save_rstd,
                            const float* gsums, float inv_total, void* dx, long long R, int C, int groups, int act, float leak, bool v4,
                            int XP, int YP, hipStream_t st) {
  return launch_apply_bwd<TX, TY, TD>(v4, (const TX*)x, (const TY*)dy, beta, save_mean, save_rstd, (const float*)nullptr, (TD*)dx, (float*)nullptr, 0.f,
                                      R, C, groups, 0, act, leak, XP, YP, gsums, inv_total, st);
}
}  // namespace

extern "C" {

int32_t acg_bn_moments(const void* x, float* moments, int64_t rows, int32_t C, int32_t x_pitch, int32_t groups, int32_t dtype, void* ws,
                       size_t wsb, acg_stream_t stream) {
  const int XP = x_pitch > 0 ? x_pitch : C;
  ACG_REQUIRE(XP >= C, ACG_ERR_INVALID_ARG, "bn_moments: pitch smaller than the row");
  if (int rc = check_bn("bn_moments", rows, C, groups)) return rc;
  ACG_REQUIRE(x && moments, ACG_ERR_INVALID_ARG, "bn_moments: null pointer");
  ACG_REQUIRE(ws && wsb >= acg_bn_workspace_bytes(rows, C, groups), ACG_ERR_WORKSPACE, "bn_moments: workspace too small");
  ACG_REQUIRE(dtype == ACG_F32 || dtype == ACG_BF16, ACG_ERR_UNSUPPORTED, "bn_moments: dtype %d (the storage type of x)", dtype);
  const long long R = rows / groups;
  const bool v4 = vec4_ok(C, x, x, ws) && XP % 4 == 0;
  if (dtype == ACG_BF16) return bn_moments_typed<__bf16>(x, moments, R, C, XP, groups, (float*)ws, v4, acg::to_stream(stream));
  return bn_moments_typed<float>(x, moments, R, C, XP, groups, (float*)ws, v4, acg::to_stream(stream));
}

int32_t acg_bn_act_fwd_moments(const void* x, const float* beta, const float* moments, void* y, float* save_mean,
                               float* save_rstd, int64_t rows, int32_t C, int32_t x_pitch, int32_t y_pitch, int32_t groups, float eps,
                               int32_t act, float leak, int32_t dtype, acg_stream_t stream) {
  const int XP = x_pitch > 0 ? x_pitch : C, YP = y_pitch > 0 ? y_pitch : C;
  ACG_REQUIRE(XP >= C && YP >= C, ACG_ERR_INVALID_ARG, "bn_act_fwd_moments: pitch smaller than the row");
  if (int rc = check_bn("bn_act_fwd_moments", rows, C, groups)) return rc;
  ACG_REQUIRE(x && beta && moments && y && save_mean && save_rstd, ACG_ERR_INVALID_ARG, "bn_act_fwd_moments: null pointer");
  ACG_REQUIRE(act == ACG_ACT_NONE || act == ACG_ACT_RELU || act == ACG_ACT_LRELU, ACG_ERR_UNSUPPORTED, "bn_act_fwd_moments: activation %d", act);
  hipStream_t st = acg::to_stream(stream);
  ACG_LAUNCH(bn_moments_to_stats, dim3((groups * C + 255) / 256), dim3(256), 0, st, moments, save_mean, save_rstd, C, groups, eps);
  if (int rc = acg::check_launch("bn_moments_to_stats")) return rc;
  const long long R = rows / groups;
  const bool v4 = vec4_ok(C, x, y, beta) && vec4_ok(C, save_mean, save_rstd, beta) && XP % 4 == 0 && YP % 4 == 0;
  ACG_WITH_TYPES(dtype, "bn_act_fwd_moments", return (bn_fwd_moments_typed<TA, TB>(x, beta, y, save_mean, save_rstd, R, C, groups, eps, act, leak, v4, XP, YP, st)));
}

int32_t acg_bn_bwd_sums(const void* x, const void* dy, const float* beta, const float* save_mean, const float* save_rstd,
                        float* sums, int64_t rows, int32_t C, int32_t x_pitch, int32_t y_pitch, int32_t groups, int32_t act, float leak,
                        int32_t dtype, void* ws, size_t wsb, acg_stream_t stream) {
  const int XP = x_pitch > 0 ? x_pitch : C, YP = y_pitch > 0 ? y_pitch : C;
  ACG_REQUIRE(XP >= C && YP >= C, ACG_ERR_INVALID_ARG, "bn_bwd_sums: pitch smaller than the row");
  if (int rc = check_bn("bn_bwd_sums", rows, C, groups)) return rc;
  ACG_REQUIRE(x && dy && beta && save_mean && save_rstd && sums, ACG_ERR_INVALID_ARG, "bn_bwd_sums: null pointer");
  ACG_REQUIRE(ws && wsb >= acg_bn_workspace_bytes(rows, C, groups), ACG_ERR_WORKSPACE, "bn_bwd_sums: workspace too small");
  const long long R = rows / groups;
  hipStream_t st = acg::to_stream(stream);
  const bool v4 = vec4_ok(C, x, dy, ws) && vec4_ok(C, save_mean, save_rstd, beta) && XP % 4 == 0 && YP % 4 == 0;
  if (dtype == ACG_DTYPE2(ACG_F32, ACG_BF16))      // the float32 head of a bf16 network: x and dy float32 (dx, bf16, is not touched here)
    return bn_bwd_sums_typed<float, float>(x, dy, beta, save_mean, save_rstd, sums, R, C, groups, act, leak, (float*)ws, v4, XP, YP, st);
  ACG_WITH_TYPES(dtype, "bn_bwd_sums", return (bn_bwd_sums_typed<TA, TB>(x, dy, beta, save_mean, save_rstd, sums, R, C, groups, act, leak, (float*)ws, v4, XP, YP, st)));
}

int32_t acg_bn_act_bwd_sums(const void* x, const void* dy, const float* beta, const float* save_mean, const float* save_rstd,
                            const float* sums, const float* local_sums, int64_t total_rows, void* dx, float* dbeta,
                            float dbeta_acc, int64_t rows, int32_t C, int32_t x_pitch, int32_t y_pitch, int32_t groups, int32_t act,
                            float leak, int32_t dtype, acg_stream_t stream) {
  const int XP = x_pitch > 0 ? x_pitch : C, YP = y_pitch > 0 ? y_pitch : C;
  ACG_REQUIRE(XP >= C && YP >= C, ACG_ERR_INVALID_ARG, "bn_act_bwd_sums: pitch smaller than the row");
  if (int rc = check_bn("bn_act_bwd_sums", rows, C, groups)) return rc;
  ACG_REQUIRE(x && dy && beta && save_mean && save_rstd && sums && local_sums && dx && dbeta, ACG_ERR_INVALID_ARG, "bn_act_bwd_sums: null pointer");
  ACG_REQUIRE(total_rows >= rows / groups, ACG_ERR_INVALID_ARG, "bn_act_bwd_sums: total_rows smaller than this rank's rows");
  hipStream_t st = acg::to_stream(stream);
  const long long R = rows / groups;
  const float inv_total = 1.0f / (float)total_rows;
  const bool v4 = vec4_ok(C, x, dy, dx) && vec4_ok(C, save_mean, save_rstd, beta) && vec4_ok(C, sums, sums, sums) && XP % 4 == 0 && YP % 4 == 0;
  int rc;
  if (dtype == ACG_DTYPE2(ACG_F32, ACG_BF16)) {
    rc = bn_bwd_apply_sums_typed<float, float, __bf16>(x, dy, beta, save_mean, save_rstd, sums, inv_total, dx, R, C, groups, act, leak, v4, XP, YP, st);
  } else {
    rc = ACG_OK;
    ACG_WITH_TYPES(dtype, "bn_act_bwd_sums", rc = (bn_bwd_apply_sums_typed<TA, TB>(x, dy, beta, save_mean, save_rstd, sums, inv_total, dx, R, C, groups, act, leak, v4, XP, YP, st)));
  }
  if (rc) return rc;
  ACG_LAUNCH(bn_dbeta_local_k, dim3((C + 255) / 256), dim3(256), 0, st, local_sums, dbeta, dbeta_acc, C, groups);
  return acg::check_launch("bn_dbeta_local_k");
}

size_t acg_bias_workspace_bytes(int64_t rows, int32_t channels) {
  (void)rows;
  return channels > 0 ? (size_t)kMaxPartialBlocks * (size_t)channels * sizeof(float) : 0;
}

int32_t acg_bias_act_fwd(const void* x, const float* bias, void* y, int64_t rows, int32_t C, int32_t x_pitch, int32_t y_pitch,
                         int32_t act, float leak, int32_t dtype, acg_stream_t stream) {
  ACG_REQUIRE(rows > 0 && C > 0, ACG_ERR_INVALID_ARG, "bias_act_fwd: non-positive size");
  ACG_REQUIRE(x && y, ACG_ERR_INVALID_ARG, "bias_act_fwd: null pointer");
  ACG_REQUIRE(act >= ACG_ACT_NONE && act <= ACG_ACT_TANH, ACG_ERR_INVALID_ARG, "bias_act_fwd: activation %d", act);
  const int xp = x_pitch > 0 ? x_pitch : C, yp = y_pitch > 0 ? y_pitch : C;
  ACG_REQUIRE(xp >= C && yp >= C, ACG_ERR_INVALID_ARG, "bias_act_fwd: pitch smaller than the row");
  ACG_WITH_TYPES(dtype, "bias_act_fwd",
                 ACG_LAUNCH((bias_act_fwd_k<TA, TB>), dim3(apply_blocks(rows, C)), dim3(256), 0, acg::to_stream(stream), (const TA*)x, bias,
                            (TB*)y, (long long)rows, C, xp, yp, act, leak));
  return acg::check_launch("bias_act_fwd");
}

int32_t acg_bias_act_bwd(const void* y, const void* dy, void* dx, float* dbias, float dbias_acc, int64_t rows, int32_t C,
                         int32_t x_pitch, int32_t y_pitch, int32_t act, float leak, int32_t dtype, void* ws, size_t wsb,
                         acg_stream_t stream) {
  const int xp = x_pitch > 0 ? x_pitch : C, yp = y_pitch > 0 ? y_pitch : C;
  ACG_REQUIRE(xp >= C && yp >= C, ACG_ERR_INVALID_ARG, "bias_act_bwd: pitch smaller than the row");
  ACG_REQUIRE(rows > 0 && C > 0, ACG_ERR_INVALID_ARG, "bias_act_bwd: non-positive size");
  ACG_REQUIRE(y && dy && (dbias || dx), ACG_ERR_INVALID_ARG, "bias_act_bwd: null pointer");
  ACG_REQUIRE(act >= ACG_ACT_NONE && act <= ACG_ACT_TANH, ACG_ERR_INVALID_ARG, "bias_act_bwd: activation %d", act);
  ACG_REQUIRE(dx || act == ACG_ACT_NONE, ACG_ERR_INVALID_ARG, "bias_act_bwd: dx NULL requires ACG_ACT_NONE");
  ACG_REQUIRE(!dbias || (ws && wsb >= acg_bias_workspace_bytes(rows, C)), ACG_ERR_WORKSPACE, "bias_act_bwd: workspace too small");
  const int nblk = partial_blocks(rows, C);
  hipStream_t st = acg::to_stream(stream);
  ACG_WITH_TYPES(dtype, "bias_act_bwd",
                 ACG_LAUNCH((bias_act_bwd_partial<TA, TB>), dim3(nblk), dim3(256), 0, st, (const TB*)y, (const TB*)dy, (TA*)dx,
                            dbias ? (float*)ws : (float*)nullptr, (long long)rows, C, xp, yp, nblk, act, leak));
  if (int rc = acg::check_launch("bias_act_bwd_partial")) return rc;
  if (!dbias) return ACG_OK;
  ACG_LAUNCH(colsum_finalize, dim3((C + 31) / 32), dim3(256), 0, st, (const float*)ws, dbias, dbias_acc, C, nblk);
  return acg::check_launch("colsum_finalize");
}

}  // extern "C"
