// NHWC implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), gfx950.
//
// One templated kernel serves the three contractions of a conv layer and, through the adjoint
// descriptor, of a TF conv2d_transpose layer (include/acgan_hip.h):
//   FWD    y [M=(b,p,q)][N=o]    = sum_{k=(tap,c)}   G[m][k]      * W[k][n]
//   DGRAD  dx[M=(b,h2,w2)][N=c]  = sum_{k=(tap,o)}   dY~[m][k]    * W^T[k][n]   per stride-parity class
//   WGRAD  dw[M=(tap,c)][N=o]    = sum_{k=(b,p,q)}   G[k][m]      * dY[k][n]
// G is the never-materialised im2col matrix: each block keeps, per gathered row, a base offset and a
// 64-bit mask of in-bounds filter taps in LDS, so the inner gather is one shift/and + one 16-byte load.
// DGRAD runs as stride_h*stride_w parity classes (blockIdx.y), each a dense stride-1 correlation over
// dY with its own tap subset - no multiplies by structural zeros (5x5/s2: 9+6+6+4 = 25 taps in total).
//
// K (or, for WGRAD, M) is indexed as (tap, channel) with the channel count padded per tap to a multiple
// of 4, so a "quad" of 4 consecutive k never straddles a tap and is one dwordx4 gather along NHWC
// channels (any Cin: 3, 6, 138, 266 included - the ragged last quad of a tap is loaded element-wise).
// LDS holds the A and B tiles as quads, [k/4][P(m) ^ k/4] float4 (column permutation P in the kernel): the
// 8 lanes of every ds_write_b128 group and the 16 lanes of every ds_read_b128 group hit distinct 16-byte
// slots, with no padding.  Operands that are contiguous along the tile column instead of along k (the
// filter in FWD, both operands in WGRAD) are loaded as 4x4 blocks and transposed in registers.  Because the MFMA contracts k-slot (lane>>5) of A with the same slot of B, the k order
// inside a K-step is free: lane half h reads quad 2t+h of both tiles and feeds its four floats to four
// consecutive MFMAs - one ds_read_b128 per 32x32 tile per 8 k.
//
// Tiling: 256 threads = 4 waves, block tile 64 x 64 x 32 (128 x 32 for N <= 32), wave tile 32 x 32.  Global loads
// run NST-1 K-steps ahead of the MFMAs through NST register stages and a double-buffered LDS tile (one barrier per
// K-step; pipeline notes in conv_f32_kernel.h).  Small grids are filled by split-K over blockIdx.z into workspace
// slabs that a second kernel sums in fixed order (deterministic; no float atomics).
#include <stdlib.h>

#include "conv_bf16_kernel.h"

using namespace acgconv;

namespace {

// out[i] = accumulate * out[i] + sum_z slabs[z][i]   (fixed summation order); `block` of `nblocks` 256-thread blocks
__device__ __forceinline__ void reduce_slabs(const float* __restrict__ slabs, float* __restrict__ out, long long numel,
                                             int splits, float accumulate, int block, int nblocks) {
  const long long stride = (long long)nblocks * 256;
  const bool al = ((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 && (numel & 3) == 0;
  const long long n4 = al ? numel / 4 : 0;
  for (long long i = (long long)block * 256 + threadIdx.x; i < n4; i += stride) {
    f4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int z = 0; z < splits; ++z) s += reinterpret_cast<const f4*>(slabs + (long long)z * numel)[i];   // 8 loads in flight, summed in z order
    f4* o = reinterpret_cast<f4*>(out) + i;
    if (accumulate != 0.f) s += accumulate * *o;
    *o = s;
  }
  for (long long i = n4 * 4 + (long long)block * 256 + threadIdx.x; i < numel; i += stride) {
    float s = 0.f;
#pragma unroll 8
    for (int z = 0; z < splits; ++z) s += slabs[(long long)z * numel + i];
    out[i] = (accumulate != 0.f ? accumulate * out[i] : 0.f) + s;
  }
}

__global__ __launch_bounds__(256) void splitk_reduce(const float* __restrict__ slabs, float* __restrict__ out,
                                                     long long numel, int splits, float accumulate) {
  reduce_slabs(slabs, out, numel, splits, accumulate, blockIdx.x, gridDim.x);
}

// bf16 activations (FWD / DGRAD of the bf16 path): out[i] = bf16(sum_z slabs[z][i]); numel is a multiple of 8 (pitch round8)
__global__ __launch_bounds__(256) void splitk_reduce_bf16(const float* __restrict__ slabs, __bf16* __restrict__ out,
                                                          long long numel, int splits) {
  const long long n4 = numel / 4, stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int z = 0; z < splits; ++z) s += reinterpret_cast<const f4*>(slabs + (long long)z * numel)[i];
    reinterpret_cast<bf4*>(out)[i] = bf4{(__bf16)s[0], (__bf16)s[1], (__bf16)s[2], (__bf16)s[3]};
  }
}

// fp32 master weights [taps][A][B] -> the two bf16 operand layouts of the bf16 conv kernels, zero padded to multiples
// of 8: rm [taps][A][B8] (k-fast along B) and tr [taps][B][A8] (k-fast along A).  All filters of a scope in ONE launch:
// entry e owns blocks [first_block[e], first_block[e+1]); the first n_rm[e] of them write rm - one 16-byte oct per
// thread and step, reading 8 consecutive floats - the rest write tr through 32 x 32 LDS tile transposes, so that both
// the float32 reads (along B) and the bf16 writes (along A) are coalesced.
struct PrepList {
  const float* src[ACG_PREP_MAX];
  __bf16* rm[ACG_PREP_MAX];
  __bf16* tr[ACG_PREP_MAX];
  int taps[ACG_PREP_MAX], A[ACG_PREP_MAX], B[ACG_PREP_MAX];
  int first_block[ACG_PREP_MAX + 1], n_rm[ACG_PREP_MAX];
};
__global__ __launch_bounds__(256) void weights_prepare_bf16(const PrepList l, int count) {
  __shared__ float tile[32][33];
  int e = 0;
  while (e + 1 < count && (int)blockIdx.x >= l.first_block[e + 1]) ++e;
  const int A = l.A[e], B = l.B[e], A8 = (A + 7) & ~7, B8 = (B + 7) & ~7, taps = l.taps[e];
  const float* __restrict__ src = l.src[e];
  const int blk = (int)blockIdx.x - l.first_block[e], nrm = l.n_rm[e], ntr = l.first_block[e + 1] - l.first_block[e] - nrm;
  if (blk < nrm) {
    const int ob = B8 >> 3;                                   // octs per (tap, a) row
    const long long nocts = (long long)taps * A * ob;
    const bool fast = (B & 7) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    for (long long i = (long long)blk * 256 + threadIdx.x; i < nocts; i += (long long)nrm * 256) {
      const long long ta = i / ob;
      const int b0 = (int)(i - ta * ob) * 8;
      float v[8];
      if (fast) {
        const f4 lo = *reinterpret_cast<const f4*>(src + ta * B + b0), hi = *reinterpret_cast<const f4*>(src + ta * B + b0 + 4);
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = b0 + u < B ? src[ta * B + b0 + u] : 0.f;
      }
      bf8 o;
#pragma unroll
      for (int u = 0; u < 8; ++u) o[u] = (__bf16)v[u];
      *reinterpret_cast<bf8*>(l.rm[e] + ta * B8 + b0) = o;
    }
    return;
  }
  // tr: tiles of 32 (a) x 32 (b) of one tap; pad columns a in [A, A8) are written as zeros by the tiles that cover them
  const int ta_n = (A8 + 31) / 32, tb_n = (B + 31) / 32;
  const long long ntiles = (long long)taps * ta_n * tb_n;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8 threads
  for (long long t = blk - nrm; t < ntiles; t += ntr) {
    const int tb = (int)(t % tb_n);
    const long long r = t / tb_n;
    const int ta = (int)(r % ta_n), tap = (int)(r / ta_n);
    const int a0 = ta * 32, b0 = tb * 32;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int a = a0 + ty + 8 * k, b = b0 + tx;
      tile[ty + 8 * k][tx] = (a < A && b < B) ? src[((long long)tap * A + a) * B + b] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int b = b0 + ty + 8 * k, a = a0 + tx;
      if (b < B && a < A8) l.tr[e][((long long)tap * B + b) * A8 + a] = (__bf16)tile[tx][ty + 8 * k];
    }
  }
}

// The slab reductions of several weight gradients in ONE launch (acg_splitk_reduce_many): a launch costs ~4-5 us in
// the step's HIP graph whatever its size, and nothing reads a weight gradient before the optimizer (or the
// all-reduce of its bucket).  Entry e owns blocks [first_block[e], first_block[e+1]); same arithmetic as above.
struct ReduceList {
  const float* slabs[ACG_REDUCE_MAX];
  float* out[ACG_REDUCE_MAX];
  long long numel[ACG_REDUCE_MAX];
  int splits[ACG_REDUCE_MAX];
  float accumulate[ACG_REDUCE_MAX];
  int first_block[ACG_REDUCE_MAX + 1];
  int* step_inc;           // acg_reduce_list::step_inc
};
__global__ __launch_bounds__(256) void splitk_reduce_many(const ReduceList l, int count) {
  if (l.step_inc != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *l.step_inc += 1;
  int e = 0;
  while (e + 1 < count && (int)blockIdx.x >= l.first_block[e + 1]) ++e;      // block-uniform scan of <= 32 entries
  reduce_slabs(l.slabs[e], l.out[e], l.numel[e], l.splits[e], l.accumulate[e], (int)blockIdx.x - l.first_block[e],
               l.first_block[e + 1] - l.first_block[e]);
}

// ---- merged input gradient of a stride-2 layer with few input channels (run_merged below) ---------------------------
// The four stride-parity classes of such an input gradient (d/conv1: 6 channels; a plain generator's last transposed layer:
// 3) each fill 6 of the 32 MFMA columns of a tile.  They share their gathered rows - class (ph, pw) of output pixel pair
// (h2, w2) reads dy rows h2 + dh for the dh with i = ph + pad - 2 dh inside the filter - so ONE stride-1 contraction over
// the union window (3 x 3 for a 5 x 5 filter) with N = 4 * Cin columns computes all four: column n = (2 ph + pw) * Cin + c
// uses the filter W'[a][b][o][n] = W[ph + pt - 2 (a + lo_h)][pw + pl - 2 (b + lo_w)][c][o], zero where that tap does not
// exist.  9 instead of 25 tap-steps per pixel quad, one epilogue instead of four.  These kernels build W' (a few KB) in front
// of the contraction; float32: [tap][o][n]; bf16: from the 'rm' copy [tap][c][o8] into the layout FWD reads, [tap][n][o8].
struct MergeGeom { int kh, kw, cin, cout, nwh, nww, lo_h, lo_w, pt, pl; };
__global__ __launch_bounds__(256) void merge_dgrad_weights_f32(const float* __restrict__ w, float* __restrict__ wm, const MergeGeom g, int total) {
  const int idx = (int)(blockIdx.x * 256 + threadIdx.x);
  if (idx >= total) return;
  const int np = 4 * g.cin, n = idx % np, o = (idx / np) % g.cout, tap = idx / (np * g.cout);
  const int a = tap / g.nww, b = tap - a * g.nww, cls = n / g.cin, c = n - cls * g.cin;
  const int i = (cls >> 1) + g.pt - 2 * (a + g.lo_h), j = (cls & 1) + g.pl - 2 * (b + g.lo_w);
  wm[idx] = (i >= 0 && i < g.kh && j >= 0 && j < g.kw) ? w[((long long)(i * g.kw + j) * g.cin + c) * g.cout + o] : 0.f;
}
__global__ __launch_bounds__(256) void merge_dgrad_weights_bf16(const __bf16* __restrict__ rm, __bf16* __restrict__ wm, const MergeGeom g, int total) {
  const int idx = (int)(blockIdx.x * 256 + threadIdx.x);
  if (idx >= total) return;
  const int np = 4 * g.cin, o8n = (g.cout + 7) & ~7, o = idx % o8n, n = (idx / o8n) % np, tap = idx / (o8n * np);
  const int a = tap / g.nww, b = tap - a * g.nww, cls = n / g.cin, c = n - cls * g.cin;
  const int i = (cls >> 1) + g.pt - 2 * (a + g.lo_h), j = (cls & 1) + g.pl - 2 * (b + g.lo_w);
  wm[idx] = (o < g.cout && i >= 0 && i < g.kh && j >= 0 && j < g.kw) ? rm[((long long)(i * g.kw + j) * g.cin + c) * o8n + o] : (__bf16)0.f;
}

int reduce_blocks(long long numel) { return (int)std::max<long long>(1, std::min<long long>(acg::ceil_div(numel, 256 * 4), 2048)); }

// ---- host-side planning ------------------------------------------------------------------------------

int validate(const acg_conv_desc* d, const char* who) {
  ACG_REQUIRE(d != nullptr, ACG_ERR_INVALID_ARG, "%s: null descriptor", who);
  ACG_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->in_c > 0 && d->out_h > 0 && d->out_w > 0 && d->out_c > 0 &&
                  d->kh > 0 && d->kw > 0 && d->stride_h > 0 && d->stride_w > 0 && d->pad_top >= 0 && d->pad_left >= 0,
              ACG_ERR_INVALID_ARG, "%s: non-positive dimension in descriptor", who);
  ACG_REQUIRE(d->in_pitch == 0 || d->in_pitch >= d->in_c, ACG_ERR_INVALID_ARG, "%s: in_pitch %d smaller than in_c %d", who, d->in_pitch, d->in_c);
  ACG_REQUIRE(d->dgrad_c >= 0 && d->dgrad_c <= d->in_c && d->adj_dgrad_c >= 0 && d->adj_dgrad_c <= d->out_c, ACG_ERR_INVALID_ARG,
              "%s: dgrad_c %d / adj_dgrad_c %d outside [0, in_c = %d] / [0, out_c = %d]", who, d->dgrad_c, d->adj_dgrad_c, d->in_c, d->out_c);
  ACG_REQUIRE(d->out_pitch == 0 || d->out_pitch >= d->out_c, ACG_ERR_INVALID_ARG, "%s: out_pitch %d smaller than out_c %d", who, d->out_pitch, d->out_c);
  ACG_REQUIRE(d->kh * d->kw <= kMaxTaps, ACG_ERR_UNSUPPORTED, "%s: %dx%d filter exceeds %d taps", who, d->kh, d->kw, kMaxTaps);
  ACG_REQUIRE(d->pad_top < d->kh && d->pad_left < d->kw, ACG_ERR_INVALID_ARG, "%s: padding not smaller than the filter", who);
  ACG_REQUIRE((d->out_h - 1) * d->stride_h - d->pad_top < d->in_h && (d->out_w - 1) * d->stride_w - d->pad_left < d->in_w,
              ACG_ERR_INVALID_ARG, "%s: output extent reads entirely outside the input", who);
  const long long lim = 1ll << 30;   // 32-bit byte offsets in the buffer descriptors
  const long long nx = (long long)d->batch * d->in_h * d->in_w * (d->in_pitch > 0 ? d->in_pitch : d->in_c);
  const long long ny = (long long)d->batch * d->out_h * d->out_w * (d->out_pitch > 0 ? d->out_pitch : d->out_c);
  const long long nw = (long long)d->kh * d->kw * ((d->in_c + 3) & ~3) * d->out_c;
  ACG_REQUIRE(nx < lim && ny < lim && nw < lim, ACG_ERR_UNSUPPORTED, "%s: tensor exceeds 2^30 elements", who);
  return ACG_OK;
}

// Tuning hooks exist in -DACG_TUNING builds only (libacgan_hip_tuning.so, `make tuning`, used by tools/): the shipped
// library has no process-wide mutable state and reads no environment variable.
#ifdef ACG_TUNING
int g_force_cfg = -1, g_force_splits = -1;   // acg_debug_conv_plan: force a tile configuration / split-K factor; -1 = heuristic
int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}
#else
constexpr int g_force_cfg = -1, g_force_splits = -1;
constexpr int env_int(const char*, int dflt) { return dflt; }
#endif

// wide_ok: the 256 x 128 LDS-DMA kernel (conv_bf16_glds.h) may be chosen - single launches only, it has no paired form
Plan make_plan(const acg_conv_desc& d, int which, bool bf16 = false, bool wide_ok = true) {
  Plan pl{};
  pl.bf16 = bf16;
  long long K;
  const int pad = bf16 ? 7 : 3;       // channels per tap are padded to the 16-byte unit: 4 floats / 8 bf16
  const long long cin_p = (d.in_c + pad) & ~pad, cout_p = (d.out_c + pad) & ~pad;
  if (which == ACG_CONV_FWD) {
    pl.M = (long long)d.batch * d.out_h * d.out_w; pl.N = d.adj_dgrad_c > 0 ? d.adj_dgrad_c : d.out_c; K = (long long)d.kh * d.kw * cin_p; pl.classes = 1;
    pl.out_numel = pl.M * (bf16 ? cout_p : (d.out_pitch > 0 ? d.out_pitch : d.out_c));
  } else if (which == ACG_CONV_DGRAD) {
    const int hc = (d.in_h + d.stride_h - 1) / d.stride_h, wc = (d.in_w + d.stride_w - 1) / d.stride_w;
    pl.M = (long long)d.batch * hc * wc; pl.N = d.dgrad_c > 0 ? d.dgrad_c : d.in_c;
    K = (long long)((d.kh + d.stride_h - 1) / d.stride_h) * ((d.kw + d.stride_w - 1) / d.stride_w) * cout_p;
    pl.classes = d.stride_h * d.stride_w;
    pl.out_numel = (long long)d.batch * d.in_h * d.in_w * (bf16 ? cin_p : (d.in_pitch > 0 ? d.in_pitch : d.in_c));
  } else {
    pl.M = (long long)d.kh * d.kw * cin_p; pl.N = d.out_c; K = (long long)d.batch * d.out_h * d.out_w; pl.classes = 1;
    pl.out_numel = (long long)d.kh * d.kw * d.in_c * d.out_c;
  }
  // channel pitch of the gathered tensor decides whether its quads are 16-byte loads
  const int ky = d.out_pitch > 0 ? d.out_pitch : d.out_c;
  const int cs = which == ACG_CONV_DGRAD ? ky : (d.in_pitch > 0 ? d.in_pitch : d.in_c);
  pl.ragged = (cs & 3) != 0;
  // dense operand rows: the filter [.., N] (FWD) or dY at its channel pitch (WGRAD)
  pl.nvec = (((which == ACG_CONV_WGRAD ? ky : (int)pl.N) | (which == ACG_CONV_FWD ? d.out_c : 0)) & 3) == 0;    // (FWD: the filter rows are out_c floats apart)
  const int bk = bf16 ? BKH : BK;
  pl.nk = (int)((K + bk - 1) / bk);
  if (pl.nk < 1) pl.nk = 1;
  // The few-MFLOP layers (d/conv6, g/sconv4, g/sconv5): one launch of the direct kernels instead of 2-16 K-splits of 1-8
  // tiles plus a slab reduction (conv_direct.hip).  All three contractions of a layer take the same decision.
  static const int direct_on = env_int("ACG_PLAN_DIRECT", 1);
  const double flops = 2.0 * d.batch * d.out_h * d.out_w * d.kh * d.kw * (double)d.in_c * d.out_c;
  // float32, at most 8 output channels (d/conv6, g/sconv5): measured wins of 4-7 us per contraction.  Sixteen channels
  // (g/sconv4) lose - every lane then walks 16 strided dword loads per pixel and the launch is latency-bound at 10-19 us
  // against 9 us tiled - and the bf16 tiled path, which does not split these layers, is already at 5 us
  // (profiles/r3/d_direct_small_convs.txt).
  if (direct_on && !bf16 && g_force_cfg < 0 && g_force_splits < 0 && d.out_c <= 8 && flops <= 8.0e6) {
    pl.direct = true; pl.cfg = 9; pl.bm = pl.bn = 64; pl.tiles = 1; pl.splits = 1; pl.ragged = false; pl.nvec = true;
    return pl;
  }
  auto tiles_for = [&](int bm, int bn) { return acg::ceil_div(pl.M, bm) * acg::ceil_div(pl.N, bn) * pl.classes; };
  // Planner (evidence: profiles/r1 tuning sweeps, tests/fit_planner.py).  Split K until ~1 block per CU for
  // FWD/DGRAD and ~2 per CU for WGRAD (long K = B*OH*OW, heavier loaders: a second resident block hides its
  // latencies), but keep >= 4 K-steps per block so the slab reduction does not take over.
  pl.cfg = pl.N <= 32 ? 2 : 3;
  if (g_force_cfg >= 0 && g_force_cfg < 4) pl.cfg = g_force_cfg <= 2 ? 2 : 3;
  pl.bm = pl.cfg == 2 ? 128 : 64; pl.bn = pl.cfg == 2 ? 32 : 64;
  if (bf16) {
    // bf16: 128x128 tiles once they fill the chip (a K-step then carries 16 MFMAs per wave for the same 8 loads per
    // thread as two 64x64 blocks carry 8), 64x64 tiles + split-K below that
    // (profiles/r2/b_tune_bf16_*.txt: 128x128 wins from ~256 tiles; weight gradients - few output tiles, parallelism
    // from split-K - take it once K is long: nk >= 256)
    static const int big_min = env_int("ACG_PLAN16_BIG_TILES", 256);
    const bool big = pl.N > 64 && pl.M >= 128 && (which == ACG_CONV_WGRAD ? pl.nk >= 256 : tiles_for(128, 128) >= big_min);
    pl.cfg = big ? 1 : 3; pl.bm = pl.bn = big ? 128 : 64;
    // at most 32 output columns (g/tconv4's 25 logits, d/conv1's 6-channel input gradient, g/conv1, the state head): the
    // 128x32 tile - no MFMA columns of padding beyond the first 32 (forward / input gradient only)
    static const int narrow = env_int("ACG_PLAN16_NARROW", 1);
    if (narrow && which != ACG_CONV_WGRAD && pl.N <= 32 && pl.M >= 128) { pl.cfg = 2; pl.bm = 128; pl.bn = 32; }
    if (g_force_cfg == 1 || g_force_cfg == 3) { pl.cfg = g_force_cfg; pl.bm = pl.bn = g_force_cfg == 1 ? 128 : 64; }
    if (g_force_cfg == 2 && which != ACG_CONV_WGRAD) { pl.cfg = 2; pl.bm = 128; pl.bn = 32; }
    // 256 x 128, operands staged by LDS-DMA, one block of 8 waves per CU (conv_bf16_glds.h): unsplit forward / input-gradient
    // contractions with more than 64 output columns whose tiles fill most of the chip AND whose K loop is long.  Measured
    // (profiles/r4/e_glds_*): +8 % inside the K loop (954 vs 883 TFLOP/s: the loop is bound by what a CU can gather from L2,
    // ~43 GB/s, not by LDS), 35.2 vs 39.9 us for a 25-K-step forward over 256 tiles - and nothing on the stride classes of a
    // transposed layer (8-18 K-steps per tile: with one block per CU the pipeline fill and the epilogue of every tile are
    // exposed where two 128 x 128 blocks per CU overlap them)
    static const int wide_min = env_int("ACG_PLAN16_WIDE_TILES", 192), wide_nk = env_int("ACG_PLAN16_WIDE_NK", 20);
    if (wide_ok && which != ACG_CONV_WGRAD && g_force_cfg < 0 && pl.N > 64 && pl.nk >= wide_nk && tiles_for(256, 128) >= wide_min) { pl.cfg = 4; pl.bm = 256; pl.bn = 128; }
    if (g_force_cfg == 4 && which != ACG_CONV_WGRAD) { pl.cfg = 4; pl.bm = 256; pl.bn = 128; }
    pl.ragged = false; pl.nvec = true;
  }
  pl.tiles = tiles_for(pl.bm, pl.bn);
  // tuning hooks (whole-step sweeps): ACG_PLAN_TARGET_FD / _W / ACG_PLAN_MIN_STEPS override the constants below
  static const int t_fd = env_int("ACG_PLAN_TARGET_FD", 0), t_w = env_int("ACG_PLAN_TARGET_W", 512), min_steps = env_int("ACG_PLAN_MIN_STEPS", 4);
  // bf16 K-steps are ~2x cheaper (matrix cores, loader-bound): two resident blocks per CU pay off there as well
  // (whole-step sweep: 507 vs 500 steps/s)
  // bf16: a K-step is ~4x cheaper than in fp32 while a slab round trip (fp32 slabs + the reduce launch) costs the same:
  // split only below half a chip of tiles, up to one block per CU (weight gradients 1.5) - profiles/r2 sweeps
  const long long target = bf16 ? (which == ACG_CONV_WGRAD ? 384 : 256) : (which == ACG_CONV_WGRAD ? t_w : (t_fd > 0 ? t_fd : 256));
  long long s = target / pl.tiles;
  if (bf16 && pl.tiles >= 128) s = (pl.tiles < 256 && pl.nk >= 64) ? 2 : 1;     // half a chip of tiles with a long K: two blocks per tile
  s = std::min<long long>(s, std::max(1, pl.nk / min_steps));
  static const int max_splits = env_int("ACG_PLAN_MAX_SPLITS", 128), max_splits_w = env_int("ACG_PLAN_MAX_SPLITS_W", 128);
  s = std::min<long long>(s, which == ACG_CONV_WGRAD ? max_splits_w : max_splits);
  // Long-K contractions that fill the chip with ONE block per CU (g/tconv3's dgrad: 256 tiles x 100 K-steps): a second
  // resident block hides the first one's load latencies (0.74 -> 0.53 us per K-step) and the slab reduction is small
  // next to a long K loop (profiles/r1/r_conv_tune_full.txt: 79.8 us unsplit, 67.1 us at 4 splits).
  s = std::max<long long>(s, 1);
  if (!bf16 && which != ACG_CONV_WGRAD && pl.tiles * s <= 256 && pl.nk / s >= 50) s *= pl.nk / s >= 100 ? 4 : 2;
  // Weight gradients: every output tile of one K range gathers the SAME pixels (at its own taps), so the blocks of a split
  // are placed on ONE XCD and share that range in its L2 (wgrad_xcd_map, conv_f32_kernel.h) - which needs the split count
  // to be a multiple of the 8 XCDs.
  static const int xcd_splits = env_int("ACG_PLAN_WGRAD_XCD", 1);
  if (xcd_splits && bf16 && which == ACG_CONV_WGRAD && s >= 6 && (s + 7) / 8 * 8 <= pl.nk) s = (s + 7) / 8 * 8;   // (fp32: the swept split counts stay - rounding them cost 1.3 % of the step)
  if (g_force_splits >= 1) s = std::min<long long>(g_force_splits, pl.nk);
  if (pl.cfg == 4) s = 1;
  pl.splits = (int)std::max<long long>(s, 1);
  return pl;
}

// One contraction, checked and planned: everything a launch needs.
struct Job {
  int which;
  Plan pl;
  ConvArgs a;
  void* ws;
  float* out;
  float accumulate;
  bool slabs_only;
};

// BatchNorm statistics out of the epilogue (ConvArgs::stats): partial blocks per group, 0 when this plan cannot provide
// them - a split contraction holds partial sums only; a tile must not straddle two groups; the parity classes of a
// DGRAD (a transposed layer's forward) must be of one size, or the smaller ones would leave partial blocks unwritten.
constexpr int kMaxStatsBlocks = 4096;
int stats_blocks(const Plan& pl, const acg_conv_desc& d, int which, int groups, int* tiles_per_group, int* run_rows = nullptr) {
  if (which == ACG_CONV_WGRAD || pl.splits != 1 || groups < 1 || pl.direct) return 0;
  const long long tiles_m = acg::ceil_div(pl.M, pl.bm);
  long long nblk = 0, tpg = 0, run = 0;
  if (which == ACG_CONV_DGRAD) {
    if (groups != 1 || d.in_h % d.stride_h || d.in_w % d.stride_w) return 0;
    tpg = tiles_m; nblk = tiles_m * pl.classes; run = pl.M;        // a run of tiles = one stride class
  } else if (groups == 1) {
    tpg = nblk = tiles_m; run = pl.M;
  } else {
    if (pl.M % groups || (pl.M / groups) % pl.bm) return 0;
    tpg = nblk = pl.M / groups / pl.bm; run = pl.M / groups;
  }
  if (nblk < 1 || nblk > kMaxStatsBlocks || run >= (1ll << 31)) return 0;
  if (tiles_per_group) *tiles_per_group = (int)tpg;
  if (run_rows) *run_rows = (int)run;
  return (int)nblk;
}

// slabs_only (weight gradients whose reduction is deferred to acg_splitk_reduce_many): the contraction leaves its
// `splits` partial slabs in the workspace and `out` is not touched.
int prepare(Job& j, int which, const float* gsrc, const float* dense, float* out, float accumulate, const acg_conv_desc* d,
            int dtype, void* ws, size_t ws_bytes, const char* who, bool slabs_only, float* stats = nullptr, int stats_groups = 0,
            int slab_layout = ACG_SLABS_ROWS, bool wide_ok = true) {
  // ACG_DTYPE2(ACG_BF16, ACG_F32): bf16 operands, the result stored as float32 (a head layer: acgan_hip.h)
  const bool out_f32 = dtype == ACG_DTYPE2(ACG_BF16, ACG_F32);
  ACG_REQUIRE(dtype == ACG_F32 || dtype == ACG_BF16 || (out_f32 && which != ACG_CONV_WGRAD), ACG_ERR_UNSUPPORTED, "%s: dtype %d", who, dtype);
  if (out_f32) dtype = ACG_BF16;
  if (int rc = validate(d, who)) return rc;
  ACG_REQUIRE(gsrc && dense && (out || slabs_only), ACG_ERR_INVALID_ARG, "%s: null tensor pointer", who);
  const Plan pl = make_plan(*d, which, dtype == ACG_BF16, wide_ok);
  ACG_REQUIRE(!slabs_only || pl.splits > 1, ACG_ERR_INVALID_ARG, "%s: this shape is not split (acg_conv2d_splits == 1): call the plain entry", who);
  const size_t need = pl.splits > 1 ? (size_t)pl.splits * (size_t)pl.out_numel * sizeof(float) : 0;
  ACG_REQUIRE(ws_bytes >= need && (need == 0 || ws != nullptr), ACG_ERR_WORKSPACE, "%s: workspace %zu bytes < required %zu", who, ws_bytes, need);
  ConvArgs a{};
  a.gsrc = gsrc; a.dense = dense; a.out = pl.splits > 1 ? (float*)ws : out; a.out_numel = pl.out_numel;
  a.accumulate = accumulate;
  a.out_f32 = out_f32 ? 1 : 0;
  const bool h = dtype == ACG_BF16;
  const int cin8 = (d->in_c + 7) & ~7, cout8 = (d->out_c + 7) & ~7;
  if (h) {   // bf16 tensors: activations at pitch round8(C); `dense` is a prepared filter copy (acg_weights_prepare_bf16)
    ACG_REQUIRE((d->in_pitch == 0 || d->in_pitch == cin8) && (d->out_pitch == 0 || d->out_pitch == cout8), ACG_ERR_INVALID_ARG,
                "%s: bf16 tensors are stored at the channel pitch round8(C)", who);
    const long long nx = (long long)d->batch * d->in_h * d->in_w * cin8, ny = (long long)d->batch * d->out_h * d->out_w * cout8;
    const long long nw = (long long)d->kh * d->kw * (which == ACG_CONV_FWD ? (long long)d->out_c * cin8 : (long long)d->in_c * cout8);
    a.g_bytes = (unsigned)((which == ACG_CONV_DGRAD ? ny : nx) * 2);
    a.d_bytes = (unsigned)((which == ACG_CONV_WGRAD ? ny : nw) * 2);
  } else {
    const long long nx = (long long)d->batch * d->in_h * d->in_w * (d->in_pitch > 0 ? d->in_pitch : d->in_c), ny = (long long)d->batch * d->out_h * d->out_w * (d->out_pitch > 0 ? d->out_pitch : d->out_c);
    const long long nw = (long long)d->kh * d->kw * d->in_c * d->out_c;
    const long long ng = which == ACG_CONV_DGRAD ? ny : nx;                       // gathered tensor
    const long long nd = which == ACG_CONV_WGRAD ? ny : nw;                       // dense operand
    a.g_bytes = (unsigned)(ng * 4); a.d_bytes = (unsigned)(nd * 4);
  }
  a.Cx = d->in_pitch > 0 ? d->in_pitch : d->in_c;
  a.Ky = d->out_pitch > 0 ? d->out_pitch : d->out_c;
  a.batch = d->batch; a.H = d->in_h; a.W = d->in_w; a.C = d->in_c; a.OH = d->out_h; a.OW = d->out_w; a.K = d->out_c;
  a.KH = d->kh; a.KW = d->kw; a.sh = d->stride_h; a.sw = d->stride_w; a.pt = d->pad_top; a.pl = d->pad_left;
  a.splits = pl.splits;
  a.Nv = which == ACG_CONV_DGRAD ? d->dgrad_c : (which == ACG_CONV_FWD ? d->adj_dgrad_c : 0);
  {     // bf16 forward / input gradient with 128 or more gathered channels in whole 64-channel chunks: chunk-major K order (option)
    static const int korder = env_int("ACG_CONV16_KORDER", 0);      // measured level (c5 +0.4 %, c3 -0.3 %, profiles/r4/f_korder_ab.txt): off; tuning builds can switch it on
    const int cp = which == ACG_CONV_DGRAD ? cout8 : cin8;
    a.korder = (h && which != ACG_CONV_WGRAD && korder && cp >= 128 && cp % 64 == 0) ? 1 : 0;
  }
  {     // strided forward convolutions walk their taps class by class (conv_f32_kernel.h, tap_class_pos): the L2 then holds a tile's window
    static const int tap_classes = env_int("ACG_CONV_TAP_CLASSES", 0);      // tuning hook: bit 0 bf16, bit 1 float32; measured: K-loop of a 5x5 / stride-2 forward -10 % at >= 128 channels, whole steps level (profiles/r4/h_bf16_kloop_load_path.txt): off
    a.tap_classes = ((tap_classes & (h ? 1 : 2)) && which == ACG_CONV_FWD && d->stride_h == 2 && d->stride_w == 2 && d->kh * d->kw > 1 && d->kh * d->kw <= 64) ? 1 : 0;
  }
  // small maps: pixel-major rows + only the taps a tile's rows can see (ConvArgs::compact); not with epilogue statistics,
  // whose partial blocks are runs of rows of one group
  // measured: pays only where more than half of the taps are dead (a 4 x 4 map under a 5 x 5 / stride-2 filter: d/conv5) - from
  // 16 pixels per class on, what the pixel-major order loses in gather reuse inside a tile eats the skipped K-steps
  static const int compact_max = env_int("ACG_PLAN_COMPACT", 4);
  {
    const int hc = (d->in_h + d->stride_h - 1) / d->stride_h, wc = (d->in_w + d->stride_w - 1) / d->stride_w;
    const int px = which == ACG_CONV_DGRAD ? hc * wc : d->out_h * d->out_w;
    a.compact = (which != ACG_CONV_WGRAD && !h && stats == nullptr && !pl.direct && px <= compact_max && d->batch >= 8 && d->kh * d->kw > 1) ? 1 : 0;
  }
  if (slab_layout != ACG_SLABS_ROWS) {      // slabs for the layer's BatchNorm in the layout its one-launch kernels read (acgan_hip.h)
    const long long orows = which == ACG_CONV_DGRAD ? (long long)d->batch * d->in_h * d->in_w : (long long)d->batch * d->out_h * d->out_w;
    const int oc = which == ACG_CONV_DGRAD ? d->in_c : d->out_c, opitch = which == ACG_CONV_DGRAD ? (h ? cin8 : a.Cx) : (h ? cout8 : a.Ky);
    ACG_REQUIRE(slab_layout == ACG_SLABS_QUADS && slabs_only && which != ACG_CONV_WGRAD, ACG_ERR_INVALID_ARG, "%s: slab layout %d", who, slab_layout);
    ACG_REQUIRE(((oc + 3) & ~3) <= opitch && orows * 4 < (1ll << 31), ACG_ERR_UNSUPPORTED, "%s: the quad slab layout needs round4(channels) <= pitch", who);
    a.slab_rows = (int)orows;
  }
  { const FastDiv fw = fast_div(d->out_w), fh = fast_div(d->out_h); a.mg_ow = fw.magic; a.sh_ow = fw.shift; a.mg_oh = fh.magic; a.sh_oh = fh.shift;
    const FastDiv fc = fast_div(which == ACG_CONV_DGRAD ? (h ? cout8 : (d->out_c + 3) & ~3) : (h ? cin8 : (d->in_c + 3) & ~3)); a.mg_cp = fc.magic; a.sh_cp = fc.shift; }
  if (stats != nullptr) {
    int tpg = 0;
    const int nblk = stats_blocks(pl, *d, which, stats_groups, &tpg);
    ACG_REQUIRE(nblk > 0 && !out_f32, ACG_ERR_UNSUPPORTED, "%s: this shape provides no BatchNorm partials (acg_conv2d_stats_blocks == 0)", who);
    a.stats = stats; a.stats_nblk = nblk; a.stats_tpg = tpg;
  }
  j.which = which; j.pl = pl; j.a = a; j.ws = ws; j.out = out; j.accumulate = accumulate; j.slabs_only = slabs_only;
  return ACG_OK;
}

int launch(const Job& j, hipStream_t st) {
  if (j.pl.direct) return launch_direct(j.which, j.a, st);
  if (j.pl.bf16) {
    if (j.which == ACG_CONV_FWD) return launch_mode16<MODE_FWD>(j.pl, j.a, st);
    if (j.which == ACG_CONV_DGRAD) return launch_mode16<MODE_DGRAD>(j.pl, j.a, st);
    return launch_mode16<MODE_WGRAD>(j.pl, j.a, st);
  }
  if (j.which == ACG_CONV_FWD) return launch_mode<MODE_FWD>(j.pl, j.a, st);
  if (j.which == ACG_CONV_DGRAD) return launch_mode<MODE_DGRAD>(j.pl, j.a, st);
  return launch_mode<MODE_WGRAD>(j.pl, j.a, st);
}

int reduce(const Job& j, hipStream_t st) {
  if (j.pl.splits > 1 && !j.slabs_only && j.pl.bf16 && j.which != ACG_CONV_WGRAD && !j.a.out_f32) {
    ACG_LAUNCH(splitk_reduce_bf16, dim3(reduce_blocks(j.pl.out_numel)), dim3(256), 0, st, (const float*)j.ws, (__bf16*)j.out, j.pl.out_numel, j.pl.splits);
    return acg::check_launch("splitk_reduce_bf16");
  }
  if (j.pl.splits > 1 && !j.slabs_only) {
    ACG_LAUNCH(splitk_reduce, dim3(reduce_blocks(j.pl.out_numel)), dim3(256), 0, st, (const float*)j.ws, j.out, j.pl.out_numel, j.pl.splits,
               j.which == ACG_CONV_WGRAD ? j.accumulate : 0.f);
    return acg::check_launch("splitk_reduce");
  }
  return ACG_OK;
}

// Merged input gradient (kernels above): eligibility and the stride-1 forward problem it runs as.
struct Merged {
  bool ok;
  acg_conv_desc syn;      // dy [B, OH, OW, Cout] * W' [nwh, nww, Cout, 4 Cin], stride 1 -> [B, OH, OW, 4 Cin]
  MergeGeom g;
  size_t w_bytes;         // W' in the workspace, behind the slabs
  long long out_numel;    // of the real result (dx at its pitch)
  int pitch;
};
Merged merged_dgrad(const acg_conv_desc& d, bool bf16) {
  Merged m{};
  static const int on = env_int("ACG_PLAN_MERGED", 1);
  if (!on || g_force_cfg >= 0 || g_force_splits >= 0) return m;
  if (d.stride_h != 2 || d.stride_w != 2 || (d.in_h & 1) || (d.in_w & 1) || d.out_h * 2 != d.in_h || d.out_w * 2 != d.in_w || 4 * d.in_c > 32) return m;
  if (d.dgrad_c > 0 && d.dgrad_c < d.in_c) return m;      // a channel limit: the class-wise kernel honours it (and leaves the other channels alone)
  auto window = [](int k, int pad, int& lo, int& hi) {
    lo = 1 << 20; hi = -(1 << 20);
    for (int ph = 0; ph < 2; ++ph)
      for (int i = 0; i < k; ++i)
        if (((ph + pad - i) & 1) == 0) { const int dh = (ph + pad - i) / 2; lo = std::min(lo, dh); hi = std::max(hi, dh); }
  };
  int lo_h, hi_h, lo_w, hi_w;
  window(d.kh, d.pad_top, lo_h, hi_h); window(d.kw, d.pad_left, lo_w, hi_w);
  const int nwh = hi_h - lo_h + 1, nww = hi_w - lo_w + 1;
  if (lo_h > 0 || lo_w > 0 || -lo_h >= nwh || -lo_w >= nww || nwh * nww > kMaxTaps || nwh * nww >= d.kh * d.kw) return m;
  if (make_plan(d, ACG_CONV_DGRAD, bf16).direct) return m;
  acg_conv_desc& y = m.syn;
  y.batch = d.batch; y.in_h = d.out_h; y.in_w = d.out_w; y.in_c = d.out_c; y.kh = nwh; y.kw = nww; y.out_c = 4 * d.in_c;
  y.stride_h = y.stride_w = 1; y.pad_top = -lo_h; y.pad_left = -lo_w; y.out_h = d.out_h; y.out_w = d.out_w;
  y.in_pitch = bf16 ? 0 : d.out_pitch; y.out_pitch = 0;
  if (validate(&y, "merged input gradient") != ACG_OK || make_plan(y, ACG_CONV_FWD, bf16).direct) return m;     // (a few-MFLOP layer: the class-wise / direct kernels)
  m.g = MergeGeom{d.kh, d.kw, d.in_c, d.out_c, nwh, nww, lo_h, lo_w, d.pad_top, d.pad_left};
  const long long welems = (long long)nwh * nww * y.out_c * (bf16 ? ((d.out_c + 7) & ~7) : d.out_c);
  m.w_bytes = (size_t)welems * (bf16 ? 2 : 4);
  m.pitch = bf16 ? ((d.in_c + 7) & ~7) : (d.in_pitch > 0 ? d.in_pitch : d.in_c);
  m.out_numel = (long long)d.batch * d.in_h * d.in_w * m.pitch;
  m.ok = true;
  return m;
}
size_t merged_slab_bytes(const Merged& m, int splits) { return splits > 1 ? (((size_t)splits * (size_t)m.out_numel * sizeof(float) + 255) & ~(size_t)255) : 0; }

int run_merged(const Merged& m, const float* dy, const float* w, float* out, const acg_conv_desc* d, int dtype, void* ws, size_t ws_bytes,
               acg_stream_t stream, const char* who, bool slabs_only, int slab_layout) {
  const bool h = acg::dt_valid(dtype) && acg::dt_first(dtype) == ACG_BF16;
  if (int rc = validate(d, who)) return rc;
  ACG_REQUIRE(dy && w && (out || slabs_only), ACG_ERR_INVALID_ARG, "%s: null tensor pointer", who);
  ACG_REQUIRE(slab_layout == ACG_SLABS_ROWS, ACG_ERR_UNSUPPORTED, "%s: this input gradient leaves its slabs in the row layout only", who);
  if (h) ACG_REQUIRE((d->in_pitch == 0 || d->in_pitch == ((d->in_c + 7) & ~7)) && (d->out_pitch == 0 || d->out_pitch == ((d->out_c + 7) & ~7)), ACG_ERR_INVALID_ARG,
                     "%s: bf16 tensors are stored at the channel pitch round8(C)", who);
  const int splits = make_plan(m.syn, ACG_CONV_FWD, h).splits;
  ACG_REQUIRE(!slabs_only || splits > 1, ACG_ERR_INVALID_ARG, "%s: this shape is not split (acg_conv2d_splits == 1): call the plain entry", who);
  const size_t slab_bytes = merged_slab_bytes(m, splits), need = slab_bytes + m.w_bytes;
  ACG_REQUIRE(ws != nullptr && ws_bytes >= need, ACG_ERR_WORKSPACE, "%s: workspace %zu bytes < required %zu", who, ws_bytes, need);
  float* const wm = reinterpret_cast<float*>(static_cast<char*>(ws) + slab_bytes);
  hipStream_t st = acg::to_stream(stream);
  const int total = (int)(m.w_bytes / (h ? 2 : 4));
  if (h) ACG_LAUNCH(merge_dgrad_weights_bf16, dim3((unsigned)acg::ceil_div(total, 256)), dim3(256), 0, st, (const __bf16*)w, (__bf16*)wm, m.g, total);
  else ACG_LAUNCH(merge_dgrad_weights_f32, dim3((unsigned)acg::ceil_div(total, 256)), dim3(256), 0, st, w, wm, m.g, total);
  if (int rc = acg::check_launch("merge_dgrad_weights")) return rc;
  Job j;
  // the forward problem as prepare() plans it; its result tensor is then redirected to dx (pixel-shuffle epilogue, ConvArgs::shuf_c)
  if (int rc = prepare(j, ACG_CONV_FWD, dy, wm, out, 0.f, &m.syn, dtype, ws, (size_t)1 << 40, who, slabs_only)) return rc;
  j.pl.out_numel = m.out_numel; j.a.out_numel = m.out_numel;
  j.a.shuf_c = d->in_c; j.a.shuf_w = d->in_w; j.a.shuf_pitch = m.pitch;
  if (int rc = launch(j, st)) return rc;
  return reduce(j, st);
}

int run(int which, const float* gsrc, const float* dense, float* out, float accumulate, const acg_conv_desc* d, int dtype,
        void* ws, size_t ws_bytes, acg_stream_t stream, const char* who, bool slabs_only = false, float* stats = nullptr,
        int stats_groups = 0, int slab_layout = ACG_SLABS_ROWS) {
  if (which == ACG_CONV_DGRAD && stats == nullptr && d != nullptr && validate(d, who) == ACG_OK) {
    const Merged m = merged_dgrad(*d, acg::dt_valid(dtype) && acg::dt_first(dtype) == ACG_BF16);
    if (m.ok) return run_merged(m, gsrc, dense, out, d, dtype, ws, ws_bytes, stream, who, slabs_only, slab_layout);
  }
  Job j;
  if (int rc = prepare(j, which, gsrc, dense, out, accumulate, d, dtype, ws, ws_bytes, who, slabs_only, stats, stats_groups, slab_layout)) return rc;
  hipStream_t st = acg::to_stream(stream);
  if (int rc = launch(j, st)) return rc;
  return reduce(j, st);
}

// Input gradient A (whichA = DGRAD, or FWD for a transposed layer) and weight gradient B of one layer: ONE launch when
// the pair kernel covers the shapes (conv_f32_pair.hip), two otherwise; same results either way.
int run_pair(int whichA, const float* gsrcA, const float* denseA, float* outA, const float* gsrcB, const float* denseB, float* outB,
             float accumulateB, const acg_conv_desc* d, int dtype, void* wsA, size_t wsbA, void* wsB, size_t wsbB, int slab_flags,
             acg_stream_t stream, const char* who) {
  if (whichA == ACG_CONV_DGRAD && d != nullptr && validate(d, who) == ACG_OK && merged_dgrad(*d, acg::dt_valid(dtype) && acg::dt_first(dtype) == ACG_BF16).ok) {
    // the merged input gradient (run_merged) is a launch of its own: same results as the separate entries, which take it too
    if (int rc = run(ACG_CONV_DGRAD, gsrcA, denseA, outA, 0.f, d, dtype, wsA, wsbA, stream, who, (slab_flags & 2) != 0, nullptr, 0,
                     (slab_flags & 4) ? ACG_SLABS_QUADS : ACG_SLABS_ROWS)) return rc;
    return run(ACG_CONV_WGRAD, gsrcB, denseB, outB, accumulateB, d, dtype, wsB, wsbB, stream, who, (slab_flags & 1) != 0);
  }
  Job ja, jb;
  const bool slabs_only_B = (slab_flags & 1) != 0, slabs_only_A = (slab_flags & 2) != 0;
  ACG_REQUIRE(!(slab_flags & 4) || slabs_only_A, ACG_ERR_INVALID_ARG, "%s: flag 4 (quad slab layout) without flag 2", who);
  // the 256 x 128 LDS-DMA kernel has no paired form: a pair keeps A on the 128 x 128 tile (ACG_PAIR_WIDE = 1, tuning builds: A on
  // the wide kernel, the weight gradient as a launch of its own)
  static const int pair_wide = env_int("ACG_PAIR_WIDE", 0);
  if (int rc = prepare(ja, whichA, gsrcA, denseA, outA, 0.f, d, dtype, wsA, wsbA, who, slabs_only_A, nullptr, 0, (slab_flags & 4) ? ACG_SLABS_QUADS : ACG_SLABS_ROWS, pair_wide != 0)) return rc;
  if (int rc = prepare(jb, ACG_CONV_WGRAD, gsrcB, denseB, outB, accumulateB, d, dtype, wsB, wsbB, who, slabs_only_B)) return rc;
  hipStream_t st = acg::to_stream(stream);
  static const int enabled = env_int("ACG_CONV_PAIR", 1);       // 0: always two launches (A/B comparison)
  if (ja.pl.direct || jb.pl.direct) {
    if (enabled && ja.pl.direct && jb.pl.direct && whichA == ACG_CONV_DGRAD) {
      if (int rc = launch_direct_pair(ja.a, jb.a, st)) return rc;
    } else {
      if (int rc = launch(ja, st)) return rc;
      if (int rc = launch(jb, st)) return rc;
    }
  } else if (enabled && g_force_cfg < 0 && ja.pl.bf16 && jb.pl.bf16 && ja.pl.cfg != 4 && (whichA == ACG_CONV_FWD || whichA == ACG_CONV_DGRAD) &&
      (long long)ja.pl.tiles * ja.pl.splits + (long long)jb.pl.tiles * jb.pl.splits < (1ll << 30)) {
    if (int rc = launch_pair16(whichA == ACG_CONV_FWD ? MODE_FWD : MODE_DGRAD, ja.pl, ja.a, jb.pl, jb.a, st)) return rc;
  } else if (enabled && g_force_cfg < 0 && pair_supported(whichA, ja.pl, jb.pl)) {
    if (int rc = launch_pair(whichA, ja.pl, ja.a, jb.pl, jb.a, st)) return rc;
  } else {
    if (int rc = launch(ja, st)) return rc;
    if (int rc = launch(jb, st)) return rc;
  }
  if (int rc = reduce(ja, st)) return rc;
  return reduce(jb, st);
}

}  // namespace

extern "C" {

int32_t acg_conv_desc_init(acg_conv_desc* d, int32_t batch, int32_t in_h, int32_t in_w, int32_t in_c, int32_t kh,
                           int32_t kw, int32_t out_c, int32_t stride, int32_t same) {
  ACG_REQUIRE(d && batch > 0 && in_h > 0 && in_w > 0 && in_c > 0 && kh > 0 && kw > 0 && out_c > 0 && stride > 0,
              ACG_ERR_INVALID_ARG, "conv_desc_init: non-positive dimension");
  d->batch = batch; d->in_h = in_h; d->in_w = in_w; d->in_c = in_c; d->out_c = out_c;
  d->kh = kh; d->kw = kw; d->stride_h = d->stride_w = stride; d->in_pitch = 0; d->out_pitch = 0; d->dgrad_c = 0; d->adj_dgrad_c = 0;
  if (same) {  // TF 'SAME' (SURVEY A.1): out = ceil(in/s), pad_before = total // 2
    d->out_h = (in_h + stride - 1) / stride; d->out_w = (in_w + stride - 1) / stride;
    const int th = std::max((d->out_h - 1) * stride + kh - in_h, 0), tw = std::max((d->out_w - 1) * stride + kw - in_w, 0);
    d->pad_top = th / 2; d->pad_left = tw / 2;
  } else {
    ACG_REQUIRE(in_h >= kh && in_w >= kw, ACG_ERR_INVALID_ARG, "conv_desc_init: VALID kernel larger than input");
    d->out_h = (in_h - kh) / stride + 1; d->out_w = (in_w - kw) / stride + 1;
    d->pad_top = d->pad_left = 0;
  }
  return ACG_OK;
}

#ifdef ACG_TUNING
int32_t acg_debug_conv_plan(int32_t cfg, int32_t splits) {
  g_force_cfg = cfg; g_force_splits = splits;
  return ACG_OK;
}
#endif

size_t acg_conv2d_workspace_bytes(const acg_conv_desc* d, int32_t which, int32_t dtype) {
  if (!d || validate(d, "conv2d_workspace_bytes") != ACG_OK || which < 0 || which > 2) return 0;
  const bool h = acg::dt_valid(dtype) && acg::dt_first(dtype) == ACG_BF16;
  if (which == ACG_CONV_DGRAD) {
    const Merged m = merged_dgrad(*d, h);
    if (m.ok) return merged_slab_bytes(m, make_plan(m.syn, ACG_CONV_FWD, h).splits) + m.w_bytes;
  }
  const Plan pl = make_plan(*d, which, h);
  return pl.splits > 1 ? (size_t)pl.splits * (size_t)pl.out_numel * sizeof(float) : 0;
}

int32_t acg_conv2d_splits(const acg_conv_desc* d, int32_t which, int32_t dtype) {
  if (!d || validate(d, "conv2d_splits") != ACG_OK || which < 0 || which > 2) return 0;
  const bool h = acg::dt_valid(dtype) && acg::dt_first(dtype) == ACG_BF16;
  if (which == ACG_CONV_DGRAD) {
    const Merged m = merged_dgrad(*d, h);
    if (m.ok) return make_plan(m.syn, ACG_CONV_FWD, h).splits;
  }
  return make_plan(*d, which, h).splits;
}

int32_t acg_conv2d_slab_layouts(const acg_conv_desc* d, int32_t which, int32_t dtype) {
  if (!d || validate(d, "conv2d_slab_layouts") != ACG_OK || which < 0 || which > 1) return 0;
  const bool h = acg::dt_valid(dtype) && acg::dt_first(dtype) == ACG_BF16;
  if (acg_conv2d_splits(d, which, dtype) < 2) return 0;
  if (which == ACG_CONV_DGRAD && merged_dgrad(*d, h).ok) return 1 << ACG_SLABS_ROWS;      // the merged input gradient scatters 2 x 2 pixels per row
  return (1 << ACG_SLABS_ROWS) | (1 << ACG_SLABS_QUADS);
}

int32_t acg_conv2d_stats_blocks(const acg_conv_desc* d, int32_t which, int32_t dtype, int32_t groups) {
  if (!d || validate(d, "conv2d_stats_blocks") != ACG_OK || which < 0 || which > 2 || (dtype != ACG_F32 && dtype != ACG_BF16)) return 0;
  return stats_blocks(make_plan(*d, which, dtype == ACG_BF16), *d, which, groups, nullptr);
}
int32_t acg_conv2d_tile(const acg_conv_desc* d, int32_t which, int32_t dtype, int32_t* tile_rows, int32_t* tile_cols) {
  if (!d || validate(d, "conv2d_tile") != ACG_OK || which < 0 || which > 2 || (dtype != ACG_F32 && dtype != ACG_BF16)) return 0;
  const Plan pl = make_plan(*d, which, dtype == ACG_BF16);
  if (tile_rows) *tile_rows = pl.bm;
  if (tile_cols) *tile_cols = pl.bn;
  return (int32_t)std::min<long long>(pl.tiles, 0x7fffffff);
}
int32_t acg_conv2d_stats_layout(const acg_conv_desc* d, int32_t which, int32_t dtype, int32_t groups, int32_t* block_rows, int32_t* run_rows) {
  if (!d || validate(d, "conv2d_stats_layout") != ACG_OK || which < 0 || which > 2 || (dtype != ACG_F32 && dtype != ACG_BF16)) return 0;
  const Plan pl = make_plan(*d, which, dtype == ACG_BF16);
  int run = 0;
  const int nblk = stats_blocks(pl, *d, which, groups, nullptr, &run);
  if (nblk > 0) {
    if (block_rows) *block_rows = pl.bm;
    if (run_rows) *run_rows = run;
  }
  return nblk;
}
int32_t acg_conv2d_fwd_stats(const void* x, const void* w, void* y, const acg_conv_desc* d, int32_t dtype, void* ws, size_t wsb,
                             float* partials, int32_t groups, acg_stream_t s) {
  ACG_REQUIRE(partials != nullptr, ACG_ERR_INVALID_ARG, "conv2d_fwd_stats: null partials");
  return run(ACG_CONV_FWD, (const float*)x, (const float*)w, (float*)y, 0.f, d, dtype, ws, wsb, s, "conv2d_fwd_stats", false, partials, groups);
}
int32_t acg_deconv2d_fwd_stats(const void* x, const void* w, void* y, const acg_conv_desc* adj, int32_t dtype, void* ws, size_t wsb,
                               float* partials, int32_t groups, acg_stream_t s) {
  ACG_REQUIRE(partials != nullptr, ACG_ERR_INVALID_ARG, "deconv2d_fwd_stats: null partials");
  return run(ACG_CONV_DGRAD, (const float*)x, (const float*)w, (float*)y, 0.f, adj, dtype, ws, wsb, s, "deconv2d_fwd_stats", false, partials, groups);
}

// Bias + activation in the epilogue of a transposed layer's forward (models.py:20-21: tanh(conv2d_transpose(x) + b), the plain
// generator's frame): available where the plan is the unsplit 128x32 tile with 16-byte gathers.
static bool deconv_bias_act_plan(const acg_conv_desc* adj, int dtype, Plan* out) {
  if (!adj || validate(adj, "deconv2d_fwd_bias_act") != ACG_OK || (dtype != ACG_F32 && dtype != ACG_BF16)) return false;
  const Plan pl = make_plan(*adj, ACG_CONV_DGRAD, dtype == ACG_BF16);
  if (out) *out = pl;
  return !pl.direct && pl.splits == 1 && pl.bm == 128 && pl.bn == 32 && !pl.ragged && g_force_cfg < 0;
}
int32_t acg_deconv2d_fwd_bias_act_ok(const acg_conv_desc* adj, int32_t dtype) { return deconv_bias_act_plan(adj, dtype, nullptr) ? 1 : 0; }
int32_t acg_deconv2d_fwd_bias_act(const void* x, const void* w, const float* bias, float* y, const acg_conv_desc* adj, int32_t act,
                                  float leak, int32_t dtype, acg_stream_t s) {
  ACG_REQUIRE(bias && y, ACG_ERR_INVALID_ARG, "deconv2d_fwd_bias_act: null pointer");
  ACG_REQUIRE(act >= ACG_ACT_NONE && act <= ACG_ACT_TANH, ACG_ERR_INVALID_ARG, "deconv2d_fwd_bias_act: activation %d", act);
  ACG_REQUIRE(deconv_bias_act_plan(adj, dtype, nullptr), ACG_ERR_UNSUPPORTED, "deconv2d_fwd_bias_act: this shape does not take the fused epilogue (acg_deconv2d_fwd_bias_act_ok == 0)");
  acg_conv_desc d = *adj;
  if (dtype == ACG_BF16) d.in_pitch = 0;      // (checked against round8 by prepare; the OUTPUT pitch is set below)
  Job j;
  if (int rc = prepare(j, ACG_CONV_DGRAD, (const float*)x, (const float*)w, y, 0.f, &d, dtype, nullptr, 0, "deconv2d_fwd_bias_act", false)) return rc;
  j.a.bias = bias; j.a.act = act; j.a.leak = leak;
  j.a.Cx = adj->in_pitch > 0 ? adj->in_pitch : adj->in_c;      // y: dense float32 rows (in_pitch of the adjoint = y's pitch)
  hipStream_t st = acg::to_stream(s);
  return dtype == ACG_BF16 ? launch_deconv_fwd_epi16(j.pl, j.a, st) : launch_deconv_fwd_epi(j.pl, j.a, st);
}

int32_t acg_conv2d_wgrad_slabs(const void* x, const void* dy, const acg_conv_desc* d, int32_t dtype, void* ws, size_t wsb,
                               acg_stream_t s) {
  return run(ACG_CONV_WGRAD, (const float*)x, (const float*)dy, nullptr, 0.f, d, dtype, ws, wsb, s, "conv2d_wgrad_slabs", true);
}
int32_t acg_deconv2d_wgrad_slabs(const void* x, const void* dy, const acg_conv_desc* adj, int32_t dtype, void* ws, size_t wsb,
                                 acg_stream_t s) {
  return run(ACG_CONV_WGRAD, (const float*)dy, (const float*)x, nullptr, 0.f, adj, dtype, ws, wsb, s, "deconv2d_wgrad_slabs", true);
}

// split-K hand-off to the consuming BatchNorm (acg_bn_act_fwd_slabs / acg_bn_act_bwd_slabs): the contraction only
int32_t acg_conv2d_fwd_slabs(const void* x, const void* w, const acg_conv_desc* d, int32_t dtype, int32_t layout, void* ws, size_t wsb, acg_stream_t s) {
  return run(ACG_CONV_FWD, (const float*)x, (const float*)w, nullptr, 0.f, d, dtype, ws, wsb, s, "conv2d_fwd_slabs", true, nullptr, 0, layout);
}
int32_t acg_conv2d_dgrad_slabs(const void* dy, const void* w, const acg_conv_desc* d, int32_t dtype, int32_t layout, void* ws, size_t wsb, acg_stream_t s) {
  return run(ACG_CONV_DGRAD, (const float*)dy, (const float*)w, nullptr, 0.f, d, dtype, ws, wsb, s, "conv2d_dgrad_slabs", true, nullptr, 0, layout);
}
int32_t acg_deconv2d_fwd_slabs(const void* x, const void* w, const acg_conv_desc* adj, int32_t dtype, int32_t layout, void* ws, size_t wsb, acg_stream_t s) {
  return run(ACG_CONV_DGRAD, (const float*)x, (const float*)w, nullptr, 0.f, adj, dtype, ws, wsb, s, "deconv2d_fwd_slabs", true, nullptr, 0, layout);
}
int32_t acg_deconv2d_dgrad_slabs(const void* dy, const void* w, const acg_conv_desc* adj, int32_t dtype, int32_t layout, void* ws, size_t wsb, acg_stream_t s) {
  return run(ACG_CONV_FWD, (const float*)dy, (const float*)w, nullptr, 0.f, adj, dtype, ws, wsb, s, "deconv2d_dgrad_slabs", true, nullptr, 0, layout);
}

int32_t acg_conv2d_bwd_pair(const void* dy, const void* w, const void* x, void* dx, float* dw, float dw_accumulate,
                            const acg_conv_desc* d, int32_t dtype, void* ws_dgrad, size_t wsb_dgrad, void* ws_wgrad, size_t wsb_wgrad,
                            int32_t wgrad_slabs_only, acg_stream_t s) {
  return run_pair(ACG_CONV_DGRAD, (const float*)dy, (const float*)w, (float*)dx, (const float*)x, (const float*)dy, dw, dw_accumulate,
                  d, dtype, ws_dgrad, wsb_dgrad, ws_wgrad, wsb_wgrad, wgrad_slabs_only, s, "conv2d_bwd_pair");
}
int32_t acg_deconv2d_bwd_pair(const void* dy, const void* w, const void* x, void* dx, float* dw, float dw_accumulate,
                              const acg_conv_desc* adj, int32_t dtype, void* ws_dgrad, size_t wsb_dgrad, void* ws_wgrad, size_t wsb_wgrad,
                              int32_t wgrad_slabs_only, acg_stream_t s) {
  // transposed layer on the adjoint descriptor: its input gradient is the adjoint's FWD, its weight gradient the
  // adjoint's WGRAD with the roles of x and dy exchanged (acg_deconv2d_dgrad / acg_deconv2d_wgrad)
  return run_pair(ACG_CONV_FWD, (const float*)dy, (const float*)w, (float*)dx, (const float*)dy, (const float*)x, dw, dw_accumulate,
                  adj, dtype, ws_dgrad, wsb_dgrad, ws_wgrad, wsb_wgrad, wgrad_slabs_only, s, "deconv2d_bwd_pair");
}

int32_t acg_splitk_reduce_many(const acg_reduce_list* list, int32_t count, acg_stream_t stream) {
  ACG_REQUIRE(list && count >= 1 && count <= ACG_REDUCE_MAX, ACG_ERR_INVALID_ARG, "splitk_reduce_many: 1..%d entries", ACG_REDUCE_MAX);
  ReduceList l{};
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    ACG_REQUIRE(list->slabs[i] && list->out[i] && list->numel[i] > 0 && list->splits[i] >= 1, ACG_ERR_INVALID_ARG,
                "splitk_reduce_many: bad entry %d", i);
    for (int j = 0; j < i; ++j)
      ACG_REQUIRE(list->out[j] != list->out[i], ACG_ERR_INVALID_ARG, "splitk_reduce_many: entries %d and %d share an output", j, i);
    l.slabs[i] = (const float*)list->slabs[i]; l.out[i] = (float*)list->out[i]; l.numel[i] = list->numel[i];
    l.splits[i] = list->splits[i]; l.accumulate[i] = list->accumulate[i];
    l.first_block[i] = blocks;
    blocks += reduce_blocks(list->numel[i]);
  }
  l.first_block[count] = blocks;
  l.step_inc = list->step_inc;
  ACG_LAUNCH(splitk_reduce_many, dim3(blocks), dim3(256), 0, acg::to_stream(stream), l, (int)count);
  return acg::check_launch("splitk_reduce_many");
}

int32_t acg_weights_prepare_bf16(const acg_prep_list* list, int32_t count, acg_stream_t stream) {
  ACG_REQUIRE(list && count >= 1 && count <= ACG_PREP_MAX, ACG_ERR_INVALID_ARG, "weights_prepare_bf16: 1..%d entries", ACG_PREP_MAX);
  PrepList l{};
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    ACG_REQUIRE(list->src[i] && list->rm[i] && list->tr[i] && list->taps[i] > 0 && list->a[i] > 0 && list->b[i] > 0, ACG_ERR_INVALID_ARG,
                "weights_prepare_bf16: bad entry %d", i);
    l.src[i] = (const float*)list->src[i]; l.rm[i] = (__bf16*)list->rm[i]; l.tr[i] = (__bf16*)list->tr[i];
    l.taps[i] = list->taps[i]; l.A[i] = list->a[i]; l.B[i] = list->b[i];
    const long long nocts = (long long)l.taps[i] * l.A[i] * (((l.B[i] + 7) & ~7) / 8);
    const long long ntiles = (long long)l.taps[i] * ((((l.A[i] + 7) & ~7) + 31) / 32) * ((l.B[i] + 31) / 32);
    l.first_block[i] = blocks;
    l.n_rm[i] = (int)std::max<long long>(1, std::min<long long>(acg::ceil_div(nocts, 256 * 2), 512));
    blocks += l.n_rm[i] + (int)std::max<long long>(1, std::min<long long>(ntiles, 1024));
  }
  l.first_block[count] = blocks;
  ACG_LAUNCH(weights_prepare_bf16, dim3(blocks), dim3(256), 0, acg::to_stream(stream), l, (int)count);
  return acg::check_launch("weights_prepare_bf16");
}

int32_t acg_conv2d_fwd(const void* x, const void* w, void* y, const acg_conv_desc* d, int32_t dtype, void* ws,
                       size_t wsb, acg_stream_t s) {
  return run(ACG_CONV_FWD, (const float*)x, (const float*)w, (float*)y, 0.f, d, dtype, ws, wsb, s, "conv2d_fwd");
}
int32_t acg_conv2d_dgrad(const void* dy, const void* w, void* dx, const acg_conv_desc* d, int32_t dtype, void* ws,
                         size_t wsb, acg_stream_t s) {
  return run(ACG_CONV_DGRAD, (const float*)dy, (const float*)w, (float*)dx, 0.f, d, dtype, ws, wsb, s, "conv2d_dgrad");
}
int32_t acg_conv2d_wgrad(const void* x, const void* dy, float* dw, float accumulate, const acg_conv_desc* d,
                         int32_t dtype, void* ws, size_t wsb, acg_stream_t s) {
  return run(ACG_CONV_WGRAD, (const float*)x, (const float*)dy, dw, accumulate, d, dtype, ws, wsb, s, "conv2d_wgrad");
}
int32_t acg_deconv2d_fwd(const void* x, const void* w, void* y, const acg_conv_desc* adj, int32_t dtype, void* ws,
                         size_t wsb, acg_stream_t s) {
  return run(ACG_CONV_DGRAD, (const float*)x, (const float*)w, (float*)y, 0.f, adj, dtype, ws, wsb, s, "deconv2d_fwd");
}
int32_t acg_deconv2d_dgrad(const void* dy, const void* w, void* dx, const acg_conv_desc* adj, int32_t dtype, void* ws,
                           size_t wsb, acg_stream_t s) {
  return run(ACG_CONV_FWD, (const float*)dy, (const float*)w, (float*)dx, 0.f, adj, dtype, ws, wsb, s, "deconv2d_dgrad");
}
int32_t acg_deconv2d_wgrad(const void* x, const void* dy, float* dw, float accumulate, const acg_conv_desc* adj,
                           int32_t dtype, void* ws, size_t wsb, acg_stream_t s) {
  // roles exchanged: the deconv's output gradient is the adjoint conv's input
  return run(ACG_CONV_WGRAD, (const float*)dy, (const float*)x, dw, accumulate, adj, dtype, ws, wsb, s, "deconv2d_wgrad");
}

}  // extern "C"
