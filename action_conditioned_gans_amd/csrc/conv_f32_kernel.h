// Kernel template of the fp32 MFMA implicit-GEMM convolution; see conv_f32.hip for the design notes.
// Included by conv_f32_{fwd,dgrad,wgrad}.hip, each of which instantiates one MODE (parallel compilation).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace acgconv {


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned 16-byte global load
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };
constexpr int BK = 32;  // k per K-step = 8 quads
constexpr int kMaxTaps = 64;

struct ConvArgs {
  const float* gsrc;   // gathered tensor: x (FWD, WGRAD) or dy (DGRAD)
  const float* dense;  // w (FWD, DGRAD) or dy (WGRAD)
  float* out;          // final tensor (splits == 1) or `splits` slabs of out_numel floats
  long long out_numel;
  unsigned g_bytes, d_bytes;   // sizes of gsrc / dense for the buffer descriptors (hardware range check)
  float accumulate;    // WGRAD with splits == 1: out = accumulate * out + value
  int batch, H, W, C, OH, OW, K, KH, KW, sh, sw, pt, pl;
  int Cx;              // channel pitch of x / dx in memory (>= C)
  int Ky;              // channel pitch of y / dy in memory (>= K)
  int splits;
};

struct alignas(16) RowInfo {
  int base;            // element offset of the row's (tap 0, channel 0) source element (may be virtual)
  unsigned mask_lo, mask_hi;  // bit t set <=> filter tap t reads inside the tensor
  int out_off;         // DGRAD: element offset of the output pixel; unused otherwise
};

__device__ __forceinline__ unsigned long long tap_mask(int lo_a, int hi_a, int lo_b, int hi_b, int nb) {
  // bits (a * nb + b) for a in [lo_a, hi_a), b in [lo_b, hi_b)
  if (hi_a <= lo_a || hi_b <= lo_b) return 0ull;
  const unsigned long long row = ((1ull << (hi_b - lo_b)) - 1ull) << lo_b;
  unsigned long long m = 0ull;
  for (int a = lo_a; a < hi_a; ++a) m |= row << (a * nb);
  return m;
}

__device__ __forceinline__ bool tap_ok(const RowInfo& ri, int t) {
  const unsigned long long mk = ((unsigned long long)ri.mask_hi << 32) | ri.mask_lo;
  return (mk >> t) & 1ull;
}

// Branch-free guarded loads through buffer descriptors.  A divergent `if (ok) load` makes hipcc wait for every
// load inside its branch (vmcnt(0) per load: the loads of a K-step serialise), and selecting the loaded VALUE
// drags the wait up to the load as well.  Instead each operand tensor is addressed through a raw buffer
// resource (base + 32-bit byte offset, hardware range check): a masked lane gets an offset beyond the buffer
// and the load returns zeros without touching memory.  One v_cndmask on a 32-bit offset per load, no 64-bit
// pointer arithmetic, and every load of a stage issues back to back; the first wait is two K-steps later.
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOob = 0xFFFFFF00u;   // >= any num_records (tensors are < 2^30 elements)

__device__ __forceinline__ f4 guarded_quad(__amdgpu_buffer_rsrc_t rs, int elem_off, bool ok) {
  const unsigned off = ok ? (unsigned)elem_off * 4u : kOob;
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}
// Ragged quad (channel count not a multiple of 4): only the first nvalid (1..3) elements exist.
__device__ __forceinline__ f4 guarded_ragged(__amdgpu_buffer_rsrc_t rs, int elem_off, bool ok, int nvalid) {
  f4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned off = (ok && e < nvalid) ? (unsigned)(elem_off + e) * 4u : kOob;
    r[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
  }
  return r;
}
__device__ __forceinline__ float guarded_scalar(__amdgpu_buffer_rsrc_t rs, int elem_off, bool ok) {
  const unsigned off = ok ? (unsigned)elem_off * 4u : kOob;
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
}

// RAGGED: gathered channel count not a multiple of 4; NVEC: dense operand rows are float4-loadable (N % 4 == 0).
// Both are compile-time so that the slow variants never share registers (and waits) with the fast one.
// BF16: operands are rounded to bf16 (RNE) as they are staged into LDS and contracted by v_mfma_f32_32x32x16_bf16
// (fp32 accumulate); tensors stay fp32 in memory.  LDS then holds 8-k "octs": quad kq lands in half (kq&1) of
// oct kq>>1, and lane half h feeds oct 2t+h of both tiles to MFMA t (2 MFMAs per 32-deep K-step instead of 16).
template <int MODE, int BM, int BN, int WM, int WN, bool RAGGED, bool NVEC, bool BF16>
__global__ __launch_bounds__(256) void conv_mfma_f32(const ConvArgs p) {
  static_assert(WM * WN == 4, "4 waves per block");
  constexpr int TA = BM / (32 * WM), TB = BN / (32 * WN);
  constexpr int QA = BM / 32, QB = BN / 32;  // quads per thread per K-step
  constexpr int NROW = (MODE == MODE_WGRAD) ? BK : BM;

  constexpr int KSLOTS = BF16 ? 4 : 8;   // 16-byte k-slots per tile column: 8 quads (fp32) or 4 octs (bf16)
  __shared__ f4 As[KSLOTS * BM];
  __shared__ f4 Bs[KSLOTS * BN];
  __shared__ RowInfo rows[NROW];
  __shared__ int tapA[kMaxTaps];
  __shared__ int tapB[kMaxTaps];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gsrc), 0, p.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dense), 0, p.d_bytes, 0x00020000);

  // ---- problem geometry (wave-uniform) -------------------------------------------------------
  // Cs: channels of the gathered tensor; Cp: Cs padded to a multiple of 4 (quad granularity per tap).
  int M, N, Kdim, Cs, Cp, ntaps;
  int ph = 0, pw = 0, i0 = 0, j0 = 0, nti = 1, ntj = 1, dp0 = 0, dq0 = 0, Hc = 0, Wc = 0;
  if constexpr (MODE == MODE_FWD) {
    Cs = p.C; Cp = (Cs + 3) & ~3; ntaps = p.KH * p.KW;
    M = p.batch * p.OH * p.OW; N = p.K; Kdim = ntaps * Cp;
  } else if constexpr (MODE == MODE_DGRAD) {
    const int cls = blockIdx.y;
    ph = cls / p.sw; pw = cls - ph * p.sw;
    Hc = ph < p.H ? (p.H - ph + p.sh - 1) / p.sh : 0;
    Wc = pw < p.W ? (p.W - pw + p.sw - 1) / p.sw : 0;
    Cs = p.K; Cp = (Cs + 3) & ~3;
    M = p.batch * Hc * Wc; N = p.C;
    i0 = (ph + p.pt) % p.sh; j0 = (pw + p.pl) % p.sw;
    nti = i0 < p.KH ? (p.KH - i0 + p.sh - 1) / p.sh : 0;
    ntj = j0 < p.KW ? (p.KW - j0 + p.sw - 1) / p.sw : 0;
    dp0 = (ph + p.pt - i0) / p.sh; dq0 = (pw + p.pl - j0) / p.sw;
    ntaps = nti * ntj; Kdim = ntaps * Cp;
  } else {
    Cs = p.C; Cp = (Cs + 3) & ~3; ntaps = p.KH * p.KW;
    M = ntaps * Cp; N = p.K; Kdim = p.batch * p.OH * p.OW;
  }
  const int tiles_n = (N + BN - 1) / BN;
  // XCD-aware order: consecutive workgroup ids are dealt round-robin over the 8 XCDs (each with its own L2), so
  // give every XCD a CONTIGUOUS run of tiles - neighbours in the run share A rows / filter columns in that L2.
  // Bijective for any grid size; placement affects speed only.
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  if (m0 >= M) return;  // block-uniform: DGRAD classes smaller than class 0

  const int nk = (Kdim + BK - 1) / BK;
  const int per = (nk + p.splits - 1) / p.splits;
  const int ks_begin = blockIdx.z * per;
  const int ks_end = min(nk, ks_begin + per);

  // ---- tap tables ------------------------------------------------------------------------------
  if (tid < kMaxTaps && tid < ntaps) {
    const int t = tid;
    if constexpr (MODE == MODE_DGRAD) {
      const int ti = t / ntj, tj = t - ti * ntj;
      tapA[t] = -(ti * p.OW + tj) * p.Ky;
      tapB[t] = ((i0 + p.sh * ti) * p.KW + (j0 + p.sw * tj)) * p.C * p.K;
    } else {
      const int i = t / p.KW, j = t - i * p.KW;
      tapA[t] = (i * p.W + j) * p.Cx;
      tapB[t] = 0;
    }
  }

  // ---- per-row gather info -------------------------------------------------------------------
  auto fill_row_fwd = [&](int r /* global row (b,p,q) */, int limit) -> RowInfo {
    RowInfo ri; ri.base = 0; ri.mask_lo = 0; ri.mask_hi = 0; ri.out_off = 0;
    if (r < limit) {
      const int q = r % p.OW; const int t2 = r / p.OW; const int pp = t2 % p.OH; const int b = t2 / p.OH;
      const int y0 = pp * p.sh - p.pt, x0 = q * p.sw - p.pl;
      ri.base = ((b * p.H + y0) * p.W + x0) * p.Cx;
      const unsigned long long m = tap_mask(max(0, -y0), min(p.KH, p.H - y0), max(0, -x0), min(p.KW, p.W - x0), p.KW);
      ri.mask_lo = (unsigned)m; ri.mask_hi = (unsigned)(m >> 32);
    }
    return ri;
  };
  if constexpr (MODE == MODE_FWD) {
    for (int r = tid; r < BM; r += 256) rows[r] = fill_row_fwd(m0 + r, M);
  } else if constexpr (MODE == MODE_DGRAD) {
    for (int r = tid; r < BM; r += 256) {
      RowInfo ri; ri.base = 0; ri.mask_lo = 0; ri.mask_hi = 0; ri.out_off = 0;
      const int m = m0 + r;
      if (m < M) {
        const int w2 = m % Wc; const int t2 = m / Wc; const int h2 = t2 % Hc; const int b = t2 / Hc;
        const int y0 = h2 + dp0, x0 = w2 + dq0;  // dY coordinates of tap (0,0)
        ri.base = ((b * p.OH + y0) * p.OW + x0) * p.Ky;
        ri.out_off = ((b * p.H + h2 * p.sh + ph) * p.W + (w2 * p.sw + pw)) * p.Cx;
        // tap (ti,tj) reads dY[y0 - ti][x0 - tj]
        const unsigned long long mk = tap_mask(max(0, y0 - p.OH + 1), min(nti, y0 + 1), max(0, x0 - p.OW + 1), min(ntj, x0 + 1), ntj);
        ri.mask_lo = (unsigned)mk; ri.mask_hi = (unsigned)(mk >> 32);
      }
      rows[r] = ri;
    }
  }
  __syncthreads();

  // ---- loaders: two register stages ---------------------------------------------------------------
  // k-fast operands (A of FWD/DGRAD, B of DGRAD) are gathered one quad per (row, k/4); the others
  // (B of FWD, A and B of WGRAD) are contiguous along the tile column, so a thread loads a 4x4 block
  // (4 k-rows x float4 of columns) and transposes it in registers into 4 quads.
  constexpr int RA = (MODE == MODE_WGRAD) ? 4 : QA;
  constexpr int RB = (MODE == MODE_DGRAD) ? QB : 4;
  f4 ra[2][RA], rb[2][RB];

  // running (tap, channel) of this thread's next k-fast quad: kp = ks*32 + 4*(tid&7); no divisions in the loop
  int kt = 0, kc = 0;
  if constexpr (MODE != MODE_WGRAD) {
    const int kp = ks_begin * BK + 4 * (tid & 7);
    kt = kp / Cp; kc = kp - kt * Cp;
  }
  // transposed loaders: thread -> (column quad jn, k quad kq); active while kq < 8
  const int jb = tid % (BN / 4), kqb = tid / (BN / 4);
  const bool actb = kqb < 8;
  constexpr bool nvec = NVEC;         // dense rows are 16-byte aligned and quads never straddle N
  int bt = 0, bc = 0;                 // FWD: running (tap, channel) of the B rows kp = ks*32 + 4*kqb
  if constexpr (MODE == MODE_FWD) {
    const int kp = ks_begin * BK + 4 * (nvec ? kqb : 0);
    bt = kp / Cp; bc = kp - bt * Cp;
  }
  const int ja = tid % (BM / 4), kqa = tid / (BM / 4);
  const bool acta = kqa < 8;
  int wg_t = 0, wg_off = 0, wg_nvalid = 0;   // WGRAD: this thread's fixed (padded) output-row quad -> (tap, channel)
  if constexpr (MODE == MODE_WGRAD) {
    const int mp = m0 + 4 * ja;
    if (acta && mp < M) {
      wg_t = mp / Cp;
      const int c = mp - wg_t * Cp;
      wg_nvalid = Cs - c;            // >= 1
      wg_off = tapA[wg_t] + c;
    }
  }

  auto transpose_into = [&](f4 (&dst)[4], const f4 (&l)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = f4{l[0][i], l[1][i], l[2][i], l[3][i]};
  };

  constexpr bool ragged = RAGGED;     // quads at the end of a tap are partial

  auto load_tiles = [&](auto stage, int ks) {
    constexpr int ST = decltype(stage)::value;
    if constexpr (MODE == MODE_WGRAD) {
      if (acta) {   // raw rows now; the 4x4 transpose happens at store time so no load is waited for here
        if constexpr (!ragged) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const RowInfo ri = rows[4 * kqa + e];
            ra[ST][e] = guarded_quad(rs_g, ri.base + wg_off, wg_nvalid > 0 && tap_ok(ri, wg_t));
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const RowInfo ri = rows[4 * kqa + e];
            ra[ST][e] = guarded_ragged(rs_g, ri.base + wg_off, wg_nvalid > 0 && tap_ok(ri, wg_t), wg_nvalid);
          }
        }
      }
    } else {
      // k-fast gather: 8 consecutive lanes walk 8 quads (128 contiguous bytes) of one gathered row
      const int rg = tid >> 3;
      const bool kv = kt < ntaps;
      const int t = kv ? kt : 0;
      const int nvalid = Cs - kc;  // >= 1
      const int aoff = tapA[t] + kc;
      if constexpr (!ragged) {
#pragma unroll
        for (int u = 0; u < QA; ++u) {
          const RowInfo ri = rows[rg + 32 * u];
          ra[ST][u] = guarded_quad(rs_g, ri.base + aoff, kv && tap_ok(ri, t));
        }
      } else {
#pragma unroll
        for (int u = 0; u < QA; ++u) {
          const RowInfo ri = rows[rg + 32 * u];
          ra[ST][u] = guarded_ragged(rs_g, ri.base + aoff, kv && tap_ok(ri, t), nvalid);
        }
      }
      if constexpr (MODE == MODE_DGRAD) {  // B[k=(tap,o)][n=c] = W[tap][c][o], contiguous along o
        const int boff = tapB[t] + kc;
        if constexpr (!ragged) {
#pragma unroll
          for (int u = 0; u < QB; ++u) {
            const int n = n0 + rg + 32 * u;
            rb[ST][u] = guarded_quad(rs_d, boff + n * p.K, kv && n < N);
          }
        } else {
#pragma unroll
          for (int u = 0; u < QB; ++u) {
            const int n = n0 + rg + 32 * u;
            rb[ST][u] = guarded_ragged(rs_d, boff + n * p.K, kv && n < N, nvalid);
          }
        }
      }
      kc += BK;
      while (kc >= Cp) { kc -= Cp; ++kt; }
    }
    if constexpr (MODE != MODE_DGRAD) {
      // dense operand: FWD W[(tap,c)][n] rows, WGRAD dY[(b,p,q)][n] rows
      if constexpr (nvec) {
        if (actb) {
          const int n = n0 + 4 * jb;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int row;
            bool ok;
            if constexpr (MODE == MODE_FWD) { row = bt * Cs + bc + e; ok = bt < ntaps && bc + e < Cs; }
            else { row = ks * BK + 4 * kqb + e; ok = row < Kdim; }
            rb[ST][e] = guarded_quad(rs_d, row * (MODE == MODE_WGRAD ? p.Ky : N) + n, ok && n < N);   // raw row; transposed at store time
          }
        }
      } else {  // ragged N (25, 5, 3, 1 ...): 4 k-rows of one column per quad, lanes along n
        constexpr int stepB = 256 / BN;
        const int n = n0 + (tid % BN), kq0 = tid / BN;
#pragma unroll
        for (int u = 0; u < QB; ++u) {
          const int kq = kq0 + stepB * u;
          f4 v;
          if constexpr (MODE == MODE_FWD) {
            int t2 = bt, c2 = bc + 4 * kq;     // bt/bc track kp = ks*32 here (kqb term is 0 when !nvec)
            while (c2 >= Cp) { c2 -= Cp; ++t2; }
#pragma unroll
            for (int e = 0; e < 4; ++e)
              v[e] = guarded_scalar(rs_d, (t2 * Cs + c2 + e) * N + n, t2 < ntaps && c2 + e < Cs && n < N);
          } else {
            const int r = ks * BK + 4 * kq;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = guarded_scalar(rs_d, (r + e) * p.Ky + n, r + e < Kdim && n < N);
          }
          rb[ST][u] = v;
        }
      }
      if constexpr (MODE == MODE_FWD) {
        bc += BK;
        while (bc >= Cp) { bc -= Cp; ++bt; }
      }
    }
  };

  // LDS column permutation: physical = P(col) ^ kq with P(32q + 4j + e) = 32q + 8e + (j ^ 4(e>>1)).
  // Conflict-free for (i) k-fast stores (8 lanes: one column, kq = 0..7), (ii) transposed stores (8 lanes:
  // columns 4j+e for 8 consecutive j) and (iii) the MFMA operand reads (32 consecutive columns, one kq).
  auto pcol = [](int col, int kq) {
    const int j = (col >> 2) & 7, e = col & 3;
    return ((col & ~31) | (e << 3) | (j ^ ((e >> 1) << 2))) ^ kq;
  };

  // write quad `kq` of tile column `col`: a 16-byte slot (fp32) or one half of the column's oct (bf16)
  auto put = [&](f4* tile, int width, int kq, int col, const f4& q) {
    if constexpr (BF16) {
      const int oct = kq >> 1;
      char* dst = reinterpret_cast<char*>(tile + oct * width + pcol(col, oct)) + (kq & 1) * 8;
      *reinterpret_cast<bf4*>(dst) = bf4{(__bf16)q[0], (__bf16)q[1], (__bf16)q[2], (__bf16)q[3]};
    } else {
      tile[kq * width + pcol(col, kq)] = q;
    }
  };

  auto store_tiles = [&](auto stage) {
    constexpr int ST = decltype(stage)::value;
    if constexpr (MODE == MODE_WGRAD) {
      if (acta) {
        f4 q[4];
        transpose_into(q, ra[ST]);
#pragma unroll
        for (int i = 0; i < 4; ++i) put(As, BM, kqa, 4 * ja + i, q[i]);
      }
    } else {
      const int kq = tid & 7, rg = tid >> 3;
#pragma unroll
      for (int u = 0; u < QA; ++u) put(As, BM, kq, rg + 32 * u, ra[ST][u]);
      if constexpr (MODE == MODE_DGRAD) {
#pragma unroll
        for (int u = 0; u < QB; ++u) put(Bs, BN, kq, rg + 32 * u, rb[ST][u]);
      }
    }
    if constexpr (MODE != MODE_DGRAD) {
      if constexpr (nvec) {
        if (actb) {
          f4 q[4];
          transpose_into(q, rb[ST]);
#pragma unroll
          for (int i = 0; i < 4; ++i) put(Bs, BN, kqb, 4 * jb + i, q[i]);
        }
      } else {
        constexpr int stepB = 256 / BN;
        const int nb = tid % BN, kq0 = tid / BN;
#pragma unroll
        for (int u = 0; u < QB; ++u) put(Bs, BN, kq0 + stepB * u, nb, rb[ST][u]);
      }
    }
  };

  // ---- main loop ----------------------------------------------------------------------------------
  f32x16 acc[TA][TB];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int wr = wave / WN, wc = wave - wr * WN;
  const int wm0 = wr * (BM / WM), wn0 = wc * (BN / WN);
  const int lrow = lane & 31, lk = lane >> 5;

  auto compute = [&]() {
    if constexpr (BF16) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int oct = 2 * t + lk;  // lane half h holds k = 8h..8h+7 of the 16-deep MFMA: oct 2t+h of BOTH tiles
        bf8 av[TA], bv[TB];
#pragma unroll
        for (int a = 0; a < TA; ++a) av[a] = *reinterpret_cast<const bf8*>(&As[oct * BM + pcol(wm0 + 32 * a + lrow, oct)]);
#pragma unroll
        for (int b = 0; b < TB; ++b) bv[b] = *reinterpret_cast<const bf8*>(&Bs[oct * BN + pcol(wn0 + 32 * b + lrow, oct)]);
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
          for (int b = 0; b < TB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[a], bv[b], acc[a][b], 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int kq = 2 * t + lk;  // lane half h consumes quad 2t+h of BOTH tiles: same k on both sides
      f4 av[TA], bv[TB];
#pragma unroll
      for (int a = 0; a < TA; ++a) av[a] = As[kq * BM + pcol(wm0 + 32 * a + lrow, kq)];
#pragma unroll
      for (int b = 0; b < TB; ++b) bv[b] = Bs[kq * BN + pcol(wn0 + 32 * b + lrow, kq)];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
          for (int b = 0; b < TB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][e], bv[b][e], acc[a][b], 0, 0, 0);
    }
  };

  // WGRAD gathers along the reduction: its 32 row infos change every K-step and must be in LDS
  // (behind a barrier) before the loads of that step are issued.
  auto wgrad_rows = [&](int ks) {
    if constexpr (MODE == MODE_WGRAD) {
      if (tid < BK) rows[tid] = fill_row_fwd(ks * BK + tid, Kdim);
    }
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  if (ks_begin < ks_end) {
    if constexpr (MODE == MODE_WGRAD) { wgrad_rows(ks_begin); __syncthreads(); }
    load_tiles(S0{}, ks_begin);
    if (ks_begin + 1 < ks_end) {
      if constexpr (MODE == MODE_WGRAD) { __syncthreads(); wgrad_rows(ks_begin + 1); __syncthreads(); }
      load_tiles(S1{}, ks_begin + 1);
    }
    if constexpr (MODE == MODE_WGRAD) __syncthreads();
  }
  auto iterate = [&](auto stage, int ks) {
    store_tiles(stage);                  // K-step ks: registers -> LDS
    if (ks + 2 < ks_end) wgrad_rows(ks + 2);
    __syncthreads();
    if (ks + 2 < ks_end) load_tiles(stage, ks + 2);   // runs two K-steps ahead of the MFMAs
    compute();
    __syncthreads();
  };
  for (int ks = ks_begin; ks < ks_end; ks += 2) {
    iterate(S0{}, ks);
    if (ks + 1 < ks_end) iterate(S1{}, ks + 1);
  }

  // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) --------------
  float* outp = p.out + (p.splits > 1 ? (long long)blockIdx.z * p.out_numel : 0ll);
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int n = n0 + wn0 + 32 * b + lrow;
        const int m = m0 + row;
        if (m < M && n < N) {
          long long idx;
          bool ok = true;
          if constexpr (MODE == MODE_DGRAD) {
            idx = (long long)rows[row].out_off + n;
          } else if constexpr (MODE == MODE_WGRAD) {
            if (Cp == Cs) {
              idx = (long long)m * N + n;
            } else {  // drop the per-tap padding rows
              const int t = m / Cp, c = m - t * Cp;
              ok = c < Cs;
              idx = ((long long)t * Cs + c) * N + n;
            }
          } else {
            idx = (long long)m * p.Ky + n;
          }
          if (ok) {
            float v = acc[a][b][r];
            if constexpr (MODE == MODE_WGRAD) {
              if (p.splits == 1 && p.accumulate != 0.f) v += p.accumulate * outp[idx];
            }
            outp[idx] = v;
          }
        }
      }
}


struct Plan {
  int cfg;       // 0: 128x128, 1: 128x64, 2: 128x32, 3: 64x64
  int bm, bn;
  long long M, N;  // per class (class 0 = largest) GEMM extents (M padded per tap for WGRAD)
  int classes, nk, splits;
  bool ragged, nvec, bf16;
  long long tiles, out_numel;
};

template <int MODE>
int launch_mode(const Plan& pl, const ConvArgs& a, hipStream_t st);

template <int MODE, bool RAGGED, bool NVEC, bool BF16>
static inline void launch_cfg(const Plan& pl, const ConvArgs& a, hipStream_t st) {
  const dim3 grid((unsigned)(acg::ceil_div(pl.M, pl.bm) * acg::ceil_div(pl.N, pl.bn)), (unsigned)pl.classes, (unsigned)pl.splits);
  switch (pl.cfg) {
    case 0: hipLaunchKernelGGL((conv_mfma_f32<MODE, 128, 128, 2, 2, RAGGED, NVEC, BF16>), grid, dim3(256), 0, st, a); break;
    case 1: hipLaunchKernelGGL((conv_mfma_f32<MODE, 128, 64, 2, 2, RAGGED, NVEC, BF16>), grid, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((conv_mfma_f32<MODE, 128, 32, 4, 1, RAGGED, NVEC, BF16>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((conv_mfma_f32<MODE, 64, 64, 2, 2, RAGGED, NVEC, BF16>), grid, dim3(256), 0, st, a); break;
  }
}
template <int MODE, bool BF16>
static inline void launch_variant(const Plan& pl, const ConvArgs& a, hipStream_t st) {
  if (pl.ragged) {
    if (pl.nvec) launch_cfg<MODE, true, true, BF16>(pl, a, st);
    else launch_cfg<MODE, true, false, BF16>(pl, a, st);
  } else {
    if (pl.nvec) launch_cfg<MODE, false, true, BF16>(pl, a, st);
    else launch_cfg<MODE, false, false, BF16>(pl, a, st);
  }
}

#define ACG_DEFINE_CONV_LAUNCH(MODE)                                                    \
  template <>                                                                           \
  int launch_mode<MODE>(const Plan& pl, const ConvArgs& a, hipStream_t st) {            \
    if (pl.bf16) launch_variant<MODE, true>(pl, a, st);                                 \
    else launch_variant<MODE, false>(pl, a, st);                                        \
    return acg::check_launch("conv_mfma_f32");                                          \
  }

}  // namespace acgconv
