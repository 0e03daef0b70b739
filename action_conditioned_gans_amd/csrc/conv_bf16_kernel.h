// Kernel template of the bf16 implicit-GEMM convolution (bf16 tensors in HBM, v_mfma_f32_32x32x16_bf16, fp32 accumulate).
// Same contractions, descriptors and pipeline as conv_f32_kernel.h; what bf16 storage changes:
//  * a 16-byte unit is an OCT of 8 channels, a K-step is 64 k (8 octs): one buffer_load_dwordx4 per oct, written to LDS
//    as it arrived - no conversion, no per-element work;
//  * every channel count is padded to a multiple of 8 IN MEMORY (activations at pitch round8(C), zero pad channels;
//    filters in the two prepared bf16 layouts of acg_weights_prepare_bf16), so there is no ragged path;
//  * FWD and DGRAD take BOTH operands k-fast (the filter copy [tap][n][c8] for FWD, [tap][c][o8] for DGRAD), in the
//    conflict-free [slot][P(col)^slot] LDS image of the fp32 kernel;
//  * WGRAD contracts over pixels, the slow axis of both of its operands: the tiles go to LDS as they lie in memory,
//    [pixel][channel] in 8-row x 32-column subtiles, and the MFMA fragments are read with ds_read_b64_tr_b16 (hardware
//    transpose, two reads per fragment) - no register transpose.
// Tiles: 64x64 and 128x128 per 256-thread block (wave tile 32x32 / 64x64), and 128x32 (four waves stacked along M, wave
// tile 32x32) for FWD / DGRAD contractions with at most 32 output columns: g/tconv4's 25 DNA logits, d/conv1's 6-channel
// input gradient, g/conv1 - on a 64-wide tile half or more of every MFMA was padding (profiles/r2: 174 and 34 TFLOP/s).
#pragma once
#include "conv_f32_kernel.h"

namespace acgconv {

typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));

constexpr int BKH = 64;   // k per K-step: 8 octs

template <int MODE, int BM, int BN>
constexpr int conv16_lds_bytes() {
  return 2 * 8 * (BM + BN) * 16 + ((MODE == MODE_WGRAD) ? 2 * 256 : BM) * (int)sizeof(RowInfo) + 2 * kMaxTaps * (int)sizeof(int);
}

__device__ __forceinline__ f4 guarded_oct(__amdgpu_buffer_rsrc_t rs, int elem_off, bool ok) {
  const unsigned off = ok ? (unsigned)elem_off * 2u : kOob;
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}

// byte offset of 16-byte chunk `ch` of row `row` in a [rows][W16 chunks] bf16 tile stored as 8-row x 32-column
// subtiles (cdna guide T10, image (a)): transposed reads are conflict-free.  Round 3: the row pair inside a subtile is
// swapped in every odd 32-column block ((row & 7) ^ ((ch >> 2) & 1)).  A ds_write_b128 group is 8 lanes = chunks 0..7 of ONE
// row, i.e. two subtiles 512 bytes apart - the same 64 bytes of the 128-byte store banking, a 2-way conflict on every
// store of the weight-gradient loader (PMC: 28 % of that kernel's LDS cycles, profiles/r3/b_conv16_pmc_wgrad_vs_fwd.txt);
// with the swap the two halves land in different 64-byte bank halves.  Reads pick the same rows, permuted.
template <int W16>
__device__ __forceinline__ int tr_off(int row, int ch) {
  return (W16 / 4) * 512 * (row >> 3) + 512 * (ch >> 2) + 64 * ((row & 7) ^ ((ch >> 2) & 1)) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

template <int MODE, int BM, int BN, bool EPI = false>
__device__ __forceinline__ void conv16_body(const ConvArgs& p, const int bx, const int by, const int bz, const int gx, char* smem) {
  constexpr int WN = BN >= 64 ? 2 : 1, WM = 4 / WN;
  constexpr int TA = BM / (32 * WM), TB = BN / (32 * WN);
  constexpr int QA = BM / 32, QB = BN / 32;          // octs per thread per K-step and operand
  constexpr int NROW = (MODE == MODE_WGRAD) ? 2 * 256 : BM;
  constexpr int NST = 3;
  constexpr int ASZ = 8 * BM, BSZ = 8 * BN;          // 16-byte cells per K-step
  f4* const As_all = reinterpret_cast<f4*>(smem);
  f4* const Bs_all = As_all + 2 * ASZ;
  RowInfo* const rows = reinterpret_cast<RowInfo*>(Bs_all + 2 * BSZ);
  int* const tapA = reinterpret_cast<int*>(rows + NROW);
  int* const tapB = tapA + kMaxTaps;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gsrc), 0, p.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dense), 0, p.d_bytes, 0x00020000);

  // ---- geometry (wave-uniform); Cp = channels of the gathered tensor = its pitch in memory (multiple of 8) ------
  const int Cin8 = (p.C + 7) & ~7, K8 = (p.K + 7) & ~7;
  int M, N, Kdim, Cp, ntaps;
  int ph = 0, pw = 0, i0 = 0, j0 = 0, nti = 1, ntj = 1, dp0 = 0, dq0 = 0, Hc = 0, Wc = 0;
  if constexpr (MODE == MODE_FWD) {
    Cp = Cin8; ntaps = p.KH * p.KW;
    M = p.batch * p.OH * p.OW; N = p.K; Kdim = ntaps * Cp;
  } else if constexpr (MODE == MODE_DGRAD) {
    const int cls = by;
    ph = cls / p.sw; pw = cls - ph * p.sw;
    Hc = ph < p.H ? (p.H - ph + p.sh - 1) / p.sh : 0;
    Wc = pw < p.W ? (p.W - pw + p.sw - 1) / p.sw : 0;
    Cp = K8;
    M = p.batch * Hc * Wc; N = p.C;
    i0 = (ph + p.pt) % p.sh; j0 = (pw + p.pl) % p.sw;
    nti = i0 < p.KH ? (p.KH - i0 + p.sh - 1) / p.sh : 0;
    ntj = j0 < p.KW ? (p.KW - j0 + p.sw - 1) / p.sw : 0;
    dp0 = (ph + p.pt - i0) / p.sh; dq0 = (pw + p.pl - j0) / p.sw;
    ntaps = nti * ntj; Kdim = ntaps * Cp;
  } else {
    Cp = Cin8; ntaps = p.KH * p.KW;
    M = ntaps * Cp; N = p.K; Kdim = p.batch * p.OH * p.OW;
  }
  if (MODE != MODE_WGRAD && p.Nv > 0) N = p.Nv;      // only the first Nv output columns (acg_conv_desc dgrad_c / adj_dgrad_c)
  const int tiles_n = (N + BN - 1) / BN;
  int bid = bx;
  if constexpr (MODE != MODE_WGRAD) {      // (weight gradients are placed by wgrad_xcd_map in the kernel wrappers)
    const int nwg = gx, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  if (m0 >= M) return;

  const int nk = (Kdim + BKH - 1) / BKH;
  const int per = (nk + p.splits - 1) / p.splits;
  const int ks_begin = bz * per;
  const int ks_end = min(nk, ks_begin + per);

  // ---- tap tables (element offsets) ------------------------------------------------------------
  if (tid < kMaxTaps && tid < ntaps) {
    const int t = tid;
    if constexpr (MODE == MODE_DGRAD) {
      const int ti = t / ntj, tj = t - ti * ntj;
      tapA[t] = -(ti * p.OW + tj) * K8;
      tapB[t] = ((i0 + p.sh * ti) * p.KW + (j0 + p.sw * tj)) * p.C * K8;      // filter copy [tap][c][o8]
    } else {
      const int i = t / p.KW, j = t - i * p.KW;
      const int u = (MODE == MODE_FWD && p.tap_classes) ? tap_class_pos(i, j, p.KH, p.KW) : t;      // position in the K order
      tapA[u] = (i * p.W + j) * Cin8;
      tapB[u] = t * p.K * Cin8;                                               // filter copy [tap][o][c8] (FWD)
    }
  }

  auto fill_row_fwd = [&](int r, int limit) -> RowInfo {
    RowInfo ri; ri.base = 0; ri.mask_lo = 0; ri.mask_hi = 0; ri.out_off = 0;
    if (r < limit) {
      const int t2 = div_fast(r, p.mg_ow, p.sh_ow), q = r - t2 * p.OW;
      const int b = div_fast(t2, p.mg_oh, p.sh_oh), pp = t2 - b * p.OH;
      const int y0 = pp * p.sh - p.pt, x0 = q * p.sw - p.pl;
      ri.base = ((b * p.H + y0) * p.W + x0) * Cin8;
      if (p.shuf_c) ri.out_off = ((b * (2 * p.OH) + 2 * pp) * p.shuf_w + 2 * q) * p.shuf_pitch;     // merged input gradient (ConvArgs::shuf_c)
      const unsigned long long m = (MODE == MODE_FWD && p.tap_classes) ? tap_mask_classes(max(0, -y0), min(p.KH, p.H - y0), max(0, -x0), min(p.KW, p.W - x0), p.KH, p.KW)
                                                                      : tap_mask(max(0, -y0), min(p.KH, p.H - y0), max(0, -x0), min(p.KW, p.W - x0), p.KW);
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
        const int y0 = h2 + dp0, x0 = w2 + dq0;
        ri.base = ((b * p.OH + y0) * p.OW + x0) * K8;
        ri.out_off = ((b * p.H + h2 * p.sh + ph) * p.W + (w2 * p.sw + pw)) * (EPI ? p.Cx : (p.slab_rows ? 4 : Cin8));      // EPI: dense float32 rows
        const unsigned long long mk = tap_mask(max(0, y0 - p.OH + 1), min(nti, y0 + 1), max(0, x0 - p.OW + 1), min(ntj, x0 + 1), ntj);
        ri.mask_lo = (unsigned)mk; ri.mask_hi = (unsigned)(mk >> 32);
      }
      rows[r] = ri;
    }
  }
  __syncthreads();

  // ---- loaders -------------------------------------------------------------------------------------
  f4 ra[NST][QA], rb[NST][QB];
  auto tap_of = [&](int kp, int& t, int& c) { t = div_fast(kp, p.mg_cp, p.sh_cp); c = kp - t * Cp; };

  // k-fast operands (FWD / DGRAD): thread -> (row group rg = tid>>3, oct slot kq = tid&7); rows rg + 32u
  RowInfo myrow[MODE == MODE_WGRAD ? 1 : QA];
  int nK[MODE == MODE_WGRAD ? 1 : QB];
  bool nOk[MODE == MODE_WGRAD ? 1 : QB];
  if constexpr (MODE != MODE_WGRAD) {
#pragma unroll
    for (int u = 0; u < QA; ++u) myrow[u] = rows[(tid >> 3) + 32 * u];
#pragma unroll
    for (int u = 0; u < QB; ++u) { const int n = n0 + (tid >> 3) + 32 * u; nK[u] = n * Cp; nOk[u] = n < N; }
  }
  const int step_t = BKH / Cp, step_c = BKH - step_t * Cp;
  int run_kt = 0, run_kc = 0;
  // K order of the gathered operand (ConvArgs::korder).  0: k = (tap, channel), channels fastest - consecutive K-steps sweep ALL
  // channels of a tap's shifted window.  1 (channels a multiple of 64): k = (64-channel chunk, tap, channel in chunk) - 25
  // consecutive K-steps re-touch the SAME half or quarter of the window's bytes at 25 shifts, so what the tiles of an XCD keep
  // live in its L2 shrinks by the chunk count (the order of a sum is free; results differ at rounding level only)
  const bool chunked = MODE != MODE_WGRAD && p.korder != 0;
  const int inc_t = chunked ? 1 : step_t, inc_c = chunked ? 0 : step_c, wrap_t = chunked ? -ntaps : 1, wrap_c = chunked ? BKH : -Cp;
  if constexpr (MODE != MODE_WGRAD) {
    if (chunked) { const int ch = ks_begin / ntaps; run_kt = ks_begin - ch * ntaps; run_kc = ch * BKH + 8 * (tid & 7); }
    else tap_of(ks_begin * BKH + 8 * (tid & 7), run_kt, run_kc);
  }

  // pixel-major operands (WGRAD): thread -> (pixel row kr0 + e * RPA, oct ja of the tile row)
  constexpr int OA = BM / 8, OB = BN / 8, RPA = 256 / OA, RPB = 256 / OB;   // octs per tile row; rows per pass
  const int ja = tid % OA, kra = tid / OA, jb = tid % OB, krb = tid / OB;
  int wg_t = 0, wg_off = 0;
  bool wg_ok = false;
  if constexpr (MODE == MODE_WGRAD) {
    const int mp = m0 + 8 * ja;
    if (mp < M) {
      wg_t = mp / Cp;
      wg_off = tapA[wg_t] + (mp - wg_t * Cp);
      wg_ok = true;
    }
  }
  const int nB = n0 + 8 * jb;
  int run_boff = 0;
  if constexpr (MODE == MODE_WGRAD) run_boff = (ks_begin * BKH + krb) * K8 + nB;

  struct Prep {
    int t, aoff, tb, brow, boff;
    bool kv;
    RowInfo wri[MODE == MODE_WGRAD ? QA : 1];
  };
  auto load_prep = [&](int ks, bool live) {
    Prep q{};
    if constexpr (MODE == MODE_WGRAD) {
#pragma unroll
      for (int e = 0; e < QA; ++e) q.wri[e] = rows[((ks - ks_begin) & 7) * BKH + kra + e * RPA];
      q.boff = run_boff; q.brow = ks * BKH + krb;
      run_boff += BKH * K8;
    } else {
      q.kv = live && run_kt < ntaps && run_kc < Cp;
      q.t = q.kv ? run_kt : 0;
      q.aoff = tapA[q.t] + run_kc;
      q.tb = tapB[q.t] + run_kc;
      // (selects, no branch: the loop body must stay ONE basic block or hipcc's counted vmcnt waits turn conservative)
      run_kt += inc_t; run_kc += inc_c;
      const bool wrap = chunked ? run_kt >= ntaps : run_kc >= Cp;
      run_kt += wrap ? wrap_t : 0; run_kc += wrap ? wrap_c : 0;
    }
    return q;
  };
  constexpr int NLD = QA + QB;
  auto load_piece = [&](auto stage, bool live, const Prep& q, auto pc) {
    constexpr int ST = decltype(stage)::value;
    constexpr int I = decltype(pc)::value;
    if constexpr (I < QA) {
      if constexpr (MODE == MODE_WGRAD) {
        const RowInfo ri = q.wri[I];
        ra[ST][I] = guarded_oct(rs_g, ri.base + wg_off, live && wg_ok && tap_ok(ri, wg_t));
      } else {
        const RowInfo ri = myrow[I];
        ra[ST][I] = guarded_oct(rs_g, ri.base + q.aoff, q.kv && tap_ok(ri, q.t));
      }
    } else {
      constexpr int U = I - QA;
      if constexpr (MODE == MODE_WGRAD) {
        rb[ST][U] = guarded_oct(rs_d, q.boff + U * RPB * K8, live && q.brow + U * RPB < Kdim && nB < N);
      } else {
        rb[ST][U] = guarded_oct(rs_d, q.tb + nK[U], q.kv && nOk[U]);
      }
    }
  };
  auto load_tiles = [&](auto stage, int ks, bool live) {
    const Prep q = load_prep(ks, live);
    static_for<0, NLD>([&](auto pc) { load_piece(stage, live, q, pc); });
  };

  auto pcol = [](int col, int kq) {
    const int j = (col >> 2) & 7, e = col & 3;
    return ((col & ~31) | (e << 3) | (j ^ ((e >> 1) << 2))) ^ kq;
  };
  auto store_tiles = [&](auto stage, int buf) {
    constexpr int ST = decltype(stage)::value;
    f4* const As = As_all + buf * ASZ;
    f4* const Bs = Bs_all + buf * BSZ;
    if constexpr (MODE == MODE_WGRAD) {
      char* const Ab = reinterpret_cast<char*>(As);
      char* const Bb = reinterpret_cast<char*>(Bs);
#pragma unroll
      for (int e = 0; e < QA; ++e) *reinterpret_cast<f4*>(Ab + tr_off<OA>(kra + e * RPA, ja)) = ra[ST][e];
#pragma unroll
      for (int e = 0; e < QB; ++e) *reinterpret_cast<f4*>(Bb + tr_off<OB>(krb + e * RPB, jb)) = rb[ST][e];
    } else {
      const int kq = tid & 7, rg = tid >> 3;
#pragma unroll
      for (int u = 0; u < QA; ++u) As[kq * BM + pcol(rg + 32 * u, kq)] = ra[ST][u];
#pragma unroll
      for (int u = 0; u < QB; ++u) Bs[kq * BN + pcol(rg + 32 * u, kq)] = rb[ST][u];
    }
  };

  // ---- main loop ------------------------------------------------------------------------------------
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

  // transposed-read addressing (WGRAD): lane 4q+p of a 16-lane group gives row q, columns 4p..4p+3 of a 4 x 16 block
  int tra[MODE == MODE_WGRAD ? TA : 1][2], trb[MODE == MODE_WGRAD ? TB : 1][2];
  if constexpr (MODE == MODE_WGRAD) {
    const int g1 = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int x = (2 * lk + s) & 3, r8 = q + 4 * s;      // row & 7 of the fragment row this lane addresses
#pragma unroll
      for (int a = 0; a < TA; ++a) {
        const int col = wm0 + 32 * a + 16 * g1 + 4 * pp, ch = col >> 3;
        tra[a][s] = (OA / 4) * 512 * lk + 64 * (r8 ^ ((ch >> 2) & 1)) + 512 * (ch >> 2) + 16 * ((ch & 3) ^ x) + 8 * ((col >> 2) & 1);
      }
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        const int col = wn0 + 32 * b + 16 * g1 + 4 * pp, ch = col >> 3;
        trb[b][s] = (OB / 4) * 512 * lk + 64 * (r8 ^ ((ch >> 2) & 1)) + 512 * (ch >> 2) + 16 * ((ch & 3) ^ x) + 8 * ((col >> 2) & 1);
      }
    }
  }

  // MFMA work of a K-step: 4 k-slices of 16; slice t takes oct 2t + (lane>>5) of both tiles
  constexpr int NT = 4, GM = TA * TB, NM = NT * GM;
  f4 av[2][TA], bv[2][TB];
  auto frag_read = [&](int buf, auto tc) {
    constexpr int T = decltype(tc)::value;
    if constexpr (MODE == MODE_WGRAD) {
      typedef s4v __attribute__((address_space(3))) * lds_s4;
      char* const Ab = reinterpret_cast<char*>(As_all + buf * ASZ) + T * 2 * (OA / 4) * 512;
      char* const Bb = reinterpret_cast<char*>(Bs_all + buf * BSZ) + T * 2 * (OB / 4) * 512;
#pragma unroll
      for (int a = 0; a < TA; ++a) {
        const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(Ab + tra[a][0]));
        const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(Ab + tra[a][1]));
        av[T & 1][a] = __builtin_bit_cast(f4, s8v{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
      }
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(Bb + trb[b][0]));
        const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(Bb + trb[b][1]));
        bv[T & 1][b] = __builtin_bit_cast(f4, s8v{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
      }
    } else {
      const f4* const As = As_all + buf * ASZ;
      const f4* const Bs = Bs_all + buf * BSZ;
      const int kq = 2 * T + lk;
#pragma unroll
      for (int a = 0; a < TA; ++a) av[T & 1][a] = As[kq * BM + pcol(wm0 + 32 * a + lrow, kq)];
#pragma unroll
      for (int b = 0; b < TB; ++b) bv[T & 1][b] = Bs[kq * BN + pcol(wn0 + 32 * b + lrow, kq)];
    }
  };
  auto mfma = [&](auto ic) {
    constexpr int I = decltype(ic)::value;
    constexpr int T = I / GM, AB = I % GM, A = AB / TB, B = AB % TB;
    acc[A][B] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, av[T & 1][A]), __builtin_bit_cast(bf8, bv[T & 1][B]), acc[A][B], 0, 0, 0);
  };

  // WGRAD row infos: 4 K-steps (256 pixel rows, one per thread) per chunk, two chunks resident
  auto wgrad_rows = [&](int chunk) {
    if constexpr (MODE == MODE_WGRAD) rows[(chunk & 1) * 256 + tid] = fill_row_fwd((ks_begin + 4 * chunk) * BKH + tid, Kdim);
  };

  if (ks_begin < ks_end) {
    if constexpr (MODE == MODE_WGRAD) { wgrad_rows(0); __syncthreads(); }
    static_for<0, NST>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      load_tiles(jc, ks_begin + J, ks_begin + J < ks_end);
    });
    store_tiles(std::integral_constant<int, 0>{}, 0);
    __syncthreads();
  }
  auto iterate = [&](auto jc, int ks, int buf) {
    constexpr int J = decltype(jc)::value;
    using SN = std::integral_constant<int, (J + 1) % NST>;
    const bool live = ks + NST < ks_end;
    frag_read(buf, std::integral_constant<int, 0>{});
    const Prep q = load_prep(ks + NST, live);
    store_tiles(SN{}, buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NM>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      mfma(ic);
      if constexpr (I % GM == 0 && I / GM + 1 < NT) frag_read(buf, std::integral_constant<int, I / GM + 1>{});
      static_for<0, NLD>([&](auto pc) {
        constexpr int P = decltype(pc)::value;
        if constexpr ((P * NM) / NLD == I) load_piece(jc, live, q, pc);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (MODE == MODE_WGRAD) {
      const int rel = ks - ks_begin + NST + 1;
      if ((rel & 3) == 0 && ks + NST + 1 < ks_end) wgrad_rows(rel >> 2);
    }
    __syncthreads();
  };
  if (ks_begin < ks_end) {
    constexpr int UNR = 2 * NST;
    int ks = ks_begin;
    while (run_unrolled<0, UNR>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      iterate(std::integral_constant<int, I % NST>{}, ks, I & 1);
      return ++ks < ks_end;
    })) {
    }
  }

  // ---- epilogue --------------------------------------------------------------------------------------
  // splits == 1: FWD / DGRAD write the bf16 activation (pitch round8), WGRAD the fp32 gradient; splits > 1: fp32 slabs
  // of out_numel elements laid out like the final tensor, summed (and rounded to bf16) by splitk_reduce.
  const bool to_bf16 = MODE != MODE_WGRAD && p.splits == 1 && !p.out_f32 && !EPI;
  if constexpr (EPI) {          // bias + activation of a layer built with normalizer_fn = None (models.py:20-21): float32 out
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      const int n = n0 + wn0 + 32 * b + lrow;
      const float bv = n < N ? p.bias[n] : 0.f;
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = acg::act_apply(p.act, acc[a][b][r] + bv, p.leak);
    }
  }
  if constexpr (MODE != MODE_WGRAD) {
    if (p.stats != nullptr) {     // BatchNorm statistics of this tile, of the bf16 values as stored (ConvArgs::stats, conv_f32_kernel.h)
      int g = 0, blk;
      if constexpr (MODE == MODE_DGRAD) { blk = by * p.stats_tpg + tm; } else { g = tm / p.stats_tpg; blk = tm - g * p.stats_tpg; }
      tile_stats_epilogue<BM, BN, WM, TA, TB>([&](int a, int b, int r) { return to_bf16 ? (float)(__bf16)acc[a][b][r] : acc[a][b][r]; }, reinterpret_cast<float*>(smem),
                                              p.stats + ((long long)g * p.stats_nblk + blk) * 2 * N, N, M, m0, n0, wm0, wn0, wr, lrow, lk, tid);
    }
  }
  float* const outf = p.out + (p.splits > 1 ? (long long)bz * p.out_numel : 0ll);
  __bf16* const outh = reinterpret_cast<__bf16*>(p.out);
  if constexpr (MODE != MODE_WGRAD) {
    // (A packed variant - neighbouring lanes trade registers over DPP so that each stores 4 bytes, half the store
    // instructions - measured SLOWER in the same session: conv stack of config 5 2246 vs 2088 us, of config 3 908 vs 864 us,
    // profiles/r2/c_epilogue_ab.txt; the plain 2-byte stores below stay.)
  }
  // Full tiles - all of them in the layers that matter - take a straight-line path: one base pointer per 32 x 32 sub-tile,
  // sixteen stores at row strides, no per-element bounds test, branch or 64-bit index multiply.  (The general loop below
  // costs ~1600 instructions and ~190 branches per thread on a 128 x 128 tile: 4.8 us of an 8.7 us one-K-step launch,
  // tools/fixed_cost_probe.py.)
  // (rows full; a ragged last column tile - 25 or 121 output channels - only masks lanes, once per 32-column sub-tile)
  const bool full = m0 + BM <= M && (MODE != MODE_WGRAD || (Cp == p.C && !(p.splits == 1 && p.accumulate != 0.f)));
  // column term of an element's offset: n, or in the quad slab layout (ConvArgs::slab_rows; float32 slabs only) the quad's
  // run of rows + n & 3
  const bool quads = MODE != MODE_WGRAD && p.slab_rows > 0;
  const bool shuf = MODE == MODE_FWD && p.shuf_c > 0;      // merged input gradient: rows by their out_off, columns by (2 x 2 pixel, channel)
  auto col_off = [&](int n) -> long long {
    if (shuf) { const int cls = n / p.shuf_c; return (long long)((cls >> 1) * p.shuf_w + (cls & 1)) * p.shuf_pitch + (n - cls * p.shuf_c); }
    return quads ? (long long)(n >> 2) * p.slab_rows * 4 + (n & 3) : (long long)n;
  };
  if (full) {
    auto store_rows = [&](auto* base, long long pitch, int a, int b) {
      using T = std::remove_pointer_t<decltype(base)>;
#pragma unroll
      for (int r = 0; r < 16; ++r) base[(long long)((r & 3) + 8 * (r >> 2)) * pitch] = (T)acc[a][b][r];
    };
    if (MODE == MODE_DGRAD || shuf) {
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        if (n0 + wn0 + 32 * b + lrow < N) {
          const long long co = col_off(n0 + wn0 + 32 * b + lrow);
#pragma unroll
          for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const long long off = rows[wm0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lk].out_off + co;
              if (to_bf16) outh[off] = (__bf16)acc[a][b][r];
              else outf[off] = acc[a][b][r];
            }
        }
      }
    } else {
      const long long pitch = MODE == MODE_WGRAD ? N : (quads ? 4 : K8);
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        if (n0 + wn0 + 32 * b + lrow < N) {
#pragma unroll
          for (int a = 0; a < TA; ++a) {
            const long long o = (long long)(m0 + wm0 + 32 * a + 4 * lk) * pitch + col_off(n0 + wn0 + 32 * b + lrow);
            if (to_bf16) store_rows(outh + o, pitch, a, b);
            else store_rows(outf + o, pitch, a, b);
          }
        }
      }
    }
    return;
  }
  // partial row tiles (the last tile of a weight gradient whose taps x channels is no multiple of the tile; tiny layers): per
  // ROW one validity test and one base index (the weight gradient's division by the padded channel count included), per
  // element only the column mask
  const bool acc_out = MODE == MODE_WGRAD && p.splits == 1 && p.accumulate != 0.f;
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lk;
      const int m = m0 + row;
      if (m >= M) continue;
      long long base;
      if (MODE == MODE_DGRAD || shuf) {
        base = rows[row].out_off;
      } else if constexpr (MODE == MODE_WGRAD) {
        const int t = div_fast(m, p.mg_cp, p.sh_cp), c = m - t * Cp;
        if (c >= p.C) continue;
        base = ((long long)t * p.C + c) * N;
      } else {
        base = (long long)m * (quads ? 4 : K8);
      }
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        const int n = n0 + wn0 + 32 * b + lrow;
        if (n < N) {
          const long long o = base + col_off(n);
          float v = acc[a][b][r];
          if (MODE == MODE_WGRAD || !to_bf16) {
            if (acc_out) v += p.accumulate * outf[o];
            outf[o] = v;
          } else {
            outh[o] = (__bf16)v;
          }
        }
      }
    }
}

template <int MODE, int BM, int BN, bool EPI = false>
__global__ __launch_bounds__(256) void conv_mfma_bf16(const ConvArgs p) {
  __shared__ __align__(16) char smem[conv16_lds_bytes<MODE, BM, BN>()];
  if constexpr (MODE == MODE_WGRAD) {      // grid (tiles, 1, splits)
    int tile, split;
    wgrad_xcd_map((int)(blockIdx.x + gridDim.x * blockIdx.z), (int)gridDim.x, (int)gridDim.z, tile, split);
    conv16_body<MODE, BM, BN>(p, tile, 0, split, (int)gridDim.x, smem);
  } else {
    conv16_body<MODE, BM, BN, EPI>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, smem);
  }
}

// A layer's input gradient (DGRAD, or FWD on the adjoint descriptor of a transposed layer) and its weight gradient out
// of ONE grid, as conv_pair_f32 does for fp32: both consume dy, neither reads the other's result; one launch boundary
// less and each contraction's blocks fill the other's tail.  Blocks [0, nA) run A with the grid (gxA, gyA, *).
template <int MODE_A, int BMA, int BNA, int BMB, int BNB>
__global__ __launch_bounds__(256) void conv_pair_bf16(const ConvArgs a, const ConvArgs b, const PairGeom g) {
  constexpr int LA = conv16_lds_bytes<MODE_A, BMA, BNA>(), LB = conv16_lds_bytes<MODE_WGRAD, BMB, BNB>();
  __shared__ __align__(16) char smem[LA > LB ? LA : LB];
  const int L = (int)blockIdx.x;
  if (L < g.nA) {
    const int t = L / g.gxA;
    conv16_body<MODE_A, BMA, BNA>(a, L - t * g.gxA, t % g.gyA, t / g.gyA, g.gxA, smem);
  } else {
    int tile, split;
    wgrad_xcd_map(L - g.nA, g.gxB, b.splits, tile, split);
    conv16_body<MODE_WGRAD, BMB, BNB>(b, tile, 0, split, g.gxB, smem);
  }
}
int launch_pair16(int modeA, const Plan& pa, const ConvArgs& a, const Plan& pb, const ConvArgs& b, hipStream_t st);

int launch_glds16(int mode, const Plan& pl, const ConvArgs& a, hipStream_t st);      // conv_bf16_glds.hip

template <int MODE>
int launch_mode16(const Plan& pl, const ConvArgs& a, hipStream_t st);

#define ACG_DEFINE_CONV16_LAUNCH(MODE)                                                                        \
  template <>                                                                                                 \
  int launch_mode16<MODE>(const Plan& pl, const ConvArgs& a, hipStream_t st) {                                \
    const dim3 grid((unsigned)(acg::ceil_div(pl.M, pl.bm) * acg::ceil_div(pl.N, pl.bn)), (unsigned)pl.classes, \
                    (unsigned)pl.splits);                                                                     \
    if (pl.bm == 256) {                                                                                       \
      if constexpr (MODE != MODE_WGRAD) return launch_glds16(MODE, pl, a, st);                                \
      else return acg::fail(ACG_ERR_UNSUPPORTED, "conv_mfma_bf16: no 256x128 weight-gradient tile");          \
    } else if (pl.bn == 32) {                                                                                 \
      if constexpr (MODE != MODE_WGRAD) ACG_LAUNCH((conv_mfma_bf16<MODE, 128, 32>), grid, dim3(256), 0, st, a); \
      else return acg::fail(ACG_ERR_UNSUPPORTED, "conv_mfma_bf16: no 128x32 weight-gradient tile");           \
    } else if (pl.bm == 128) ACG_LAUNCH((conv_mfma_bf16<MODE, 128, 128>), grid, dim3(256), 0, st, a);         \
    else ACG_LAUNCH((conv_mfma_bf16<MODE, 64, 64>), grid, dim3(256), 0, st, a);                               \
    return acg::check_launch("conv_mfma_bf16");                                                               \
  }

}  // namespace acgconv
