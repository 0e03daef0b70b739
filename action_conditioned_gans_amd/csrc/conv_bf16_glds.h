// bf16 implicit-GEMM convolution, FWD / DGRAD, 256 x 128 tile, operands staged by LDS-DMA (round 4).
//
// conv_bf16_kernel.h stages a K-step HBM -> VGPR -> ds_write_b128 -> LDS (three register stages, two LDS buffers, two blocks
// of 235 VGPRs per CU): its 128 x 128 K-loop runs at ~850 TFLOP/s with the matrix pipe 35 % busy - 32 KB of ds_write_b128 per
// block K-step (~79 B/clk/CU) and 64 KB of fragment reads against 512 MFMA cycles (profiles/r2/a_conv16_*).  Here:
//  * `buffer_load_dwordx4 ... lds` (__builtin_amdgcn_raw_ptr_buffer_load_lds): the gathered octs go straight from the
//    memory pipeline into LDS - no staging registers, no ds_write instructions, no store-side bank conflicts.  A masked lane
//    (padding tap, row or column beyond the tensor) carries an out-of-range offset: the buffer descriptor's range check
//    makes the hardware write zeros, exactly as the register loads did;
//  * the LDS destination of one wave-instruction is a linear 1 KB run (wave-uniform base + lane * 16), so the image is
//    chosen for the DMA: [row][8 octs] row-major, a wave-instruction = 8 consecutive rows x their 8 octs - full 128-byte
//    lines of each gathered row - and the swizzle lives on the SOURCE side: the lane writing physical slot s of row r
//    fetches logical oct s ^ ((r >> 1) & 7).  Fragment reads (ds_read_b128, row = lane & 31) then touch 16 distinct
//    16-byte bank positions per 16-lane service group: conflict-free without padding;
//  * 512 threads = 8 waves as 4 (M) x 2 (N), wave tile 64 x 64 (16 MFMAs of 32x32x16 per wave and K-step, as the 128 x 128
//    tile), ONE block per CU: three LDS buffers of 48 KB; a K-step is 48 wave-instructions of DMA, 6 per wave;
//  * counted waits: a wave keeps the 6 DMA instructions of K-step ks + 2 in flight across the barrier
//    (`s_waitcnt vmcnt(6)` + raw `s_barrier`: __syncthreads() would drain them, cdna_hip_programming.md section 5); the tail
//    issues fully masked (zero) loads so that the count stays exact;
//  * K-loop schedule (template VAR; 2 ships): the waves of a SIMD (w, w + 4) belong to two groups one barrier apart; a K-step is a
//    load phase (16 fragment reads, the offsets of the next DMA pieces) and a compute phase (16 MFMAs at raised priority with the 6
//    DMA pieces issued BETWEEN them), so one wave of each SIMD multiplies while the other reads.  Measured against the in-step
//    loop (VAR 0): -3...-9 % per launch; what bounds all three variants is outside the CU (profiles/r4/h_bf16_kloop_load_path.txt).
// Used for unsplit, unpaired FWD / DGRAD contractions with more than 64 output columns whose 256 x 128 tiles fill the chip
// (make_plan cfg 4); everything else - and every weight gradient - stays on conv_bf16_kernel.h.
#pragma once
#include "conv_bf16_kernel.h"

namespace acgconv {

constexpr int GBM = 256, GBN = 128, GNT = 512, GNBUF = 3;
constexpr int GCELLS = 8 * (GBM + GBN);          // 16-byte cells per K-step buffer
constexpr int conv16g_lds_bytes() { return GNBUF * GCELLS * 16 + GBM * (int)sizeof(RowInfo) + 2 * kMaxTaps * (int)sizeof(int); }

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int MODE, int VAR = 0>
__device__ __forceinline__ void conv16g_body(const ConvArgs& p, char* smem) {
  static_assert(MODE == MODE_FWD || MODE == MODE_DGRAD, "forward / input gradient only");
  constexpr int WM = 4, TA = 2, TB = 2;          // 8 waves as 4 (M) x 2 (N)
  f4* const cells = reinterpret_cast<f4*>(smem);                      // [GNBUF][A: GBM x 8 | B: GBN x 8]
  RowInfo* const rows = reinterpret_cast<RowInfo*>(cells + GNBUF * GCELLS);
  int* const tapA = reinterpret_cast<int*>(rows + GBM);
  int* const tapB = tapA + kMaxTaps;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gsrc), 0, p.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dense), 0, p.d_bytes, 0x00020000);

  // ---- geometry (as conv16_body) --------------------------------------------------------------------------------
  const int Cin8 = (p.C + 7) & ~7, K8 = (p.K + 7) & ~7;
  int M, N, Kdim, Cp, ntaps;
  int ph = 0, pw = 0, i0 = 0, j0 = 0, nti = 1, ntj = 1, dp0 = 0, dq0 = 0, Hc = 0, Wc = 0;
  if constexpr (MODE == MODE_FWD) {
    Cp = Cin8; ntaps = p.KH * p.KW;
    M = p.batch * p.OH * p.OW; N = p.K; Kdim = ntaps * Cp;
  } else {
    const int cls = blockIdx.y;
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
  }
  if (p.Nv > 0) N = p.Nv;
  const int tiles_n = (N + GBN - 1) / GBN;
  int bid = blockIdx.x;
  {      // XCD-aware tile order: each XCD a contiguous run of tiles (bijective for any grid size)
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * GBM, n0 = tn * GBN;
  if (m0 >= M) return;
  const int nk = (Kdim + BKH - 1) / BKH;

  // ---- tap tables and row infos -------------------------------------------------------------------------------------
  if (tid < kMaxTaps && tid < ntaps) {
    const int t = tid;
    if constexpr (MODE == MODE_DGRAD) {
      const int ti = t / ntj, tj = t - ti * ntj;
      tapA[t] = -(ti * p.OW + tj) * K8;
      tapB[t] = ((i0 + p.sh * ti) * p.KW + (j0 + p.sw * tj)) * p.C * K8;      // filter copy [tap][c][o8]
    } else {
      const int i = t / p.KW, j = t - i * p.KW;
      const int u = p.tap_classes ? tap_class_pos(i, j, p.KH, p.KW) : t;      // position in the K order
      tapA[u] = (i * p.W + j) * Cin8;
      tapB[u] = t * p.K * Cin8;                                               // filter copy [tap][o][c8]
    }
  }
  if (tid < GBM) {
    RowInfo ri; ri.base = 0; ri.mask_lo = 0; ri.mask_hi = 0; ri.out_off = 0;
    const int m = m0 + tid;
    if (m < M) {
      if constexpr (MODE == MODE_FWD) {
        const int t2 = div_fast(m, p.mg_ow, p.sh_ow), q = m - t2 * p.OW;
        const int b = div_fast(t2, p.mg_oh, p.sh_oh), pp = t2 - b * p.OH;
        const int y0 = pp * p.sh - p.pt, x0 = q * p.sw - p.pl;
        ri.base = ((b * p.H + y0) * p.W + x0) * Cin8;
        const unsigned long long mk = p.tap_classes ? tap_mask_classes(max(0, -y0), min(p.KH, p.H - y0), max(0, -x0), min(p.KW, p.W - x0), p.KH, p.KW)
                                                    : tap_mask(max(0, -y0), min(p.KH, p.H - y0), max(0, -x0), min(p.KW, p.W - x0), p.KW);
        ri.mask_lo = (unsigned)mk; ri.mask_hi = (unsigned)(mk >> 32);
      } else {
        const int w2 = m % Wc; const int t2 = m / Wc; const int h2 = t2 % Hc; const int b = t2 / Hc;
        const int y0 = h2 + dp0, x0 = w2 + dq0;
        ri.base = ((b * p.OH + y0) * p.OW + x0) * K8;
        ri.out_off = ((b * p.H + h2 * p.sh + ph) * p.W + (w2 * p.sw + pw)) * Cin8;
        const unsigned long long mk = tap_mask(max(0, y0 - p.OH + 1), min(nti, y0 + 1), max(0, x0 - p.OW + 1), min(ntj, x0 + 1), ntj);
        ri.mask_lo = (unsigned)mk; ri.mask_hi = (unsigned)(mk >> 32);
      }
    }
    rows[tid] = ri;
  }
  __syncthreads();

  // ---- loader: wave w owns the 8-row chunks w, w + 8, ... of both tiles; lane -> (row chunk * 8 + lane / 8, slot lane & 7) --
  // rows of one chunk set differ by multiples of 64, so (row >> 1) & 7 - and with it the lane's logical oct - is one value
  const int lrow8 = lane >> 3, slot = lane & 7;
  const int oct = slot ^ (((8 * wave + lrow8) >> 1) & 7);
  RowInfo myrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) myrow[i] = rows[8 * (wave + 8 * i) + lrow8];
  int nK[2];
  bool nOk[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int n = n0 + 8 * (wave + 8 * i) + lrow8; nK[i] = n * Cp; nOk[i] = n < N; }
  const int step_t = BKH / Cp, step_c = BKH - step_t * Cp;
  const bool chunked = p.korder != 0;          // ConvArgs::korder (conv_bf16_kernel.h): taps fastest inside a 64-channel chunk
  const int inc_t = chunked ? 1 : step_t, inc_c = chunked ? 0 : step_c, wrap_t = chunked ? -ntaps : 1, wrap_c = chunked ? BKH : -Cp;
  int run_kt = chunked ? 0 : div_fast(8 * oct, p.mg_cp, p.sh_cp), run_kc = 8 * oct - (chunked ? 0 : run_kt * Cp);

  unsigned offs[6];
  auto prep = [&](bool live) {
    // this lane's (tap, channel) of the K-step, then the next one's
    const bool kv = live && run_kt < ntaps && run_kc < Cp;
    const int t = kv ? run_kt : 0;
    const int aoff = tapA[t] + run_kc, boff = tapB[t] + run_kc;
    run_kt += inc_t; run_kc += inc_c;
    const bool wrap = chunked ? run_kt >= ntaps : run_kc >= Cp;
    run_kt += wrap ? wrap_t : 0; run_kc += wrap ? wrap_c : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      offs[i] = (kv && tap_ok(myrow[i], t)) ? (unsigned)(myrow[i].base + aoff) * 2u : kOob;
      asm volatile("" : "+v"(offs[i]));    // opaque: hipcc otherwise turns the select into a divergent branch around the load -
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      offs[4 + i] = (kv && nOk[i]) ? (unsigned)(boff + nK[i]) * 2u : kOob;
      asm volatile("" : "+v"(offs[4 + i]));      // - and a wave that skips a branch would break the counted vmcnt below
    }
  };
  auto fire = [&](int buf, auto ic) {
    constexpr int I = decltype(ic)::value;
    char* const bufp = reinterpret_cast<char*>(cells + buf * GCELLS);
    if constexpr (I < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_ptr_t)(bufp + 1024 * (wave + 8 * I)), 16, offs[I], 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (lds_ptr_t)(bufp + GBM * 128 + 1024 * (wave + 8 * (I - 4))), 16, offs[I], 0, 0, 0);
  };
  auto issue = [&](int buf, bool live) {
    prep(live);
    static_for<0, 6>([&](auto ic) { fire(buf, ic); });
  };

  // ---- main loop ----------------------------------------------------------------------------------------------------
  f32x16 acc[TA][TB];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int wr = wave >> 1, wc = wave & 1;
  const int wm0 = wr * 64, wn0 = wc * 64;
  const int lrow = lane & 31, lk = lane >> 5;
  // fragment cell of (tile row r, logical oct q): r * 8 + (q ^ ((r >> 1) & 7)); the row part per fragment is loop-invariant
  int ca[TA], xa[TA], cb[TB], xb[TB];
#pragma unroll
  for (int a = 0; a < TA; ++a) { const int r = wm0 + 32 * a + lrow; ca[a] = r * 8; xa[a] = (r >> 1) & 7; }
#pragma unroll
  for (int b = 0; b < TB; ++b) { const int r = wn0 + 32 * b + lrow; cb[b] = GBM * 8 + r * 8; xb[b] = (r >> 1) & 7; }

  issue(0, 0 < nk);
  issue(1, 1 < nk);
  if constexpr (VAR == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (its loop waits for K-step ks + 1 from compute phase ks - 1 on)
  else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int buf = 0;
  if constexpr (VAR == 0) {
  for (int ks = 0; ks < nk; ++ks) {
    const int nbuf = buf == 0 ? 2 : buf - 1;                 // (ks + 2) % 3: read last in iteration ks - 1, free since its barrier
    issue(nbuf, ks + 2 < nk);
    const f4* const cur = cells + buf * GCELLS;
    f4 av[2][TA], bv[2][TB];
    auto frag_read = [&](auto tc) {
      constexpr int T = decltype(tc)::value;
      const int kq = 2 * T + lk;
#pragma unroll
      for (int a = 0; a < TA; ++a) av[T & 1][a] = cur[ca[a] + (kq ^ xa[a])];
#pragma unroll
      for (int b = 0; b < TB; ++b) bv[T & 1][b] = cur[cb[b] + (kq ^ xb[b])];
    };
    frag_read(std::integral_constant<int, 0>{});
    static_for<0, 16>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      constexpr int T = I / 4, A = (I % 4) / TB, B = I % TB;
      acc[A][B] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, av[T & 1][A]), __builtin_bit_cast(bf8, bv[T & 1][B]), acc[A][B], 0, 0, 0);
      if constexpr (I % 4 == 0 && T + 1 < 4) frag_read(std::integral_constant<int, T + 1>{});
    });
    // K-step ks + 1 must have landed (this wave's share: all but the 6 instructions just issued), and every wave must be
    // done reading `buf` before the next iteration's DMA overwrites the buffer behind it
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");      // (lgkmcnt: this wave's fragment reads of `buf` have returned)
    __builtin_amdgcn_s_barrier();
    buf = buf == 2 ? 0 : buf + 1;
  }
  } else if constexpr (VAR == 2) {
  // As VAR 1 below (two wave groups one phase apart), but the six DMA pieces of K-step ks + 2 are issued BETWEEN the MFMAs of the
  // compute phase (where an LDS-DMA issue costs least, MI355X_MICROARCH.md cycle constants); the load phase only computes their
  // offsets and reads the fragments.  Group 0 waits for its pieces of K-step ks + 1 at the end of compute phase ks (vmcnt(6): the
  // six just issued stay in flight), group 1 - whose compute phase ends one barrier later - at the end of its load phase ks (vmcnt(0)).
  const int grp = wave >> 2;
  if (grp == 1) __builtin_amdgcn_s_barrier();
  for (int ks = 0; ks < nk; ++ks) {
    const int nbuf = buf == 0 ? 2 : buf - 1;
    const f4* const cur = cells + buf * GCELLS;
    f4 av[4][TA], bv[4][TB];
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      const int kq = 2 * T + lk;
#pragma unroll
      for (int a = 0; a < TA; ++a) av[T][a] = cur[ca[a] + (kq ^ xa[a])];
#pragma unroll
      for (int b = 0; b < TB; ++b) bv[T][b] = cur[cb[b] + (kq ^ xb[b])];
    }
    prep(ks + 2 < nk);
    if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    static_for<0, 16>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      constexpr int T = I / 4, A = (I % 4) / TB, B = I % TB;
      acc[A][B] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, av[T][A]), __builtin_bit_cast(bf8, bv[T][B]), acc[A][B], 0, 0, 0);
      constexpr int F = I == 1 ? 0 : I == 4 ? 1 : I == 6 ? 2 : I == 9 ? 3 : I == 11 ? 4 : I == 14 ? 5 : -1;
      if constexpr (F >= 0) {
        __builtin_amdgcn_sched_barrier(0);
        fire(nbuf, std::integral_constant<int, F>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    __builtin_amdgcn_s_setprio(0);
    if (grp == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    buf = buf == 2 ? 0 : buf + 1;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  } else {
  // Two wave groups one phase apart (waves w and w + 4 share a SIMD): a K-step is a LOAD phase (the K-step's 16 fragment reads, the
  // 6 DMA pieces of K-step ks + 2, the counted wait for K-step ks + 1) and a COMPUTE phase (16 MFMAs at raised priority), each
  // closed by a block barrier; group 1 runs one barrier behind, so one wave of every SIMD loads while the other multiplies.
  // RAW: a wave waits for its DMA of K-step ks + 1 at the end of its load phase ks, i.e. before a barrier every reader passes
  // before its load phase ks + 1.  WAR: the fragment reads of a buffer are retired (lgkmcnt(0)) before the barrier that closes
  // the load phase, and the DMA that refills it is issued two phases later by the same group, one phase later by the other.
  const int grp = wave >> 2;
  if (grp == 1) __builtin_amdgcn_s_barrier();
  for (int ks = 0; ks < nk; ++ks) {
    const int nbuf = buf == 0 ? 2 : buf - 1;
    const f4* const cur = cells + buf * GCELLS;
    f4 av[4][TA], bv[4][TB];
#pragma unroll
    for (int T = 0; T < 4; ++T) {
      const int kq = 2 * T + lk;
#pragma unroll
      for (int a = 0; a < TA; ++a) av[T][a] = cur[ca[a] + (kq ^ xa[a])];
#pragma unroll
      for (int b = 0; b < TB; ++b) bv[T][b] = cur[cb[b] + (kq ^ xb[b])];
    }
    issue(nbuf, ks + 2 < nk);
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    static_for<0, 16>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      constexpr int T = I / 4, A = (I % 4) / TB, B = I % TB;
      acc[A][B] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, av[T][A]), __builtin_bit_cast(bf8, bv[T][B]), acc[A][B], 0, 0, 0);
    });
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    buf = buf == 2 ? 0 : buf + 1;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the two masked tail K-steps: nothing of ours may still write LDS
  __syncthreads();

  // ---- epilogue (as conv16_body: bf16 activation at pitch round8, or float32 for a head layer; optional BatchNorm partials) --
  const bool to_bf16 = !p.out_f32;
  if (p.stats != nullptr) {
    int g = 0, blk;
    if constexpr (MODE == MODE_DGRAD) { blk = (int)blockIdx.y * p.stats_tpg + tm; } else { g = tm / p.stats_tpg; blk = tm - g * p.stats_tpg; }
    tile_stats_epilogue<GBM, GBN, WM, TA, TB>([&](int a, int b, int r) { return to_bf16 ? (float)(__bf16)acc[a][b][r] : acc[a][b][r]; }, reinterpret_cast<float*>(smem),
                                              p.stats + ((long long)g * p.stats_nblk + blk) * 2 * N, N, M, m0, n0, wm0, wn0, wr, lrow, lk, tid);
  }
  float* const outf = p.out;
  __bf16* const outh = reinterpret_cast<__bf16*>(p.out);
  const long long pitch = MODE == MODE_DGRAD ? Cin8 : K8;
  if (m0 + GBM <= M) {
    // full row tiles (every tile of the layers this kernel is planned for): straight-line stores, one base per 32 x 32
    // sub-tile; a ragged last column tile only masks lanes
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      const int n = n0 + wn0 + 32 * b + lrow;
      if (n < N) {
#pragma unroll
        for (int a = 0; a < TA; ++a) {
          if constexpr (MODE == MODE_DGRAD) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const long long off = (long long)rows[wm0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lk].out_off + n;
              if (to_bf16) outh[off] = (__bf16)acc[a][b][r];
              else outf[off] = acc[a][b][r];
            }
          } else {
            const long long o = (long long)(m0 + wm0 + 32 * a + 4 * lk) * pitch + n;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const long long off = o + (long long)((r & 3) + 8 * (r >> 2)) * pitch;
              if (to_bf16) outh[off] = (__bf16)acc[a][b][r];
              else outf[off] = acc[a][b][r];
            }
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < TB; ++b) {
    const int n = n0 + wn0 + 32 * b + lrow;
    if (n >= N) continue;
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m0 + row >= M) continue;
        const long long off = (MODE == MODE_DGRAD ? (long long)rows[row].out_off : (long long)(m0 + row) * pitch) + n;
        if (to_bf16) outh[off] = (__bf16)acc[a][b][r];
        else outf[off] = acc[a][b][r];
      }
  }
}

template <int MODE, int VAR = 0>
__global__ __launch_bounds__(GNT) void conv_glds_bf16(const ConvArgs p) {
  __shared__ __align__(16) char smem[conv16g_lds_bytes()];
  conv16g_body<MODE, VAR>(p, smem);
}

int launch_glds16(int mode, const Plan& pl, const ConvArgs& a, hipStream_t st);

}  // namespace acgconv
