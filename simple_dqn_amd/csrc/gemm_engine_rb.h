// gemm_engine_rb.h — register-blocked tile routine of the engine for the THROUGHPUT regime (B >= 128: thousands of
// 32x32 tiles per launch, BASELINE.json configs[2]).
//
// gemm_tile (gemm_engine.h) feeds every v_mfma_f32_32x32x2_f32 with one fresh A and one fresh B value per lane: right
// when a launch is a handful of latency chains (B = 32), wrong when it is matrix-pipe bound — two thirds of the B = 256
// step were operand movement and instruction issue (tools/exp/README.md).  Here a wave owns RM x RN accumulators
// (a (32 RM) x (32 RN) block of C), so every operand fragment is reused RN (A) / RM (B) times from registers:
//   * RM + RN fragment loads feed RM * RN MFMAs per k-step (2x2: 1 operand value per MFMA instead of 2);
//   * operands contiguous along m / n are INTERLEAVED over the sub-tiles (lane i owns rows m0 + RM i + r, r < RM), so
//     ONE RM- (RN-) wide vector load per k-slot serves all sub-tiles: 16 x dwordx2/x4 per chunk instead of 32-64 dword
//     loads (any row -> sub-tile assignment is valid: the epilogue applies the same map);
//   * k-contiguous operands (im2col rows, dgrad weights) stay 4 x 16-byte loads of the lane's own row per sub-tile,
//     k-slot assignment as in gemm_tile (kslot());
//   * the chunk is software-pipelined in HALVES (16 k): while the 8 x RM x RN MFMAs of one half issue (>= 2048 cycles
//     for 2x2) the loads of the next half are in flight in a second register set — two ~32-register fragment sets
//     instead of one per chunk keep the kernel at <= ~168 VGPRs (3 waves per SIMD);
//   * NW waves of a workgroup still split K and are combined through LDS in fixed order (deterministic), one
//     sub-tile at a time so the LDS footprint stays NW panels.
// Same problem structs, same epilogues (P::store), same exact-fp32 arithmetic per output element: the k-order of every
// accumulator is identical to gemm_tile's, so B >= 128 results are bit-identical to the unblocked routine.
#pragma once

namespace sdqn {

template <class P, class = void> struct rb_m { static constexpr int value = 1; };
template <class P> struct rb_m<P, decltype((void)P::RBM)> { static constexpr int value = P::RBM; };
template <class P, class = void> struct rb_n { static constexpr int value = 1; };
template <class P> struct rb_n<P, decltype((void)P::RBN)> { static constexpr int value = P::RBN; };
template <class P, class = void> struct is_rb { static constexpr bool value = false; };
template <class P> struct is_rb<P, decltype((void)P::RBM)> { static constexpr bool value = true; };
// the same problem computed with RM x RN accumulators per wave
template <class P, int RM, int RN> struct RB : P { static constexpr int RBM = RM, RBN = RN; };

template <int N> struct vecf { typedef float type __attribute__((ext_vector_type(N))); };
template <> struct vecf<1> { typedef float type; };

template <int N>
__device__ __forceinline__ void ld_vec(const float* p, float* dst) {
  if constexpr (N == 1) dst[0] = *p;
  else {
    typedef float vt __attribute__((ext_vector_type(N)));
    const vt v = *reinterpret_cast<const vt*>(p);
#pragma unroll
    for (int i = 0; i < N; ++i) dst[i] = v[i];
  }
}

template <class P, int NW, int NT>
__device__ __forceinline__ void gemm_tile_rb(const StepArgs& a, int bx, int by, int bz, float* smem) {
  constexpr int RM = rb_m<P>::value, RN = rb_n<P>::value;
  typedef typename P::aoff_t aoff_t;
  typedef typename a_elem<P>::type AT; typedef typename b_elem<P>::type BT;
  static_assert(sizeof(AT) == 4 && sizeof(BT) == 4, "register-blocked routine: fp32 operands");
  static_assert(P::B_K || P::B_REG, "B is k-contiguous or a plain row-major matrix in every problem");
  // interleaved sub-tiles (one vector load per k-slot) where the operand is contiguous along m / n
  constexpr bool A_IL = !P::A_K && !P::A_U8 && RM > 1;
  constexpr bool B_IL = !P::B_K && RN > 1;

  SDQN_STAMP(0);
#ifdef SDQN_TIMING
  if (g_sdqn_dbg && threadIdx.x == 0)       // where this workgroup runs: HW_ID (wave / simd / cu / sh / se) and XCC_ID
    g_sdqn_dbg[((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8) + 7] =
        ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = bx * 32 * RM, n0 = by * 32 * RN;
  int z, ks, kbeg, kend;
  P::ksplit(a, bz, z, ks, kbeg, kend);
  if (NW * 64 < NT && wave >= NW) kend = kbeg;
  const int M = P::M(a), N = P::N(a);
  const int hb = lane >> 5, i = lane & 31;
  const bool hi = lane >= 32;
  auto mrow = [&](int r, int ii) { return A_IL ? m0 + RM * ii + r : m0 + 32 * r + ii; };       // row of sub-tile r, tile row ii
  auto ncol = [&](int c, int jj) { return B_IL ? n0 + RN * jj + c : n0 + 32 * c + jj; };

  // ---- per-lane operand geometry ----------------------------------------------------------------------------------
  aoff_t arow[RM]; int bcol[RN];
#pragma unroll
  for (int r = 0; r < RM; ++r) { const int m = mrow(r, i); arow[r] = P::a_row(a, z, m < M ? m : M - 1); }
#pragma unroll
  for (int c = 0; c < RN; ++c) { const int n = ncol(c, i); bcol[c] = P::b_col(a, z, n < N ? n : N - 1); }
  if constexpr (A_IL) {                      // vector base = row of sub-tile 0, kept inside the matrix
    const int m = m0 + RM * i; arow[0] = P::a_row(a, z, m + RM <= M ? m : M - RM);
  }
  if constexpr (B_IL) { const int n = n0 + RN * i; bcol[0] = P::b_col(a, z, n + RN <= N ? n : N - RN); }
  const AT* abase = P::a_ptr(a, z);
  const BT* bbase = P::b_ptr(a, z);
  (void)abase; (void)bbase;

  // ---- fragment loaders: half hf (0 / 1) of the 32-deep chunk at kc = MFMA steps t = 8 hf .. 8 hf + 7 ---------------
  // fa[r][tt], fb[c][tt]: value of sub-tile r (c) for step t = 8 hf + tt, i.e. logical k = kc + kslot(t, hb)
  auto load_a = [&](int kc, int hf, aoff_t cv, float (*fa)[8]) {
    if constexpr (P::A_K) {
#pragma unroll
      for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const f4 v = P::a_load4(a, z, arow[r] + P::a_col(a, z, kc + 8 * (2 * hf + jj) + 4 * hb));
          fa[r][4 * jj] = v.x; fa[r][4 * jj + 1] = v.y; fa[r][4 * jj + 2] = v.z; fa[r][4 * jj + 3] = v.w;
        }
    } else if constexpr (P::A_REG) {
      const bool full = kc + 32 <= kend;
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) {
        const int k = kc + kslot(8 * hf + tt, hb);
        const int kk = full ? k : (k < kend ? k : kend - 1);
        float v[RM];
        if constexpr (A_IL) ld_vec<RM>(abase + (size_t)arow[0] + (size_t)kk * P::A_LD, v);
        else {
#pragma unroll
          for (int r = 0; r < RM; ++r) v[r] = abase[(size_t)arow[r] + (size_t)kk * P::A_LD];
        }
#pragma unroll
        for (int r = 0; r < RM; ++r) fa[r][tt] = (full || k < kend) ? v[r] : 0.0f;
      }
    } else if constexpr (a_group4<P>::value) {                  // conv1 wgrad: u8 ring, 4 consecutive k per 16-byte load
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * hf + jj;
        const aoff_t c = pick_half(cv, 8 * j, hi);
        const bool ok = kc + 8 * j + 4 * hb < kend;
#pragma unroll
        for (int r = 0; r < RM; ++r) {
          const f4 v = P::a_load_group4(a, z, arow[r] + c);
          fa[r][4 * jj] = ok ? v.x : 0.0f; fa[r][4 * jj + 1] = ok ? v.y : 0.0f;
          fa[r][4 * jj + 2] = ok ? v.z : 0.0f; fa[r][4 * jj + 3] = ok ? v.w : 0.0f;
        }
      }
    } else {                                                    // conv2 / conv3 wgrad: gathered im2col rows, lanes along (r, s, c)
      static_assert(!P::A_U8, "u8 im2col rows go through the group-of-4 loader");
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) {
        const int t = 8 * hf + tt;
        const aoff_t c = pick_half(cv, kslot(t, 0), hi);
        const bool ok = kc + kslot(t, hb) < kend;
        float v[RM];
        if constexpr (A_IL) ld_vec<RM>(abase + (uint32_t)(arow[0] + c), v);
        else {
#pragma unroll
          for (int r = 0; r < RM; ++r) v[r] = abase[(uint32_t)(arow[r] + c)];
        }
#pragma unroll
        for (int r = 0; r < RM; ++r) fa[r][tt] = ok ? v[r] : 0.0f;
      }
    }
  };
  auto load_b = [&](int kc, int hf, float (*fb)[8]) {
    if constexpr (P::B_K) {
#pragma unroll
      for (int c = 0; c < RN; ++c)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const f4 v = P::b_load4(a, z, bcol[c] + P::b_row(a, z, kc + 8 * (2 * hf + jj) + 4 * hb));
          fb[c][4 * jj] = v.x; fb[c][4 * jj + 1] = v.y; fb[c][4 * jj + 2] = v.z; fb[c][4 * jj + 3] = v.w;
        }
    } else {
      const bool full = kc + 32 <= kend;
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) {
        const int k = kc + kslot(8 * hf + tt, hb);
        const int kk = full ? k : (k < kend ? k : kend - 1);
        float v[RN];
        if constexpr (B_IL) ld_vec<RN>(bbase + (size_t)bcol[0] + (size_t)kk * P::B_LD, v);
        else {
#pragma unroll
          for (int c = 0; c < RN; ++c) v[c] = bbase[(size_t)bcol[c] + (size_t)kk * P::B_LD];
        }
#pragma unroll
        for (int c = 0; c < RN; ++c) fb[c][tt] = (full || k < kend) ? v[c] : 0.0f;
      }
    }
  };
  // index decomposition of the gathered im2col operand: lane <-> k = kc + (l & 31), once per chunk (both halves)
  auto col_of_chunk = [&](int kc) -> aoff_t {
    if constexpr (!P::A_K && !P::A_REG) { const int kl = kc + i; return P::a_col(a, z, kl < kend ? kl : kbeg); }
    else return (aoff_t)0;
  };

  f32x16 acc[RM][RN];
#pragma unroll
  for (int r = 0; r < RM; ++r)
#pragma unroll
    for (int c = 0; c < RN; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[r][c][q] = 0.0f;

  auto mfma_half = [&](float (*fa)[8], float (*fb)[8]) {
#pragma unroll
    for (int tt = 0; tt < 8; ++tt)
#pragma unroll
      for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int c = 0; c < RN; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[r][tt], fb[c][tt], acc[r][c], 0, 0, 0);
  };

  // ---- main loop: two fragment sets, the next half's loads fly under the current half's MFMAs ----------------------
  float fa0[RM][8], fb0[RN][8], fa1[RM][8], fb1[RN][8];
  int kc = kbeg + (wave < NW ? wave : 0) * 32;
  if (kc < kend) {
    aoff_t cv = col_of_chunk(kc);
    load_a(kc, 0, cv, fa0); load_b(kc, 0, fb0);
    SDQN_STAMP(1);
    bool first = true; (void)first;
    while (true) {
      load_a(kc, 1, cv, fa1); load_b(kc, 1, fb1);
      mfma_half(fa0, fb0);
      const int kn = kc + NW * 32;
      const bool more = kn < kend;                               // wave-uniform
      if (more) { cv = col_of_chunk(kn); load_a(kn, 0, cv, fa0); load_b(kn, 0, fb0); }
      mfma_half(fa1, fb1);
#ifdef SDQN_TIMING
      if (first) { asm volatile("" :: "v"(acc[0][0][0])); SDQN_STAMP(2); first = false; }
#endif
      if (!more) break;
      kc = kn;
    }
  }
#ifdef SDQN_TIMING
  asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[RM - 1][RN - 1][15]));
  SDQN_STAMP(3);
#endif

  // ---- epilogue ------------------------------------------------------------------------------------------------------
  if constexpr (NW > 1) {
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
      for (int c = 0; c < RN; ++c) {
        if (wave < NW) {
          float* cw = smem + wave * PANEL;
#pragma unroll
          for (int q = 0; q < 16; ++q) cw[((q & 3) + 8 * (q >> 2) + 4 * hb) * 33 + i] = acc[r][c][q];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 1024; e += NT) {
          const int ml = e >> 5, nl = e & 31;
          float v = smem[ml * 33 + nl];
#pragma unroll
          for (int w = 1; w < NW; ++w) v += smem[w * PANEL + ml * 33 + nl];
          const int m = mrow(r, ml), n = ncol(c, nl);
          if (m < M && n < N) P::store(a, z, ks, m, n, v);
        }
        __syncthreads();                                         // panels are rewritten by the next sub-tile
      }
  } else {
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int ml = (q & 3) + 8 * (q >> 2) + 4 * hb;
        const int m = mrow(r, ml);
#pragma unroll
        for (int c = 0; c < RN; ++c) {
          const int n = ncol(c, i);
          if (m < M && n < N) P::store(a, z, ks, m, n, acc[r][c][q]);
        }
      }
  }
  SDQN_STAMP(4);
}


// ---- the same blocking for the fp16-mode forward / dgrad stages (packed-fp16 MFMA, both operands k-contiguous) -------------
// At B >= 128 those launches are operand-traffic bound (their time is flat in the number of K-split waves per tile,
// tools/sweep_nw.py): every 32 x 32 tile re-reads its 32 A rows and 32 B rows for 2 MFMAs per 32 k.  RM x RN accumulators
// per wave reuse each half8 fragment RN (A) / RM (B) times: 2x2 halves the bytes per MFMA.  One 16-k step = RM + RN 16-byte
// loads per lane + RM * RN v_mfma_f32_32x32x16_f16; the next step's fragments are loaded before the current step's MFMAs
// issue.  Sub-tiles are blocked (rows m0 + 32 r + i): the operands are gathered row-wise anyway.  K split over NW waves and
// combined in fixed order one sub-tile at a time, epilogue = P::store (half activations / deltas), as in gemm_tile_h.
template <class P, int NW, int NT>
__device__ __forceinline__ void gemm_tile_hb(const StepArgs& a, int bx, int by, int bz, float* smem) {
  constexpr int RM = rb_m<P>::value, RN = rb_n<P>::value;
  typedef typename P::aoff_t aoff_t;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = bx * 32 * RM, n0 = by * 32 * RN;
  int z, ks, kbeg, kend;
  P::ksplit(a, bz, z, ks, kbeg, kend);
  if (NW * 64 < NT && wave >= NW) kend = kbeg;
  const int M = P::M(a), N = P::N(a);
  const int i = lane & 31, hb = lane >> 5, h8 = hb * 8;
  aoff_t arow[RM]; int bcol[RN];
#pragma unroll
  for (int r = 0; r < RM; ++r) { const int m = m0 + 32 * r + i; arow[r] = P::a_row(a, z, m < M ? m : M - 1); }
#pragma unroll
  for (int c = 0; c < RN; ++c) { const int n = n0 + 32 * c + i; bcol[c] = P::b_col(a, z, n < N ? n : N - 1); }
  f32x16 acc[RM][RN];
#pragma unroll
  for (int r = 0; r < RM; ++r)
#pragma unroll
    for (int c = 0; c < RN; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[r][c][q] = 0.0f;
  auto load = [&](int k, half8* fa, half8* fb) {          // fragments of the 16-k step starting at k (all K are multiples of 32)
#pragma unroll
    for (int r = 0; r < RM; ++r) fa[r] = P::a_load8(a, z, arow[r] + P::a_col(a, z, k + h8));
#pragma unroll
    for (int c = 0; c < RN; ++c) fb[c] = P::b_load8(a, z, bcol[c] + P::b_row(a, z, k + h8));
  };
  auto mma = [&](const half8* fa, const half8* fb) {
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
      for (int c = 0; c < RN; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[r], fb[c], acc[r][c], 0, 0, 0);
  };
  half8 fa0[RM], fb0[RN], fa1[RM], fb1[RN];
  int kc = kbeg + (wave < NW ? wave : 0) * 32;
  if (kc < kend) {
    load(kc, fa0, fb0);
    while (true) {
      load(kc + 16, fa1, fb1);
      mma(fa0, fb0);
      const int kn = kc + NW * 32;
      const bool more = kn < kend;
      if (more) load(kn, fa0, fb0);
      mma(fa1, fb1);
      if (!more) break;
      kc = kn;
    }
  }
  if constexpr (NW > 1) {
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
      for (int c = 0; c < RN; ++c) {
        if (wave < NW) {
          float* cw = smem + wave * PANEL;
#pragma unroll
          for (int q = 0; q < 16; ++q) cw[((q & 3) + 8 * (q >> 2) + 4 * hb) * 33 + i] = acc[r][c][q];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 1024; e += NT) {
          const int ml = e >> 5, nl = e & 31;
          float v = smem[ml * 33 + nl];
#pragma unroll
          for (int w = 1; w < NW; ++w) v += smem[w * PANEL + ml * 33 + nl];
          const int m = m0 + 32 * r + ml, n = n0 + 32 * c + nl;
          if (m < M && n < N) P::store(a, z, ks, m, n, v);
        }
        __syncthreads();
      }
  } else {
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
      for (int c = 0; c < RN; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int m = m0 + 32 * r + (q & 3) + 8 * (q >> 2) + 4 * hb, n = n0 + 32 * c + i;
          if (m < M && n < N) P::store(a, z, ks, m, n, acc[r][c][q]);
        }
  }
}

}  // namespace sdqn
