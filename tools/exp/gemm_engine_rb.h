// gemm_engine_rb.h — register-blocked tile routine of the engine for the float16 mode's forward stages at B >= 128 when a launch is
// taken off the half block-tile routine (`bt:<id>` = -1: round 3's kernels, the same-box reference of tools/sweep_bt.py; sdqn_kernels_ext.hip).  A wave owns RM x RN accumulators (a (32 RM) x (32 RN) block of C), so
// every operand fragment is reused RN (A) / RM (B) times from registers; NW waves of a workgroup still split K and are combined through
// LDS in fixed order (deterministic), one sub-tile at a time.  Same problem structs, same epilogues (P::store).
// (The float32 form of this routine — gemm_tile_rb, option "rb:<id>" — was measured slower than the block-tile engine in every launch of
// the B = 256 step in rounds 3-4 and left the tree in round 5: tools/exp/experiments_r04.patch, tools/exp/README.md.)
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
