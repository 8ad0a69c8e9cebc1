// gemm_engine.h — the one tile engine behind every GEMM-shaped stage (problems.h).
//
// CDNA4 mapping: 256-thread workgroup = 4 wave64s arranged WM x WN x WK.  Each wave owns one
// 32x32 fp32 accumulator tile (16 VGPR/AGPR per lane) fed by v_mfma_f32_32x32x2_f32
// (exact fp32, bitwise an fmaf chain — required for the 1e-4 Q-value parity).  The WK waves
// of a tile split every 32-deep K-tile between them and are summed through LDS in the
// epilogue.  K-tiles are staged global -> VGPR -> LDS (double-buffered, one barrier per
// tile): the loader performs the separable im2col gather A(m,k) = srcA[row(m) + col(k)], so
// the LDS image is a dense [k][m] / [k][n] panel (row pitch +1 dword: conflict-free for both
// the lane-along-k stores and the lane-along-m MFMA operand reads).
//
// MFMA operand maps (guide §3): lane l holds A[i = l&31][k = l>>5], B[k = l>>5][j = l&31];
// D register r of lane l is D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31].
#pragma once
#include <hip/hip_runtime.h>
#include "problems.h"

namespace sdqn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <class P>
__global__ void __launch_bounds__(256) gemm_kernel(const StepArgs a) {
  constexpr int WM = P::WM, WN = P::WN, WK = P::WK;
  static_assert(WM * WN * WK == 4, "4 waves per workgroup");
  constexpr int BM = 32 * WM, BN = 32 * WN, BK = 32;
  constexpr int LDA = BM + 1, LDB = BN + 1, LDC = BN + 1;
  constexpr int AE = BM * BK / 256, BE = BN * BK / 256;
  constexpr int SM_AB = 2 * BK * (LDA + LDB);
  constexpr int SM_C = WK * BM * LDC;
  constexpr int SM = SM_AB > SM_C ? SM_AB : SM_C;
  __shared__ float smem[SM];
  float* As = smem;
  float* Bs = smem + 2 * BK * LDA;
  typedef typename P::aoff_t aoff_t;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  int z, ks, kbeg, kend;
  P::ksplit(a, blockIdx.z, z, ks, kbeg, kend);
  const int M = P::M(a), N = P::N(a);

  // ---- loader geometry: which (m|n, k) elements of a K-tile this thread stages ------------
  // lane-along-k (A_K): k_l fixed = t & 31, rows t/32 + 8j.   lane-along-m: m_l fixed, k = t/BM + j*(256/BM)
  aoff_t arow[P::A_K ? AE : 1];
  int brow_or_col[P::B_K ? BE : 1];
  if constexpr (P::A_K) {
#pragma unroll
    for (int j = 0; j < AE; ++j) { int m = m0 + (t >> 5) + 8 * j; arow[j] = P::a_row(a, z, m < M ? m : M - 1); }
  } else {
    int m = m0 + (t % BM); arow[0] = P::a_row(a, z, m < M ? m : M - 1);
  }
  if constexpr (P::B_K) {
#pragma unroll
    for (int j = 0; j < BE; ++j) { int n = n0 + (t >> 5) + 8 * j; brow_or_col[j] = P::b_col(a, z, n < N ? n : N - 1); }
  } else {
    int n = n0 + (t % BN); brow_or_col[0] = P::b_col(a, z, n < N ? n : N - 1);
  }

  float ra[AE], rb[BE];
  auto load_tile = [&](int kt) {
    if constexpr (P::A_K) {
      const int k = kt + (t & 31); const bool ok = k < kend;
      const aoff_t c = ok ? P::a_col(a, z, k) : (aoff_t)0;
#pragma unroll
      for (int j = 0; j < AE; ++j) ra[j] = ok ? P::a_load(a, z, arow[j] + c) : 0.0f;
    } else {
#pragma unroll
      for (int j = 0; j < AE; ++j) {
        const int k = kt + t / BM + j * (256 / BM);
        ra[j] = k < kend ? P::a_load(a, z, arow[0] + P::a_col(a, z, k)) : 0.0f;
      }
    }
    if constexpr (P::B_K) {
      const int k = kt + (t & 31); const bool ok = k < kend;
      const int r = ok ? P::b_row(a, z, k) : 0;
#pragma unroll
      for (int j = 0; j < BE; ++j) rb[j] = ok ? P::b_load(a, z, r + brow_or_col[j]) : 0.0f;
    } else {
#pragma unroll
      for (int j = 0; j < BE; ++j) {
        const int k = kt + t / BN + j * (256 / BN);
        rb[j] = k < kend ? P::b_load(a, z, P::b_row(a, z, k) + brow_or_col[0]) : 0.0f;
      }
    }
  };
  auto store_tile = [&](int buf) {
    float* Ab = As + buf * BK * LDA;
    float* Bb = Bs + buf * BK * LDB;
    if constexpr (P::A_K) {
#pragma unroll
      for (int j = 0; j < AE; ++j) Ab[(t & 31) * LDA + (t >> 5) + 8 * j] = ra[j];
    } else {
#pragma unroll
      for (int j = 0; j < AE; ++j) Ab[(t / BM + j * (256 / BM)) * LDA + (t % BM)] = ra[j];
    }
    if constexpr (P::B_K) {
#pragma unroll
      for (int j = 0; j < BE; ++j) Bb[(t & 31) * LDB + (t >> 5) + 8 * j] = rb[j];
    } else {
#pragma unroll
      for (int j = 0; j < BE; ++j) Bb[(t / BN + j * (256 / BN)) * LDB + (t % BN)] = rb[j];
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

  const int T = (kend - kbeg + BK - 1) / BK;
  if (T > 0) {
    load_tile(kbeg);
    store_tile(0);
    __syncthreads();
    constexpr int KW = BK / WK;
    for (int it = 0; it < T; ++it) {
      const bool more = it + 1 < T;
      if (more) load_tile(kbeg + (it + 1) * BK);           // global gathers in flight under the MFMAs
      const float* Ab = As + (it & 1) * BK * LDA + wm * 32 + (lane & 31);
      const float* Bb = Bs + (it & 1) * BK * LDB + wn * 32 + (lane & 31);
#pragma unroll
      for (int kk = 0; kk < KW; kk += 2) {
        const int k2 = wk * KW + kk + (lane >> 5);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ab[k2 * LDA], Bb[k2 * LDB], acc, 0, 0, 0);
      }
      if (more) store_tile((it + 1) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue: WK partial tiles -> LDS -> summed in fixed order -> P::store (lanes along n) ----
  float* Cs = smem;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    Cs[(wk * BM + wm * 32 + row) * LDC + wn * 32 + (lane & 31)] = acc[r];
  }
  __syncthreads();
  for (int e = t; e < BM * BN; e += 256) {
    const int ml = e / BN, nl = e - ml * BN;
    float v = Cs[ml * LDC + nl];
#pragma unroll
    for (int w = 1; w < WK; ++w) v += Cs[(w * BM + ml) * LDC + nl];
    if (m0 + ml < M && n0 + nl < N) P::store(a, z, ks, m0 + ml, n0 + nl, v);
  }
}

template <class P>
inline hipError_t launch_gemm(const StepArgs& a, hipStream_t stream) {
  constexpr int BM = 32 * P::WM, BN = 32 * P::WN;
  dim3 grid((P::M(a) + BM - 1) / BM, (P::N(a) + BN - 1) / BN, P::nbz(a));
  hipLaunchKernelGGL(gemm_kernel<P>, grid, dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace sdqn
