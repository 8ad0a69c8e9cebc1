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

  // ---- loader geometry: every thread stages BM/32 (A) + BN/32 (B) 4-vectors of a K-tile ------------
  // The vector always runs along the operand's memory-contiguous dimension (16 B/lane global loads,
  // 4 B/lane for the u8 ring), and so do the lanes:
  //   contiguous k (A_K/B_K):  k4 = t & 7 -> k = kt + 4*k4 .. +3,  x = (t >> 3) + 32*j
  //   contiguous x (m or n):   x4 = t % (BX/4) -> x = 4*x4 .. +3,   k = kt + t / (BX/4) + (1024/BX)*j
  constexpr int AJ = BM / 32, BJ = BN / 32;
  aoff_t arow[P::A_K ? AJ : 1];
  int bcol[P::B_K ? BJ : 1];
  if constexpr (P::A_K) {
#pragma unroll
    for (int j = 0; j < AJ; ++j) { const int m = m0 + (t >> 3) + 32 * j; arow[j] = P::a_row(a, z, m < M ? m : M - 1); }
  } else {
    const int m = m0 + 4 * (t % (BM / 4)); arow[0] = P::a_row(a, z, m + 3 < M ? m : M - 4);
  }
  if constexpr (P::B_K) {
#pragma unroll
    for (int j = 0; j < BJ; ++j) { const int n = n0 + (t >> 3) + 32 * j; bcol[j] = P::b_col(a, z, n < N ? n : N - 1); }
  } else {
    const int n = n0 + 4 * (t % (BN / 4)); bcol[0] = P::b_col(a, z, n + 3 < N ? n : N - 4);
  }

  f4 ra[AJ], rb[BJ];
  const f4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
  auto load_tile = [&](int kt) {
    if constexpr (P::A_K) {
      const int k = kt + 4 * (t & 7); const bool ok = k < kend;
      const aoff_t c = ok ? P::a_col(a, z, k) : (aoff_t)0;
#pragma unroll
      for (int j = 0; j < AJ; ++j) ra[j] = ok ? P::a_load4(a, z, arow[j] + c) : zero4;
    } else {
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const int k = kt + t / (BM / 4) + j * (1024 / BM);
        ra[j] = k < kend ? P::a_load4(a, z, arow[0] + P::a_col(a, z, k)) : zero4;
      }
    }
    if constexpr (P::B_K) {
      const int k = kt + 4 * (t & 7); const bool ok = k < kend;
      const int r = ok ? P::b_row(a, z, k) : 0;
#pragma unroll
      for (int j = 0; j < BJ; ++j) rb[j] = ok ? P::b_load4(a, z, r + bcol[j]) : zero4;
    } else {
#pragma unroll
      for (int j = 0; j < BJ; ++j) {
        const int k = kt + t / (BN / 4) + j * (1024 / BN);
        rb[j] = k < kend ? P::b_load4(a, z, P::b_row(a, z, k) + bcol[0]) : zero4;
      }
    }
  };
  auto store_tile = [&](int buf) {
    float* Ab = As + buf * BK * LDA;
    float* Bb = Bs + buf * BK * LDB;
    if constexpr (P::A_K) {
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        float* d = Ab + (4 * (t & 7)) * LDA + (t >> 3) + 32 * j;
        d[0] = ra[j].x; d[LDA] = ra[j].y; d[2 * LDA] = ra[j].z; d[3 * LDA] = ra[j].w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        float* d = Ab + (t / (BM / 4) + j * (1024 / BM)) * LDA + 4 * (t % (BM / 4));
        d[0] = ra[j].x; d[1] = ra[j].y; d[2] = ra[j].z; d[3] = ra[j].w;
      }
    }
    if constexpr (P::B_K) {
#pragma unroll
      for (int j = 0; j < BJ; ++j) {
        float* d = Bb + (4 * (t & 7)) * LDB + (t >> 3) + 32 * j;
        d[0] = rb[j].x; d[LDB] = rb[j].y; d[2 * LDB] = rb[j].z; d[3 * LDB] = rb[j].w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < BJ; ++j) {
        float* d = Bb + (t / (BN / 4) + j * (1024 / BN)) * LDB + 4 * (t % (BN / 4));
        d[0] = rb[j].x; d[1] = rb[j].y; d[2] = rb[j].z; d[3] = rb[j].w;
      }
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
