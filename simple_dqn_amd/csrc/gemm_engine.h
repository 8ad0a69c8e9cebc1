// gemm_engine.h — the one tile engine behind every GEMM-shaped stage (problems.h).
//
// Regime: at B=32 every stage is a few hundred MFLOP — far too small to fill 256 CUs with classic
// block-tiled GEMMs, and a per-K-tile __syncthreads pipeline is a serial chain of memory latencies
// (measured: ~10 us per stage for ~1 us of MFMA work).  So the engine is organised for latency:
//
//   * one workgroup = ONE 32x32 fp32 output tile (a single v_mfma_f32_32x32x2_f32 accumulator:
//     exact fp32, bitwise an fmaf chain — needed for the 1e-4 Q-value parity);
//   * its NW wave64s (1..16) split the reduction dimension: wave w owns the 32-deep K-chunks
//     w, w+NW, ... and runs them autonomously — NO workgroup barrier in the main loop, every wave
//     has its next chunk's global loads in flight while it issues the current chunk's 16 MFMAs;
//   * operands whose memory-contiguous dimension is m/n (weights [k][n], deltas [m][f], the wgrad
//     im2col rows) are loaded straight into the MFMA operand layout (lane l <- [k + (l>>5)][x0 + (l&31)],
//     two coalesced 128-B rows per instruction, no LDS);
//   * operands contiguous along k (im2col patches, activations, dgrad weights) are fetched with
//     16 B/lane vector loads (4 B/lane for the u8 ring) and transposed through a WAVE-PRIVATE LDS
//     panel [k][x] (pitch 33: conflict-free for both the k-major stores and the x-major reads);
//   * the NW partial tiles are summed through LDS in a fixed order (deterministic), then P::store.
//
// MFMA operand maps (guide §3): lane l holds A[i = l&31][k = l>>5], B[k = l>>5][j = l&31];
// D register r of lane l is D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31].
#pragma once
#include <hip/hip_runtime.h>
#include "problems.h"

namespace sdqn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PANEL = 32 * 33;      // one [32 k][32 x] fp32 panel, pitch 33

__device__ __forceinline__ void wave_lds_sync() {
  // wave-private LDS hand-off between lanes of ONE wave: LDS ops of a wave execute in order, so only
  // the compiler has to be kept from reordering the stores past the loads
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <class P, int NW>
__global__ void __launch_bounds__(NW * 64) gemm_kernel(const StepArgs a) {
  constexpr int NPAN = (P::A_K ? 1 : 0) + (P::B_K ? 1 : 0);
  constexpr int WAVE_LDS = (NPAN > 0 ? NPAN : 1) * PANEL;
  __shared__ float smem[NW * WAVE_LDS];
  typedef typename P::aoff_t aoff_t;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // provably wave-uniform -> SGPR index math
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  int z, ks, kbeg, kend;
  P::ksplit(a, blockIdx.z, z, ks, kbeg, kend);
  const int M = P::M(a), N = P::N(a);
  float* pa = smem + wave * WAVE_LDS;
  float* pb = pa + (P::A_K ? PANEL : 0);

  // ---- per-lane operand geometry ---------------------------------------------------------------
  // k-contiguous operand: lane -> (k4 = l & 7, x = (l >> 3) + 8j), one 4-vector per j < 4
  // x-contiguous operand: lane -> (x = l & 31, k = i + 16*(l >> 5)), one scalar per i < 16 (MFMA layout).
  //   Both candidate k's of step i are wave-uniform, so the (n,p,q) im2col decomposition of the
  //   reduction index runs on the scalar unit and the lanes only select (v_cndmask) + add.
  aoff_t arow[P::A_K ? 4 : 1];
  int bcol[P::B_K ? 4 : 1];
  if constexpr (P::A_K) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int m = m0 + (lane >> 3) + 8 * j; arow[j] = P::a_row(a, z, m < M ? m : M - 1); }
  } else {
    const int m = m0 + (lane & 31); arow[0] = P::a_row(a, z, m < M ? m : M - 1);
  }
  if constexpr (P::B_K) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int n = n0 + (lane >> 3) + 8 * j; bcol[j] = P::b_col(a, z, n < N ? n : N - 1); }
  } else {
    const int n = n0 + (lane & 31); bcol[0] = P::b_col(a, z, n < N ? n : N - 1);
  }

  float ra[16], rb[16];                       // next chunk's operands (4 x f4, or 16 scalars in MFMA layout)
  auto load_chunk = [&](int kc) {
    if constexpr (P::A_K) {
      const aoff_t c = P::a_col(a, z, kc + 4 * (lane & 7));
#pragma unroll
      for (int j = 0; j < 4; ++j) { const f4 v = P::a_load4(a, z, arow[j] + c); ra[4 * j] = v.x; ra[4 * j + 1] = v.y; ra[4 * j + 2] = v.z; ra[4 * j + 3] = v.w; }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k0 = kc + i, k1 = kc + 16 + i;
        const bool ok0 = k0 < kend, ok1 = k1 < kend;
        const aoff_t c0 = P::a_col(a, z, ok0 ? k0 : kbeg), c1 = P::a_col(a, z, ok1 ? k1 : kbeg);
        const bool hi = lane >= 32;
        ra[i] = (hi ? ok1 : ok0) ? P::a_load(a, z, arow[0] + (hi ? c1 : c0)) : 0.0f;
      }
    }
    if constexpr (P::B_K) {
      const int r = P::b_row(a, z, kc + 4 * (lane & 7));
#pragma unroll
      for (int j = 0; j < 4; ++j) { const f4 v = P::b_load4(a, z, r + bcol[j]); rb[4 * j] = v.x; rb[4 * j + 1] = v.y; rb[4 * j + 2] = v.z; rb[4 * j + 3] = v.w; }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k0 = kc + i, k1 = kc + 16 + i;
        const bool ok0 = k0 < kend, ok1 = k1 < kend;
        const int r0 = P::b_row(a, z, ok0 ? k0 : kbeg), r1 = P::b_row(a, z, ok1 ? k1 : kbeg);
        const bool hi = lane >= 32;
        rb[i] = (hi ? ok1 : ok0) ? P::b_load(a, z, (hi ? r1 : r0) + bcol[0]) : 0.0f;
      }
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

  int kc = kbeg + wave * 32;                              // wave-uniform
  if (kc < kend) load_chunk(kc);
  while (kc < kend) {
    // ---- move the fetched chunk to its MFMA operands (through the wave-private panels if k-contiguous)
    float fa[16], fb[16];
    if constexpr (P::A_K) {
      float* d = pa + (4 * (lane & 7)) * 33 + (lane >> 3);
#pragma unroll
      for (int j = 0; j < 4; ++j) { d[8 * j] = ra[4 * j]; d[33 + 8 * j] = ra[4 * j + 1]; d[66 + 8 * j] = ra[4 * j + 2]; d[99 + 8 * j] = ra[4 * j + 3]; }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) fa[i] = ra[i];
    }
    if constexpr (P::B_K) {
      float* d = pb + (4 * (lane & 7)) * 33 + (lane >> 3);
#pragma unroll
      for (int j = 0; j < 4; ++j) { d[8 * j] = rb[4 * j]; d[33 + 8 * j] = rb[4 * j + 1]; d[66 + 8 * j] = rb[4 * j + 2]; d[99 + 8 * j] = rb[4 * j + 3]; }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) fb[i] = rb[i];
    }
    const int knext = kc + NW * 32;
    if (knext < kend) load_chunk(knext);                 // next chunk's global loads fly under the MFMAs
    if constexpr (P::A_K || P::B_K) wave_lds_sync();
    if constexpr (P::A_K) {
      const float* s = pa + (lane >> 5) * (16 * 33) + (lane & 31);
#pragma unroll
      for (int i = 0; i < 16; ++i) fa[i] = s[33 * i];
    }
    if constexpr (P::B_K) {
      const float* s = pb + (lane >> 5) * (16 * 33) + (lane & 31);
#pragma unroll
      for (int i = 0; i < 16; ++i) fb[i] = s[33 * i];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[i], acc, 0, 0, 0);
    if constexpr (P::A_K || P::B_K) wave_lds_sync();      // panel reads done before the next chunk's stores
    kc = knext;
  }

  // ---- epilogue: NW partial tiles -> LDS -> summed in fixed order -> P::store (lanes along n) ----
  if constexpr (NW > 1) {
    float* cw = smem + wave * WAVE_LDS;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      cw[row * 33 + (lane & 31)] = acc[r];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += NW * 64) {
      const int ml = e >> 5, nl = e & 31;
      float v = smem[ml * 33 + nl];
#pragma unroll
      for (int w = 1; w < NW; ++w) v += smem[w * WAVE_LDS + ml * 33 + nl];
      if (m0 + ml < M && n0 + nl < N) P::store(a, z, ks, m0 + ml, n0 + nl, v);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ml = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), nl = lane & 31;
      if (m0 + ml < M && n0 + nl < N) P::store(a, z, ks, m0 + ml, n0 + nl, acc[r]);
    }
  }
}

template <class P, int NW>
inline hipError_t launch_gemm(const StepArgs& a, hipStream_t stream) {
  dim3 grid((P::M(a) + 31) / 32, (P::N(a) + 31) / 32, P::nbz(a));
  hipLaunchKernelGGL((gemm_kernel<P, NW>), grid, dim3(NW * 64), 0, stream, a);
  return hipGetLastError();
}

}  // namespace sdqn
