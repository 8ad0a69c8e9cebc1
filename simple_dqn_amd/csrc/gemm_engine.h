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

// offset of step i for this lane's half: lanes 0-31 take lane i's value, lanes 32-63 lane (i+16)'s.
// v_readlane (VALU -> SGPR, no LDS round trip) x2 + v_cndmask; i is a compile-time constant after unrolling.
__device__ __forceinline__ int pick_half(int v, int i, bool hi) {
  const int lo = __builtin_amdgcn_readlane(v, i), up = __builtin_amdgcn_readlane(v, i + 16);
  return hi ? up : lo;
}
__device__ __forceinline__ int64_t pick_half(int64_t v, int i, bool hi) {
  const int l0 = __builtin_amdgcn_readlane((int)(v & 0xFFFFFFFF), i), l1 = __builtin_amdgcn_readlane((int)(v >> 32), i);
  const int u0 = __builtin_amdgcn_readlane((int)(v & 0xFFFFFFFF), i + 16), u1 = __builtin_amdgcn_readlane((int)(v >> 32), i + 16);
  const int64_t lo = ((int64_t)l1 << 32) | (uint32_t)l0, up = ((int64_t)u1 << 32) | (uint32_t)u0;
  return hi ? up : lo;
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

  typename P::Epi epi;
  if constexpr (NW == 1) P::epi_begin(a, m0, n0, lane, epi);

  float ra[16], rb[16];                       // next chunk's operands (4 x f4, or 16 scalars in MFMA layout)
  auto load_chunk = [&](int kc) {
    if constexpr (P::A_K) {
      const aoff_t c = P::a_col(a, z, kc + 4 * (lane & 7));
#pragma unroll
      for (int j = 0; j < 4; ++j) { const f4 v = P::a_load4(a, z, arow[j] + c); ra[4 * j] = v.x; ra[4 * j + 1] = v.y; ra[4 * j + 2] = v.z; ra[4 * j + 3] = v.w; }
    } else {
      // one im2col decomposition per lane per chunk (lane <-> k = kc + (l & 31)); step i fetches its
      // offset from lane i (+16 for the upper half-wave) with v_readlane
      const int kl = kc + (lane & 31);
      const aoff_t cv = P::a_col(a, z, kl < kend ? kl : kbeg);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bool hi = lane >= 32;
        const aoff_t c = pick_half(cv, i, hi);
        ra[i] = kc + i + (hi ? 16 : 0) < kend ? P::a_load(a, z, arow[0] + c) : 0.0f;
      }
    }
    if constexpr (P::B_K) {
      const int r = P::b_row(a, z, kc + 4 * (lane & 7));
#pragma unroll
      for (int j = 0; j < 4; ++j) { const f4 v = P::b_load4(a, z, r + bcol[j]); rb[4 * j] = v.x; rb[4 * j + 1] = v.y; rb[4 * j + 2] = v.z; rb[4 * j + 3] = v.w; }
    } else {
      const int kl = kc + (lane & 31);
      const int rv = P::b_row(a, z, kl < kend ? kl : kbeg);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bool hi = lane >= 32;
        const int r = pick_half(rv, i, hi);
        rb[i] = kc + i + (hi ? 16 : 0) < kend ? P::b_load(a, z, r + bcol[0]) : 0.0f;
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
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r];
    P::store16(a, z, ks, m0, n0, lane, M, N, v, epi);          // lane holds rows (r&3)+8(r>>2)+4(l>>5), column l&31
  }
}

template <class P, int NW>
inline hipError_t launch_gemm(const StepArgs& a, hipStream_t stream) {
  dim3 grid((P::M(a) + 31) / 32, (P::N(a) + 31) / 32, P::nbz(a));
  hipLaunchKernelGGL((gemm_kernel<P, NW>), grid, dim3(NW * 64), 0, stream, a);
  return hipGetLastError();
}

}  // namespace sdqn
