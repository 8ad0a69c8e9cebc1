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
//     im2col rows) are loaded straight into the MFMA operand layout (lane l <- [kslot][x0 + (l&31)],
//     two coalesced 128-B rows per instruction, no LDS);
//   * operands contiguous along k (im2col patches, activations, dgrad weights) need NO staging either: the MFMA's
//     k-slot <-> logical-k assignment is free as long as A and B agree, and with kslot(t, h) = 8*(t>>2) + 4h + (t&3)
//     a lane's 16 operand values are four 16 B loads of its own row (4 B for the u8 ring);
//   * the NW partial tiles are summed through LDS in a fixed order (deterministic), then P::store.
//
// MFMA operand maps (guide §3): lane l holds A[i = l&31][k = l>>5], B[k = l>>5][j = l&31];
// D register r of lane l is D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31].
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>
#include "problems.h"
#include "launch.h"

namespace sdqn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// element type of an x-contiguous operand (float unless the problem says otherwise: the fp16-mode wgrads read half)
template <class P, class = void> struct a_elem { typedef float type; };
template <class P> struct a_elem<P, decltype((void)sizeof(typename P::AT))> { typedef typename P::AT type; };
template <class P, class = void> struct b_elem { typedef float type; };
template <class P> struct b_elem<P, decltype((void)sizeof(typename P::BT))> { typedef typename P::BT type; };
// k-contiguous operands: direct (4 x 16 B loads of the lane's own row: 32 cache lines per instruction, no LDS) or
// staged (lanes (k4, x): 8 rows x 128 B per instruction, transposed through a wave-private LDS panel).  Direct wins
// when the rows of a tile are close together and the launch is latency-bound (conv fwd/dgrad at B = 32); staged wins
// for the fc4 GEMMs (rows 2-12 KB apart) and in the throughput regime (B >= 128).  Measured: profiles/README.md.
template <class P, class = void> struct stages_lds { static constexpr bool value = false; };
template <class P> struct stages_lds<P, decltype((void)P::STAGE_LDS)> { static constexpr bool value = P::STAGE_LDS; };
template <class P> struct Staged : P { static constexpr bool STAGE_LDS = true; };      // same problem, staged operands
// A_GROUP4: the irregular (im2col-row) A operand is affine inside every aligned group of 4 consecutive k
// (conv1: 20 x 20 output positions, x fastest, stride 4 bytes in the u8 frame; 20 and 400 are multiples of 4), so a
// lane needs 4 gathered offsets per chunk instead of 16 and fetches its 4 bytes with ONE unaligned 16-byte load
template <class P, class = void> struct a_group4 { static constexpr bool value = false; };
template <class P> struct a_group4<P, decltype((void)P::A_GROUP4)> { static constexpr bool value = P::A_GROUP4; };
template <class P, class = void> struct uses_f16_wgrad { static constexpr bool value = false; };
template <class P> struct uses_f16_wgrad<P, decltype((void)P::F16_WGRAD)> { static constexpr bool value = P::F16_WGRAD; };
// SIGNALS: thread 0 calls P::signal(a, bx, by, bz) right after the barrier that follows the main loops of all waves of the tile
// (every operand load of the tile has returned by then) — round 3's write-after-read hand-off between fc4_dgrad and fc4_wgrad
template <class P, class = void> struct signals { static constexpr bool value = false; };
template <class P> struct signals<P, decltype((void)P::SIGNALS)> { static constexpr bool value = P::SIGNALS; };
// PRELOAD: P::preload(a) runs first thing in the kernel and touches every argument field the problem will read, so that hipcc issues
// ALL their scalar loads in one batch behind one wait.  Left alone it fetches each field of the 464-byte by-value StepArgs where it is
// first used: four or five DEPENDENT round trips to a kernel-argument segment that is cold at every launch (round 3, sdqn_kernels_r3.hip)
template <class P, class = void> struct has_preload { static constexpr bool value = false; };
template <class P> struct has_preload<P, decltype((void)P::PRELOAD)> { static constexpr bool value = P::PRELOAD; };
template <class P, class = void> struct has_preload_multi { static constexpr bool value = false; };
template <class P> struct has_preload_multi<P, decltype((void)P::PRELOAD_MULTI)> { static constexpr bool value = P::PRELOAD_MULTI; };
template <class P, class = void> struct uses_f16_mfma { static constexpr bool value = false; };
template <class P> struct uses_f16_mfma<P, decltype((void)P::F16_MFMA)> { static constexpr bool value = P::F16_MFMA; };

constexpr int PANEL = 32 * 33;      // one [32 k][32 x] fp32 panel, pitch 33

__device__ __forceinline__ void wave_lds_sync() {
  // wave-private LDS hand-off between lanes of ONE wave: LDS ops of a wave execute in order, so only
  // the compiler has to be kept from reordering the stores past the loads
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// value held by lane s (lanes 0-31) / lane s+4 (lanes 32-63): the k-slot of the upper half-wave is 4 further.
// v_readlane (VALU -> SGPR, no LDS round trip) x2 + v_cndmask; s is a compile-time constant after unrolling.
__device__ __forceinline__ int pick_half(int v, int s, bool hi) {
  const int lo = __builtin_amdgcn_readlane(v, s), up = __builtin_amdgcn_readlane(v, s + 4);
  return hi ? up : lo;
}
__device__ __forceinline__ int64_t pick_half(int64_t v, int s, bool hi) {
  const int i = s;
  const int l0 = __builtin_amdgcn_readlane((int)(v & 0xFFFFFFFF), i), l1 = __builtin_amdgcn_readlane((int)(v >> 32), i);
  const int u0 = __builtin_amdgcn_readlane((int)(v & 0xFFFFFFFF), i + 4), u1 = __builtin_amdgcn_readlane((int)(v >> 32), i + 4);
  const int64_t lo = ((int64_t)l1 << 32) | (uint32_t)l0, up = ((int64_t)u1 << 32) | (uint32_t)u0;
  return hi ? up : lo;
}

#ifdef SDQN_TIMING
// phase stamps (s_memtime) of wave 0 of every workgroup: dbg[(block*8 + phase)]; NOT in the product build
__device__ unsigned long long* g_sdqn_dbg = nullptr;
#define SDQN_STAMP(ph) do { if (g_sdqn_dbg && threadIdx.x == 0) g_sdqn_dbg[((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8) + (ph)] = clock64(); } while (0)
// per-WAVE stamps of the latency engine's tile routine (round 5: which operand lands when, tools/landing_hist.py):
// wdbg[((block * 16 + wave) * 4 + phase)], phase 0 = before the first operand load, 1 = all loads issued, 2 = A landed, 3 = B landed;
// + the XCC the workgroup ran on in wdbg[blocks * 64 + block]
__device__ unsigned long long* g_sdqn_wdbg = nullptr;
__device__ unsigned g_sdqn_wdbg_blocks = 0;
// (the stamps are TAKEN into registers where they belong and WRITTEN at the end of the tile: a store, or the load of the buffer pointer,
//  next to the operand loads would sit in the same in-order vmcnt queue and turn every stamp into "everything before me has landed")
#define SDQN_WSTAMP_DECL unsigned long long wst_[4] = {0, 0, 0, 0}
#define SDQN_WSTAMP(ph) do { if (wst_[ph] == 0) wst_[ph] = clock64(); } while (0)
#define SDQN_WSTAMP_FLUSH do { unsigned long long* wd_ = g_sdqn_wdbg; if (wd_ && (threadIdx.x & 63) == 0) { \
  const size_t b_ = (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)); const unsigned nb_ = g_sdqn_wdbg_blocks; \
  if (b_ < nb_ && (threadIdx.x >> 6) < 16) { \
    for (int q_ = 0; q_ < 4; ++q_) wd_[(b_ * 16 + (threadIdx.x >> 6)) * 4 + q_] = wst_[q_]; \
    if (threadIdx.x == 0) { unsigned x_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x_)); wd_[(size_t)nb_ * 64 + b_] = x_; } } } } while (0)
#else
#define SDQN_STAMP(ph) do {} while (0)
#define SDQN_WSTAMP_DECL do {} while (0)
#define SDQN_WSTAMP(ph) do {} while (0)
#define SDQN_WSTAMP_FLUSH do {} while (0)
#endif

// LDS floats one workgroup of NW waves needs (only the fixed-order combine of the NW partial tiles uses LDS)
template <class P, int NW>
constexpr int tile_lds() {
  return NW * PANEL * (stages_lds<P>::value && (P::A_K || P::B_K) ? ((P::A_K ? 1 : 0) + (P::B_K ? 1 : 0)) : 1);
}

// k-slot assignment shared by BOTH operands of every problem: MFMA step t (0..15) of a 32-deep chunk, half-wave h,
// consumes logical k = kc + kslot(t, h).  Any bijection works as long as A and B agree; this one makes a k-contiguous
// operand exactly four 16-byte loads of the lane's own row (j = t >> 2 selects the load, e = t & 3 the component):
__device__ __forceinline__ constexpr int kslot(int t, int h) { return 8 * (t >> 2) + 4 * h + (t & 3); }

template <class T>
__device__ __forceinline__ T ld_byte_off(const T* base, uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

// One 32x32 output tile (bx, by, bz) of problem P computed by the first NW waves of a workgroup of NT threads
// (waves >= NW idle through the epilogue barrier).  NW == 1 inside a wider workgroup is handled by the caller
// (one tile per wave, no barrier): see gemm_multi_kernel.
template <class P, int NW, int NT>
__device__ __forceinline__ void gemm_tile(const StepArgs& a, int bx, int by, int bz, float* smem) {
  constexpr bool STG = stages_lds<P>::value;
  constexpr bool STG_A = STG && P::A_K, STG_B = STG && P::B_K;
  constexpr int WAVE_LDS = PANEL * ((STG_A || STG_B) ? ((STG_A ? 1 : 0) + (STG_B ? 1 : 0)) : 1);
  typedef typename P::aoff_t aoff_t;

  SDQN_STAMP(0);
  SDQN_WSTAMP_DECL;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // provably wave-uniform -> SGPR index math
  const int m0 = bx * 32, n0 = by * 32;
  int z, ks, kbeg, kend;
  P::ksplit(a, bz, z, ks, kbeg, kend);
  if (NW * 64 < NT && wave >= NW) kend = kbeg;          // surplus waves of a wider workgroup: no chunks
  const int M = P::M(a), N = P::N(a);
  const int hb = lane >> 5;                             // half-wave: which k-slots this lane feeds
  const bool hi = lane >= 32;

  // ---- per-lane operand geometry: lane l always owns row m0 + (l & 31) of A and column n0 + (l & 31) of B ----------
  const int mrow = m0 + (lane & 31), ncol = n0 + (lane & 31);
  const aoff_t arow = P::a_row(a, z, mrow < M ? mrow : M - 1);
  const int bcol = P::b_col(a, z, ncol < N ? ncol : N - 1);
  typedef typename a_elem<P>::type AT; typedef typename b_elem<P>::type BT;
  const AT* abase = P::a_ptr(a, z);
  const BT* bbase = P::b_ptr(a, z);
  // plain row-major [k][x] operands: per-lane pointer fixed for the whole tile (row 0 / row 4 for the two half-waves)
  const AT* areg = nullptr; const BT* breg = nullptr;
  if constexpr (!P::A_K && P::A_REG) areg = abase + (uint32_t)arow + (hi ? 4 * P::A_LD : 0);
  if constexpr (!P::B_K && P::B_REG) breg = bbase + (uint32_t)bcol + (hi ? 4 * P::B_LD : 0);
  // the same per-lane position as a 32-bit BYTE offset from the (uniform) base pointer: `global_load v, v_off, s[base]`
  // instead of three VALU instructions of 64-bit pointer arithmetic per load (all operand buffers are < 4 GB)
  uint32_t aoffb = 0, boffb = 0;
  if constexpr (!P::A_K && P::A_REG) aoffb = (uint32_t)sizeof(AT) * ((uint32_t)arow + (hi ? 4u * P::A_LD : 0u));
  if constexpr (!P::B_K && P::B_REG) boffb = (uint32_t)sizeof(BT) * ((uint32_t)bcol + (hi ? 4u * P::B_LD : 0u));
  (void)abase; (void)bbase; (void)areg; (void)breg; (void)aoffb; (void)boffb;

  // staged operands: lane -> (k4 = l & 7, rows (l >> 3) + 8j) for the loads, wave-private panels [k][x] (pitch 33)
  float* pan_a = smem + (wave < NW ? wave : 0) * WAVE_LDS;
  float* pan_b = pan_a + (STG_A ? PANEL : 0);
  aoff_t srow[STG_A ? 4 : 1]; int scol[STG_B ? 4 : 1];
  if constexpr (STG_A) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int m = m0 + (lane >> 3) + 8 * j; srow[j] = P::a_row(a, z, m < M ? m : M - 1); }
  }
  if constexpr (STG_B) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int n = n0 + (lane >> 3) + 8 * j; scol[j] = P::b_col(a, z, n < N ? n : N - 1); }
  }
  (void)pan_a; (void)pan_b; (void)srow; (void)scol;

  typename P::Epi epi;
  if constexpr (NW == 1) P::epi_begin(a, m0, n0, lane, epi);

  // One chunk's 16 MFMA operand values of this lane, straight from global memory into registers (no LDS staging):
  //   k-contiguous operand : 4 x 16-byte loads (4 bytes for the u8 ring) of the lane's own row at k = kc + 8j + 4h
  //   x-contiguous, regular: 16 dword loads off the per-lane pointer at immediate row offsets kslot(t, 0) * LD
  //   x-contiguous, im2col : one index decomposition per lane per chunk (lane <-> k = kc + (l & 31)), distributed with
  //                          v_readlane; offsets of out-of-range k are clamped in bounds, loads are UNconditional and the
  //                          value is selected afterwards (a conditional load costs a branch + vmcnt(0) per element)
  auto load_a = [&](int kc, float* dst) {
    if constexpr (P::A_K) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f4 v = P::a_load4(a, z, arow + P::a_col(a, z, kc + 8 * j + 4 * hb));
        dst[4 * j] = v.x; dst[4 * j + 1] = v.y; dst[4 * j + 2] = v.z; dst[4 * j + 3] = v.w;
      }
    } else if constexpr (P::A_REG) {
      if (kc + 32 <= kend) {
        if constexpr (sizeof(AT) * P::A_LD >= 2048) {          // every row offset beyond the 12-bit immediate: 32-bit offsets
          const uint32_t kcb = (uint32_t)sizeof(AT) * (uint32_t)kc * (uint32_t)P::A_LD;          // wave-uniform
#pragma unroll
          for (int t = 0; t < 16; ++t) dst[t] = (float)ld_byte_off(abase, aoffb + kcb + (uint32_t)(sizeof(AT) * kslot(t, 0) * P::A_LD));
        } else {
          const AT* p = areg + (size_t)kc * P::A_LD;
#pragma unroll
          for (int t = 0; t < 16; ++t) dst[t] = (float)p[(size_t)kslot(t, 0) * P::A_LD];
        }
      } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) { const int k = kc + kslot(t, hb); dst[t] = k < kend ? (float)areg[(size_t)(kc + kslot(t, 0)) * P::A_LD] : 0.0f; }
      }
    } else if constexpr (a_group4<P>::value) {
      const int kl = kc + (lane & 31);
      const aoff_t cv = P::a_col(a, z, kl < kend ? kl : kbeg);
#pragma unroll
      for (int j = 0; j < 4; ++j) {                                         // k = kc + 8j + 4h + e, e = 0..3
        const aoff_t c = pick_half(cv, 8 * j, hi);
        const bool ok = kc + 8 * j + 4 * hb < kend;                         // Kt is a multiple of 4: whole groups
        const f4 v = P::a_load_group4(a, z, arow + c);
        dst[4 * j] = ok ? v.x : 0.0f; dst[4 * j + 1] = ok ? v.y : 0.0f; dst[4 * j + 2] = ok ? v.z : 0.0f; dst[4 * j + 3] = ok ? v.w : 0.0f;
      }
    } else {
      const int kl = kc + (lane & 31);
      const aoff_t cv = P::a_col(a, z, kl < kend ? kl : kbeg);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const aoff_t c = pick_half(cv, kslot(t, 0), hi);
        const bool ok = kc + kslot(t, hb) < kend;
        float v;
        if constexpr (P::A_U8) v = P::a_load(a, z, arow + c);
        else v = (float)abase[(uint32_t)(arow + c)];                      // uniform base + 32-bit lane offset
        dst[t] = ok ? v : 0.0f;
      }
    }
  };
  auto load_b = [&](int kc, float* dst) {
    if constexpr (P::B_K) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f4 v = P::b_load4(a, z, bcol + P::b_row(a, z, kc + 8 * j + 4 * hb));
        dst[4 * j] = v.x; dst[4 * j + 1] = v.y; dst[4 * j + 2] = v.z; dst[4 * j + 3] = v.w;
      }
    } else if constexpr (P::B_REG) {
      if (kc + 32 <= kend) {
        if constexpr (sizeof(BT) * P::B_LD >= 2048) {
          const uint32_t kcb = (uint32_t)sizeof(BT) * (uint32_t)kc * (uint32_t)P::B_LD;
#pragma unroll
          for (int t = 0; t < 16; ++t) dst[t] = (float)ld_byte_off(bbase, boffb + kcb + (uint32_t)(sizeof(BT) * kslot(t, 0) * P::B_LD));
        } else {
          const BT* p = breg + (size_t)kc * P::B_LD;
#pragma unroll
          for (int t = 0; t < 16; ++t) dst[t] = (float)p[(size_t)kslot(t, 0) * P::B_LD];
        }
      } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) { const int k = kc + kslot(t, hb); dst[t] = k < kend ? (float)breg[(size_t)(kc + kslot(t, 0)) * P::B_LD] : 0.0f; }
      }
    } else {
      const int kl = kc + (lane & 31);
      const int rv = P::b_row(a, z, kl < kend ? kl : kbeg);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int r = pick_half(rv, kslot(t, 0), hi);
        const float v = (float)bbase[(uint32_t)(r + bcol)];               // always in bounds (clamped k)
        dst[t] = kc + kslot(t, hb) < kend ? v : 0.0f;
      }
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

  int kc = kbeg + (wave < NW ? wave : 0) * 32;            // wave-uniform
  SDQN_STAMP(1);
  SDQN_STAMP(2);
  // No software prefetch across chunks: one register set keeps the kernels at ~64 VGPRs, so 7-8 waves per SIMD are
  // resident and hide each other's load latency (most waves own a single chunk at B = 32 anyway).
  // staged operands are software-prefetched one chunk ahead (4 x f4 per operand): registers -> panel -> fragment
  f4 pra[STG_A ? 4 : 1], prb[STG_B ? 4 : 1];
  auto stage_load = [&](int kk) {
    if constexpr (STG_A) {
      const aoff_t c = P::a_col(a, z, kk + 4 * (lane & 7));
#pragma unroll
      for (int j = 0; j < 4; ++j) pra[j] = P::a_load4(a, z, srow[j] + c);
    }
    if constexpr (STG_B) {
      const int r = P::b_row(a, z, kk + 4 * (lane & 7));
#pragma unroll
      for (int j = 0; j < 4; ++j) prb[j] = P::b_load4(a, z, r + scol[j]);
    }
  };
  (void)pra; (void)prb;
  if constexpr (STG_A || STG_B) { SDQN_WSTAMP(0); if (kc < kend) stage_load(kc); SDQN_WSTAMP(1); }
  while (kc < kend) {
    float fa[16], fb[16];
    SDQN_WSTAMP(0);
    if constexpr (!STG_A) load_a(kc, fa);
    if constexpr (!STG_B) load_b(kc, fb);
    if constexpr (STG_A || STG_B) {
      if constexpr (STG_A) {
        float* d = pan_a + (4 * (lane & 7)) * 33 + (lane >> 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) { d[8 * j] = pra[j].x; d[33 + 8 * j] = pra[j].y; d[66 + 8 * j] = pra[j].z; d[99 + 8 * j] = pra[j].w; }
      }
      if constexpr (STG_B) {
        float* d = pan_b + (4 * (lane & 7)) * 33 + (lane >> 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) { d[8 * j] = prb[j].x; d[33 + 8 * j] = prb[j].y; d[66 + 8 * j] = prb[j].z; d[99 + 8 * j] = prb[j].w; }
      }
      if (kc + NW * 32 < kend) stage_load(kc + NW * 32);   // next chunk's loads fly under this chunk's MFMAs
      wave_lds_sync();
      if constexpr (STG_A) {
#pragma unroll
        for (int t = 0; t < 16; ++t) fa[t] = pan_a[kslot(t, 0) * 33 + hb * (4 * 33) + (lane & 31)];
      }
      if constexpr (STG_B) {
#pragma unroll
        for (int t = 0; t < 16; ++t) fb[t] = pan_b[kslot(t, 0) * 33 + hb * (4 * 33) + (lane & 31)];
      }
      wave_lds_sync();                                     // panel reads issued before the next chunk's stores
    }
#ifdef SDQN_TIMING
    SDQN_WSTAMP(1);
    asm volatile("" :: "v"(fa[0]), "v"(fa[3]), "v"(fa[4]), "v"(fa[8]), "v"(fa[12]), "v"(fa[15]));      // A landed (its loads were issued first: in-order return)
    SDQN_WSTAMP(2);
    asm volatile("" :: "v"(fa[0]), "v"(fb[0]), "v"(fa[15]), "v"(fb[15]));      // operands landed
    SDQN_WSTAMP(3);
    SDQN_STAMP(3);
#endif
#pragma unroll
    for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[t], acc, 0, 0, 0);
    kc += NW * 32;
  }

  // ---- epilogue: NW partial tiles -> LDS -> summed in fixed order -> P::store (lanes along n) ----
#ifdef SDQN_TIMING
  asm volatile("" :: "v"(acc[0]), "v"(acc[15]));
  SDQN_STAMP(4);
#endif
  if constexpr (NW > 1) {
    if (wave < NW) {
      float* cw = smem + wave * WAVE_LDS;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        cw[row * 33 + (lane & 31)] = acc[r];
      }
    }
    __syncthreads();
    SDQN_STAMP(5);
    if constexpr (signals<P>::value) { if (threadIdx.x == 0) P::signal(a, bx, by, bz); }
    for (int e = threadIdx.x; e < 1024; e += NT) {
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
    P::store16(a, z, ks, m0, n0, lane, M, N, v, epi);     // lane holds rows (r&3)+8(r>>2)+4(l>>5), column l&31
  }
  SDQN_STAMP(6);
  SDQN_WSTAMP_FLUSH;
}

// ---- fp16-mode tile body: packed-fp16 MFMA -------------------------------------------------------------------
// v_mfma_f32_32x32x16_f16: lane (i = l & 31, h = l >> 5) feeds 8 consecutive k (one 16-B load) of row i of A and of
// column i of B into k-slots 8h..8h+7; both operands of these problems are k-contiguous in memory (NHWC activations /
// padded deltas; TRANSPOSED half weight copies for forward, master-layout half copies for dgrad), so there is no LDS
// staging at all: 2 loads + 1 MFMA per 16 k.  A and B use the same slot <-> k assignment, so the result does not
// depend on the hardware's internal k numbering.  fp32 accumulation, same K-split / fixed-order combine as above.
template <class P, int NW, int NT>
__device__ __forceinline__ void gemm_tile_h(const StepArgs& a, int bx, int by, int bz, float* smem) {
  constexpr int WAVE_LDS = PANEL;
  typedef typename P::aoff_t aoff_t;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = bx * 32, n0 = by * 32;
  int z, ks, kbeg, kend;
  P::ksplit(a, bz, z, ks, kbeg, kend);
  if (NW * 64 < NT && wave >= NW) kend = kbeg;
  const int M = P::M(a), N = P::N(a);
  const int i = lane & 31, h8 = (lane >> 5) * 8;
  const aoff_t arow = P::a_row(a, z, m0 + i < M ? m0 + i : M - 1);
  const int bcol = P::b_col(a, z, n0 + i < N ? n0 + i : N - 1);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  for (int kc = kbeg + (wave < NW ? wave : 0) * 32; kc < kend; kc += NW * 32) {
    const half8 fa0 = P::a_load8(a, z, arow + P::a_col(a, z, kc + h8));
    const half8 fb0 = P::b_load8(a, z, bcol + P::b_row(a, z, kc + h8));
    const half8 fa1 = P::a_load8(a, z, arow + P::a_col(a, z, kc + 16 + h8));
    const half8 fb1 = P::b_load8(a, z, bcol + P::b_row(a, z, kc + 16 + h8));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0, fb0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1, fb1, acc, 0, 0, 0);
  }
  if constexpr (NW > 1) {
    if (wave < NW) {
      float* cw = smem + wave * WAVE_LDS;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        cw[row * 33 + (lane & 31)] = acc[r];
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += NT) {
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



// ---- fp16-mode weight gradients on packed-fp16 MFMA ---------------------------------------------------------------
// C(m, n) = sum_k A(k, m) B(k, n) with k = (sample, output position): BOTH operands are contiguous along m / n in memory
// (NHWC activations, dense deltas), while v_mfma_f32_32x32x16_f16 wants 8 consecutive k per lane.  So every 32-deep
// K-chunk goes through a wave-private LDS transpose: lane (k = l & 31, part = l >> 5) fetches 16 consecutive halves
// (32 B, two 16-byte loads) of row k of each operand and stores them as [k][x] rows (pitch 40 halves: 16-byte aligned,
// conflict-free for the 8-lane groups of ds_write_b128); lane (i = l & 31, h = l >> 5) then reads its 8 k-values of
// column i with 8 ds_read_u16 (lanes along x: 64 contiguous bytes per half-wave, no conflicts) per MFMA operand.
// 4 loads + 4 wide stores + 32 narrow reads + 2 MFMAs per chunk instead of 32 loads + 16 fp32 MFMAs (1024 pipe cycles).
// conv1's A operand needs no transpose: the 4 output positions of an aligned k-group are 4 bytes apart in the u8 frame
// (A_GROUP4), so a lane builds its 8 k-values from two unaligned 16-byte ring loads, normalised and rounded to half like
// the forward pass does.  fp32 accumulation, K split over NW waves and combined in fixed order, same epilogues as the
// fp32-MFMA wgrads (loss scale divided out, fused RMSProp + half-copy refresh for fc4).
constexpr int HW_PITCH = 40;                                   // halves per staged k-row
constexpr int HW_TILE = 32 * HW_PITCH;                         // one [32 k][32 x] half tile: 2560 B
constexpr int HW_WAVE_LDS = (2 * HW_TILE * 2 + 3) / 4 > PANEL ? (2 * HW_TILE * 2 + 3) / 4 : PANEL;   // floats per wave: two half tiles, later one fp32 combine panel
static_assert(2 * HW_TILE * 2 <= HW_WAVE_LDS * 4 && PANEL <= HW_WAVE_LDS, "staging tiles / combine panel fit the per-wave region");
__device__ __forceinline__ half8 ld_half8(const half_t* p) { return *reinterpret_cast<const half8*>(p); }

template <class P, int NW, int NT>
__device__ __forceinline__ void gemm_tile_hw(const StepArgs& a, int bx, int by, int bz, float* smem) {
  typedef typename P::aoff_t aoff_t;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = bx * 32, n0 = by * 32;
  int z, ks, kbeg, kend;
  P::ksplit(a, bz, z, ks, kbeg, kend);
  if (NW * 64 < NT && wave >= NW) kend = kbeg;
  const int M = P::M(a), N = P::N(a);
  const int i = lane & 31, hb = lane >> 5;
  const bool hi = lane >= 32;
  // NW == 1 inside a wider workgroup (one tile per wave, gemm_multi_kernel): every wave stages in its own region
  half_t* sa = reinterpret_cast<half_t*>(smem + (NW == 1 ? wave : (wave < NW ? wave : 0)) * HW_WAVE_LDS);
  half_t* sb = sa + HW_TILE;
  const half_t* bbase = P::b_ptr(a, z);
  // staging role: this lane fetches halves [16 * hb, 16 * hb + 16) of row k = kc + i of the chunk
  const int bcol16 = P::b_col(a, z, n0 + 16 * hb);
  aoff_t arow16 = 0, arow_own = 0;
  if constexpr (P::A_U8) arow_own = P::a_row(a, z, m0 + i < M ? m0 + i : M - 1);          // conv1: own row, direct
  else arow16 = P::a_row(a, z, m0 + 16 * hb);
  (void)arow16; (void)arow_own;

  typename P::Epi epi;
  if constexpr (NW == 1) P::epi_begin(a, m0, n0, lane, epi);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

  half8 pa0, pa1, pb0, pb1;                                    // staged rows of the NEXT chunk (prefetched)
  aoff_t cvn = 0;                                             // conv1: patch origin of k = kc + i of the next chunk
  auto stage_load = [&](int kc) {
    const int kl = kc + i;
    const int kk = kl < kend ? kl : kbeg;
    const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    const half_t* pb = bbase + (size_t)kk * P::B_LD + bcol16;
    pb0 = ld_half8(pb); pb1 = ld_half8(pb + 8);
    if (kl >= kend) { pb0 = zero; pb1 = zero; }
    if constexpr (P::A_U8) cvn = P::a_col(a, z, kk);
    else {
      const half_t* pa;
      if constexpr (P::A_REG) pa = P::a_ptr(a, z) + (size_t)kk * P::A_LD + (size_t)arow16;
      else pa = P::a_ptr(a, z) + (uint32_t)(arow16 + P::a_col(a, z, kk));
      pa0 = ld_half8(pa); pa1 = ld_half8(pa + 8);
      if (kl >= kend) { pa0 = zero; pa1 = zero; }
    }
  };
  int kc = kbeg + (wave < NW ? wave : 0) * 32;
  if (kc < kend) stage_load(kc);
  while (kc < kend) {
    // staged rows -> LDS ([k][x], pitch 40 halves)
    *reinterpret_cast<half8*>(sb + i * HW_PITCH + 16 * hb) = pb0;
    *reinterpret_cast<half8*>(sb + i * HW_PITCH + 16 * hb + 8) = pb1;
    if constexpr (!P::A_U8) {
      *reinterpret_cast<half8*>(sa + i * HW_PITCH + 16 * hb) = pa0;
      *reinterpret_cast<half8*>(sa + i * HW_PITCH + 16 * hb + 8) = pa1;
    }
    const aoff_t cv = cvn;
    const int kcur = kc;
    kc += NW * 32;
    if (kc < kend) stage_load(kc);                              // next chunk's loads fly under the transposes + MFMAs
    wave_lds_sync();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      half8 fa, fb;
#pragma unroll
      for (int j = 0; j < 8; ++j) fb[j] = sb[(16 * s + 8 * hb + j) * HW_PITCH + i];
      if constexpr (P::A_U8) {
        // k = kcur + 16 s + 8 h + 4 g + e: patch origins of the two aligned k-groups from lanes 16 s + 4 g (+ 8 for h = 1)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int src = 16 * s + 4 * g;
          const int l0 = __builtin_amdgcn_readlane((int)(cv & 0xFFFFFFFF), src), l1 = __builtin_amdgcn_readlane((int)((int64_t)cv >> 32), src);
          const int u0 = __builtin_amdgcn_readlane((int)(cv & 0xFFFFFFFF), src + 8), u1 = __builtin_amdgcn_readlane((int)((int64_t)cv >> 32), src + 8);
          const int64_t lo = ((int64_t)l1 << 32) | (uint32_t)l0, up = ((int64_t)u1 << 32) | (uint32_t)u0;
          const int64_t c = hi ? up : lo;
          const bool ok = kcur + 16 * s + 8 * hb + 4 * g < kend;
          const f4 v = P::a_load_group4(a, z, arow_own + c);
          fa[4 * g] = ok ? (half_t)v.x : (half_t)0.0f; fa[4 * g + 1] = ok ? (half_t)v.y : (half_t)0.0f;
          fa[4 * g + 2] = ok ? (half_t)v.z : (half_t)0.0f; fa[4 * g + 3] = ok ? (half_t)v.w : (half_t)0.0f;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) fa[j] = sa[(16 * s + 8 * hb + j) * HW_PITCH + i];
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
    }
    wave_lds_sync();                                            // tile reads issued before the next chunk's stores
  }

  if constexpr (NW > 1) {
    if (wave < NW) {                                            // the wave's own staging region becomes its combine panel
      float* cw = smem + wave * HW_WAVE_LDS;
#pragma unroll
      for (int r = 0; r < 16; ++r) cw[((r & 3) + 8 * (r >> 2) + 4 * hb) * 33 + i] = acc[r];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += NT) {
      const int ml = e >> 5, nl = e & 31;
      float v = smem[ml * 33 + nl];
#pragma unroll
      for (int w = 1; w < NW; ++w) v += smem[w * HW_WAVE_LDS + ml * 33 + nl];
      if (m0 + ml < M && n0 + nl < N) P::store(a, z, ks, m0 + ml, n0 + nl, v);
    }
  } else {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r];
    P::store16(a, z, ks, m0, n0, lane, M, N, v, epi);
  }
}

template <class P, int NW, int NT>
__device__ __forceinline__ void run_tile(const StepArgs& a, int bx, int by, int bz, float* smem) {
  if constexpr (uses_f16_wgrad<P>::value) gemm_tile_hw<P, NW, NT>(a, bx, by, bz, smem);
  else if constexpr (uses_f16_mfma<P>::value) gemm_tile_h<P, NW, NT>(a, bx, by, bz, smem);
  else gemm_tile<P, NW, NT>(a, bx, by, bz, smem);
}
template <class P, int NW>
constexpr int tile_lds_any() {
  if constexpr (uses_f16_wgrad<P>::value) return NW * HW_WAVE_LDS;   // staging tiles, reused as combine panels
  else return uses_f16_mfma<P>::value ? NW * PANEL : tile_lds<P, NW>();
}
// rows / columns of C one wave-tile covers
template <class P> constexpr int tile_m() { return 32; }
template <class P> constexpr int tile_n() { return 32; }

// (xcd_tile_id / xcd_tile_id_range, the XCD-aware workgroup -> tile maps: problems.h — host + device, tests/emul executes them)

template <class P, int NW>
__global__ void __launch_bounds__(NW * 64) gemm_kernel(const StepArgs a) {
  __shared__ float smem[tile_lds_any<P, NW>()];
  if constexpr (has_preload<P>::value) P::preload(a, gridDim.x, gridDim.y, gridDim.z);
  const int gx = gridDim.x, gy = gridDim.y;
  const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const int t = (a.xcd_map & 1) ? xcd_tile_id(lin, gx * gy * gridDim.z) : lin;
  const int bz = t / (gx * gy), r = t - bz * (gx * gy);
  run_tile<P, NW, NW * 64>(a, r % gx, r / gx, bz, smem);
}

// ---- several independent problems in ONE launch -------------------------------------------------------
// Stages that depend on the same producer (e.g. conv3_dgrad, conv3_wgrad and fc4_wgrad all wait for
// fc4_dgrad only) run as one grid: workgroup ranges dispatch to different problem structs, so their latency
// chains overlap without cross-stream event waits and two launch ramps disappear.  A problem with NW == 1
// (fc4_wgrad at B <= 32) gets one TILE PER WAVE of the 1024-thread workgroup.
struct MultiDims { int n[3]; int gx[3], gy[3]; };       // workgroups per problem and its (x, y) tile grid

template <class P, int NW, int NT>
__device__ __forceinline__ void multi_dispatch(const StepArgs& a, const MultiDims& d, int which, int local, float* smem) {
  if constexpr (NW == 1) {
    // one tile per wave; the launch owns the tile range [a.f4w_first, a.f4w_first + a.f4w_count)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = local * (NT / 64) + wave;
    const int per_z = d.gx[which] * d.gy[which];
    if (t < a.f4w_count) {
      const int tile = a.f4w_first + t;
      const int bz = tile / per_z, r = tile - bz * per_z;
      run_tile<P, 1, 64>(a, r % d.gx[which], r / d.gx[which], bz, smem);    // (n-fastest tile order measured: -0.4 %)
    }
  } else {
    const int per_z = d.gx[which] * d.gy[which];
    const int bz = local / per_z, r = local - bz * per_z;
    run_tile<P, NW, NT>(a, r % d.gx[which], r / d.gx[which], bz, smem);
  }
}

template <int NT, class P0, int NW0, class P1, int NW1, class P2, int NW2>
__global__ void __launch_bounds__(NT) gemm_multi_kernel(const StepArgs a, const MultiDims d) {
  // (a one-tile-per-wave problem needs one region per wave of the workgroup)
  constexpr int L0 = tile_lds_any<P0, NW0>() * (NW0 == 1 ? NT / 64 : 1), L1 = tile_lds_any<P1, NW1>() * (NW1 == 1 ? NT / 64 : 1),
                L2 = tile_lds_any<P2, NW2>() * (NW2 == 1 ? NT / 64 : 1);
  constexpr int L = L0 > L1 ? (L0 > L2 ? L0 : L2) : (L1 > L2 ? L1 : L2);
  __shared__ float smem[L];
  if constexpr (has_preload_multi<P1>::value) P1::preload_multi(a, d);      // (one statement for the whole launch: every field of all its problems)
  const int b = blockIdx.x;                               // problem choice is workgroup-uniform; XCD-contiguous runs per problem
  const int xm = a.xcd_map;                               // bit i: problem i of the launch uses the XCD-contiguous map
  if (b < d.n[0]) multi_dispatch<P0, NW0, NT>(a, d, 0, (xm & 1) ? xcd_tile_id_range(b, 0, d.n[0]) : b, smem);
  else if (b < d.n[0] + d.n[1]) multi_dispatch<P1, NW1, NT>(a, d, 1, (xm & 2) ? xcd_tile_id_range(b, d.n[0], d.n[1]) : b - d.n[0], smem);
  else multi_dispatch<P2, NW2, NT>(a, d, 2, (xm & 4) ? xcd_tile_id_range(b, d.n[0] + d.n[1], d.n[2]) : b - d.n[0] - d.n[1], smem);
}

struct NoProblem {            // placeholder third problem for two-problem launches (never dispatched: n[2] = 0)
  static constexpr bool A_K = false, B_K = false; typedef int aoff_t; struct Epi {};
  static constexpr bool A_REG = false, A_U8 = false, B_REG = false; static constexpr int A_LD = 0, B_LD = 0;
  SDQN_HD static const float* a_ptr(const StepArgs&, int) { return nullptr; }
  SDQN_HD static const float* b_ptr(const StepArgs&, int) { return nullptr; }
  SDQN_HD static int M(const StepArgs&) { return 0; }
  SDQN_HD static int N(const StepArgs&) { return 0; }
  SDQN_HD static int nbz(const StepArgs&) { return 0; }
  SDQN_HD static void ksplit(const StepArgs&, int, int& z, int& ks, int& kb, int& ke) { z = ks = kb = ke = 0; }
  SDQN_HD static int a_row(const StepArgs&, int, int) { return 0; }
  SDQN_HD static int a_col(const StepArgs&, int, int) { return 0; }
  SDQN_HD static float a_load(const StepArgs&, int, int) { return 0.f; }
  SDQN_HD static f4 a_load4(const StepArgs&, int, int) { f4 o; o.x = o.y = o.z = o.w = 0.f; return o; }
  SDQN_HD static f4 b_load4(const StepArgs&, int, int) { f4 o; o.x = o.y = o.z = o.w = 0.f; return o; }
  SDQN_HD static int b_row(const StepArgs&, int, int) { return 0; }
  SDQN_HD static int b_col(const StepArgs&, int, int) { return 0; }
  SDQN_HD static float b_load(const StepArgs&, int, int) { return 0.f; }
  SDQN_HD static void store(const StepArgs&, int, int, int, int, float) {}
  __device__ static void epi_begin(const StepArgs&, int, int, int, Epi&) {}
  __device__ static void store16(const StepArgs&, int, int, int, int, int, int, int, const float*, Epi&) {}
};

template <class P, int NW>
inline hipError_t launch_gemm(const StepArgs& a, hipStream_t stream) {
  dim3 grid((P::M(a) + tile_m<P>() - 1) / tile_m<P>(), (P::N(a) + tile_n<P>() - 1) / tile_n<P>(), P::nbz(a));
  SDQN_LAUNCH((gemm_kernel<P, NW>), grid, dim3(NW * 64), 0, stream, a);
  return hipGetLastError();
}

template <int NT, class P, int NW>
inline void multi_fill(const StepArgs& a, MultiDims& d, int i) {
  d.gx[i] = (P::M(a) + tile_m<P>() - 1) / tile_m<P>(); d.gy[i] = (P::N(a) + tile_n<P>() - 1) / tile_n<P>();
  const int tiles = d.gx[i] * d.gy[i] * P::nbz(a);
  d.n[i] = NW == 1 ? (a.f4w_count + NT / 64 - 1) / (NT / 64) : tiles;    // NW == 1: one tile per wave, range from StepArgs
}
template <int NT, class P0, int NW0, class P1, int NW1, class P2, int NW2>
inline hipError_t launch_multi(const StepArgs& a, bool has1, bool has2, hipStream_t stream) {
  static_assert(NW0 * 64 <= NT && NW1 * 64 <= NT && NW2 * 64 <= NT, "waves per tile exceed the workgroup");
  MultiDims d; memset(&d, 0, sizeof d);
  multi_fill<NT, P0, NW0>(a, d, 0);
  if (has1) multi_fill<NT, P1, NW1>(a, d, 1);
  if (has2) multi_fill<NT, P2, NW2>(a, d, 2);
  if (d.n[0] + d.n[1] + d.n[2] == 0) return hipSuccess;
  SDQN_LAUNCH((gemm_multi_kernel<NT, P0, NW0, P1, NW1, P2, NW2>), dim3(d.n[0] + d.n[1] + d.n[2]), dim3(NT), 0, stream, a, d);
  return hipGetLastError();
}

}  // namespace sdqn
