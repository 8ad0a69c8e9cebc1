// gemm_engine_sk.h — CHUNK-GRANULAR work assignment ("stream-K") for the block-tile routine's forward launches at B >= 128 (float32).
//
// What bounds a block-tile launch is its most loaded SIMDs (profiles/r04_pmc_sq_b256.txt): conv3_fwd has 392 blocks of 18 chunks
// for 256 CUs — 136 CUs run two blocks, 120 run one, every wave is matrix-bound two thirds of its life, and the launch takes the time of TWO
// blocks (36 chunk-times per SIMD) for 1.53 blocks of work per CU.  Here the launch has G = 512 workgroups (two per CU, all resident) and the
// U = blocks x chunks units are dealt out evenly: workgroup g owns units [g U / G, (g + 1) U / G) — the tail (or a middle) piece of one block,
// whole blocks, the head piece of the next.
//   * a piece that does not start at the block's chunk 0 is the FIRST thing its workgroup computes; its 64 x 64 partial sum goes to scratch
//     slot g with write-through stores, then flag[g] = epoch (every thread waits for its own stores, barrier, one relaxed agent-scope store);
//   * the HEAD piece (chunks 0 .. c) is the LAST thing its workgroup computes; its owner polls the flags of the workgroups that follow it
//     (bounded: a time-out sets the sticky word the host checks and the launch ends with a wrong block instead of hanging), reads their
//     partials with agent-scope loads and adds them in piece order — head + next + next: k ascending, ONE fixed order per (grid, G) — then
//     runs the block's epilogue.  A waiter only ever waits for higher-numbered workgroups' FIRST pieces: no cycle, whatever is resident.
//   * same panels, loaders, fragment maps and k-slot order as bt_tile (bt_map.h); unconditional clamped ring loads (run-time chunk ranges).
// Scratch: StepArgs::slab1 (conv1's split-K slabs: written by bwd1, consumed by the update — idle during the forward pass; >= 10 MB at
// B >= 128 for G x 16 KB = 8 MB), StepArgs::f4d_flags / f4d_epoch (one epoch per train step: only launches of a train step may use this form).
// Results: a whole block is bit-identical to bt_tile's; a split block sums the same chunks in the same order but in two or three partial
// accumulators (last-bit differences).
//
// STATUS (last GPU minutes of round 4; experiments build, menu entry 9 of bt:1 / bt:2): the hand-off WORKS — conv2_fwd and conv3_fwd, alone and
// together (one flag region per launch of the step), reproduce bt_tile's gradients to 5e-7 ... 9e-7 — but the launches are still slower than
// bt_tile: first run conv3_fwd 26.3 vs 23.6 us and conv2_fwd 37.0 vs 30.1 (hipcc gave the ring loads the registers of the fragments just
// multiplied: every fragment read behind a barrier waited for the loads issued two MFMAs earlier); with the fragments read ahead of the
// loads and kept live across their issue (census: no wait in front of the reads any more, the previous set drained early in the chunk
// instead) 24.2 vs 22.8 and 33.0 vs 29.4 (same box); third run, the piece loop instantiated for every chunk count 1 .. MAXCH and inlined
// (straight-line code with bt_tile's exact vmcnt(4 / 5) ladder, 73-93 KB of code per kernel, no scratch): 23.1 vs 22.6 and 32.3 vs 28.9 —
// correct, balanced, bt_tile-quality loops, and STILL not faster.  Two workgroups per CU with 13.8 chunks each need 28.2 k matrix cycles per
// SIMD (11.8 us); the launch takes ~20: a SIMD reaches ~60 % matrix utilisation whether it holds one, two or three of these waves.  The
// balance was not the bound; what keeps two ready-looking waves from filling one matrix pipe is the open question for round 5.
#pragma once
#include "gemm_engine_bt.h"
#include "problems_wt.h"      // wt_store

namespace sdqn {

constexpr int SK_GROUPS = 512;                          // workgroups per launch: two per CU
constexpr int SK_PART = 64 * 64;                        // floats of one partial block

template <class P_, int MAXCH_, int D_ = 2>       // MAXCH_: chunks of a whole block (K / 32)
struct SkCfg : BtCfg<P_, 64, 64, 2, 2, D_> { static constexpr int MAXCH = MAXCH_; };

__device__ __forceinline__ float sk_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// chunks [c0, c1) of block (bx, by, bz) -> this wave's 32 x 32 accumulator (bt_tile's loop, SM = SN = 1, unconditional ring loads)
// NIT: the piece's chunk count as a COMPILE-TIME value (sk_kernel dispatches over 1 .. MAXCH): the loop below is then straight-line code with
// exact wait counts, like bt_tile's — the rolled form lost them at its back edge and drained the ring every chunk (tools/exp/README.md)
template <class C, int NIT>
__device__ __forceinline__ void sk_piece(const StepArgs& a, int bx, int by, int bz, int c0, float* smem, f32x16& acc,
                                      int& z_out, int& ks_out) {
  const int c1 = c0 + NIT;
  typedef typename C::P P;
  typedef typename P::aoff_t aoff_t;
  constexpr int BM = C::BM, BN = C::BN, WN = C::WN, D = C::D;
  constexpr bool AK = P::A_K, BKC = P::B_K;
  constexpr int PA = bt::passes(BM), PB = bt::passes(BN);
  static_assert(C::SM == 1 && C::SN == 1, "one 32 x 32 sub-tile per wave");
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int m0 = bx * BM, n0 = by * BN;
  int z, ks, kb0, ke0;
  P::ksplit(a, bz, z, ks, kb0, ke0);
  z_out = z; ks_out = ks;
  const int kbeg = kb0 + c0 * bt::BK;
  int kend = kb0 + c1 * bt::BK; if (kend > ke0) kend = ke0;
  const int M = P::M(a), N = P::N(a);
  aoff_t ag[PA]; int bg[PB];
  if constexpr (AK) {
#pragma unroll
    for (int p = 0; p < PA; ++p) { const int m = m0 + bt::km_item_row(tid, p); ag[p] = P::a_row(a, z, m < M ? m : M - 1); }
  } else {
    const int m = m0 + bt::mk_item_x(BM, tid);
    ag[0] = P::a_row(a, z, m + 4 <= M ? m : M - 4);
  }
  if constexpr (BKC) {
#pragma unroll
    for (int p = 0; p < PB; ++p) { const int n = n0 + bt::km_item_row(tid, p); bg[p] = P::b_col(a, z, n < N ? n : N - 1); }
  } else {
    const int n = n0 + bt::mk_item_x(BN, tid);
    bg[0] = P::b_col(a, z, n + 4 <= N ? n : N - 4);
  }
  float4 ra[D][PA], rb[D][PB];
  // (every chunk of these problems is a whole one — K is a multiple of 32, launch_sk checks it — and the ring only ever asks for chunks inside
  //  [kbeg, kend): no zero fill, no select on a loaded value in front of the LDS stores)
  auto gload = [&](int kc, float4* qa, float4* qb) {
    if constexpr (AK) {
      const aoff_t c = P::a_col(a, z, kc + bt::km_item_k(tid));
#pragma unroll
      for (int p = 0; p < PA; ++p) qa[p] = f4_to_float4(P::a_load4(a, z, ag[p] + c));
    } else {
#pragma unroll
      for (int p = 0; p < PA; ++p) qa[p] = f4_to_float4(P::a_load4(a, z, ag[0] + P::a_col(a, z, kc + bt::mk_item_k(BM, tid, p))));
    }
    if constexpr (BKC) {
      const int r = P::b_row(a, z, kc + bt::km_item_k(tid));
#pragma unroll
      for (int p = 0; p < PB; ++p) qb[p] = f4_to_float4(P::b_load4(a, z, bg[p] + r));
    } else {
#pragma unroll
      for (int p = 0; p < PB; ++p) qb[p] = f4_to_float4(P::b_load4(a, z, bg[0] + P::b_row(a, z, kc + bt::mk_item_k(BN, tid, p))));
    }
  };
  auto lds_store = [&](const float4* qa, const float4* qb, float* As, float* Bs) {
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int o = AK ? bt::km_off(bt::km_item_row(tid, p), bt::km_item_k(tid)) : bt::mk_off(BM, bt::mk_item_k(BM, tid, p), bt::mk_item_x(BM, tid));
      *reinterpret_cast<float4*>(As + o) = qa[p];
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int o = BKC ? bt::km_off(bt::km_item_row(tid, p), bt::km_item_k(tid)) : bt::mk_off(BN, bt::mk_item_k(BN, tid, p), bt::mk_item_x(BN, tid));
      *reinterpret_cast<float4*>(Bs + o) = qb[p];
    }
  };
  float fa[16], fb[16];
  auto read_frags = [&](const float* As, const float* Bs) {
    { const int x = wm * 32 + i;
      if constexpr (AK) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 v = *reinterpret_cast<const float4*>(As + bt::km_off(x, 8 * j + 4 * h));
          fa[4 * j] = v.x; fa[4 * j + 1] = v.y; fa[4 * j + 2] = v.z; fa[4 * j + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) fa[t] = As[bt::mk_off(BM, bt::kslot(t, 0), x) + h * (4 * BM)];
      } }
    { const int x = wn * 32 + i;
      if constexpr (BKC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 v = *reinterpret_cast<const float4*>(Bs + bt::km_off(x, 8 * j + 4 * h));
          fb[4 * j] = v.x; fb[4 * j + 1] = v.y; fb[4 * j + 2] = v.z; fb[4 * j + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) fb[t] = Bs[bt::mk_off(BN, bt::kslot(t, 0), x) + h * (4 * BN)];
      } }
  };
  auto mma = [&]() {
#pragma unroll
    for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[t], acc, 0, 0, 0);
  };
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
  constexpr int nit = NIT;
#pragma unroll
  for (int d = 0; d < D; ++d) gload(kbeg + (d < nit ? d : nit - 1) * bt::BK, ra[d], rb[d]);
  lds_store(ra[0], rb[0], smem, smem + C::AF);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < nit; ++t) {
    const int d = t % D;
    float* cur = smem + (t & 1) * C::STAGE;
    float* nxt = smem + ((t + 1) & 1) * C::STAGE;
    if (t + D < nit) gload(kbeg + (t + D) * bt::BK, ra[d], rb[d]);           // (compile-time condition after unrolling)
    read_frags(cur, cur + C::AF);
    mma();
    if (t + 1 < nit) { lds_store(ra[(d + 1) % D], rb[(d + 1) % D], nxt, nxt + C::AF); __syncthreads(); }
  }
}

// piece length -> instantiation
template <class C, int N>
__device__ __forceinline__ void sk_piece_dispatch(int len, const StepArgs& a, int bx, int by, int bz, int c0, float* smem, f32x16& acc, int& z, int& ks) {
  if constexpr (N >= 1) {
    if (len == N) sk_piece<C, N>(a, bx, by, bz, c0, smem, acc, z, ks);
    else sk_piece_dispatch<C, N - 1>(len, a, bx, by, bz, c0, smem, acc, z, ks);
  }
}

template <class C>
__global__ void __launch_bounds__(bt::NT) sk_kernel(const StepArgs a, const int gx, const int gy, const int ntiles, const int nch, const int flag_base) {
  typedef typename C::P P;
  static_assert(!bt_gated<P>::value && !has_store_tile<P>::value && sizeof(typename P::Epi) <= 1, "plain-store problems (the forward convs)");
  __shared__ __attribute__((aligned(16))) float smem[C::LDS];
  if constexpr (has_preload<P>::value) P::preload(a, gridDim.x, (unsigned)gx, (unsigned)gy);
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WN, wn = wave - wm * C::WN;
  const int g = blockIdx.x, G = gridDim.x;
  const long long U = (long long)ntiles * nch;
  auto ubeg = [&](int w) { return (int)(((long long)w * U) / G); };
  int u0 = ubeg(g);
  const int u1 = ubeg(g + 1);
  float* const part = a.slab1;
  unsigned* const flags = a.f4d_flags + flag_base;       // one region of G words per launch of the step (they share the step's epoch)
  const unsigned epoch = a.f4d_epoch;
  const int M = P::M(a), N = P::N(a);
  while (u0 < u1) {
    const int tile = u0 / nch, c0 = u0 - tile * nch;
    const int len = (nch - c0 < u1 - u0) ? nch - c0 : u1 - u0, c1 = c0 + len;
    const int per_z = gx * gy, bz = tile / per_z, r = tile - bz * per_z, bx = r % gx, by = r / gx;
    f32x16 acc; int z, ks;
    sk_piece_dispatch<C, C::MAXCH>(len, a, bx, by, bz, c0, smem, acc, z, ks);
    if (c0 > 0) {
      // not the head of its block (always this workgroup's FIRST piece): partial -> slot g, then the flag
      float* slot = part + (size_t)g * SK_PART + wave * 1024 + lane;
#pragma unroll
      for (int q = 0; q < 16; ++q) wt_store(slot + q * 64, acc[q]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + g, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (c1 < nch) {
        // head of a split block: the following workgroups' first pieces complete it, in order
        int cc = c1, w = g + 1;
        while (cc < nch && w < G) {
          if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(flags + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
              __builtin_amdgcn_s_sleep(4);
              if (++spins > 2000000) { __hip_atomic_store(a.f4d_flags + (NIN4 / 32) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
          }
          __syncthreads();
          const float* slot = part + (size_t)w * SK_PART + wave * 1024 + lane;
          float pv[16];
#pragma unroll
          for (int q = 0; q < 16; ++q) pv[q] = sk_ld(slot + q * 64);
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[q] = acc[q] + pv[q];
          const int wl = ubeg(w + 1) - ubeg(w);                 // (its whole range, or what was left of this block)
          cc += (nch - cc < wl) ? nch - cc : wl;
          ++w;
        }
      }
      const int ms = bx * C::BM + wm * 32, ns = by * C::BN + wn * 32;
      if (ms < M && ns < N) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int m = ms + bt::acc_row(q, h), n = ns + i;
          if (m < M && n < N) P::store(a, z, ks, m, n, acc[q]);
        }
      }
    }
    u0 += len;
    __syncthreads();                                            // the LDS stages are reused by the next piece
  }
}

template <class C>
inline hipError_t launch_sk(const StepArgs& a, hipStream_t stream, int flag_region) {       // flag_region: 0 .. 2 (1 568 flag words are allocated)
  typedef typename C::P P;
  int gx, gy, gz; bt_grid<C>(a, gx, gy, gz);
  const int ntiles = gx * gy * gz;
  if (ntiles == 0) return hipSuccess;
  if (!a.slab1 || !a.f4d_flags || a.f4d_epoch == 0) return hipErrorInvalidValue;
  int z, ks, kb, ke; P::ksplit(a, 0, z, ks, kb, ke);
  if ((ke - kb) % bt::BK != 0) return hipErrorInvalidValue;          // whole chunks only
  const int nch = (ke - kb) / bt::BK;
  if (nch > C::MAXCH) return hipErrorInvalidValue;
  if (flag_region < 0 || flag_region > 2) return hipErrorInvalidValue;
  SDQN_LAUNCH((sk_kernel<C>), dim3(SK_GROUPS), dim3(bt::NT), 0, stream, a, gx, gy, ntiles, nch, flag_region * SK_GROUPS);
  return hipGetLastError();
}

}  // namespace sdqn
