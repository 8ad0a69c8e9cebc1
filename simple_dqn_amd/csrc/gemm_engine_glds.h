// gemm_engine_glds.h — the block-tile routine with its operand panels fetched STRAIGHT INTO LDS (global_load_lds_dwordx4), float32, B >= 128.
//
// Why (tools/exp/mfma_lds.hip, profiles/r04_mfma_lds.txt): what a chunk of bt_tile waits for is the global -> VGPR -> LDS staging of its 16 KB
// — the chunk body alone reaches 81-83 % of the matrix rate at two or three workgroups per CU, 69-70 % with L2-resident operands through the
// register ring, 55-60 % when they stream from memory (the product launches).  The same micro-kernel with direct-to-LDS loads: 74 % and 63 %
// at two workgroups per CU, 69 % / 62 % at ONE (ring: 55 % / 35 %).  This header is that loop on the real problems:
//   * panels are LANE-LINEAR images (bt_map.h: g_* maps) — a wave's load instruction writes lane l's 16 bytes at base + 16 l, so the layout
//     is chosen by which item a lane fetches: KM panels without padding, the bank swizzle applied to the SOURCE addresses; MK panels as before;
//   * THREE LDS stages: the loads of chunk t + 2 are issued at the top of chunk t into the stage chunk t - 1 was read from (free since the
//     barrier that ended it); `s_waitcnt vmcnt(loads per chunk)` in front of the barrier that ends chunk t — chunk t + 1 has landed, chunk
//     t + 2 stays in flight; no staging registers, no ds_write, no vmcnt -> ds_write dependency;
//   * the loads are inline asm (M0 = the instruction's LDS base): with the builtin hipcc puts `vmcnt(0)` in front of the next LDS read, and an
//     asm statement is invisible to its wait-count bookkeeping — the counted waits here are explicit, the compiler's own (gate / epilogue
//     prefetch loads issued before the loop) can only over-wait: vmcnt retires in order;
//   * whole chunks only (K ranges that are multiples of 32: every conv / fc problem of the step at B a multiple of 32 — launch_gl checks);
//   * same fragment order, MFMA order and epilogues as bt_tile: bit-identical results.
// STATUS (written in the last hours of round 4; experiments build, menu entry 13 of bt:<id>): maps validated on the CPU (tests/emul variants
// 3 / 4), instruction order read in the ISA, and its FIRST GPU run is green — every stage and gradient bit-identical to bt_tile's, launch by
// launch and all together, at B = 256 and 160 (test_direct_to_lds_panels_match_the_block_tile_routine).  Not faster yet (same box): fc4_dgrad
// 14.2 us (bt_tile 12.8), conv3_fwd 23.3 (22.6), conv2_fwd 30.6 (28.7), bwd2 40.8 (37.6), bwd3 64 (40.5: 54 KB of LDS per workgroup), fc4_fwd
// with 7 K slabs 18.7 (bt_tile 21.0, latency engine 18.1) — the loop is rolled (run-time chunk counts for every problem here): the address
// arithmetic of the panel loads is redone every chunk and only two chunks are in flight, where the micro-kernel that promised 62-74 % had
// neither cost.  Since then (no GPU left in the round: census only): the forward convs and the dgrads take their chunk count as a template
// argument (GlCfg<..., NITC>) — fully unrolled, 22 VALU instructions and one vmcnt(4) per chunk, 34 VGPRs — and the round's very last GPU
// call measured them: conv3_fwd 22.57 us (bt_tile 22.55), fc4_dgrad 12.77 (12.78), conv2_fwd 31.7 (28.8: three workgroups per CU instead of
// four), bit-identical again.  So on the real operands the direct-to-LDS path EQUALS the register ring instead of beating it as it does in the
// micro-kernel (whose 16 KB per chunk are one contiguous block): what these launches wait for is not how the bytes get from L2 into LDS but how
// fast this access pattern (64 + 64 row pieces of 128 B per chunk and workgroup) is served at all.  Round 5: TCP / TCC counters on exactly that.
#pragma once
#include "gemm_engine_bt.h"

namespace sdqn {

// NITC_: the problem's chunk count when it is a compile-time fact (K / 32 of the forward convs and the dgrads; 0 = run-time: K slabs / splits).
// With it the chunk loop is fully unrolled — the stage rotation, the k offsets and most of the loads' address arithmetic fold into immediates
// (the rolled loop redoes them every chunk: the first GPU run's handicap against bt_tile's unrolled loops).
template <class P_, int BM_, int BN_, int WM_, int WN_, int NITC_ = 0>
struct GlCfg {
  typedef P_ P;
  static constexpr int NITC = NITC_;
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr int KIND = 5;
  static constexpr int SM = BM / (32 * WM), SN = BN / (32 * WN);
  static_assert(WM * WN * 64 == bt::NT, "four waves per workgroup");
  static_assert(SM >= 1 && SN >= 1 && SM * 32 * WM == BM && SN * 32 * WN == BN, "block = wave grid x sub-tiles of 32 x 32");
  static_assert(BM % 32 == 0 && BN % 32 == 0, "whole 256-slot passes");
  static constexpr int AF = bt::g_panel_floats(P::A_K, BM), BF = bt::g_panel_floats(P::B_K, BN);
  static constexpr int STAGE = AF + BF;                 // floats per LDS stage
  static constexpr int LDS = 3 * STAGE;
  static constexpr int PA = bt::g_passes(BM), PB = bt::g_passes(BN);
  static constexpr int LPC = PA + PB;                   // load instructions per thread and chunk
};

// one 16-byte load per lane, landing at lds_base + 16 * lane (lds_base wave-uniform)
__device__ __forceinline__ void glds16(const float* gp, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gp), "s"(lds_base) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void gl_wait() {
  static_assert(N >= 0 && N <= 8, "loads per chunk");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

template <class C>
__device__ __forceinline__ void gl_tile(const StepArgs& a, int bx, int by, int bz, float* smem) {
  typedef typename C::P P;
  typedef typename P::aoff_t aoff_t;
  constexpr int BM = C::BM, BN = C::BN, SM = C::SM, SN = C::SN, WN = C::WN, PA = C::PA, PB = C::PB;
  constexpr bool AK = P::A_K, BKC = P::B_K;
  static_assert(sizeof(typename a_elem<P>::type) == 4 && sizeof(typename b_elem<P>::type) == 4, "fp32 operands");
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int m0 = bx * BM, n0 = by * BN;
  int z, ks, kbeg, kend;
  P::ksplit(a, bz, z, ks, kbeg, kend);
  const int M = P::M(a), N = P::N(a);
  const float* abase = P::a_ptr(a, z);
  const float* bbase = P::b_ptr(a, z);

  // ---- loader geometry: slot p = 256 pass + tid of a panel; what it fetches (bt_map.h: g_* maps) --------------------------------------
  aoff_t ag[PA]; int bg[PB];
  const int p0 = tid;
  if constexpr (AK) {
#pragma unroll
    for (int ps = 0; ps < PA; ++ps) { const int m = m0 + bt::g_km_slot_x(p0 + 256 * ps); ag[ps] = P::a_row(a, z, m < M ? m : M - 1); }
  } else {
    const int m = m0 + bt::g_mk_slot_x(BM, p0);
    ag[0] = P::a_row(a, z, m + 4 <= M ? m : M - 4);
  }
  if constexpr (BKC) {
#pragma unroll
    for (int ps = 0; ps < PB; ++ps) { const int n = n0 + bt::g_km_slot_x(p0 + 256 * ps); bg[ps] = P::b_col(a, z, n < N ? n : N - 1); }
  } else {
    const int n = n0 + bt::g_mk_slot_x(BN, p0);
    bg[0] = P::b_col(a, z, n + 4 <= N ? n : N - 4);
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;       // byte address of the workgroup's LDS block
  // chunk kc -> stage st (0 .. 2): PA + PB load instructions per thread
  auto issue = [&](int kc, int st) {
    const unsigned sa = lds0 + (unsigned)(st * C::STAGE) * 4u, sb = sa + (unsigned)C::AF * 4u;
    if constexpr (AK) {
      const aoff_t col = P::a_col(a, z, kc + 4 * bt::g_km_slot_q(p0));                          // (q is the same in every pass: 256 | 128)
#pragma unroll
      for (int ps = 0; ps < PA; ++ps) glds16(abase + (ag[ps] + col), sa + (unsigned)(256 * ps + 64 * wave) * 16u);
    } else {
#pragma unroll
      for (int ps = 0; ps < PA; ++ps)
        glds16(abase + (ag[0] + P::a_col(a, z, kc + bt::g_mk_slot_k(BM, p0 + 256 * ps))), sa + (unsigned)(256 * ps + 64 * wave) * 16u);
    }
    if constexpr (BKC) {
      const int r = P::b_row(a, z, kc + 4 * bt::g_km_slot_q(p0));
#pragma unroll
      for (int ps = 0; ps < PB; ++ps) glds16(bbase + (bg[ps] + r), sb + (unsigned)(256 * ps + 64 * wave) * 16u);
    } else {
#pragma unroll
      for (int ps = 0; ps < PB; ++ps)
        glds16(bbase + (bg[0] + P::b_row(a, z, kc + bt::g_mk_slot_k(BN, p0 + 256 * ps))), sb + (unsigned)(256 * ps + 64 * wave) * 16u);
    }
  };

  f32x16 acc[SM][SN];
#pragma unroll
  for (int sm = 0; sm < SM; ++sm)
#pragma unroll
    for (int sn = 0; sn < SN; ++sn)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[sm][sn][q] = 0.0f;
  auto compute = [&](const float* As, const float* Bs) {
    float fa[SM][16], fb[SN][16];
#pragma unroll
    for (int sm = 0; sm < SM; ++sm) {
      const int x = (wm * SM + sm) * 32 + i;
      if constexpr (AK) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 v = *reinterpret_cast<const float4*>(As + 4 * bt::g_km_slot(x, 2 * j + h));
          fa[sm][4 * j] = v.x; fa[sm][4 * j + 1] = v.y; fa[sm][4 * j + 2] = v.z; fa[sm][4 * j + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) fa[sm][t] = As[bt::mk_off(BM, bt::kslot(t, 0), x) + h * (4 * BM)];
      }
    }
#pragma unroll
    for (int sn = 0; sn < SN; ++sn) {
      const int x = (wn * SN + sn) * 32 + i;
      if constexpr (BKC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 v = *reinterpret_cast<const float4*>(Bs + 4 * bt::g_km_slot(x, 2 * j + h));
          fb[sn][4 * j] = v.x; fb[sn][4 * j + 1] = v.y; fb[sn][4 * j + 2] = v.z; fb[sn][4 * j + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) fb[sn][t] = Bs[bt::mk_off(BN, bt::kslot(t, 0), x) + h * (4 * BN)];
      }
    }
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int sm = 0; sm < SM; ++sm)
#pragma unroll
        for (int sn = 0; sn < SN; ++sn) acc[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sm][t], fb[sn][t], acc[sm][sn], 0, 0, 0);
  };

  constexpr bool GATED = bt_gated<P>::value;
  float gate[GATED ? SM : 1][GATED ? SN : 1][16];
  if constexpr (GATED) {
#pragma unroll
    for (int sm = 0; sm < SM; ++sm)
#pragma unroll
      for (int sn = 0; sn < SN; ++sn)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int m = m0 + (wm * SM + sm) * 32 + bt::acc_row(q, h), n = n0 + (wn * SN + sn) * 32 + i;
          gate[sm][sn][q] = P::gate_load(a, z, m < M ? m : M - 1, n < N ? n : N - 1);
        }
  }
  (void)gate;
  typename P::Epi epi[SM][SN];
  auto epi_prefetch = [&]() {
#pragma unroll
    for (int sm = 0; sm < SM; ++sm)
#pragma unroll
      for (int sn = 0; sn < SN; ++sn) P::epi_begin(a, m0 + (wm * SM + sm) * 32, n0 + (wn * SN + sn) * 32, lane, epi[sm][sn]);
  };

  if constexpr (C::NITC > 0) {
    constexpr int nit = C::NITC;                                   // (launch_gl checks that the problem's K range is exactly NITC chunks)
    issue(kbeg, 0);
    issue(kbeg + (nit > 1 ? 1 : 0) * bt::BK, 1);
    gl_wait<C::LPC>();
    __syncthreads();
#pragma unroll
    for (int t = 0; t < nit; ++t) {
      issue(kbeg + (t + 2 < nit ? t + 2 : nit - 1) * bt::BK, (t + 2) % 3);
      if (t + 1 == nit) epi_prefetch();
      const float* cur = smem + (t % 3) * C::STAGE;
      compute(cur, cur + C::AF);
      gl_wait<C::LPC>();
      __syncthreads();
    }
    gl_wait<0>();
  } else {
  const int nit = (kend - kbeg) / bt::BK;                          // whole chunks (launch_gl)
  if (nit > 0) {
    issue(kbeg, 0);
    issue(kbeg + (nit > 1 ? 1 : 0) * bt::BK, 1);
    gl_wait<C::LPC>();                                             // chunk 0 has landed (chunk 1 may still fly)
    __syncthreads();
    int st = 0;                                                    // stage of chunk t
    for (int t = 0; t < nit; ++t) {
      const int st2 = st == 0 ? 2 : st - 1;                        // (t + 2) % 3
      issue(kbeg + (t + 2 < nit ? t + 2 : nit - 1) * bt::BK, st2);
      if (t + 1 == nit) epi_prefetch();
      const float* cur = smem + st * C::STAGE;
      compute(cur, cur + C::AF);
      gl_wait<C::LPC>();                                           // chunk t + 1 has landed; chunk t + 2 stays in flight
      __syncthreads();
      st = st == 2 ? 0 : st + 1;
    }
    gl_wait<0>();                                                  // nothing of this workgroup may land in LDS after it has gone
  } else epi_prefetch();
  }

#pragma unroll
  for (int sm = 0; sm < SM; ++sm)
#pragma unroll
    for (int sn = 0; sn < SN; ++sn) {
      const int ms = m0 + (wm * SM + sm) * 32, ns = n0 + (wn * SN + sn) * 32;
      if (ms >= M || ns >= N) continue;
      if constexpr (has_store_tile<P>::value) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = acc[sm][sn][q];
        P::store_tile(a, ms, ns, lane, v);
      } else if constexpr (sizeof(typename P::Epi) > 1) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = acc[sm][sn][q];
        P::store16(a, z, ks, ms, ns, lane, M, N, v, epi[sm][sn]);
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int m = ms + bt::acc_row(q, h), n = ns + i;
          if (m < M && n < N) {
            if constexpr (GATED) P::store_gated(a, z, ks, m, n, acc[sm][sn][q], gate[sm][sn][q]);
            else P::store(a, z, ks, m, n, acc[sm][sn][q]);
          }
        }
      }
    }
}

template <class C>
__global__ void __launch_bounds__(bt::NT) gl_kernel(const StepArgs a, const int gx, const int gy) {
  __shared__ __attribute__((aligned(16))) float smem[C::LDS];
  if constexpr (has_preload<typename C::P>::value) C::P::preload(a, gridDim.x, (unsigned)gx, (unsigned)gy);
  const int t = (a.xcd_map & 1) ? xcd_tile_id((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int per_z = gx * gy, bz = t / per_z, r = t - bz * per_z;
  gl_tile<C>(a, r % gx, r / gx, bz, smem);
}
// whole chunks only: every K range of the problem (each split) must be a multiple of 32
template <class C>
inline bool gl_whole_chunks(const StepArgs& a) {
  typedef typename C::P P;
  for (int bz = 0; bz < P::nbz(a); ++bz) {
    int z, ks, kb, ke; P::ksplit(a, bz, z, ks, kb, ke);
    if ((ke - kb) % bt::BK != 0) return false;
    if (C::NITC > 0 && (ke - kb) != C::NITC * bt::BK) return false;
  }
  return true;
}
template <class C>
inline hipError_t launch_gl(const StepArgs& a, hipStream_t stream) {
  int gx, gy, gz; bt_grid<C>(a, gx, gy, gz);
  if (gx * gy * gz == 0) return hipSuccess;
  if (!gl_whole_chunks<C>(a)) return hipErrorInvalidValue;
  SDQN_LAUNCH((gl_kernel<C>), dim3(gx * gy * gz), dim3(bt::NT), 0, stream, a, gx, gy);
  return hipGetLastError();
}
template <class C0, class C1, class C2>
__global__ void __launch_bounds__(bt::NT) gl_multi_kernel(const StepArgs a, const MultiDims d) {
  constexpr int L01 = C0::LDS > C1::LDS ? C0::LDS : C1::LDS, L = L01 > C2::LDS ? L01 : C2::LDS;
  __shared__ __attribute__((aligned(16))) float smem[L];
  if constexpr (has_preload_multi<typename C1::P>::value) C1::P::preload_multi(a, d);
  const int b = blockIdx.x, xm = a.xcd_map;
  if (b < d.n[0]) { const int l = (xm & 1) ? xcd_tile_id_range(b, 0, d.n[0]) : b, pz = d.gx[0] * d.gy[0], bz = l / pz, r = l - bz * pz; gl_tile<C0>(a, r % d.gx[0], r / d.gx[0], bz, smem); }
  else if (b < d.n[0] + d.n[1]) { const int l = (xm & 2) ? xcd_tile_id_range(b, d.n[0], d.n[1]) : b - d.n[0], pz = d.gx[1] * d.gy[1], bz = l / pz, r = l - bz * pz; gl_tile<C1>(a, r % d.gx[1], r / d.gx[1], bz, smem); }
  else { const int l = (xm & 4) ? xcd_tile_id_range(b, d.n[0] + d.n[1], d.n[2]) : b - d.n[0] - d.n[1], pz = d.gx[2] * d.gy[2], bz = l / pz, r = l - bz * pz; gl_tile<C2>(a, r % d.gx[2], r / d.gx[2], bz, smem); }
}
template <class C0, class C1, class C2>
inline hipError_t launch_gl_multi(const StepArgs& a, bool has0, bool has1, bool has2, hipStream_t stream) {
  MultiDims d; memset(&d, 0, sizeof d);
  int gz;
  if (has0) { if (!gl_whole_chunks<C0>(a)) return hipErrorInvalidValue; bt_grid<C0>(a, d.gx[0], d.gy[0], gz); d.n[0] = d.gx[0] * d.gy[0] * gz; }
  if (has1) { if (!gl_whole_chunks<C1>(a)) return hipErrorInvalidValue; bt_grid<C1>(a, d.gx[1], d.gy[1], gz); d.n[1] = d.gx[1] * d.gy[1] * gz; }
  if (has2) { if (!gl_whole_chunks<C2>(a)) return hipErrorInvalidValue; bt_grid<C2>(a, d.gx[2], d.gy[2], gz); d.n[2] = d.gx[2] * d.gy[2] * gz; }
  if (d.n[0] + d.n[1] + d.n[2] == 0) return hipSuccess;
  SDQN_LAUNCH((gl_multi_kernel<C0, C1, C2>), dim3(d.n[0] + d.n[1] + d.n[2]), dim3(bt::NT), 0, stream, a, d);
  return hipGetLastError();
}

}  // namespace sdqn
