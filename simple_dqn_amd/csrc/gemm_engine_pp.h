// gemm_engine_pp.h — PING-PONG form of the block-tile routine (gemm_engine_bt.h) for the throughput regime (B >= 128, float32).
//
// bt_tile runs ONE wave per SIMD and workgroup; a chunk is a serial string for that wave — fragment reads (LDS latency in front of the
// first MFMA), 16 MFMAs (1 024 matrix-pipe cycles), LDS stores of the next chunk, the workgroup barrier — and with one or two workgroups
// per CU (196 ... 650 blocks on 256 CUs is all these layers offer at B = 256) nothing covers the ~500 cycles around the MFMAs: the
// matrix pipes are busy 47-63 % of a launch (rocprofv3 PMC, tools/exp/pmc_sq_b256.sh), fc4_dgrad on bt_tile spends 2 170 cycles per chunk.
// Here a workgroup has EIGHT waves, two per SIMD, as two groups of four with the same BM x BN wave grid:
//   * group g owns the chunks c = g (mod 2) of the workgroup's K range, with its own accumulators; the two partial sums are added once
//     at the end (even chunks + odd chunks, fixed order: deterministic);
//   * phase p (one per chunk, ONE barrier per phase): group p & 1 issues the 16 MFMAs of chunk p out of fragment registers it filled
//     during phase p - 1, while the OTHER group reads its fragments of chunk p + 1 out of LDS — a SIMD's matrix pipe always has one
//     of its two waves in a pure MFMA segment and the other one's LDS latency, waits and barrier arrival cost it nothing (the pairing
//     MI355X_MICROARCH.md "Two waves per SIMD" describes: compute segment against load segment, separated by s_barrier);
//   * all 512 threads are loaders (half the float4 per thread): in phase p chunk p + 2 (issued D phases ago) is stored into LDS stage
//     p & 1 — free since the barrier that ended phase p - 1, when its last readers had chunk p's fragments in registers — and its
//     register set is re-issued for chunk p + 2 + D; chunk p + 1 sits in the other stage, being read;
//   * same panels, fragment maps and k-slot order as bt_tile (bt_map.h), same problem structs and epilogues; the epilogue runs on
//     group 0 after the partial sums are combined through the (then idle) stage memory.
// Per output element the sum is (chunks 0, 2, 4, ... in order) + (chunks 1, 3, 5, ... in order): another partition of the same fp32 sum
// than bt_tile's, so results differ from it in the last bits — every block shape of THIS routine produces the same bits.
//
// VERDICT (round 4, MI355X, B = 256; tools/exp/README.md): correct on the first run (gradients 5e-7 of bt_tile's), SLOWER in every launch —
// fc4_dgrad 18.6 us (bt_tile 17.0, latency engine 15.1), conv3_fwd 26.4 (23.2), conv2_fwd 35.1 (29.3), bwd3 60 (40), bwd2 62 (40: 129-139
// VGPRs leave one 512-thread workgroup per CU).  s_memtime stamps (tools/exp/bt_stamps.py) put a chunk at ~1 900 cycles in BOTH routines
// for fc4_dgrad (1 024 of them matrix time), whatever the prefetch depth: the phases wait for operand delivery, not for each other, and a
// second wave per SIMD doubles the address / staging VALU work without shortening that.  Experiments build only (menu entries 10-12).
#pragma once
#include "gemm_engine_bt.h"

namespace sdqn {

template <class P_, int BM_, int BN_, int WM_, int WN_, int D_ = 2>
struct PpCfg {
  typedef P_ P;
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, D = D_;
  static constexpr int KIND = 4;                        // ping-pong (pp_tile)
  static_assert(D >= 2 && D <= 3, "register sets of chunks in flight");
  static constexpr int SM = BM / (32 * WM), SN = BN / (32 * WN);
  static_assert(WM * WN == 4, "each group is a 4-wave grid");
  static_assert(SM >= 1 && SN >= 1 && SM * 32 * WM == BM && SN * 32 * WN == BN, "block = wave grid x sub-tiles of 32 x 32");
  static constexpr int AF = bt::panel_floats(P::A_K, BM), BF = bt::panel_floats(P::B_K, BN);
  static constexpr int STAGE = AF + BF;                 // floats per LDS stage (one chunk's two panels)
  static constexpr int COMB = 4 * SM * SN * 16 * 64;    // floats of group 1's accumulators on their way to group 0
  static constexpr int LDS = 2 * STAGE > COMB ? 2 * STAGE : COMB;
};

template <class C>
__device__ __forceinline__ void pp_tile(const StepArgs& a, int bx, int by, int bz, float* smem) {
  typedef typename C::P P;
  typedef typename P::aoff_t aoff_t;
  constexpr int BM = C::BM, BN = C::BN, SM = C::SM, SN = C::SN, WN = C::WN, D = C::D;
  constexpr bool AK = P::A_K, BKC = P::B_K;
  constexpr int PA = bt::pp_passes(BM), PB = bt::pp_passes(BN);
  static_assert(sizeof(typename a_elem<P>::type) == 4 && sizeof(typename b_elem<P>::type) == 4, "fp32 operands");

  SDQN_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave8 >> 2, wave = wave8 & 3;           // group = chunk parity this wave computes; wave of the group's grid
  const int wm = wave / WN, wn = wave - wm * WN;
  const int m0 = bx * BM, n0 = by * BN;
  int z, ks, kbeg, kend;
  P::ksplit(a, bz, z, ks, kbeg, kend);
  const int M = P::M(a), N = P::N(a);

  // ---- loader geometry (bt_map.h: pp_* items; all 512 threads) ---------------------------------------------------------------------
  aoff_t ag[PA]; int bg[PB];
  if constexpr (AK) {
#pragma unroll
    for (int p = 0; p < PA; ++p) { const int m = m0 + bt::pp_km_row(bt::pp_item(BM, tid, p)); ag[p] = P::a_row(a, z, m < M ? m : M - 1); }
  } else {
    const int m = m0 + bt::pp_mk_x(BM, bt::pp_item(BM, tid, 0));
    ag[0] = P::a_row(a, z, m + 4 <= M ? m : M - 4);
  }
  if constexpr (BKC) {
#pragma unroll
    for (int p = 0; p < PB; ++p) { const int n = n0 + bt::pp_km_row(bt::pp_item(BN, tid, p)); bg[p] = P::b_col(a, z, n < N ? n : N - 1); }
  } else {
    const int n = n0 + bt::pp_mk_x(BN, bt::pp_item(BN, tid, 0));
    bg[0] = P::b_col(a, z, n + 4 <= N ? n : N - 4);
  }
  float4 ra[D][PA], rb[D][PB];
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // unconditional loads with clamped k (a chunk past the end is a chunk of zeros that nobody stores)
  auto gload = [&](int c, float4* qa, float4* qb) {
    const int kc = kbeg + c * bt::BK;
    if constexpr (AK) {
      const int k = kc + bt::pp_km_k(bt::pp_item(BM, tid, 0));
      const aoff_t col = P::a_col(a, z, k < kend ? k : kbeg);
#pragma unroll
      for (int p = 0; p < PA; ++p) { qa[p] = f4_to_float4(P::a_load4(a, z, ag[p] + col)); if (k >= kend) qa[p] = zero4; }
    } else {
#pragma unroll
      for (int p = 0; p < PA; ++p) {
        const int k = kc + bt::pp_mk_k(BM, bt::pp_item(BM, tid, p));
        qa[p] = f4_to_float4(P::a_load4(a, z, ag[0] + P::a_col(a, z, k < kend ? k : kbeg)));
        if (k >= kend) qa[p] = zero4;
      }
    }
    if constexpr (BKC) {
      const int k = kc + bt::pp_km_k(bt::pp_item(BN, tid, 0));
      const int r = P::b_row(a, z, k < kend ? k : kbeg);
#pragma unroll
      for (int p = 0; p < PB; ++p) { qb[p] = f4_to_float4(P::b_load4(a, z, bg[p] + r)); if (k >= kend) qb[p] = zero4; }
    } else {
#pragma unroll
      for (int p = 0; p < PB; ++p) {
        const int k = kc + bt::pp_mk_k(BN, bt::pp_item(BN, tid, p));
        qb[p] = f4_to_float4(P::b_load4(a, z, bg[0] + P::b_row(a, z, k < kend ? k : kbeg)));
        if (k >= kend) qb[p] = zero4;
      }
    }
  };
  auto lds_store = [&](const float4* qa, const float4* qb, float* st) {
    float* As = st; float* Bs = st + C::AF;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int it = bt::pp_item(BM, tid, p);
      const int o = AK ? bt::km_off(bt::pp_km_row(it), bt::pp_km_k(it)) : bt::mk_off(BM, bt::pp_mk_k(BM, it), bt::pp_mk_x(BM, it));
      *reinterpret_cast<float4*>(As + o) = qa[p];
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int it = bt::pp_item(BN, tid, p);
      const int o = BKC ? bt::km_off(bt::pp_km_row(it), bt::pp_km_k(it)) : bt::mk_off(BN, bt::pp_mk_k(BN, it), bt::pp_mk_x(BN, it));
      *reinterpret_cast<float4*>(Bs + o) = qb[p];
    }
  };

  // ---- fragments of one chunk (bt_tile's maps) and its 16 MFMA steps ------------------------------------------------------------------
  float fa[SM][16], fb[SN][16];
  auto read_frags = [&](const float* st) {
    const float* As = st; const float* Bs = st + C::AF;
#pragma unroll
    for (int sm = 0; sm < SM; ++sm) {
      const int x = (wm * SM + sm) * 32 + i;
      if constexpr (AK) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 v = *reinterpret_cast<const float4*>(As + bt::km_off(x, 8 * j + 4 * h));
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
          const float4 v = *reinterpret_cast<const float4*>(Bs + bt::km_off(x, 8 * j + 4 * h));
          fb[sn][4 * j] = v.x; fb[sn][4 * j + 1] = v.y; fb[sn][4 * j + 2] = v.z; fb[sn][4 * j + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) fb[sn][t] = Bs[bt::mk_off(BN, bt::kslot(t, 0), x) + h * (4 * BN)];
      }
    }
  };
  f32x16 acc[SM][SN];
#pragma unroll
  for (int sm = 0; sm < SM; ++sm)
#pragma unroll
    for (int sn = 0; sn < SN; ++sn)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[sm][sn][q] = 0.0f;
  auto mfmas = [&]() {
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int sm = 0; sm < SM; ++sm)
#pragma unroll
        for (int sn = 0; sn < SN; ++sn) acc[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sm][t], fb[sn][t], acc[sm][sn], 0, 0, 0);
  };

  // a dgrad's gating activations: fetched now by the group that will run the epilogue (bt_tile)
  constexpr bool GATED = bt_gated<P>::value;
  float gate[GATED ? SM : 1][GATED ? SN : 1][16];
  if constexpr (GATED) {
    if (grp == 0) {
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
  }
  (void)gate;

  typename P::Epi epi[SM][SN];
  const int nch = (kend - kbeg + bt::BK - 1) / bt::BK;
  float* const st0 = smem; float* const st1 = smem + C::STAGE;
  if (nch > 0) {
    // prologue: chunks 0 and 1 into their stages, chunks 2 .. D + 1 in flight, group 0 takes chunk 0's fragments
#pragma unroll
    for (int d = 0; d < D; ++d) gload(d, ra[d], rb[d]);
    lds_store(ra[0], rb[0], st0);
    gload(D, ra[0], rb[0]);                                               // set 0 is free again: chunk D
    if (nch > 1) lds_store(ra[1 % D], rb[1 % D], st1);
    gload(D + 1, ra[1 % D], rb[1 % D]);                                   // chunk D + 1
    __syncthreads();
    if (grp == 0) read_frags(st0);
    __syncthreads();                                                      // stage 0 is rewritten in phase 0
    SDQN_STAMP(1);
    // phase p: chunk p computed by group p & 1; the other group fetches chunk p + 1's fragments; chunk p + 2 goes into stage p & 1 (its register
    // set (p + 2) % D was loaded D - 1 phases ago and is re-issued for chunk p + 2 + D)
    for (int p0 = 0; p0 < nch; p0 += 2 * D) {
#pragma unroll
      for (int u = 0; u < 2 * D; ++u) {
        const int p = p0 + u;                                             // u & 1 == p & 1, (u + 2) % D == (p + 2) % D  (2 D | p0)
        if (p < nch) {                                                    // workgroup-uniform
          float* cur = (u & 1) ? st1 : st0;                               // stage of chunk p (and of chunk p + 2)
          float* oth = (u & 1) ? st0 : st1;                               // stage of chunk p + 1
          const int s = (u + 2) % D;
          if (grp == (u & 1)) {
            if (p + 2 < nch) lds_store(ra[s], rb[s], cur);
            gload(p + 2 + D, ra[s], rb[s]);
            if (p + 1 >= nch) {                                           // last phase: what the epilogue reads flies under the MFMAs
              if (grp == 0) {
#pragma unroll
                for (int sm = 0; sm < SM; ++sm)
#pragma unroll
                  for (int sn = 0; sn < SN; ++sn) P::epi_begin(a, m0 + (wm * SM + sm) * 32, n0 + (wn * SN + sn) * 32, lane, epi[sm][sn]);
              }
            }
            mfmas();
          } else {
            if (p + 1 < nch) read_frags(oth);
            if (p + 2 < nch) lds_store(ra[s], rb[s], cur);
            gload(p + 2 + D, ra[s], rb[s]);
            if (p + 1 >= nch && grp == 0) {
#pragma unroll
              for (int sm = 0; sm < SM; ++sm)
#pragma unroll
                for (int sn = 0; sn < SN; ++sn) P::epi_begin(a, m0 + (wm * SM + sm) * 32, n0 + (wn * SN + sn) * 32, lane, epi[sm][sn]);
            }
          }
          __syncthreads();
#ifdef SDQN_TIMING
          if (p == 3) SDQN_STAMP(2);
          if (p == 7) SDQN_STAMP(3);
          if (p == 11) SDQN_STAMP(4);
#endif
        }
      }
    }
    SDQN_STAMP(5);
  } else if (grp == 0) {
#pragma unroll
    for (int sm = 0; sm < SM; ++sm)
#pragma unroll
      for (int sn = 0; sn < SN; ++sn) P::epi_begin(a, m0 + (wm * SM + sm) * 32, n0 + (wn * SN + sn) * 32, lane, epi[sm][sn]);
  }

  // ---- the two partial sums: group 1's accumulators through LDS, group 0 adds (even chunks + odd chunks) ----------------------------
  // (the barrier that ended the last phase: no fragment read is outstanding, the stages are idle)
  if (nch > 1) {
    float* cb = smem + (size_t)wave * (SM * SN * 16 * 64) + lane;
    if (grp == 1) {
#pragma unroll
      for (int sm = 0; sm < SM; ++sm)
#pragma unroll
        for (int sn = 0; sn < SN; ++sn)
#pragma unroll
          for (int q = 0; q < 16; ++q) cb[((sm * SN + sn) * 16 + q) * 64] = acc[sm][sn][q];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int sm = 0; sm < SM; ++sm)
#pragma unroll
        for (int sn = 0; sn < SN; ++sn)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[sm][sn][q] = acc[sm][sn][q] + cb[((sm * SN + sn) * 16 + q) * 64];
    }
  }
  SDQN_STAMP(6);
  if (grp != 0) return;

  // ---- epilogue (group 0): every sub-tile through the problem's own 32 x 32 epilogue, as bt_tile ---------------------------------------
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
  SDQN_STAMP(7);
}

template <class C>
__global__ void __launch_bounds__(bt::NT2) pp_kernel(const StepArgs a, const int gx, const int gy) {
  __shared__ __attribute__((aligned(16))) float smem[C::LDS];
  if constexpr (has_preload<typename C::P>::value) C::P::preload(a, gridDim.x, (unsigned)gx, (unsigned)gy);
  const int t = (a.xcd_map & 1) ? xcd_tile_id((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;       // (bt_kernel)
  const int per_z = gx * gy, bz = t / per_z, r = t - bz * per_z;
  pp_tile<C>(a, r % gx, r / gx, bz, smem);
}
template <class C>
inline hipError_t launch_pp(const StepArgs& a, hipStream_t stream) {
  int gx, gy, gz; bt_grid<C>(a, gx, gy, gz);
  if (gx * gy * gz == 0) return hipSuccess;
  SDQN_LAUNCH((pp_kernel<C>), dim3(gx * gy * gz), dim3(bt::NT2), 0, stream, a, gx, gy);
  return hipGetLastError();
}
// several independent problems in ONE launch (bwd3, bwd2), as bt_multi_kernel
template <class C0, class C1, class C2>
__global__ void __launch_bounds__(bt::NT2) pp_multi_kernel(const StepArgs a, const MultiDims d) {
  constexpr int L01 = C0::LDS > C1::LDS ? C0::LDS : C1::LDS, L = L01 > C2::LDS ? L01 : C2::LDS;
  __shared__ __attribute__((aligned(16))) float smem[L];
  if constexpr (has_preload_multi<typename C1::P>::value) C1::P::preload_multi(a, d);
  const int b = blockIdx.x, xm = a.xcd_map;
  if (b < d.n[0]) { const int l = (xm & 1) ? xcd_tile_id_range(b, 0, d.n[0]) : b, pz = d.gx[0] * d.gy[0], bz = l / pz, r = l - bz * pz; pp_tile<C0>(a, r % d.gx[0], r / d.gx[0], bz, smem); }
  else if (b < d.n[0] + d.n[1]) { const int l = (xm & 2) ? xcd_tile_id_range(b, d.n[0], d.n[1]) : b - d.n[0], pz = d.gx[1] * d.gy[1], bz = l / pz, r = l - bz * pz; pp_tile<C1>(a, r % d.gx[1], r / d.gx[1], bz, smem); }
  else { const int l = (xm & 4) ? xcd_tile_id_range(b, d.n[0] + d.n[1], d.n[2]) : b - d.n[0] - d.n[1], pz = d.gx[2] * d.gy[2], bz = l / pz, r = l - bz * pz; pp_tile<C2>(a, r % d.gx[2], r / d.gx[2], bz, smem); }
}
template <class C0, class C1, class C2>
inline hipError_t launch_pp_multi(const StepArgs& a, bool has0, bool has1, bool has2, hipStream_t stream) {
  MultiDims d; memset(&d, 0, sizeof d);
  int gz;
  if (has0) { bt_grid<C0>(a, d.gx[0], d.gy[0], gz); d.n[0] = d.gx[0] * d.gy[0] * gz; }
  if (has1) { bt_grid<C1>(a, d.gx[1], d.gy[1], gz); d.n[1] = d.gx[1] * d.gy[1] * gz; }
  if (has2) { bt_grid<C2>(a, d.gx[2], d.gy[2], gz); d.n[2] = d.gx[2] * d.gy[2] * gz; }
  if (d.n[0] + d.n[1] + d.n[2] == 0) return hipSuccess;
  SDQN_LAUNCH((pp_multi_kernel<C0, C1, C2>), dim3(d.n[0] + d.n[1] + d.n[2]), dim3(bt::NT2), 0, stream, a, d);
  return hipGetLastError();
}

}  // namespace sdqn
