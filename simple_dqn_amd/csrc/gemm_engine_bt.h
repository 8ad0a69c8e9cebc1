// gemm_engine_bt.h — BLOCK-TILE routine of the tile engine: the throughput regime (B >= 128, BASELINE.json configs[2]).
//
// gemm_tile (gemm_engine.h) gives every 32 x 32 output tile its own workgroup whose waves split K and fetch their operand
// fragments straight from L2: right for a launch that is a handful of latency chains (B = 32), wrong for thousands of tiles —
// every tile re-reads its 32 A rows and 32 B rows (8 FLOP per L2 byte), the fragment-shaped loads touch 32 cache lines per
// instruction, and the B = 256 step sat at 45 % of the fp32 matrix peak with both fused backward launches at 35 %.
// Here (maps and layouts: bt_map.h):
//   * ONE workgroup of 4 waves owns a BM x BN block of C (64 x 64 ... 128 x 128) for the WHOLE K range of its split;
//   * per 32-deep chunk the workgroup stages one A panel and one B panel in LDS with full-line 16-byte loads (each element
//     leaves L2 once per workgroup: 16-32 FLOP per L2 byte) through a ring of D register sets: the global loads of chunk c + D
//     are issued before the MFMAs of chunk c, chunk c + 1 (issued D - 1 chunks ago: landed) is written to the other LDS stage
//     behind them, ONE barrier per chunk.  Round 1's first cut of this routine (tools/exp/big_tiles.patch) had one chunk in
//     flight and scalar LDS traffic and lost to the latency engine: a workgroup waited a loaded-L2 round trip per 0.43 us of
//     matrix work.  D chunks in flight per workgroup x several workgroups per CU (37 KB of LDS at 64 x 64) cover it;
//   * wave (wm, wn) multiplies its SM x SN sub-tiles out of LDS (ds_read_b128 of its own row for k-contiguous panels,
//     conflict-free ds_read_b32 for x-contiguous ones) on v_mfma_f32_32x32x2_f32 — exact fp32, one accumulator per sub-tile
//     over the whole K range (no cross-wave combine: the sum order is k ascending in k-slot order, chunk after chunk);
//   * same problem structs (problems.h) and the same epilogues (P::store16: ReLU, dgrad masks, padded + dense delta scatter,
//     split-K slabs, fused RMSProp of W4, write-through variants) as the latency engine.
// Results differ from gemm_tile's in the last bits only (another partition of the same fp32 sum).
#pragma once
#include <type_traits>
#include "gemm_engine.h"
#include "bt_map.h"

namespace sdqn {

// X_ = 0: v_mfma_f32_32x32x2_f32 (exact fp32: an fmaf chain).  X_ = 9 / 6: the same fp32 operands on PACKED-bf16 MFMA through exact
// three-way splits (problems.h: split_bf16x3 — hi + mid + lo == x exactly): x y = sum_{i,j} x_i y_j with every one of the 9 partial
// products exact in fp32 (8 x 8 significant bits), accumulated in fp32 by v_mfma_f32_32x32x16_bf16 at 16x the fp32 matrix rate:
// 9 instructions of 32 cycles per 16 k instead of 8 of 64 (0.56x the matrix-pipe time); X_ = 6 drops the three products below
// 2^-24 of x y (lo x mid, mid x lo, lo x lo: 0.375x).  The dominant hi x hi products have their own accumulator, the small cross
// terms a second one, added once in the epilogue (small + main): no cross term is ever rounded against the running main sum.
// CPI_ = chunks per barrier interval (1 or 2): with 2 a stage holds two consecutive 32-deep chunks (a 64-deep slice of K), each wave
// issues 32 MFMAs per sub-tile between barriers and the per-interval costs — the barrier itself, the LDS-read latency in front of the
// first MFMA, the staging stores — are paid once per 64 k.  rocprofv3 PMC at B = 256 (tools/exp/pmc_sq_b256.sh): 1.3-2.6 waves per SIMD
// and the matrix pipe busy 47-63 % of a launch — a lone wave per SIMD pays those costs serially.
// problems whose epilogue gates the result with a stored activation (the dgrads) say so: P::GATED, P::gate_load, P::store_gated
template <class P, class = void> struct bt_gated { static constexpr bool value = false; };
template <class P> struct bt_gated<P, decltype((void)P::GATED)> { static constexpr bool value = P::GATED; };

template <class P_, int BM_, int BN_, int WM_, int WN_, int D_ = 2, int X_ = 0, int CPI_ = 1, int UNC_ = 0>
struct BtCfg {
  typedef P_ P;
  // UNC: the ring loads are UNCONDITIONAL (interval index clamped to the last one) — for problems whose chunk count is a run-time value
  // (the weight gradients' K slabs, fc4 forward's K split).  With the guarded form `if (t + D < nit) gload(...)` hipcc has to cover the path
  // WITHOUT new loads and waits with vmcnt(3 .. 0) in front of the LDS stores — on the path with loads that drains the loads just issued:
  // one exposed memory round trip per chunk (conv3_wgrad: 2 640 cycles per chunk for 1 024 of matrix time, tools/exp/bt_stamps.py)
  static constexpr bool UNC = UNC_ != 0;
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr int KIND = 0;                        // bt_run_tile: fp32 panels (bt_tile)
  static constexpr int D = D_;                          // chunks in flight per workgroup: D register sets of (BM + BN) / 32 float4 per thread
  static constexpr int X = X_;
  static constexpr int CPI = CPI_;
  static_assert(CPI == 1 || CPI == 2, "chunks per barrier interval");
  static_assert(X == 0 || X == 6 || X == 9 || X == 19 || X == 16, "fp32 MFMA, or 6 / 9 exact bf16 partial products (19 / 16: TIMING ONLY, no split)");
  static_assert(D >= 1 && D <= 4, "prefetch depth");
  static constexpr int SM = BM / (32 * WM), SN = BN / (32 * WN);
  static_assert(WM * WN * 64 == bt::NT, "four waves per workgroup");
  static_assert(SM >= 1 && SN >= 1 && SM * 32 * WM == BM && SN * 32 * WN == BN, "block = wave grid x sub-tiles of 32 x 32");
  static constexpr int AF = bt::panel_floats(P::A_K, BM), BF = bt::panel_floats(P::B_K, BN);
  static constexpr int PANELS = AF + BF;                // floats of one chunk's two panels
  static constexpr int STAGE = CPI * PANELS;            // floats per LDS stage
  static constexpr int LDS = 2 * STAGE;                 // double-buffered
};

template <class P, class = void> struct has_store_tile { static constexpr bool value = false; };
template <class P> struct has_store_tile<P, decltype((void)P::STORE_TILE)> { static constexpr bool value = P::STORE_TILE; };

__device__ __forceinline__ float4 f4_to_float4(const f4& v) { return make_float4(v.x, v.y, v.z, v.w); }

typedef __bf16 bt_bf16x8 __attribute__((ext_vector_type(8)));
// eight fp32 values -> three bf16 fragments with hi + mid + lo == x exactly (round-to-nearest splits, exact residuals)
__device__ __forceinline__ void split8_bf16x3(const float* x, bt_bf16x8& hi, bt_bf16x8& mid, bt_bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h = (__bf16)x[e]; const float r1 = x[e] - (float)h;
    const __bf16 m = (__bf16)r1; const float r2 = r1 - (float)m;
    hi[e] = h; mid[e] = m; lo[e] = (__bf16)r2;
  }
}

template <class C>
__device__ __forceinline__ void bt_tile(const StepArgs& a, int bx, int by, int bz, float* smem) {
  typedef typename C::P P;
  typedef typename P::aoff_t aoff_t;
  constexpr int BM = C::BM, BN = C::BN, SM = C::SM, SN = C::SN, WN = C::WN;
  constexpr bool AK = P::A_K, BKC = P::B_K;
  constexpr int PA = bt::passes(BM), PB = bt::passes(BN);
  static_assert(sizeof(typename a_elem<P>::type) == 4 && sizeof(typename b_elem<P>::type) == 4, "fp32 operands (the fp16 mode has its own routine)");

  SDQN_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int m0 = bx * BM, n0 = by * BN;
  int z, ks, kbeg, kend;
  P::ksplit(a, bz, z, ks, kbeg, kend);
  const int M = P::M(a), N = P::N(a);

  // ---- loader geometry: the part of every address that does not change from chunk to chunk -------------------------------
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

  // ---- one chunk's panels: global -> registers (ra, rb) -> LDS stage ---------------------------------------------------------
  // loads are unconditional with clamped k (a conditional load costs a branch and a vmcnt(0) each), out-of-range k is zeroed
  constexpr int D = C::D;
  constexpr int CPI = C::CPI;
  float4 ra[D][CPI * PA], rb[D][CPI * PB];
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto gload = [&](int kc, float4* qa, float4* qb) {
    if constexpr (AK) {
      const int k = kc + bt::km_item_k(tid);
      const aoff_t c = P::a_col(a, z, k < kend ? k : kbeg);
#pragma unroll
      for (int p = 0; p < PA; ++p) { qa[p] = f4_to_float4(P::a_load4(a, z, ag[p] + c)); if (k >= kend) qa[p] = zero4; }
    } else {
#pragma unroll
      for (int p = 0; p < PA; ++p) {
        const int k = kc + bt::mk_item_k(BM, tid, p);
        qa[p] = f4_to_float4(P::a_load4(a, z, ag[0] + P::a_col(a, z, k < kend ? k : kbeg)));
        if (k >= kend) qa[p] = zero4;
      }
    }
    if constexpr (BKC) {
      const int k = kc + bt::km_item_k(tid);
      const int r = P::b_row(a, z, k < kend ? k : kbeg);
#pragma unroll
      for (int p = 0; p < PB; ++p) { qb[p] = f4_to_float4(P::b_load4(a, z, bg[p] + r)); if (k >= kend) qb[p] = zero4; }
    } else {
#pragma unroll
      for (int p = 0; p < PB; ++p) {
        const int k = kc + bt::mk_item_k(BN, tid, p);
        qb[p] = f4_to_float4(P::b_load4(a, z, bg[0] + P::b_row(a, z, k < kend ? k : kbeg)));
        if (k >= kend) qb[p] = zero4;
      }
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

  f32x16 acc[SM][SN];
#pragma unroll
  for (int sm = 0; sm < SM; ++sm)
#pragma unroll
    for (int sn = 0; sn < SN; ++sn)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[sm][sn][q] = 0.0f;

  constexpr int X = C::X;
  f32x16 accs[X ? SM : 1][X ? SN : 1];                                  // the small cross terms of the bf16x3 mode
  if constexpr (X != 0) {
#pragma unroll
    for (int sm = 0; sm < SM; ++sm)
#pragma unroll
      for (int sn = 0; sn < SN; ++sn)
#pragma unroll
        for (int q = 0; q < 16; ++q) accs[sm][sn][q] = 0.0f;
  }
  // packed-bf16 form of one chunk: two 16-deep steps; lane (i, h) feeds k = 16 s + 8 h .. + 7 of row / column i of BOTH operands
  // (the same slot <-> k assignment for A and B, so the result does not depend on the hardware's k numbering)
  auto compute_x3 = [&](const float* As, const float* Bs) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      bt_bf16x8 a1[SM], a2[SM], a3[SM], b1[SN], b2[SN], b3[SN];
#pragma unroll
      for (int sm = 0; sm < SM; ++sm) {
        const int x = (wm * SM + sm) * 32 + i;
        float v[8];
        if constexpr (AK) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float4 q = *reinterpret_cast<const float4*>(As + bt::km_off(x, 16 * st + 8 * h + 4 * j));
            v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = As[bt::mk_off(BM, 16 * st + e, x) + h * (8 * BM)];
        }
        split8_bf16x3(v, a1[sm], a2[sm], a3[sm]);
      }
#pragma unroll
      for (int sn = 0; sn < SN; ++sn) {
        const int x = (wn * SN + sn) * 32 + i;
        float v[8];
        if constexpr (BKC) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float4 q = *reinterpret_cast<const float4*>(Bs + bt::km_off(x, 16 * st + 8 * h + 4 * j));
            v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = Bs[bt::mk_off(BN, 16 * st + e, x) + h * (8 * BN)];
        }
        split8_bf16x3(v, b1[sn], b2[sn], b3[sn]);
      }
#pragma unroll
      for (int sm = 0; sm < SM; ++sm)
#pragma unroll
        for (int sn = 0; sn < SN; ++sn) {
          f32x16 m = acc[sm][sn], q = accs[sm][sn];
          if constexpr (X == 9 || X == 19) {
            q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[sm], b3[sn], q, 0, 0, 0);
            q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[sm], b3[sn], q, 0, 0, 0);
            q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[sm], b2[sn], q, 0, 0, 0);
          }
          q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[sm], b3[sn], q, 0, 0, 0);
          m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[sm], b1[sn], m, 0, 0, 0);
          q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[sm], b1[sn], q, 0, 0, 0);
          q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[sm], b2[sn], q, 0, 0, 0);
          q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[sm], b2[sn], q, 0, 0, 0);
          q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[sm], b1[sn], q, 0, 0, 0);
          acc[sm][sn] = m; accs[sm][sn] = q;
        }
    }
  };
  auto compute = [&](const float* As, const float* Bs) {
    if constexpr (X != 0) compute_x3(As, Bs);
    else {
    float fa[SM][16], fb[SN][16];
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
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int sm = 0; sm < SM; ++sm)
#pragma unroll
        for (int sn = 0; sn < SN; ++sn) acc[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sm][t], fb[sn][t], acc[sm][sn], 0, 0, 0);
    }
  };

  // a dgrad's gating activations (delta = (W^T delta') . 1[a > 0]): fetched now, under the K loop, not as dependent loads in the epilogue
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

  // ---- main loop: register set (t mod D) holds interval t = chunks [CPI t, CPI t + CPI); two LDS stages; one barrier per interval ----------
  typename P::Epi epi[SM][SN];
  const int nch = (kend - kbeg + bt::BK - 1) / bt::BK;
  const int nit = (nch + CPI - 1) / CPI;                           // (an odd tail chunk is zero-filled by the loaders: k >= kend)
  auto epi_prefetch = [&]() {
    // whatever the epilogue reads besides the accumulators (W4 / RMSProp state of the fused optimizer) flies under the last interval
#pragma unroll
    for (int sm = 0; sm < SM; ++sm)
#pragma unroll
      for (int sn = 0; sn < SN; ++sn) P::epi_begin(a, m0 + (wm * SM + sm) * 32, n0 + (wn * SN + sn) * 32, lane, epi[sm][sn]);
  };
  auto gload_it = [&](int t, float4* qa, float4* qb) {
#pragma unroll
    for (int cc = 0; cc < CPI; ++cc) gload(kbeg + (t * CPI + cc) * bt::BK, qa + cc * PA, qb + cc * PB);
  };
  auto store_it = [&](const float4* qa, const float4* qb, float* st) {
#pragma unroll
    for (int cc = 0; cc < CPI; ++cc) lds_store(qa + cc * PA, qb + cc * PB, st + cc * C::PANELS, st + cc * C::PANELS + C::AF);
  };
  if (nit > 0) {
#pragma unroll
    for (int d = 0; d < D; ++d) { if constexpr (C::UNC) gload_it(d < nit ? d : nit - 1, ra[d], rb[d]); else if (d < nit) gload_it(d, ra[d], rb[d]); }
    store_it(ra[0], rb[0], smem);
    __syncthreads();
    SDQN_STAMP(1);
    // (unconditional ring loads with a clamped interval index — what the half routines below use — measured 1 % SLOWER here: 217.5 vs
    //  215.4 us per step at B = 256; the fp32 forward problems have compile-time chunk counts and exact waits already)
    for (int t0 = 0; t0 < nit; t0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int t = t0 + d;                                          // workgroup-uniform control flow throughout
        if (t < nit) {
          float* cur = smem + (t & 1) * C::STAGE;
          float* nxt = smem + ((t + 1) & 1) * C::STAGE;
          if constexpr (C::UNC) gload_it(t + D < nit ? t + D : nit - 1, ra[d], rb[d]);       // (clamped: a re-fetch of the last interval that nobody stores)
          else if (t + D < nit) gload_it(t + D, ra[d], rb[d]);          // set d is free: interval t went to LDS one iteration ago
          if (t + 1 == nit) epi_prefetch();
#pragma unroll
          for (int cc = 0; cc < CPI; ++cc) compute(cur + cc * C::PANELS, cur + cc * C::PANELS + C::AF);
          if (t + 1 < nit) { store_it(ra[(d + 1) % D], rb[(d + 1) % D], nxt); __syncthreads(); }
#ifdef SDQN_TIMING
          if (t == 3) SDQN_STAMP(2);
          if (t == 7) SDQN_STAMP(3);
          if (t == 11) SDQN_STAMP(4);
#endif
        }
      }
    }
    SDQN_STAMP(5);
  } else epi_prefetch();

  // ---- epilogue: every sub-tile through the problem's own 32 x 32 epilogue ----------------------------------------------------
#pragma unroll
  for (int sm = 0; sm < SM; ++sm)
#pragma unroll
    for (int sn = 0; sn < SN; ++sn) {
      const int ms = m0 + (wm * SM + sm) * 32, ns = n0 + (wn * SN + sn) * 32;
      if (ms >= M || ns >= N) continue;                               // (wave-uniform) sub-tile entirely outside C
      if constexpr (X != 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[sm][sn][q] = accs[sm][sn][q] + acc[sm][sn][q];
      }
      if constexpr (has_store_tile<P>::value) {                      // the problem's own 32 x 32 epilogue (fc4_wgrad + fused RMSProp, full tiles)
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = acc[sm][sn][q];
        P::store_tile(a, ms, ns, lane, v);
      } else if constexpr (sizeof(typename P::Epi) > 1) {            // an epilogue with prefetched state
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = acc[sm][sn][q];
        P::store16(a, z, ks, ms, ns, lane, M, N, v, epi[sm][sn]);
      } else {
        // P::store of the MOST DERIVED problem (the inherited default store16 would call the base problem's plain stores and lose
        // a write-through variant); 32 lanes along n: one 128-byte row per accumulator register
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


// ---- the float16 mode's forward / dgrad stages (packed-fp16 MFMA, both operands k-contiguous halves: problems_h16.h) --------------------
// Same structure with half panels: chunks of 64 k (a 128-byte row chunk = 8 lanes x 16 bytes, like the fp32 chunk), LDS rows of 72 halves
// (= 36 dwords: the fp32 panel's bank picture), a lane's fragment of a 16-deep step = ONE ds_read_b128, four v_mfma_f32_32x32x16_f16 per
// sub-tile and chunk.  At B >= 128 these launches were operand-traffic bound on the wave-tile routines (every 64 x 64 wave block fetched
// its own rows from L2: round 3's register-blocked routine, tools/exp/gemm_engine_rb.h); here a row leaves L2 once per workgroup.
template <class P_, int BM_, int BN_, int WM_, int WN_, int D_ = 2, int BK_ = 64>
struct BtCfgH {
  typedef P_ P;
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, D = D_, BK = BK_;       // BK halves of k per chunk (64 or 128)
  static constexpr int SM = BM / (32 * WM), SN = BN / (32 * WN);
  static_assert(WM * WN * 64 == bt::NT && SM >= 1 && SN >= 1 && SM * 32 * WM == BM && SN * 32 * WN == BN, "block = 4 waves x sub-tiles of 32 x 32");
  static_assert(uses_f16_mfma<P>::value, "a problems_h16.h forward / dgrad problem: both operands k-contiguous halves (a_load8 / b_load8)");
  static_assert(BK == 64 || BK == 128, "chunk depth");
  // a row of a chunk = BK / 8 pieces of 8 halves; thread -> (row tid / (BK / 8) + rows-per-pass * p, piece tid % (BK / 8)); row pitch BK + 8
  // halves (36 or 68 dwords: consecutive rows shift by 36 / 4 banks: conflict-free 16-byte fragment reads either way)
  static constexpr int PPR = BK / 8, RPP = bt::NT / PPR, PITCH = BK + 8;
  static constexpr int PA = BM / RPP, PB = BN / RPP;
  static_assert(PA * RPP == BM && PB * RPP == BN, "whole passes");
  static constexpr int AH = BM * PITCH, BH = BN * PITCH;
  static constexpr int STAGE = AH + BH;                 // halves per LDS stage
  static constexpr int LDS = 2 * STAGE / 2;             // floats (the kernels declare float arrays): double-buffered
};

template <class C>
__device__ __forceinline__ void bt_tile_h(const StepArgs& a, int bx, int by, int bz, float* smem_f) {
  typedef typename C::P P;
  typedef typename P::aoff_t aoff_t;
  constexpr int BM = C::BM, BN = C::BN, SM = C::SM, SN = C::SN, WN = C::WN, D = C::D, BK = C::BK;
  constexpr int PA = C::PA, PB = C::PB, PPR = C::PPR, RPP = C::RPP, PITCH = C::PITCH;
  half_t* smem = reinterpret_cast<half_t*>(smem_f);
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int m0 = bx * BM, n0 = by * BN;
  int z, ks, kbeg, kend;
  P::ksplit(a, bz, z, ks, kbeg, kend);
  const int M = P::M(a), N = P::N(a);
  const int irow = tid / PPR, ik = (tid - irow * PPR) * 8;           // this thread's piece: rows irow + RPP p, halves ik .. ik + 7
  aoff_t ag[PA]; int bg[PB];
#pragma unroll
  for (int p = 0; p < PA; ++p) { const int m = m0 + irow + RPP * p; ag[p] = P::a_row(a, z, m < M ? m : M - 1); }
#pragma unroll
  for (int p = 0; p < PB; ++p) { const int n = n0 + irow + RPP * p; bg[p] = P::b_col(a, z, n < N ? n : N - 1); }
  half8 ra[D][PA], rb[D][PB];
  const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  auto gload = [&](int kc, half8* qa, half8* qb) {
    const int k = kc + ik, kk = k < kend ? k : kbeg;
    const aoff_t ca = P::a_col(a, z, kk); const int rbk = P::b_row(a, z, kk);
#pragma unroll
    for (int p = 0; p < PA; ++p) { qa[p] = P::a_load8(a, z, ag[p] + ca); if (k >= kend) qa[p] = zero8; }
#pragma unroll
    for (int p = 0; p < PB; ++p) { qb[p] = P::b_load8(a, z, bg[p] + rbk); if (k >= kend) qb[p] = zero8; }
  };
  auto lds_store = [&](const half8* qa, const half8* qb, half_t* As, half_t* Bs) {
#pragma unroll
    for (int p = 0; p < PA; ++p) *reinterpret_cast<half8*>(As + (irow + RPP * p) * PITCH + ik) = qa[p];
#pragma unroll
    for (int p = 0; p < PB; ++p) *reinterpret_cast<half8*>(Bs + (irow + RPP * p) * PITCH + ik) = qb[p];
  };
  f32x16 acc[SM][SN];
#pragma unroll
  for (int sm = 0; sm < SM; ++sm)
#pragma unroll
    for (int sn = 0; sn < SN; ++sn)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[sm][sn][q] = 0.0f;
  // a dgrad's gating activations (delta = (W^T delta') . 1[a > 0]): fetched now, under the K loop, not as dependent loads in the epilogue
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
  auto compute = [&](const half_t* As, const half_t* Bs) {
#pragma unroll
    for (int st = 0; st < BK / 16; ++st) {
      half8 fa[SM], fb[SN];
#pragma unroll
      for (int sm = 0; sm < SM; ++sm) fa[sm] = *reinterpret_cast<const half8*>(As + ((wm * SM + sm) * 32 + i) * PITCH + 16 * st + 8 * h);
#pragma unroll
      for (int sn = 0; sn < SN; ++sn) fb[sn] = *reinterpret_cast<const half8*>(Bs + ((wn * SN + sn) * 32 + i) * PITCH + 16 * st + 8 * h);
#pragma unroll
      for (int sm = 0; sm < SM; ++sm)
#pragma unroll
        for (int sn = 0; sn < SN; ++sn) acc[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[sm], fb[sn], acc[sm][sn], 0, 0, 0);
    }
  };
  const int nch = (kend - kbeg + BK - 1) / BK;
  if (nch > 0) {
    // loads are issued UNCONDITIONALLY (a chunk index past the end re-reads the last chunk and is never stored): with a guarded load and
    // a run-time chunk count (fc4 forward's K split) the compiler waited for every load in flight before each LDS store (bt_tile_hw)
    const int klast = kbeg + (nch - 1) * BK;
    auto chunk_k = [&](int c) { const int k = kbeg + c * BK; return k < klast ? k : klast; };
#pragma unroll
    for (int d = 0; d < D; ++d) gload(chunk_k(d), ra[d], rb[d]);
    lds_store(ra[0], rb[0], smem, smem + C::AH);
    __syncthreads();
    for (int c0 = 0; c0 < nch; c0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int c = c0 + d;
        half_t* cur = smem + (c & 1) * C::STAGE;
        half_t* nxt = smem + ((c + 1) & 1) * C::STAGE;
        gload(chunk_k(c + D), ra[d], rb[d]);
        if (c < nch) compute(cur, cur + C::AH);
        if (c + 1 < nch) { lds_store(ra[(d + 1) % D], rb[(d + 1) % D], nxt, nxt + C::AH); __syncthreads(); }
      }
    }
  }
#pragma unroll
  for (int sm = 0; sm < SM; ++sm)
#pragma unroll
    for (int sn = 0; sn < SN; ++sn) {
      const int ms = m0 + (wm * SM + sm) * 32, ns = n0 + (wn * SN + sn) * 32;
      if (ms >= M || ns >= N) continue;
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

template <class C>
__global__ void __launch_bounds__(bt::NT) bt_kernel_h(const StepArgs a, const int gx, const int gy) {
  __shared__ __attribute__((aligned(16))) float smem[C::LDS];
  const int t = (a.xcd_map & 1) ? xcd_tile_id((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;       // (bt_kernel)
  const int per_z = gx * gy, bz = t / per_z, r = t - bz * per_z;
  bt_tile_h<C>(a, r % gx, r / gx, bz, smem);
}
template <class C>
inline hipError_t launch_bt_h(const StepArgs& a, hipStream_t stream) {
  typedef typename C::P P;
  const int gx = (P::M(a) + C::BM - 1) / C::BM, gy = (P::N(a) + C::BN - 1) / C::BN, gz = P::nbz(a);
  if (gx * gy * gz == 0) return hipSuccess;
  SDQN_LAUNCH((bt_kernel_h<C>), dim3(gx * gy * gz), dim3(bt::NT), 0, stream, a, gx, gy);
  return hipGetLastError();
}

// ---- the float16 mode's WEIGHT GRADIENTS (problems_h16.h: *WgradH): G[m][n] = sum_k A[k][m] * D[k][n], both operands k-major halves ------
// (k = sample / output position: the long axis; a row of A is the im2col patch of one position, contiguous along m in runs of >= 32).
// Packed-fp16 MFMA wants 8 consecutive k per lane for BOTH operands — a transpose of what memory holds.  The wave-tile routine
// (gemm_tile_hw) transposes every 32 x 32 tile through wave-private LDS with 2-byte accesses; here a workgroup stages ONE k-major image
// [64 k][BM] / [64 k][BN] per chunk with 16-byte loads and stores exactly as memory has it, and the fragments come out of LDS through
// gfx950's transpose read: ds_read_b64_tr_b16 hands lane 16 g + t the 4 halves in[16 g + 4 j + (t >> 2)][t & 3], j = 0..3, of the
// 8-byte pieces the 16 lanes of its group address (tools/exp/tr16_probe.hip) — lanes 4 j .. 4 j + 3 point at the four 4-column pieces
// of k-row j, and lane t receives column t of a [4 k][16] block: 4 consecutive k of ONE m.  Two reads = the 8-k fragment of a
// v_mfma_f32_32x32x16_f16.  Row pitch BM + 16 halves: the four 32-byte row pieces a 16-lane group touches fall in disjoint banks.
template <class P_, int BM_, int BN_, int WM_, int WN_, int D_ = 4>
struct BtCfgHW {
  typedef P_ P;
  static constexpr int KIND = 3;                        // bt_run_tile: bt_tile_hw
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, BK = 64, D = D_;     // D chunks of global loads in flight per thread
  static constexpr int SM = BM / (32 * WM), SN = BN / (32 * WN);
  static_assert(WM * WN * 64 == bt::NT && SM >= 1 && SN >= 1 && SM * 32 * WM == BM && SN * 32 * WN == BN, "block = 4 waves x sub-tiles of 32 x 32");
  static_assert(!P::A_K && !P::B_K && (P::A_U8 || sizeof(typename a_elem<P>::type) == 2) && sizeof(typename b_elem<P>::type) == 2, "a *WgradH problem: k-major half operands (conv1: A from the bytes)");
  static constexpr int PITA = BM + 16, PITB = BN + 16;                // halves
  static constexpr int AH = BK * PITA, BH = BK * PITB, STAGE = AH + BH;
  static constexpr int LDS = 2 * STAGE / 2;             // floats: double-buffered
  static constexpr int IA = BK * (BM / 8) / bt::NT, IB = BK * (BN / 8) / bt::NT;     // 16-byte pieces per thread and chunk
  static_assert(IA * bt::NT == BK * (BM / 8) && IB * bt::NT == BK * (BN / 8), "whole pieces per thread");
};

typedef __fp16 bt_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ half8 bt_tr_frag(const half_t* p0, const half_t* p1) {      // two transpose reads -> the 8 consecutive k of one lane
  const bt_fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) bt_fp16x4*)p0);
  const bt_fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) bt_fp16x4*)p1);
  half8 f;
  f[0] = (half_t)lo[0]; f[1] = (half_t)lo[1]; f[2] = (half_t)lo[2]; f[3] = (half_t)lo[3];
  f[4] = (half_t)hi[0]; f[5] = (half_t)hi[1]; f[6] = (half_t)hi[2]; f[7] = (half_t)hi[3];
  return f;
}

template <class C>
__device__ __forceinline__ void bt_tile_hw(const StepArgs& a, int bx, int by, int bz, float* smem_f) {
  typedef typename C::P P;
  typedef typename P::aoff_t aoff_t;
  constexpr int BM = C::BM, BN = C::BN, SM = C::SM, SN = C::SN, WN = C::WN, BK = C::BK, IA = C::IA, IB = C::IB, D = C::D;
  half_t* smem = reinterpret_cast<half_t*>(smem_f);
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int m0 = bx * BM, n0 = by * BN;
  int z, ks, kbeg, kend;
  P::ksplit(a, bz, z, ks, kbeg, kend);
  const int M = P::M(a), N = P::N(a);
  const half_t* bbase = P::b_ptr(a, z);
  // A pieces: 8 halves — or, for conv1 (A_U8), the 8 BYTES of a patch row (converted to half(x / 255), as the forward pass does, when the
  // piece is stored to LDS)
  struct APiece { half8 v; };
  struct UPiece { uint32_t lo, hi; };
  typedef typename std::conditional<P::A_U8, UPiece, APiece>::type piece_t;
  auto a_fetch = [&](aoff_t off) {
    piece_t r;
    if constexpr (P::A_U8) { const uint32_t* q = reinterpret_cast<const uint32_t*>(a.src + off); r.lo = q[0]; r.hi = q[1]; }
    else r.v = ld_half8(reinterpret_cast<const half_t*>(P::a_ptr(a, z)) + off);
    return r;
  };
  auto a_zero = [&]() { piece_t r; if constexpr (P::A_U8) { r.lo = 0u; r.hi = 0u; } else { r.v = half8{0, 0, 0, 0, 0, 0, 0, 0}; } return r; };
  auto a_halves = [&](const piece_t& r) {
    if constexpr (P::A_U8) {
      half8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = (half_t)norm_u8((r.lo >> (8 * e)) & 255u); o[4 + e] = (half_t)norm_u8((r.hi >> (8 * e)) & 255u); }
      return o;
    } else return r.v;
  };
  // staging pieces of this thread: piece q = tid + NT * p -> k-row q / (BM / 8), 8 columns from 8 * (q % (BM / 8))
  aoff_t ag[IA]; int bg[IB]; int ak[IA], bk[IB], alds[IA], blds[IB];
#pragma unroll
  for (int p = 0; p < IA; ++p) {
    const int q = tid + bt::NT * p; ak[p] = q / (BM / 8); const int c8 = 8 * (q - ak[p] * (BM / 8)), m = m0 + c8;
    ag[p] = P::a_row(a, z, m < M ? m : M - 8); alds[p] = ak[p] * C::PITA + c8;
  }
#pragma unroll
  for (int p = 0; p < IB; ++p) {
    const int q = tid + bt::NT * p; bk[p] = q / (BN / 8); const int c8 = 8 * (q - bk[p] * (BN / 8)), n = n0 + c8;
    bg[p] = P::b_col(a, z, n < N ? n : N - 8); blds[p] = bk[p] * C::PITB + c8;
  }
  piece_t ra[D][IA]; half8 rb[D][IB];
  const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  auto gload = [&](int kc, piece_t* qa, half8* qb) {
#pragma unroll
    for (int p = 0; p < IA; ++p) {
      const int k = kc + ak[p], kk = k < kend ? k : kbeg;
      qa[p] = a_fetch(ag[p] + P::a_col(a, z, kk));
      if (k >= kend) qa[p] = a_zero();
    }
#pragma unroll
    for (int p = 0; p < IB; ++p) {
      const int k = kc + bk[p], kk = k < kend ? k : kbeg;
      qb[p] = ld_half8(bbase + (bg[p] + P::b_row(a, z, kk)));
      if (k >= kend) qb[p] = zero8;
    }
  };
  auto lds_store = [&](const piece_t* qa, const half8* qb, half_t* As, half_t* Bs) {
#pragma unroll
    for (int p = 0; p < IA; ++p) *reinterpret_cast<half8*>(As + alds[p]) = a_halves(qa[p]);
#pragma unroll
    for (int p = 0; p < IB; ++p) *reinterpret_cast<half8*>(Bs + blds[p]) = qb[p];
  };
  typename P::Epi epi;
  constexpr bool EPI = sizeof(typename P::Epi) > 1;
  if constexpr (EPI) { static_assert(SM == 1 && SN == 1, "one epilogue state per wave"); P::epi_begin(a, m0 + wm * 32, n0 + wn * 32, lane, epi); }
  f32x16 acc[SM][SN];
#pragma unroll
  for (int sm = 0; sm < SM; ++sm)
#pragma unroll
    for (int sn = 0; sn < SN; ++sn)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[sm][sn][q] = 0.0f;
  // fragment addresses: 16-lane group G = lane >> 4 covers columns 16 (G & 1) .. + 15 and k 8 (G >> 1) .. + 7 of a step; lane t of the
  // group points at k-row (t >> 2) (+ 4 for the second read), columns 4 (t & 3) .. + 3
  const int t = lane & 15, G = lane >> 4;
  const int frow = 8 * (G >> 1) + (t >> 2), fcol = 16 * (G & 1) + 4 * (t & 3);
  auto compute = [&](const half_t* As, const half_t* Bs) {
#pragma unroll
    for (int st = 0; st < BK / 16; ++st) {
      half8 fa[SM], fb[SN];
#pragma unroll
      for (int sm = 0; sm < SM; ++sm) {
        const half_t* p = As + (16 * st + frow) * C::PITA + (wm * SM + sm) * 32 + fcol;
        fa[sm] = bt_tr_frag(p, p + 4 * C::PITA);
      }
#pragma unroll
      for (int sn = 0; sn < SN; ++sn) {
        const half_t* p = Bs + (16 * st + frow) * C::PITB + (wn * SN + sn) * 32 + fcol;
        fb[sn] = bt_tr_frag(p, p + 4 * C::PITB);
      }
#pragma unroll
      for (int sm = 0; sm < SM; ++sm)
#pragma unroll
        for (int sn = 0; sn < SN; ++sn) acc[sm][sn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[sm], fb[sn], acc[sm][sn], 0, 0, 0);
    }
  };
  const int nch = (kend - kbeg + BK - 1) / BK;
  if (nch > 0) {
    // every load is issued UNCONDITIONALLY (a chunk index past the end re-reads the last chunk and is never stored): the number of loads
    // in flight at each LDS store is then a compile-time constant and the compiler waits for exactly the oldest set (vmcnt = 4 (D - 1)),
    // not for everything — with guarded loads it emitted vmcnt(0) and every iteration paid a full round trip
    const int klast = kbeg + (nch - 1) * BK;
    auto chunk_k = [&](int c) { const int k = kbeg + c * BK; return k < klast ? k : klast; };
#pragma unroll
    for (int d = 0; d < D; ++d) gload(chunk_k(d), ra[d], rb[d]);
    lds_store(ra[0], rb[0], smem, smem + C::AH);
    __syncthreads();
    for (int c0 = 0; c0 < nch; c0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int c = c0 + d;
        half_t* cur = smem + (c & 1) * C::STAGE;
        half_t* nxt = smem + ((c + 1) & 1) * C::STAGE;
        gload(chunk_k(c + D), ra[d], rb[d]);                                   // (set d held chunk c: in LDS since the last iteration)
        if (c < nch) compute(cur, cur + C::AH);
        if (c + 1 < nch) { lds_store(ra[(d + 1) % D], rb[(d + 1) % D], nxt, nxt + C::AH); __syncthreads(); }
      }
    }
  }
#pragma unroll
  for (int sm = 0; sm < SM; ++sm)
#pragma unroll
    for (int sn = 0; sn < SN; ++sn) {
      const int ms = m0 + (wm * SM + sm) * 32, ns = n0 + (wn * SN + sn) * 32;
      if (ms >= M || ns >= N) continue;
      if constexpr (EPI) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = acc[sm][sn][q];
        P::store16(a, z, ks, ms, ns, lane, M, N, v, epi);
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int m = ms + bt::acc_row(q, h), n = ns + i;
          if (m < M && n < N) P::store(a, z, ks, m, n, acc[sm][sn][q]);
        }
      }
    }
}

// workgroups of problem P at block size BM x BN
template <class C> inline void bt_grid(const StepArgs& a, int& gx, int& gy, int& gz) {
  typedef typename C::P P;
  gx = (P::M(a) + C::BM - 1) / C::BM; gy = (P::N(a) + C::BN - 1) / C::BN; gz = P::nbz(a);
}

template <class C>
__global__ void __launch_bounds__(bt::NT) bt_kernel(const StepArgs a, const int gx, const int gy) {
  __shared__ __attribute__((aligned(16))) float smem[C::LDS];
  if constexpr (has_preload<typename C::P>::value) C::P::preload(a, gridDim.x, (unsigned)gx, (unsigned)gy);
  // XCD-contiguous block map (StepArgs::xcd_map bit 0): workgroup b runs on XCD b % 8; every XCD gets ONE contiguous run of block ids
  // (m-fastest), so the blocks that share a B panel (and neighbouring A rows) share an L2 instead of fetching it into eight
  const int t = (a.xcd_map & 1) ? xcd_tile_id((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int per_z = gx * gy, bz = t / per_z, r = t - bz * per_z;
  bt_tile<C>(a, r % gx, r / gx, bz, smem);
}

template <class C>
inline hipError_t launch_bt(const StepArgs& a, hipStream_t stream) {
  int gx, gy, gz; bt_grid<C>(a, gx, gy, gz);
  if (gx * gy * gz == 0) return hipSuccess;
  SDQN_LAUNCH((bt_kernel<C>), dim3(gx * gy * gz), dim3(bt::NT), 0, stream, a, gx, gy);
  return hipGetLastError();
}

// ---- several independent problems in ONE launch (bwd3 = fc4_wgrad || conv3_dgrad || conv3_wgrad, bwd2 = conv2_dgrad || conv2_wgrad):
// block-id ranges dispatch to different configurations; the LDS footprint is the largest one's
template <class C>
__device__ __forceinline__ void bt_run_tile(const StepArgs& a, int bx, int by, int bz, float* smem) {
  if constexpr (C::KIND == 3) bt_tile_hw<C>(a, bx, by, bz, smem); else
  bt_tile<C>(a, bx, by, bz, smem);
}
template <class C0, class C1, class C2>
__global__ void __launch_bounds__(bt::NT) bt_multi_kernel(const StepArgs a, const MultiDims d) {
  constexpr int L01 = C0::LDS > C1::LDS ? C0::LDS : C1::LDS, L = L01 > C2::LDS ? L01 : C2::LDS;
  __shared__ __attribute__((aligned(16))) float smem[L];
  if constexpr (has_preload_multi<typename C1::P>::value) C1::P::preload_multi(a, d);
  const int b = blockIdx.x, xm = a.xcd_map;              // bit i: problem i on the XCD-contiguous block map
  if (b < d.n[0]) { const int l = (xm & 1) ? xcd_tile_id_range(b, 0, d.n[0]) : b, pz = d.gx[0] * d.gy[0], bz = l / pz, r = l - bz * pz; bt_run_tile<C0>(a, r % d.gx[0], r / d.gx[0], bz, smem); }
  else if (b < d.n[0] + d.n[1]) { const int l = (xm & 2) ? xcd_tile_id_range(b, d.n[0], d.n[1]) : b - d.n[0], pz = d.gx[1] * d.gy[1], bz = l / pz, r = l - bz * pz; bt_run_tile<C1>(a, r % d.gx[1], r / d.gx[1], bz, smem); }
  else { const int l = (xm & 4) ? xcd_tile_id_range(b, d.n[0] + d.n[1], d.n[2]) : b - d.n[0] - d.n[1], pz = d.gx[2] * d.gy[2], bz = l / pz, r = l - bz * pz; bt_run_tile<C2>(a, r % d.gx[2], r / d.gx[2], bz, smem); }
}

template <class C0, class C1, class C2>
inline hipError_t launch_bt_multi(const StepArgs& a, bool has0, bool has1, bool has2, hipStream_t stream) {
  MultiDims d; memset(&d, 0, sizeof d);
  int gz;
  if (has0) { bt_grid<C0>(a, d.gx[0], d.gy[0], gz); d.n[0] = d.gx[0] * d.gy[0] * gz; }
  if (has1) { bt_grid<C1>(a, d.gx[1], d.gy[1], gz); d.n[1] = d.gx[1] * d.gy[1] * gz; }
  if (has2) { bt_grid<C2>(a, d.gx[2], d.gy[2], gz); d.n[2] = d.gx[2] * d.gy[2] * gz; }
  if (d.n[0] + d.n[1] + d.n[2] == 0) return hipSuccess;
  SDQN_LAUNCH((bt_multi_kernel<C0, C1, C2>), dim3(d.n[0] + d.n[1] + d.n[2]), dim3(bt::NT), 0, stream, a, d);
  return hipGetLastError();
}

}  // namespace sdqn
