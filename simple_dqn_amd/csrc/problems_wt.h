// problems_wt.h — the write-through (sc1) epilogue variants of the fp32 problems (round 3), shared by the latency-regime launches
// (sdqn_kernels_r3.hip) and the block-tile engine of the throughput regime (sdqn_kernels_bt.hip).
#pragma once
#include "gemm_engine.h"

namespace sdqn {

// ---- write-through epilogues ----------------------------------------------------------------------------------------------------
// A kernel boundary writes back every dirty L2 line its predecessor left; a stage whose output leaves with write-through (sc1) stores
// while it still computes has nothing left to flush (tools/exp/handoff_r3.hip section D: ~0.03 us per MB at the next boundary; in the
// step, where the next launch waits for exactly those bytes, a1 alone was worth 0.7 us).  Same problems, same arithmetic, only the
// store instruction of the epilogue differs: results are bit-identical.  LaunchTune::wt selects them per launch.
#ifndef SDQN_NT_W4
#define SDQN_NT_W4 0
#endif
constexpr bool NT_W4 = SDQN_NT_W4 != 0;      // experiment: non-temporal loads of the streamed W4 operand
#ifndef SDQN_PRELOAD
#define SDQN_PRELOAD 1
#endif
#define SDQN_TOUCH(...) asm volatile("" :: __VA_ARGS__)
// the whole bwd3 / bwd2 launch in ONE statement (an asm statement takes at most 30 operands): every pointer and scalar any of its problems reads
#define SDQN_PRELOAD_MULTI_DEF \
  static constexpr bool PRELOAD_MULTI = SDQN_PRELOAD != 0; \
  __device__ static void preload_multi(const StepArgs& a, const MultiDims& d) { \
    SDQN_TOUCH("s"(a.d3p), "s"(a.theta[0]), "s"(a.a2), "s"(a.d2p), "s"(a.d2), "s"(a.d3), "s"(a.slab3), "s"(a.a3), "s"(a.d4), "s"(a.theta_w), "s"(a.state), \
               "s"(a.g), "s"(a.a1), "s"(a.d1), "s"(a.slab2), "s"(a.B), "s"(a.tps2), "s"(a.tps3), "s"(a.fuse_rms), "s"(a.f4w_first), "s"(a.f4w_count), \
               "s"(a.xcd_map), "s"(a.bsz), "s"(a.rho), "s"(a.one_minus_rho), "s"(a.lr), "s"(a.eps), "s"(d.n[0]), "s"(d.n[1]), "s"(d.gx[1])); \
  }
#ifdef SDQN_WT_PLAIN      // experiment build only: every "write-through" epilogue store is a plain store (what do 4-byte sc1 stores cost at B = 256?)
__device__ __forceinline__ void wt_store(float* p, float v) { *p = v; }
#else
__device__ __forceinline__ void wt_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
struct Conv2FwdWT : Conv2Fwd {
  static constexpr bool PRELOAD = SDQN_PRELOAD != 0;
  __device__ static void preload(const StepArgs& a, unsigned g0, unsigned g1, unsigned g2) { SDQN_TOUCH("s"(a.a1), "s"(a.a2), "s"(a.theta[0]), "s"(a.theta[1]), "s"(a.B), "s"(a.nz), "s"(a.xcd_map), "s"(g0), "s"(g1), "s"(g2)); }

  __device__ static void store(const StepArgs& a, int z, int, int m, int n, float v) { wt_store(&a.a2[((int64_t)z * M(a) + m) * K2 + n], fmaxf(v, 0.0f)); }
};
struct Conv3FwdWT : Conv3Fwd {
  static constexpr bool PRELOAD = SDQN_PRELOAD != 0;
  __device__ static void preload(const StepArgs& a, unsigned g0, unsigned g1, unsigned g2) { SDQN_TOUCH("s"(a.a2), "s"(a.a3), "s"(a.theta[0]), "s"(a.theta[1]), "s"(a.B), "s"(a.nz), "s"(a.xcd_map), "s"(g0), "s"(g1), "s"(g2)); }

  __device__ static void store(const StepArgs& a, int z, int, int m, int n, float v) { wt_store(&a.a3[((int64_t)z * M(a) + m) * K3 + n], fmaxf(v, 0.0f)); }
};
__device__ __forceinline__ f4 ld4_nt(const float* p) {          // streamed-once operand (W4: every element is read by exactly one workgroup)
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)); f4 o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w; return o;
}
struct Fc4FwdWT : Fc4Fwd {
  static constexpr bool PRELOAD = SDQN_PRELOAD != 0;
  __device__ static void preload(const StepArgs& a, unsigned g0, unsigned g1, unsigned g2) { SDQN_TOUCH("s"(a.a3), "s"(a.slab4), "s"(a.theta[0]), "s"(a.theta[1]), "s"(a.B), "s"(a.nz), "s"(a.S4), "s"(a.xcd_map), "s"(g0), "s"(g1), "s"(g2)); }

  __device__ static f4 b_load4(const StepArgs& a, int z, int o) { return NT_W4 ? ld4_nt(a.theta[z] + OFF4 + o) : ld4(a.theta[z] + OFF4 + o); }
  __device__ static void store(const StepArgs& a, int z, int ks, int m, int n, float v) { wt_store(&a.slab4[(((int64_t)ks * 2 + z) * a.B + m) * NFC + n], v); }
};
struct Fc4DgradWT : Fc4Dgrad {
  static constexpr bool PRELOAD = SDQN_PRELOAD != 0;
  __device__ static void preload(const StepArgs& a, unsigned g0, unsigned g1, unsigned g2) { SDQN_TOUCH("s"(a.d4), "s"(a.theta[0]), "s"(a.a3), "s"(a.d3p), "s"(a.d3), "s"(a.B), "s"(a.xcd_map), "s"(g0), "s"(g1), "s"(g2)); }

  __device__ static f4 b_load4(const StepArgs& a, int, int o) { return NT_W4 ? ld4_nt(a.theta[0] + OFF4 + o) : ld4(a.theta[0] + OFF4 + o); }
  // GATED (block-tile routine; the latency engine calls store): the gating activation is fetched before the K loop
  static constexpr bool GATED = true;
  __device__ static float gate_load(const StepArgs& a, int, int m, int n) { return a.a3[(int64_t)m * NIN4 + n]; }
  __device__ static void store_gated(const StepArgs& a, int, int, int m, int n, float v, float g) {
    const int pix = n >> 6, f = n & 63, p = pix / Q3, q = pix - p * Q3;
    const float dv = g > 0.0f ? v : 0.0f;
    wt_store(&a.d3p[((m * PD3 + p + 2) * PD3 + q + 2) * K3 + f], dv);
    wt_store(&a.d3[(int64_t)m * NIN4 + n], dv);
  }
  __device__ static void store(const StepArgs& a, int z, int ks, int m, int n, float v) { store_gated(a, z, ks, m, n, v, gate_load(a, z, m, n)); }
};
struct Conv3DgradWT : Conv3Dgrad {
  static constexpr bool PRELOAD = SDQN_PRELOAD != 0;
  SDQN_PRELOAD_MULTI_DEF
  __device__ static void preload(const StepArgs& a, unsigned g0, unsigned g1, unsigned g2) { SDQN_TOUCH("s"(a.d3p), "s"(a.theta[0]), "s"(a.a2), "s"(a.d2p), "s"(a.d2), "s"(a.B), "s"(g0), "s"(g1), "s"(g2)); }

  // GATED (block-tile routine): the gating activation is fetched before the K loop instead of as a dependent load per element in the epilogue
  static constexpr bool GATED = true;
  __device__ static float gate_load(const StepArgs& a, int, int m, int c) { return a.a2[(int64_t)m * K2 + c]; }
  __device__ static void store_gated(const StepArgs& a, int, int, int m, int c, float v, float g) {
    const float dv = g > 0.0f ? v : 0.0f;
    wt_store(&a.d2p[prow2(m) + c], dv);
    wt_store(&a.d2[(int64_t)m * K2 + c], dv);
  }
  __device__ static void store(const StepArgs& a, int z, int ks, int m, int c, float v) { store_gated(a, z, ks, m, c, v, gate_load(a, z, m, c)); }
};
struct Conv3WgradWT : Conv3Wgrad {
  static constexpr bool PRELOAD = SDQN_PRELOAD != 0;
  SDQN_PRELOAD_MULTI_DEF
  __device__ static void preload(const StepArgs& a, unsigned g0, unsigned g1, unsigned g2) { SDQN_TOUCH("s"(a.a2), "s"(a.d3), "s"(a.slab3), "s"(a.tps3), "s"(a.B), "s"(g0), "s"(g1), "s"(g2)); }

  __device__ static void store(const StepArgs& a, int, int ks, int m, int n, float v) { wt_store(&a.slab3[(int64_t)ks * NW3 + m * K3 + n], v); }
};
struct Conv2DgradWT : Conv2Dgrad {
  static constexpr bool PRELOAD = SDQN_PRELOAD != 0;
  SDQN_PRELOAD_MULTI_DEF
  __device__ static void preload(const StepArgs& a, unsigned g0, unsigned g1, unsigned g2) { SDQN_TOUCH("s"(a.d2p), "s"(a.theta[0]), "s"(a.a1), "s"(a.d1), "s"(a.B), "s"(g0), "s"(g1), "s"(g2)); }

  static constexpr bool GATED = true;
  __device__ static int out_off(int z, int m, int c) {
    const int py = z >> 1, px = z & 1;
    const int n = m / 100, pix = m - n * 100, i = pix / 10, j = pix - i * 10;
    return ((n * P1 + 2 * i + py) * Q1 + 2 * j + px) * K1 + c;
  }
  __device__ static float gate_load(const StepArgs& a, int z, int m, int c) { return a.a1[out_off(z, m, c)]; }
  __device__ static void store_gated(const StepArgs& a, int z, int, int m, int c, float v, float g) { wt_store(&a.d1[out_off(z, m, c)], g > 0.0f ? v : 0.0f); }
  __device__ static void store(const StepArgs& a, int z, int ks, int m, int c, float v) { store_gated(a, z, ks, m, c, v, gate_load(a, z, m, c)); }
};
struct Conv2WgradWT : Conv2Wgrad {
  static constexpr bool PRELOAD = SDQN_PRELOAD != 0;
  __device__ static void preload(const StepArgs& a, unsigned g0, unsigned g1, unsigned g2) { SDQN_TOUCH("s"(a.a1), "s"(a.d2), "s"(a.slab2), "s"(a.tps2), "s"(a.B), "s"(g0), "s"(g1), "s"(g2)); }

  __device__ static void store(const StepArgs& a, int, int ks, int m, int n, float v) { wt_store(&a.slab2[(int64_t)ks * NW2 + m * K2 + n], v); }
};
struct Conv1WgradWT : Conv1Wgrad {
  __device__ static void store(const StepArgs& a, int, int ks, int m, int n, float v) { wt_store(&a.slab1[(int64_t)ks * NW1 + m * K1 + n], v); }
};
struct Fc4WgradWT : Fc4Wgrad {
  __device__ static void preload(const StepArgs& a, unsigned g0, unsigned g1, unsigned g2) { SDQN_TOUCH("s"(a.a3), "s"(a.d4), "s"(a.theta_w), "s"(a.state), "s"(a.g), "s"(a.fuse_rms), "s"(a.f4w_first), "s"(a.f4w_count), "s"(a.B)); SDQN_TOUCH("s"(a.bsz), "s"(a.rho), "s"(a.one_minus_rho), "s"(a.lr), "s"(a.eps), "s"(g0), "s"(g1), "s"(g2)); }
          // the 12.8 MB of new W4 + RMSProp state (or the 6.4 MB gradient) leave write-through
  __device__ static void store(const StepArgs& a, int, int, int m, int n, float v) {          // K-split form (B > 32)
    const int64_t e = OFF4 + (int64_t)m * NFC + n;
    if (a.fuse_rms) { float st = a.state[e]; const float w = rms_step(a.theta_w[e], st, v, a.bsz, a.rho, a.one_minus_rho, a.lr, a.eps); wt_store(&a.theta_w[e], w); wt_store(&a.state[e], st); }
    else wt_store(&a.g[e], v);
  }
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi& e) {
    const uint32_t base = epi_base(m0, n0, lane);
    auto stw = [](float* p, uint32_t off, float x) { wt_store(reinterpret_cast<float*>(reinterpret_cast<char*>(p) + off), x); };
    if (!a.fuse_rms) {
#pragma unroll
      for (int r = 0; r < 16; ++r) stw(a.g, base + epi_row(r), v[r]);
      return;
    }
#pragma unroll
    for (int r = 0; r < 16; r += 2) rms_step2(e.w[r], e.w[r + 1], e.st[r], e.st[r + 1], v[r], v[r + 1], a.bsz, a.rho, a.one_minus_rho, a.lr, a.eps);
#pragma unroll
    for (int r = 0; r < 16; ++r) { stw(a.theta_w, base + epi_row(r), e.w[r]); stw(a.state, base + epi_row(r), e.st[r]); }
  }
};

// fc4_wgrad on the block-tile engine (B >= 128): the fused RMSProp epilogue in four groups of four accumulator rows — 8 loads, the
// arithmetic, 8 write-through stores per group — instead of the single-wave form's 32 loads prefetched under the whole K loop
// (156 VGPRs: the multi-problem launch takes the register count of its largest problem, and conv3_dgrad / conv3_wgrad beside it would
// run at 2 workgroups per CU).  Other resident workgroups cover the four round trips.  Same operations per element as rms_step.
struct Fc4WgradBT : Fc4WgradWT {
  struct Epi {};
  __device__ static void epi_begin(const StepArgs&, int, int, int, Epi&) {}
  static constexpr bool STORE_TILE = true;
  __device__ static void store_tile(const StepArgs& a, int m0, int n0, int lane, const float* v) {
    const uint32_t base = epi_base(m0, n0, lane);
    auto stw = [](float* p, uint32_t off, float x) { wt_store(reinterpret_cast<float*>(reinterpret_cast<char*>(p) + off), x); };
    if (!a.fuse_rms) {
#pragma unroll
      for (int r = 0; r < 16; ++r) stw(a.g, base + epi_row(r), v[r]);
      return;
    }
    const float* __restrict__ tw = a.theta_w; const float* __restrict__ sp = a.state;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float w[4], st[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { w[e] = ldb(tw, base + epi_row(4 * g + e)); st[e] = ldb(sp, base + epi_row(4 * g + e)); }
      rms_step2(w[0], w[1], st[0], st[1], v[4 * g], v[4 * g + 1], a.bsz, a.rho, a.one_minus_rho, a.lr, a.eps);
      rms_step2(w[2], w[3], st[2], st[3], v[4 * g + 2], v[4 * g + 3], a.bsz, a.rho, a.one_minus_rho, a.lr, a.eps);
#pragma unroll
      for (int e = 0; e < 4; ++e) { stw(a.theta_w, base + epi_row(4 * g + e), w[e]); stw(a.state, base + epi_row(4 * g + e), st[e]); }
      __builtin_amdgcn_sched_barrier(0);      // keep the groups apart: hoisting all 32 loads is exactly the register bill this form avoids
    }
  }
};

// ---- plane mode of the block-tile engine (bt_tile_xp): where a problem's B operand (always a weight matrix here) lives as bf16 planes ----
template <class P, int OFF, int K> struct XPF : P {       // forward conv: transposed planes [n][K] of net z
  __device__ static const unsigned short* bp(const StepArgs& a, int z) { return a.wpt[z] + OFF; }
  __device__ static int bp_col(const StepArgs&, int, int n) { return n * K; }
  __device__ static int bp_row(const StepArgs&, int, int k) { return k; }
};
template <class P, int OFF> struct XPD : P {              // dgrad: master-layout planes, the problem's own k-contiguous B index functions
  __device__ static const unsigned short* bp(const StepArgs& a, int) { return a.wpm + OFF; }
  __device__ static int bp_col(const StepArgs& a, int z, int n) { return P::b_col(a, z, n); }
  __device__ static int bp_row(const StepArgs& a, int z, int k) { return P::b_row(a, z, k); }
};

}  // namespace sdqn
