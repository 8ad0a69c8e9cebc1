// problems_h16.h — the fp16 mode (--datatype float16, BASELINE.json configs[4]) of the GEMM-shaped stages.
//
// Precision contract (ours; Neon's fp16 backend is GPU-only and unpinned — oracle/dqn_numpy.py half_activations=True
// restates exactly this): activations, deltas and the weight operands of the forward / dgrad GEMMs are IEEE half;
// every accumulation is fp32; master weights, gradients and optimizer state are fp32; deltas are stored as
// half(delta * loss_scale) with a static power-of-two loss scale that the wgrad epilogues divide out again.
//
// Forward and dgrad run on packed-fp16 MFMA (gemm_tile_h: both operands k-contiguous, no LDS staging): they reuse the
// index math of the fp32 problem structs and only swap the operand pointers / layouts.  Wgrads stay on the fp32 MFMA
// engine (fp32 accumulation of the gradient) and read their half operands through the typed x-contiguous loaders.
#pragma once
#include "problems.h"

#if defined(__HIPCC__)
namespace sdqn {

__device__ __forceinline__ half8 ldh8(const half_t* p) { return *reinterpret_cast<const half8*>(p); }
__device__ __forceinline__ half8 ldh8_u8(const uint8_t* p) {        // 8 ring bytes (one patch row) -> 8 normalised halves
  uint2 w;                                                           // patch rows are only 4-byte aligned: two dword loads
  w.x = *reinterpret_cast<const uint32_t*>(p); w.y = *reinterpret_cast<const uint32_t*>(p + 4);
  half8 o;
  o[0] = (half_t)norm_u8(w.x & 255u); o[1] = (half_t)norm_u8((w.x >> 8) & 255u);
  o[2] = (half_t)norm_u8((w.x >> 16) & 255u); o[3] = (half_t)norm_u8(w.x >> 24);
  o[4] = (half_t)norm_u8(w.y & 255u); o[5] = (half_t)norm_u8((w.y >> 8) & 255u);
  o[6] = (half_t)norm_u8((w.y >> 16) & 255u); o[7] = (half_t)norm_u8(w.y >> 24);
  return o;
}

// ---- forward: A = half activations (k-contiguous), B = transposed half weights wht[z][OFF + n*K + k] ----------------
struct Conv1FwdH : Conv1Fwd {
  static constexpr bool F16_MFMA = true;
  __device__ static half8 a_load8(const StepArgs& a, int, aoff_t o) { return ldh8_u8(a.src + o); }
  __device__ static int b_row(const StepArgs&, int, int k) { return k; }
  __device__ static int b_col(const StepArgs&, int, int n) { return n * CRS1; }
  __device__ static half8 b_load8(const StepArgs& a, int z, int o) { return ldh8(a.wht[z] + OFF1 + o); }
  __device__ static void store(const StepArgs& a, int z, int, int m, int n, float v) {
    a.h_a1[((int64_t)z * M(a) + m) * K1 + n] = (half_t)fmaxf(v, 0.0f);
  }
};
struct Conv2FwdH : Conv2Fwd {
  static constexpr bool F16_MFMA = true;
  __device__ static half8 a_load8(const StepArgs& a, int, aoff_t o) { return ldh8(a.h_a1 + o); }
  __device__ static int b_row(const StepArgs&, int, int k) { return k; }
  __device__ static int b_col(const StepArgs&, int, int n) { return n * CRS2; }
  __device__ static half8 b_load8(const StepArgs& a, int z, int o) { return ldh8(a.wht[z] + OFF2 + o); }
  __device__ static void store(const StepArgs& a, int z, int, int m, int n, float v) {
    a.h_a2[((int64_t)z * M(a) + m) * K2 + n] = (half_t)fmaxf(v, 0.0f);
  }
};
struct Conv3FwdH : Conv3Fwd {
  static constexpr bool F16_MFMA = true;
  __device__ static half8 a_load8(const StepArgs& a, int, aoff_t o) { return ldh8(a.h_a2 + o); }
  __device__ static int b_row(const StepArgs&, int, int k) { return k; }
  __device__ static int b_col(const StepArgs&, int, int n) { return n * CRS3; }
  __device__ static half8 b_load8(const StepArgs& a, int z, int o) { return ldh8(a.wht[z] + OFF3 + o); }
  __device__ static void store(const StepArgs& a, int z, int, int m, int n, float v) {
    a.h_a3[((int64_t)z * M(a) + m) * K3 + n] = (half_t)fmaxf(v, 0.0f);
  }
};
struct Fc4FwdH : Fc4Fwd {              // slabs stay fp32 (summed + ReLU'd by the head kernel)
  static constexpr bool F16_MFMA = true;
  __device__ static half8 a_load8(const StepArgs& a, int, aoff_t o) { return ldh8(a.h_a3 + o); }
  __device__ static int b_row(const StepArgs&, int, int k) { return k; }
  __device__ static int b_col(const StepArgs&, int, int n) { return n * NIN4; }
  __device__ static half8 b_load8(const StepArgs& a, int z, int o) { return ldh8(a.wht[z] + OFF4 + o); }
};

// ---- dgrad: A = half (loss-scaled) deltas, B = half weights in the master layout wh[0] -------------------------------
struct Fc4DgradH : Fc4Dgrad {
  static constexpr bool F16_MFMA = true;
  __device__ static half8 a_load8(const StepArgs& a, int, aoff_t o) { return ldh8(a.h_d4 + o); }
  __device__ static half8 b_load8(const StepArgs& a, int, int o) { return ldh8(a.wh[0] + OFF4 + o); }
  // GATED (block-tile routines): the activation whose sign gates the delta is fetched BEFORE the K loop (it does not depend on it) instead of
  // as a dependent load per element in the epilogue
  static constexpr bool GATED = true;
  __device__ static float gate_load(const StepArgs& a, int, int m, int n) { return (float)a.h_a3[(int64_t)m * NIN4 + n]; }
  __device__ static void store_gated(const StepArgs& a, int, int, int m, int n, float v, float g) {
    int pix = n >> 6, f = n & 63, p = pix / Q3, q = pix - p * Q3;
    const half_t dv = g > 0.0f ? (half_t)v : (half_t)0.0f;
    a.h_d3p[((m * PD3 + p + 2) * PD3 + q + 2) * K3 + f] = dv;
    a.h_d3[(int64_t)m * NIN4 + n] = dv;
  }
  __device__ static void store(const StepArgs& a, int z, int ks, int m, int n, float v) { store_gated(a, z, ks, m, n, v, gate_load(a, z, m, n)); }
};
struct Conv3DgradH : Conv3Dgrad {
  static constexpr bool F16_MFMA = true;
  __device__ static half8 a_load8(const StepArgs& a, int, aoff_t o) { return ldh8(a.h_d3p + o); }
  __device__ static half8 b_load8(const StepArgs& a, int, int o) { return ldh8(a.wh[0] + OFF3 + o); }
  static constexpr bool GATED = true;
  __device__ static float gate_load(const StepArgs& a, int, int m, int c) { return (float)a.h_a2[(int64_t)m * K2 + c]; }
  __device__ static void store_gated(const StepArgs& a, int, int, int m, int c, float v, float g) {
    const half_t dv = g > 0.0f ? (half_t)v : (half_t)0.0f;
    a.h_d2p[prow2(m) + c] = dv;
    a.h_d2[(int64_t)m * K2 + c] = dv;
  }
  __device__ static void store(const StepArgs& a, int z, int ks, int m, int c, float v) { store_gated(a, z, ks, m, c, v, gate_load(a, z, m, c)); }
};
struct Conv2DgradH : Conv2Dgrad {
  static constexpr bool F16_MFMA = true;
  __device__ static half8 a_load8(const StepArgs& a, int, aoff_t o) { return ldh8(a.h_d2p + o); }
  __device__ static half8 b_load8(const StepArgs& a, int, int o) { return ldh8(a.wh[0] + OFF2 + o); }
  static constexpr bool GATED = true;
  __device__ static int out_off(int z, int m, int c) {
    int py = z >> 1, px = z & 1;
    int n = m / 100, pix = m - n * 100, i = pix / 10, j = pix - i * 10;
    return ((n * P1 + 2 * i + py) * Q1 + 2 * j + px) * K1 + c;
  }
  __device__ static float gate_load(const StepArgs& a, int z, int m, int c) { return (float)a.h_a1[out_off(z, m, c)]; }
  __device__ static void store_gated(const StepArgs& a, int z, int, int m, int c, float v, float g) { a.h_d1[out_off(z, m, c)] = g > 0.0f ? (half_t)v : (half_t)0.0f; }
  __device__ static void store(const StepArgs& a, int z, int ks, int m, int c, float v) { store_gated(a, z, ks, m, c, v, gate_load(a, z, m, c)); }
};

// ---- wgrad: fp32 MFMA engine, half operands, loss scale divided out in the epilogue ---------------------------------
// refresh of the two half copies of a conv / fc4 weight (row k, column n of its (K x N) internal matrix)
__device__ __forceinline__ void refresh_half(const StepArgs& a, int off, int K, int N, int k, int n, float w) {
  a.wh_w[off + (int64_t)k * N + n] = (half_t)w;
  a.wht_w[off + (int64_t)n * K + k] = (half_t)w;
}
struct Fc4WgradH : Fc4Wgrad {
  typedef half_t AT; typedef half_t BT;
  __device__ static const half_t* a_ptr(const StepArgs& a, int) { return a.h_a3; }
  __device__ static const half_t* b_ptr(const StepArgs& a, int) { return a.h_d4; }
  __device__ static void store(const StepArgs& a, int, int, int m, int n, float v) {
    const int64_t e = OFF4 + (int64_t)m * NFC + n;
    const float g = v * a.inv_loss_scale;
    if (a.fuse_rms) {
      float st = a.state[e];
      const float w = rms_step(a.theta_w[e], st, g, a.bsz, a.rho, a.one_minus_rho, a.lr, a.eps);
      a.theta_w[e] = w; a.state[e] = st;
      refresh_half(a, OFF4, NIN4, NFC, m, n, w);
    } else a.g[e] = g;
  }
  // single-wave epilogue (B <= 32), as Fc4Wgrad's: theta / state loads issued at kernel entry; the transposed half copy
  // is written as 4-half (8 B) groups — a lane's 16 accumulator rows are 4 runs of 4 consecutive k'
  struct Epi { float w[16], st[16]; };
  __device__ static void epi_begin(const StepArgs& a, int m0, int n0, int lane, Epi& e) {
    if (!a.fuse_rms) return;
    const float* __restrict__ tw = a.theta_w; const float* __restrict__ sp = a.state;
    const uint32_t base = epi_base(m0, n0, lane);                 // 32-bit byte offsets from the uniform base (Fc4Wgrad)
#pragma unroll
    for (int r = 0; r < 16; ++r) { e.w[r] = ldb(tw, base + epi_row(r)); e.st[r] = ldb(sp, base + epi_row(r)); }
  }
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi& e) {
    const int n = n0 + (lane & 31);
    const uint32_t base = epi_base(m0, n0, lane);
    if (!a.fuse_rms) {
      float* __restrict__ gp = a.g;
#pragma unroll
      for (int r = 0; r < 16; ++r) stb(gp, base + epi_row(r), v[r] * a.inv_loss_scale);
      return;
    }
#pragma unroll
    for (int r = 0; r < 16; r += 2)
      rms_step2(e.w[r], e.w[r + 1], e.st[r], e.st[r + 1], v[r] * a.inv_loss_scale, v[r + 1] * a.inv_loss_scale, a.bsz, a.rho, a.one_minus_rho, a.lr, a.eps);
    float* __restrict__ tw = a.theta_w; float* __restrict__ sp = a.state;
    half_t* __restrict__ whp = a.wh_w;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      stb(tw, base + epi_row(r), e.w[r]); stb(sp, base + epi_row(r), e.st[r]);
      *reinterpret_cast<half_t*>(reinterpret_cast<char*>(whp) + (base + epi_row(r)) / 2) = (half_t)e.w[r];
    }
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      half4 hv; hv[0] = (half_t)e.w[4 * g]; hv[1] = (half_t)e.w[4 * g + 1]; hv[2] = (half_t)e.w[4 * g + 2]; hv[3] = (half_t)e.w[4 * g + 3];
      *reinterpret_cast<half4*>(a.wht_w + OFF4 + (int64_t)n * NIN4 + m0 + 8 * g + 4 * (lane >> 5)) = hv;
    }
  }
};
struct Conv3WgradH : Conv3Wgrad {
  typedef half_t AT; typedef half_t BT;
  __device__ static const half_t* a_ptr(const StepArgs& a, int) { return a.h_a2; }
  __device__ static const half_t* b_ptr(const StepArgs& a, int) { return a.h_d3; }
  __device__ static void store(const StepArgs& a, int, int ks, int m, int n, float v) { a.slab3[(int64_t)ks * NW3 + m * K3 + n] = v * a.inv_loss_scale; }
  struct Epi {};
  __device__ static void epi_begin(const StepArgs&, int, int, int, Epi&) {}
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi&) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int ml = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); if (m0 + ml < M && n0 + (lane & 31) < N) store(a, z, ks, m0 + ml, n0 + (lane & 31), v[r]); }
  }
};
struct Conv2WgradH : Conv2Wgrad {
  typedef half_t AT; typedef half_t BT;
  __device__ static const half_t* a_ptr(const StepArgs& a, int) { return a.h_a1; }
  __device__ static const half_t* b_ptr(const StepArgs& a, int) { return a.h_d2; }
  __device__ static void store(const StepArgs& a, int, int ks, int m, int n, float v) { a.slab2[(int64_t)ks * NW2 + m * K2 + n] = v * a.inv_loss_scale; }
  struct Epi {};
  __device__ static void epi_begin(const StepArgs&, int, int, int, Epi&) {}
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi&) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int ml = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); if (m0 + ml < M && n0 + (lane & 31) < N) store(a, z, ks, m0 + ml, n0 + (lane & 31), v[r]); }
  }
};
struct Conv1WgradH : Conv1Wgrad {      // A' = fp32-normalised u8 patches (as in fp32 mode), B = half d1
  typedef half_t BT;
  __device__ static const half_t* b_ptr(const StepArgs& a, int) { return a.h_d1; }
  __device__ static void store(const StepArgs& a, int, int ks, int m, int n, float v) { a.slab1[(int64_t)ks * NW1 + m * K1 + n] = v * a.inv_loss_scale; }
  struct Epi {};
  __device__ static void epi_begin(const StepArgs&, int, int, int, Epi&) {}
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi&) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int ml = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); if (m0 + ml < M && n0 + (lane & 31) < N) store(a, z, ks, m0 + ml, n0 + (lane & 31), v[r]); }
  }
};

// ---- the same weight gradients on packed-fp16 MFMA (gemm_tile_hw: wave-private LDS transposes) ---------------------
struct Fc4WgradHW : Fc4WgradH { static constexpr bool F16_WGRAD = true; };
struct Conv3WgradHW : Conv3WgradH { static constexpr bool F16_WGRAD = true; };
struct Conv2WgradHW : Conv2WgradH { static constexpr bool F16_WGRAD = true; };
struct Conv1WgradHW : Conv1WgradH { static constexpr bool F16_WGRAD = true; };    // A = half(x / 255) like the forward pass

}  // namespace sdqn
#endif
