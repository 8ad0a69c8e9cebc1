// sdqn_kernels.hip — device code of the DQN train step for gfx950 (MI355X / CDNA4).
//   * gemm_kernel<P> instantiations (gemm_engine.h / problems.h): conv1..3, fc4, all dgrad/wgrad
//   * head_kernel   : fc4 slab reduce + ReLU, fc5 of both nets, max_a Q', TD target, delta, cost term,
//                     clip, fc5 dgrad            (deepqnetwork.py:120-159 without any host round trip)
//   * update_kernel : split-K slab reduction + fc5 wgrad + RMSProp (deepqnetwork.py:165, A9/A10)
//   * gather_kernel : standalone replay gather of (s, a, r, s', t) (replay_memory.py:71-78)
#include "gemm_engine.h"
#include "kernels.h"

namespace sdqn {

static const char* k_names[K_COUNT] = {
  "conv1_fwd(gather+norm+conv+relu)", "conv2_fwd", "conv3_fwd", "fc4_fwd(splitK)", "head(fc5+td+delta)",
  "fc4_dgrad", "fc4_wgrad", "conv3_dgrad", "conv3_wgrad", "conv2_dgrad", "conv2_wgrad",
  "conv1_wgrad", "update(reduce+fc5wgrad+rmsprop)", "rccl_allreduce", "replay_gather_u8"};
const char* kernel_name(int id) { return (id >= 0 && id < K_COUNT) ? k_names[id] : "?"; }

hipError_t launch_kernel(int id, const StepArgs& a, hipStream_t s) {
  switch (id) {
    case K_CONV1_FWD: return launch_gemm<Conv1Fwd>(a, s);
    case K_CONV2_FWD: return launch_gemm<Conv2Fwd>(a, s);
    case K_CONV3_FWD: return launch_gemm<Conv3Fwd>(a, s);
    case K_FC4_FWD: return launch_gemm<Fc4Fwd>(a, s);
    case K_FC4_DGRAD: return launch_gemm<Fc4Dgrad>(a, s);
    case K_FC4_WGRAD: return launch_gemm<Fc4Wgrad>(a, s);
    case K_CONV3_DGRAD: return launch_gemm<Conv3Dgrad>(a, s);
    case K_CONV3_WGRAD: return launch_gemm<Conv3Wgrad>(a, s);
    case K_CONV2_DGRAD: return launch_gemm<Conv2Dgrad>(a, s);
    case K_CONV2_WGRAD: return launch_gemm<Conv2Wgrad>(a, s);
    case K_CONV1_WGRAD: return launch_gemm<Conv1Wgrad>(a, s);
    default: return hipErrorInvalidValue;
  }
}

// ------------------------------------------------------------------------------------------------
// head: one 512-thread workgroup (8 wave64s) per sample; thread j owns hidden unit j.
__global__ void __launch_bounds__(512) head_kernel(const StepArgs a, const HeadArgs h) {
  const int n = blockIdx.x, j = threadIdx.x, lane = j & 63, wave = j >> 6;
  __shared__ float red[2][MAX_ACTIONS][8];
  __shared__ float sh_q[2][MAX_ACTIONS];
  __shared__ float sh_dc;
  __shared__ int sh_act;
  float a4v[2] = {0.0f, 0.0f};
  for (int z = 0; z < a.nz; ++z) {
    float v = 0.0f;
    for (int s = 0; s < a.S4; ++s) v += a.slab4[(((int64_t)s * 2 + z) * a.B + n) * NFC + j];   // fixed order
    v = fmaxf(v, 0.0f);                                                                          // Rectlin, :89
    a4v[z] = v;
    a.a4[((int64_t)z * a.B + n) * NFC + j] = v;
    const float* W5 = a.theta[z] + OFF5;
    for (int act = 0; act < a.A; ++act) {                                                        // Affine(A), :91
      float p = W5[act * NFC + j] * v;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) p += __shfl_xor(p, off, 64);                       // wavefront reduction
      if (lane == 0) red[z][act][wave] = p;
    }
  }
  __syncthreads();
  if (j < a.nz * a.A) {
    const int z = j / a.A, act = j - z * a.A;
    float q = 0.0f;
    for (int w = 0; w < 8; ++w) q += red[z][act][w];
    sh_q[z][act] = q;
    h.q[((int64_t)z * a.B + n) * a.A + act] = q;
  }
  if (!h.train) return;
  __syncthreads();
  if (j == 0) {
#pragma clang fp contract(off)
    int act, term; int64_t rew;
    if (a.from_ring) { const MetaRec rec = h.meta[a.idx[n]]; act = rec.action; rew = rec.reward; term = rec.terminal; }
    else { act = h.st_actions[n]; rew = h.st_rewards[n]; term = h.st_terminals[n]; }
    float m = sh_q[1][0];
    for (int k = 1; k < a.A; ++k) m = fmaxf(m, sh_q[1][k]);                                     // be.max(postq, axis=0), :124
    double rr = (double)rew;                                                                     // np.clip(rewards, ..), :136
    rr = rr < h.min_reward ? h.min_reward : (rr > h.max_reward ? h.max_reward : rr);
    const double y = term ? rr : rr + h.discount * (double)m;                                    // :139-143 (host float math)
    const float d = sh_q[0][act] - (float)y;                                                     // get_errors, :149
    h.cost_terms[n] = 0.5f * (d * d);                                                            // get_cost summand, :154
    float dc = d;
    if (h.clip_error != 0.0f) dc = fminf(fmaxf(d, -h.clip_error), h.clip_error);                 // :158-159
    h.maxq[n] = m;
    sh_dc = dc; sh_act = act;
  }
  __syncthreads();
  const float dc = sh_dc; const int act = sh_act;
  // fc5 dgrad: delta4 = W5^T delta * 1[a4 > 0]; delta is non-zero on the taken action only
  a.d4[(int64_t)n * NFC + j] = a4v[0] > 0.0f ? a.theta[0][OFF5 + act * NFC + j] * dc : 0.0f;
  if (j < a.A) h.dq[(int64_t)n * a.A + j] = (j == act) ? dc : 0.0f;
}

hipError_t launch_head(const StepArgs& a, const HeadArgs& h, hipStream_t s) {
  hipLaunchKernelGGL(head_kernel, dim3(a.B), dim3(512), 0, s, a, h);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
__device__ inline float rms_apply(float w, float& st, float gsum, const UpdateArgs& u) {
#pragma clang fp contract(off)
  const float g = gsum / u.bsz;                                    // A9: grad / be.bsz
  st = u.rho * st + (g * g) * u.one_minus_rho;                     // A10
  return w - (g * u.lr) / (sqrtf(st + u.eps) + u.eps);
}

__global__ void __launch_bounds__(256) update_kernel(const UpdateArgs u) {
  const int64_t NP4 = (OFF5 + (int64_t)u.A * NFC) / 4;
  for (int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x; i4 < NP4; i4 += (int64_t)gridDim.x * 256) {
    const int64_t e = i4 * 4;
    float4 gs;
    if (u.mode == 2 || (e >= OFF4 && e < OFF5)) {
      gs = *reinterpret_cast<const float4*>(u.g + e);              // fc4 wgrad wrote g directly; mode 2: all-reduced g
    } else if (e < OFF4) {
      const int L = e < OFF2 ? 0 : (e < OFF3 ? 1 : 2);
      const int64_t off = e - (L == 0 ? OFF1 : (L == 1 ? OFF2 : OFF3));
      const int64_t nw = L == 0 ? NW1 : (L == 1 ? NW2 : NW3);
      gs = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s = 0; s < u.ns[L]; ++s) {                          // fixed order: deterministic
        const float4 v = *reinterpret_cast<const float4*>(u.slab[L] + (int64_t)s * nw + off);
        gs.x += v.x; gs.y += v.y; gs.z += v.z; gs.w += v.w;
      }
    } else {                                                       // fc5 wgrad: delta . a4^T  (A x 512, tiny)
      const int64_t o = e - OFF5;
      const int act = (int)(o / NFC), j0 = (int)(o - (int64_t)act * NFC);
      gs = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int n = 0; n < u.B; ++n) {
        const float d = u.dq[(int64_t)n * u.A + act];
        const float4 v = *reinterpret_cast<const float4*>(u.a4 + (int64_t)n * NFC + j0);
        gs.x += d * v.x; gs.y += d * v.y; gs.z += d * v.z; gs.w += d * v.w;
      }
    }
    if (u.mode != 2 && !(e >= OFF4 && e < OFF5)) *reinterpret_cast<float4*>(u.g + e) = gs;
    if (u.mode != 1) {
      float4 w = *reinterpret_cast<float4*>(u.theta + e);
      float4 st = *reinterpret_cast<float4*>(u.state + e);
      w.x = rms_apply(w.x, st.x, gs.x, u); w.y = rms_apply(w.y, st.y, gs.y, u);
      w.z = rms_apply(w.z, st.z, gs.z, u); w.w = rms_apply(w.w, st.w, gs.w, u);
      *reinterpret_cast<float4*>(u.theta + e) = w;
      *reinterpret_cast<float4*>(u.state + e) = st;
    }
  }
  if (u.mode != 2 && blockIdx.x == 0 && threadIdx.x == 0) {        // get_cost: mean over the batch, :154
    float c = 0.0f;
    for (int n = 0; n < u.B; ++n) c += u.cost_terms[n];
    c = c / (float)u.B;
    u.cost_out[0] = c;
    u.cost_accum[0] += (double)c;
  }
}

hipError_t launch_update(const UpdateArgs& u, hipStream_t s) {
  const int64_t NP4 = (OFF5 + (int64_t)u.A * NFC) / 4;
  int blocks = (int)((NP4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(update_kernel, dim3(blocks), dim3(256), 0, s, u);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// standalone replay gather: 16 B/lane coalesced loads of the 5 contiguous frames screens[i-4 : i+1],
// each written to prestates[k] (frames 0..3) and poststates[k] (frames 1..4).  7056 = 441 * 16.
__global__ void __launch_bounds__(256) gather_kernel(const GatherArgs g) {
  const int n = blockIdx.y;
  const int64_t index = g.idx[n];
  constexpr int V = FRAME / 16;                                    // 441 uint4 per frame
  const uint4* src = reinterpret_cast<const uint4*>(g.ring + (index - C0) * (int64_t)FRAME);
  uint4* pre = reinterpret_cast<uint4*>(g.pre + (int64_t)n * STATE);
  uint4* post = reinterpret_cast<uint4*>(g.post + (int64_t)n * STATE);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < (C0 + 1) * V; i += gridDim.x * 256) {
    const uint4 v = src[i];
    if (i < C0 * V) pre[i] = v;
    if (i >= V) post[i - V] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {                       // replay_memory.py:76-78
    const MetaRec rec = g.meta[index];
    g.actions[n] = rec.action; g.rewards[n] = rec.reward; g.terminals[n] = rec.terminal;
  }
}

hipError_t launch_gather(const GatherArgs& g, hipStream_t s) {
  hipLaunchKernelGGL(gather_kernel, dim3(9, g.B), dim3(256), 0, s, g);
  return hipGetLastError();
}

}  // namespace sdqn
