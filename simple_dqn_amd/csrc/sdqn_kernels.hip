// sdqn_kernels.hip — device code of the DQN train step for gfx950 (MI355X / CDNA4).
//   * gemm_kernel<P> instantiations (gemm_engine.h / problems.h): conv1..3, fc4, all dgrad/wgrad
//   * head_kernel   : fc4 slab reduce + ReLU, fc5 of both nets, max_a Q', TD target, delta, cost term,
//                     clip, fc5 dgrad            (deepqnetwork.py:120-159 without any host round trip)
//   * update_kernel : split-K slab reduction + fc5 wgrad + RMSProp (deepqnetwork.py:165, A9/A10)
//   * gather_kernel : standalone replay gather of (s, a, r, s', t) (replay_memory.py:71-78)
#include "gemm_engine.h"
#include "problems_h16.h"
#include "kernels.h"
#include "update_body.h"

namespace sdqn {

static const char* k_names[K_COUNT] = {
  "conv1_fwd(gather+norm+conv+relu)", "conv2_fwd", "conv3_fwd", "fc4_fwd(splitK)", "head(fc5+td+delta)",
  "fc4_dgrad", "fc4_wgrad", "conv3_dgrad", "conv3_wgrad", "conv2_dgrad", "conv2_wgrad",
  "conv1_wgrad", "update(reduce+fc5wgrad+rmsprop)", "rccl_allreduce", "replay_gather_u8", "prep(idx+meta)",
  "bwd3(conv3_dgrad+conv3_wgrad+fc4_wgrad)", "bwd2(conv2_dgrad+conv2_wgrad+fc4_wgrad)", "bwd1(conv1_wgrad+fc4_wgrad)",
  "batchnorm(layer fwd/bwd)", "fc4_dgrad+fc4_wgrad(+rmsprop W4)", "bwd3(conv3_dgrad+conv3_wgrad)", "update(i)+conv1_fwd(i+1)",
  "head+fc4_dgrad", "wgrads(fc4+conv3+conv2)", "act(conv1..fc5, one state)"};
const char* kernel_name(int id) { return (id >= 0 && id < K_COUNT) ? k_names[id] : "?"; }
LaunchEvents& launch_events() { static thread_local LaunchEvents le; return le; }

template <class P>
static hipError_t launch_nw(int nw, const StepArgs& a, hipStream_t s) {
  switch (nw) {
    case 2: return launch_gemm<P, 2>(a, s);
    case 4: return launch_gemm<P, 4>(a, s);
    case 8: return launch_gemm<P, 8>(a, s);
    case 16: return launch_gemm<P, 16>(a, s);
    default: return hipErrorInvalidValue;
  }
}

// Waves per workgroup = how many 32-deep K-chunks run concurrently on one output tile (gemm_engine.h).
// Everything that is NOT the default fp32 step lives in sdqn_kernels_ext.hip (fp16 mode, option "hoist", the register-blocked
// experiments): hipcc's schedule of the default kernels depends on what else is instantiated in their translation unit
// (measured: -1 % step rate when the new variants shared this file), so this file stays what round 1 tuned.
hipError_t launch_kernel_ext(int id, const StepArgs& a, const LaunchTune& t, hipStream_t s, bool* handled);

hipError_t launch_kernel_r3(int id, const StepArgs& a, const LaunchTune& t, hipStream_t s, bool* handled);
hipError_t launch_kernel_bt(int id, const StepArgs& a, const LaunchTune& t, hipStream_t s, bool* handled);
hipError_t launch_kernel_ss(int id, const StepArgs& a, const LaunchTune& t, hipStream_t s, bool* handled);

hipError_t launch_kernel(int id, const StepArgs& a, const LaunchTune& t, hipStream_t s) {
  if (a.B >= 128 || a.h16) {   // throughput regime (and float16 at any batch size): the sample-stationary convolution chains (round 6;
    bool handled = false;      // sdqn_kernels_ss.hip), then — B >= 128 — the block-tile engine (round 4; sdqn_kernels_bt.hip) take what they implement
    hipError_t e = launch_kernel_ss(id, a, t, s, &handled);
    if (handled) return e;
    e = launch_kernel_bt(id, a, t, s, &handled);        // (B < 128: float16's conv1 kernels on request only)
    if (handled) return e;
  }
  if (t.r3 || t.wt) {          // round-3 launch variants live in their own translation unit (same reason as sdqn_kernels_ext.hip)
    bool handled = false;
    const hipError_t e = launch_kernel_r3(id, a, t, s, &handled);
    if (handled) return e;
  }
  if (a.h16) {
    bool handled = false;
    const hipError_t e = launch_kernel_ext(id, a, t, s, &handled);
    if (handled) return e;
  }
  if (a.bn) {                  // --batch_norm forward: raw linear outputs (same tilings as the default problems)
    switch (id) {
      case K_CONV1_FWD: return launch_gemm<Conv1FwdRaw, 8>(a, s);
      case K_CONV2_FWD: return a.B >= 128 ? launch_gemm<Staged<Conv2FwdRaw>, 8>(a, s) : launch_gemm<Conv2FwdRaw, 16>(a, s);
      case K_CONV3_FWD: return a.B >= 128 ? launch_gemm<Staged<Conv3FwdRaw>, 8>(a, s) : launch_gemm<Staged<Conv3FwdRaw>, 9>(a, s);
      default: break;
    }
  }
  if (id >= 0 && id < 12 && t.nw_override[id] > 0) {          // tuning hook (sdqn_net_set_option "nw:<id>")
    const int nw = t.nw_override[id];
    switch (id) {
      case K_CONV1_FWD: return launch_nw<Conv1Fwd>(nw, a, s);
      case K_CONV2_FWD: return launch_nw<Conv2Fwd>(nw, a, s);
      case K_CONV3_FWD: if (nw == 9) return launch_gemm<Staged<Conv3Fwd>, 9>(a, s); return launch_nw<Conv3Fwd>(nw, a, s);    // 9 = round 1's choice
      case K_FC4_FWD: return launch_nw<Fc4Fwd>(nw, a, s);
      case K_FC4_DGRAD: return launch_nw<Fc4Dgrad>(nw, a, s);
      case K_CONV3_DGRAD: return launch_nw<Conv3Dgrad>(nw, a, s);
      case K_CONV3_WGRAD: return launch_nw<Conv3Wgrad>(nw, a, s);
      case K_CONV2_DGRAD: return launch_nw<Conv2Dgrad>(nw, a, s);
      case K_CONV2_WGRAD: return launch_nw<Conv2Wgrad>(nw, a, s);
      case K_CONV1_WGRAD: return launch_nw<Conv1Wgrad>(nw, a, s);
      default: break;
    }
  }
  if (a.B >= 128) {            // throughput regime (thousands of tiles per launch): tools/sweep_nw.py at B = 256
    switch (id) {
      case K_CONV1_FWD: return launch_gemm<Conv1Fwd, 8>(a, s);
      case K_CONV2_FWD: return launch_gemm<Staged<Conv2Fwd>, 8>(a, s);
      case K_CONV3_FWD: return launch_gemm<Staged<Conv3Fwd>, 8>(a, s);
      case K_FC4_FWD: return launch_gemm<Staged<Fc4Fwd>, 8>(a, s);
      case K_FC4_DGRAD: return launch_gemm<Staged<Fc4Dgrad>, 4>(a, s);
      case K_CONV3_DGRAD: return launch_gemm<Staged<Conv3Dgrad>, 8>(a, s);
      case K_CONV2_DGRAD: return launch_gemm<Staged<Conv2Dgrad>, 8>(a, s);
      case K_BWD3:
        if (a.f4w_count > 0) return launch_multi<512, Fc4Wgrad, 8, Staged<Conv3Dgrad>, 8, Conv3Wgrad, 8>(a, true, true, s);
        return launch_multi<512, NoProblem, 2, Staged<Conv3Dgrad>, 8, Conv3Wgrad, 8>(a, true, true, s);
      case K_BWD2: return launch_multi<512, NoProblem, 2, Staged<Conv2Dgrad>, 8, Conv2Wgrad, 8>(a, true, true, s);
      default: break;
    }
  }
  switch (id) {
    case K_CONV1_FWD: return launch_gemm<Conv1Fwd, 8>(a, s);        // K = 256  -> 8 chunks
    case K_CONV2_FWD: return launch_gemm<Conv2Fwd, 16>(a, s);       // K = 512  -> 16 chunks
    case K_CONV3_FWD: return launch_gemm<Conv3Fwd, 16>(a, s);       // K = 576 -> 18 chunks over 16 waves (staged, 9 waves x 2 chunks: -0.4 %)
    case K_FC4_FWD: return launch_gemm<Staged<Fc4Fwd>, 14>(a, s);   // K = 3136 -> 98 chunks = S4(7) x 14; rows 12.5 KB apart: staged
    case K_FC4_DGRAD: return launch_gemm<Staged<Fc4Dgrad>, 16>(a, s);   // K = 512; rows 2 KB apart: staged
    case K_FC4_WGRAD:                                               // K = B
      if (a.B <= 32) return launch_gemm<Fc4Wgrad, 1>(a, s);
      if (a.B <= 64) return launch_gemm<Fc4Wgrad, 2>(a, s);
      if (a.B <= 128) return launch_gemm<Fc4Wgrad, 4>(a, s);
      return launch_gemm<Fc4Wgrad, 8>(a, s);
    case K_CONV3_DGRAD: return launch_gemm<Staged<Conv3Dgrad>, 8>(a, s);   // K = 576 (same split as inside K_BWD3: bit-identical)
    case K_CONV3_WGRAD: return launch_gemm<Conv3Wgrad, 8>(a, s);    // K = B*49 split over slabs
    case K_CONV2_DGRAD: return launch_gemm<Conv2Dgrad, 8>(a, s);    // K = 256 per parity class
    case K_CONV2_WGRAD: return launch_gemm<Conv2Wgrad, 8>(a, s);
    case K_CONV1_WGRAD: return launch_gemm<Conv1Wgrad, 16>(a, s);
    // multi-problem launches.  512-thread workgroups (8 waves): at <= 90 VGPRs two of them are resident per CU,
    // so every tile of the launch is resident at once and the problems' latency chains overlap.  Fc4Wgrad is the
    // first problem so its memory-bound read-modify-write stream starts earliest (one tile per wave at B <= 32).
    case K_BWD3:
      // (problem order = dispatch order: the long conv3 tiles first, the streaming fc4 tiles fill in behind them: +1 % step rate)
      if (a.B <= 32 && a.f4w_count > 0) return launch_multi<512, Staged<Conv3Dgrad>, 8, Conv3Wgrad, 8, Fc4Wgrad, 1>(a, true, true, s);
      if (a.f4w_count > 0) return launch_multi<512, Fc4Wgrad, 8, Staged<Conv3Dgrad>, 8, Conv3Wgrad, 8>(a, true, true, s);
      return launch_multi<512, NoProblem, 2, Staged<Conv3Dgrad>, 8, Conv3Wgrad, 8>(a, true, true, s);
    case K_BWD2:
      if (a.B <= 32 && a.f4w_count > 0) return launch_multi<512, Fc4Wgrad, 1, Conv2Dgrad, 8, Conv2Wgrad, 8>(a, true, true, s);
      return launch_multi<512, NoProblem, 2, Conv2Dgrad, 8, Conv2Wgrad, 8>(a, true, true, s);
    case K_BWD1:
      if (a.B <= 32 && a.f4w_count > 0) return launch_multi<1024, Fc4Wgrad, 1, Conv1Wgrad, 16, NoProblem, 2>(a, true, false, s);
      return launch_multi<1024, NoProblem, 2, Conv1Wgrad, 16, NoProblem, 2>(a, true, false, s);
    default: return hipErrorInvalidValue;
  }
}

// ------------------------------------------------------------------------------------------------
// head: one 512-thread workgroup (8 wave64s) per sample; thread j owns hidden unit j.
// AMAX = compile-time bound on num_actions (4 / 8 / 18): every load below is unconditional with a clamped index and
// a select — a conditional load costs hipcc a branch, a scalar pointer re-load and a wait EACH (36 of them measured
// ~3000 cycles here), and the LDS footprint follows the bucket.
// HOIST (option "hoist" only; compiled out of the default kernel — even this one branch was measurable): one extra workgroup
// fetches the next step's indexes from their pinned host slot into HBM.
// QSYS (acting path, round 4): h.q is HOST memory (mapped, pinned) and every Q-value leaves with a system-scope store the moment it is
// summed, so the host can poll for it instead of paying a D2H copy packet + a stream synchronisation (sdqn_api_act.hip: predict_state).
template <int AMAX, bool BN, bool HOIST = false, bool QSYS = false>
__global__ void __launch_bounds__(512) head_kernel(const StepArgs a, const HeadArgs h) {
  SDQN_STAMP(0);
  if constexpr (HOIST) {
    if ((int)blockIdx.x >= a.B) {
      for (int k = threadIdx.x; k < h.next_B; k += 512) h.next_idx_dev[k] = h.next_idx_pinned[k];
      return;
    }
  }
  const int n = blockIdx.x, j = threadIdx.x, lane = j & 63, wave = j >> 6;
  __shared__ float prod[2 * AMAX][NFC];               // 16 KB (A <= 4) .. 72 KB (A <= 18)
  __shared__ float sh_q[2][AMAX];
  __shared__ float sh_dc;
  __shared__ int sh_act;
  // ---- everything this thread will ever load is issued up front (one memory round trip) ----------------
  const int nz = a.nz, A = a.A;
  const float* __restrict__ th0 = a.theta[0];
  const float* __restrict__ th1 = a.theta[nz > 1 ? 1 : 0];
  const float* __restrict__ slab = a.slab4;
  float a4v[2] = {0.0f, 0.0f};
  const int64_t sstride = (int64_t)2 * a.B * NFC;
  float t[2][7];
  if constexpr (BN) {                                // --batch_norm: a4 was activated by the BatchNorm pass (bn_kernels.hip)
#pragma unroll
    for (int z = 0; z < 2; ++z) a4v[z] = a.a4[((int64_t)(z < nz ? z : 0) * a.B + n) * NFC + j];
  } else if (a.S4 == 7) {                            // the built-in split: 14 independent loads in flight
#pragma unroll
    for (int z = 0; z < 2; ++z)
#pragma unroll
      for (int s = 0; s < 7; ++s)
        t[z][s] = slab[s * sstride + ((int64_t)(z < nz ? z : 0) * a.B + n) * NFC + j];
  }
  float w5[2][AMAX];
#pragma unroll
  for (int act = 0; act < AMAX; ++act) {
    const int ac = act < A ? act : A - 1;
    w5[0][act] = th0[OFF5 + ac * NFC + j];                                                        // Affine(A) rows, :91
    w5[1][act] = th1[OFF5 + ac * NFC + j];
  }
  // minibatch metadata last: thread 0 only, and nothing above waits behind it
  int m_act = 0, m_term = 0; int64_t m_rew = 0;
  if (h.train && j == 0) { m_act = h.st_actions[n]; m_rew = h.st_rewards[n]; m_term = h.st_terminals[n]; }
  m_act = m_act < A ? m_act : A - 1;                 // memory safety only: the host rejects out-of-range actions before launching
  if constexpr (BN) {
  } else if (a.S4 == 7) {
#pragma unroll
    for (int z = 0; z < 2; ++z) { float v = 0.0f;
#pragma unroll
      for (int s = 0; s < 7; ++s) v += t[z][s];                                                   // fixed order
      a4v[z] = v; }
  } else {
#pragma unroll
    for (int z = 0; z < 2; ++z) { float v = 0.0f;
      if (z < nz) for (int s = 0; s < a.S4; ++s) v += slab[s * sstride + ((int64_t)z * a.B + n) * NFC + j];
      a4v[z] = v; }
  }
#ifdef SDQN_TIMING
  asm volatile("" :: "v"(a4v[0]), "v"(a4v[1]), "v"(w5[0][0]));
  SDQN_STAMP(1);
#endif
  // fc5 (Affine(A), :91) for both nets: every thread contributes w5[z][act][j] * a4[z][j]; the 2A block-wide sums go
  // through LDS — products transposed into prod[row][512], then wave w reduces rows w, w+8, ... (8 values per lane +
  // ONE 6-step butterfly per row).  A chain of __shfl_xor (= ds_bpermute, ~100 cycles each) per action on every wave
  // measured 5400 cycles here.
#pragma unroll
  for (int z = 0; z < 2; ++z) {                      // static indices only: w5 / a4v stay in registers
    if (z >= nz) continue;
    const float v = fmaxf(a4v[z], 0.0f);                                                          // Rectlin, :89
    a4v[z] = v;
    if constexpr (!BN) a.a4[((int64_t)z * a.B + n) * NFC + j] = v;
#pragma unroll
    for (int act = 0; act < AMAX; ++act)
      if (act < A) prod[z * A + act][j] = w5[z][act] * v;
  }
  SDQN_STAMP(2);
  __syncthreads();
  SDQN_STAMP(3);
  for (int row = wave; row < nz * A; row += 8) {
    float p = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) p += prod[row][lane + 64 * k];                                    // fixed order
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) p += __shfl_xor(p, off, 64);                          // one wavefront reduction
    if (lane == 0) {
      const int z = row / A, act = row - z * A;
      sh_q[z][act] = p;
      if constexpr (QSYS) __hip_atomic_store(&h.q[((int64_t)z * a.B + n) * A + act], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      else h.q[((int64_t)z * a.B + n) * A + act] = p;
    }
  }
  if (!h.train) return;
  __syncthreads();
  SDQN_STAMP(4);
  if (j == 0) {
    const int act = m_act, term = m_term; const int64_t rew = m_rew;
    float m = sh_q[1][0];
    for (int k = 1; k < A; ++k) m = fmaxf(m, sh_q[1][k]);                                       // be.max(postq, axis=0), :124
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
  SDQN_STAMP(5);
  __syncthreads();
  SDQN_STAMP(6);
  const float dc = sh_dc; const int act = sh_act;
  // fc5 dgrad: delta4 = W5^T delta * 1[a4 > 0]; delta is non-zero on the taken action only (W5 row already in registers)
  float wa = 0.0f;
#pragma unroll
  for (int k = 0; k < AMAX; ++k) wa = (k == act) ? w5[0][k] : wa;
  const float d4v = a4v[0] > 0.0f ? wa * dc : 0.0f;
  if (a.h16) a.h_d4[(int64_t)n * NFC + j] = (half_t)(d4v * a.loss_scale);     // fp16 mode: loss-scaled half delta
  else a.d4[(int64_t)n * NFC + j] = d4v;
  if (j < A) h.dq[(int64_t)n * A + j] = (j == act) ? dc : 0.0f;
  SDQN_STAMP(7);
}

#ifdef SDQN_TIMING
hipError_t set_timing_buffer_rb(unsigned long long* p);
hipError_t set_timing_buffer_r3(unsigned long long* p);
hipError_t set_timing_buffer_bt(unsigned long long* p);
hipError_t set_timing_buffer_ss(unsigned long long* p);
hipError_t set_timing_buffer(unsigned long long* p) {
  hipError_t e = set_timing_buffer_rb(p);
  if (e == hipSuccess) e = set_timing_buffer_r3(p);
  if (e == hipSuccess) e = set_timing_buffer_bt(p);
  if (e == hipSuccess) e = set_timing_buffer_ss(p);
  return e != hipSuccess ? e : hipMemcpyToSymbol(HIP_SYMBOL(g_sdqn_dbg), &p, sizeof p);
}
hipError_t set_wave_timing_buffer_rb(unsigned long long* const* p, const unsigned* nb, hipStream_t s);
hipError_t set_wave_timing_buffer_r3(unsigned long long* const* p, const unsigned* nb, hipStream_t s);
hipError_t set_wave_timing_buffer_bt(unsigned long long* const* p, const unsigned* nb, hipStream_t s);
// stream-ordered (the pointer and block count are read from PINNED host words the caller keeps alive): armed right in front of ONE launch of
// a real train step and disarmed behind it, so the stamped launch sees the caches exactly as its predecessors in the step left them
hipError_t set_wave_timing_buffer(unsigned long long* const* p, const unsigned* nb, hipStream_t s) {
  hipError_t e = set_wave_timing_buffer_rb(p, nb, s);
  if (e == hipSuccess) e = set_wave_timing_buffer_r3(p, nb, s);
  if (e == hipSuccess) e = set_wave_timing_buffer_bt(p, nb, s);
  if (e == hipSuccess) e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_sdqn_wdbg_blocks), nb, sizeof *nb, 0, hipMemcpyHostToDevice, s);
  return e != hipSuccess ? e : hipMemcpyToSymbolAsync(HIP_SYMBOL(g_sdqn_wdbg), p, sizeof *p, 0, hipMemcpyHostToDevice, s);
}
#endif

hipError_t launch_head(const StepArgs& a, const HeadArgs& h, hipStream_t s, bool q_system_scope) {
  if (q_system_scope && !a.bn && !h.train) {          // acting path: Q-values straight into mapped host memory
    if (a.A <= 4) SDQN_LAUNCH((head_kernel<4, false, false, true>), dim3(a.B), dim3(512), 0, s, a, h);
    else if (a.A <= 8) SDQN_LAUNCH((head_kernel<8, false, false, true>), dim3(a.B), dim3(512), 0, s, a, h);
    else SDQN_LAUNCH((head_kernel<MAX_ACTIONS, false, false, true>), dim3(a.B), dim3(512), 0, s, a, h);
    return hipGetLastError();
  }
  if (a.bn) SDQN_LAUNCH((head_kernel<MAX_ACTIONS, true>), dim3(a.B), dim3(512), 0, s, a, h);     // --batch_norm (not tuned per bucket)
  else if (a.A <= 4) SDQN_LAUNCH((head_kernel<4, false>), dim3(a.B), dim3(512), 0, s, a, h);
  else if (a.A <= 8) SDQN_LAUNCH((head_kernel<8, false>), dim3(a.B), dim3(512), 0, s, a, h);
  else SDQN_LAUNCH((head_kernel<MAX_ACTIONS, false>), dim3(a.B), dim3(512), 0, s, a, h);
  return hipGetLastError();
}

template <bool OVF>
__global__ void __launch_bounds__(256) update_kernel(const UpdateArgs u) {
  __shared__ float4 part[8][32];
  __shared__ float cost_sh[4096];
#ifndef SDQN_NO_UPDATE_PRELOAD
  // every argument field in flight at once (one wait): left alone hipcc fetches the fields of the by-value UpdateArgs where each is first
  // used — seven dependent round trips to a cold kernel-argument segment before the first vector load of this latency-bound launch
  asm volatile("" :: "s"(u.theta), "s"(u.state), "s"(u.g), "s"(u.slab[0]), "s"(u.slab[1]), "s"(u.slab[2]), "s"(u.dq), "s"(u.a4), "s"(u.cost_terms),
               "s"(u.cost_out), "s"(u.cost_accum), "s"(u.w1p), "s"(u.next.meta), "s"(u.next.idx), "s"(u.next.actions), "s"(u.next.rewards),
               "s"(u.next.terminals), "s"(u.ns[0]), "s"(u.ns[1]), "s"(u.ns[2]), "s"(u.B), "s"(u.A), "s"(u.mode), "s"(u.skip_fc4), "s"(u.next.B),
               "s"(u.next.idx_in_valid), "s"(u.bsz), "s"(u.rho), "s"(u.one_minus_rho), "s"(u.lr));
#endif
  update_body<OVF>(u, (int)blockIdx.x, (int)gridDim.x, part, cost_sh);
}

hipError_t launch_update(const UpdateArgs& u, hipStream_t s) {
  int dense = 2;                                                   // hosts the ride-along prep and the cost mean
  if (!u.skip_fc4) { dense = (NW4 / 4 + 255) / 256; if (dense > 1792) dense = 1792; }
  const dim3 grid(CONV_BLOCKS + u.A * FC5_BLOCKS_PER_ACTION + dense);
  if (u.ovf_flag) SDQN_LAUNCH(update_kernel<true>, grid, dim3(256), 0, s, u);
  else SDQN_LAUNCH(update_kernel<false>, grid, dim3(256), 0, s, u);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// fp16 mode: rebuild wh (master layout) and wht (per-layer transposed, k contiguous) from an fp32 parameter buffer
__global__ void __launch_bounds__(256) refresh16_kernel(const float* theta, half_t* wh, half_t* wht) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < OFF5; e += (int64_t)gridDim.x * 256) {
    const int L = e < OFF2 ? 0 : (e < OFF3 ? 1 : (e < OFF4 ? 2 : 3));
    const int off = L == 0 ? OFF1 : (L == 1 ? OFF2 : (L == 2 ? OFF3 : OFF4));
    const int K = L == 0 ? CRS1 : (L == 1 ? CRS2 : (L == 2 ? CRS3 : NIN4));
    const int N = L == 0 ? K1 : (L == 1 ? K2 : (L == 2 ? K3 : NFC));
    const int64_t r = e - off; const int k = (int)(r / N), n = (int)(r - (int64_t)k * N);
    const half_t w = (half_t)theta[e];
    wh[e] = w; wht[off + (int64_t)n * K + k] = w;
  }
}
hipError_t launch_refresh16(const float* theta, half_t* wh, half_t* wht, hipStream_t s) {
  hipLaunchKernelGGL(refresh16_kernel, dim3(2048), dim3(256), 0, s, theta, wh, wht);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// fp16 mode, data parallel: the gradient crosses xGMI as IEEE half (3.37 MB instead of 6.74 MB per rank and
// step).  g * scale -> half before the all-reduce (scale = a power of two that keeps R summed ranks inside the half
// range), half -> fp32 / scale after it; accumulation in the optimizer stays fp32.  A non-finite value after the
// all-reduce (a half overflow on any rank becomes inf on EVERY rank through the sum) raises the step's overflow flag:
// the apply-only update then leaves parameters and optimizer state untouched on all ranks alike (the usual
// mixed-precision "skip the step" rule) and the skipped-step counter goes up.
// Dynamic payload scale, kept on the device (state = {flag, log2 scale, good steps}; identical on every rank because every
// rank sees the same all-reduced values): halved after an overflow, doubled after 200 clean steps, 2^0 .. 2^15 — RMSProp
// turns a gradient that was flushed to zero into a missing +-lr/sqrt(1-rho) step, so the scale should sit as high as the
// summed gradient allows.  The update launch (update_kernel<true>, block 0) moves it; these two passes only read it.
__global__ void __launch_bounds__(256) grad_to_half_kernel(const float* __restrict__ g, half_t* __restrict__ gh, int64_t n, int* state) {
  const float scale = ldexpf(1.0f, state[1]);
  if (blockIdx.x == 0 && threadIdx.x == 0) state[0] = 0;                 // this step's flag (only the from-half pass, a later launch, sets it)
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 4 <= n) {
      const float4 v = *reinterpret_cast<const float4*>(g + i);
      typedef _Float16 half4 __attribute__((ext_vector_type(4)));
      half4 h; h[0] = (half_t)(v.x * scale); h[1] = (half_t)(v.y * scale); h[2] = (half_t)(v.z * scale); h[3] = (half_t)(v.w * scale);
      *reinterpret_cast<half4*>(gh + i) = h;
    } else for (int64_t k = i; k < n; ++k) gh[k] = (half_t)(g[k] * scale);
  }
}
__global__ void __launch_bounds__(256) grad_from_half_kernel(const half_t* __restrict__ gh, float* __restrict__ g, int64_t n, int* state) {
  const float inv_scale = ldexpf(1.0f, -state[1]);
  int* flag = state;
  bool bad = false;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    const int64_t e = i + 4 <= n ? i + 4 : n;
    for (int64_t k = i; k < e; ++k) { const float v = (float)gh[k]; bad |= !(fabsf(v) <= 65504.0f); g[k] = v * inv_scale; }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
hipError_t launch_grad_to_half(const float* g, half_t* gh, int64_t n, int* state, hipStream_t s) {
  hipLaunchKernelGGL(grad_to_half_kernel, dim3(512), dim3(256), 0, s, g, gh, n, state);
  return hipGetLastError();
}
hipError_t launch_grad_from_half(const half_t* gh, float* g, int64_t n, int* state, hipStream_t s) {
  hipLaunchKernelGGL(grad_from_half_kernel, dim3(512), dim3(256), 0, s, gh, g, n, state);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// prep: the sampled indexes arrive in a pinned host slot; one tiny workgroup copies them into device
// memory and gathers (a, r, t) = (actions, rewards, terminals)[idx] (replay_memory.py:76-78) so that no
// later kernel of the step touches host memory.
__global__ void __launch_bounds__(256) prep_kernel(const PrepArgs p, double* zero8) {
  if (zero8 && threadIdx.x == 0) *zero8 = 0.0;                    // the cost accumulator of a train_many call (instead of a memset launch)
  const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(PrepArgs, idx_in);      // (first kernel parameter)
  for (int n = threadIdx.x; n < p.B; n += 256) {
    const int64_t i = p.idx_in_valid ? *reinterpret_cast<const int64_t*>(ka + 8 * (n & 31)) : p.idx_pinned[n];
    p.idx[n] = i;
    const MetaRec rec = p.meta[i];
    p.actions[n] = rec.action; p.rewards[n] = rec.reward; p.terminals[n] = rec.terminal;
  }
}
hipError_t launch_prep(const PrepArgs& p, hipStream_t s, double* zero8) {
  SDQN_LAUNCH(prep_kernel, dim3(1), dim3(256), 0, s, p, zero8);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// standalone replay gather: 16 B/lane coalesced loads of the 5 contiguous frames screens[i-4 : i+1],
// each written to prestates[k] (frames 0..3) and poststates[k] (frames 1..4).  7056 = 441 * 16.
// INLINE (B <= 256, what getMinibatch() uses): the sampled indexes travel IN the kernel-argument block — they are host data at
// launch time anyway (replay_memory.py:54-68 samples on the host) — so a workgroup's index is one scalar load from the
// argument segment instead of a pointer load followed by a dependent global load: one memory round trip less on the launch's
// critical path (the whole launch is three of them), no pinned index slot, no slot-release event in the stream.
struct IdxBlock { int64_t v[256]; };
template <bool INLINE>
__global__ void __launch_bounds__(256) gather_kernel(const GatherArgs g, const IdxBlock ib) {
  const int n = blockIdx.y;
  const int64_t index = INLINE ? ib.v[n] : g.idx[n];
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

hipError_t launch_gather(const GatherArgs& g, hipStream_t s, const int64_t* host_idx) {
  if (host_idx && g.B <= 256) {
    IdxBlock ib;
    memcpy(ib.v, host_idx, (size_t)g.B * sizeof(int64_t));
    SDQN_LAUNCH(gather_kernel<true>, dim3(9, g.B), dim3(256), 0, s, g, ib);
  } else {
    IdxBlock ib; ib.v[0] = 0;
    SDQN_LAUNCH(gather_kernel<false>, dim3(9, g.B), dim3(256), 0, s, g, ib);
  }
  return hipGetLastError();
}

}  // namespace sdqn
