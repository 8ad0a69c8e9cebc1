// sdqn_kernels_ss.hip — the SAMPLE-STATIONARY convolution launches (round 6): own translation unit, like every other family of launch variants.
//   float32, B >= 128 (conv_ss.h):   conv2_fwd  a1 [2][B][20][20][32] -> a2 [2][B][81][64]   4 x 4 stride 2   (deepqnetwork.py:85)
//                                    conv3_fwd  a2 [2][B][9][9][64]   -> a3 [2][B][49][64]   3 x 3 stride 1   (deepqnetwork.py:87)
//     no batch-norm (the raw-output problems stay on the latency engine).  LaunchTune::bt[id] == 0: this routine where its workgroups fill the
//     chip (below), else the block-tile engine's built-in shape; 7: this routine always; 8: always, and never chained with the other layer;
//     other menu entries > 0: the block-tile engine's block shapes (sdqn_kernels_bt.hip; entry 6 = its built-in 64 x 64).  When both layers
//     run here they are ONE launch (conv_ss_chain_kernel: a workgroup's conv3 follows its own conv2 behind a barrier).
//   float16, any B (conv_ssh.h):     conv1 -> conv2 -> conv3 as one launch (conv_ssh_chain_kernel; K_CONV1_FWD and K_CONV3_FWD then launch nothing),
//     and — B >= 128, where the two dgrads are launches of their own — conv3_dgrad -> conv2_dgrad as one launch (conv_ssh_dgrad_chain_kernel; deepqnetwork.py:162).  Menu entries
//     0 / 7: these launches; 8: with plain (write-back) stores; 6 (any other): the packed-fp16 routines they replace.
#include <stdlib.h>
#include "conv_ss.h"
#include "conv_ssh.h"
#include "kernels.h"

namespace sdqn {

// two samples per workgroup when one per workgroup would not fit the chip in one round (nz B > 256 workgroups), else one
// (the last three numbers: pixel / row / sample padding of the LDS image in floats — conflict-free fragment reads, tools/exp/ss_bank_search.py)
typedef ss::Cfg<P1, Q1, K1, 4, 4, ST2, P2, Q2, 2, 4, 20, 56> C2S2;
typedef ss::Cfg<P1, Q1, K1, 4, 4, ST2, P2, Q2, 1, 4, 20, 0> C2S1;
typedef ss::Cfg<P2, Q2, K2, 3, 3, 1, P3, Q3, 2, 8, 48, 16> C3S2;
typedef ss::Cfg<P2, Q2, K2, 3, 3, 1, P3, Q3, 1, 8, 48, 0> C3S1;

// does launch `id` (conv2 / conv3 forward) run on this routine?  LaunchTune::bt[id]: 0 = where its workgroups fill the chip, 7 / 8 = always
static bool ss_takes(int id, const StepArgs& a, const LaunchTune& t) {
  if (a.B < 128 || a.bn || a.h16) return false;
  if ((t.bt[id] != 0 && t.bt[id] != 7 && t.bt[id] != 8) || t.nw_override[id] > 0) return false;
  // one workgroup per CU, NS whole samples each: the routine pays when its workgroups fill (nearly) whole rounds of the chip's 256 CUs —
  // B = 128 and 256 with both nets, B = 256 alone (predict) — and loses to the block-tile engine's finer blocks in between (measured,
  // conv2 / conv3 forward, us: B = 160: 23.8 / 17.4 against 20.7 / 14.0; B = 256: 26.1 / 18.9 against 28.7 / 22.3): below 80 % it declines
  const int ns = a.nz * a.B > 256 ? 2 : 1;
  const int wgs = a.nz * ((a.B + ns - 1) / ns), rounds = (wgs + 255) / 256;
  return t.bt[id] != 0 || wgs * 5 >= rounds * 256 * 4;
}
// conv2 and conv3 forward both on this routine: ONE launch (conv_ss_chain_kernel) at K_CONV2_FWD, nothing at K_CONV3_FWD — unless menu
// entry 8 asks for the two launches (tests, same-box A/B)
static bool ss_chains(const StepArgs& a, const LaunchTune& t) {
  return ss_takes(K_CONV2_FWD, a, t) && ss_takes(K_CONV3_FWD, a, t) && t.bt[K_CONV2_FWD] != 8 && t.bt[K_CONV3_FWD] != 8;
}

// float16 mode, ANY batch size: conv2 -> conv3 forward as one launch (conv_ssh.h).  No fill rule here — the launch is data movement and
// latency, not matrix time, and wins wherever it was measured (fused-loop steps/s, chain against the launches it replaces: B = 32 18 894 vs
// 17 594, 64 13 399 vs 12 006, 100 10 487 vs 9 115, 160 12 181 vs 11 711, 192 11 786 vs 11 138).  Menu entries bt:1 / bt:2: 0 / 7 = this
// launch, 8 = with plain (write-back) stores, anything else the packed-fp16 routines (latency engine / block tiles: launch_single_h)
static bool ssh_takes(const StepArgs& a, const LaunchTune& t) {
  if (!a.h16 || a.bn) return false;
  for (int id : {K_CONV2_FWD, K_CONV3_FWD}) if ((t.bt[id] != 0 && t.bt[id] != 7 && t.bt[id] != 8) || t.nw_override[id] > 0) return false;
  return t.bt[K_CONV2_FWD] == t.bt[K_CONV3_FWD];
}

// float16 mode, B >= 128 (where the two dgrads are launches of their own: step structure h16_block_tile): conv3_dgrad -> conv2_dgrad as one
// launch (conv_ssh.h: one workgroup per sample of the online net), same menu on bt:7 / bt:9.  Measured with the forward chain on, steps/s:
// B = 128 14 761 vs 13 641, 160 13 451 vs 12 181, 192 13 059 vs 11 786, 256 11 900 vs 10 575
static bool ssh_dgrad_takes(const StepArgs& a, const LaunchTune& t) {
  if (!a.h16 || a.B < 128 || a.bn) return false;
  for (int id : {K_CONV3_DGRAD, K_CONV2_DGRAD}) if ((t.bt[id] != 0 && t.bt[id] != 7 && t.bt[id] != 8) || t.nw_override[id] > 0) return false;
  return t.bt[K_CONV3_DGRAD] == t.bt[K_CONV2_DGRAD];
}

// float16: conv1 rides in FRONT of the forward chain (the workgroup computes its own samples' a1 from the frames: conv_ssh.h, C1) wherever the
// chain runs and nothing asks for a conv1 launch of its own (bt:0 = 0, no nw override): then K_CONV1_FWD launches nothing
static bool ssh_c1(const StepArgs& a, const LaunchTune& t) {
  return ssh_takes(a, t) && t.bt[K_CONV1_FWD] == 0 && t.nw_override[K_CONV1_FWD] == 0 && a.idx_t == nullptr && a.src != nullptr;
}

hipError_t launch_kernel_ss(int id, const StepArgs& a, const LaunchTune& t, hipStream_t s, bool* handled) {
  *handled = false;
  if (id == K_CONV1_FWD) {
    if (a.h16 && ssh_c1(a, t)) *handled = true;                  // (rides in the conv2 launch)
    return hipSuccess;
  }
  if (id == K_CONV3_DGRAD || id == K_CONV2_DGRAD) {
    if (!ssh_dgrad_takes(a, t)) return hipSuccess;
    *handled = true;
    if (id == K_CONV2_DGRAD) return hipSuccess;                  // (rode in the conv3_dgrad launch)
    ssh::DArgs c; c.d3p = a.h_d3p; c.w3 = a.wh[0] + OFF3; c.w2 = a.wh[0] + OFF2; c.a2 = a.h_a2; c.a1 = a.h_a1; c.d2 = a.h_d2; c.d1 = a.h_d1; c.B = a.B;
    return t.bt[K_CONV3_DGRAD] != 8 ? ssh::launch_dgrad_chain<true>(c, s) : ssh::launch_dgrad_chain<false>(c, s);
  }
  if (id != K_CONV2_FWD && id != K_CONV3_FWD) return hipSuccess;
  if (a.h16) {
    if (!ssh_takes(a, t)) return hipSuccess;
    *handled = true;
    if (id == K_CONV3_FWD) return hipSuccess;                    // (rode in the conv2 launch)
    const int ns = a.nz * a.B > 256 ? 2 : 1, z1 = a.nz > 1 ? 1 : 0;
    ssh::Args c; c.a1 = a.h_a1; c.a2 = a.h_a2; c.a3 = a.h_a3; c.B = a.B; c.G = (a.B + ns - 1) / ns;
    c.w2[0] = a.wht[0] + OFF2; c.w2[1] = a.wht[z1] + OFF2; c.w3[0] = a.wht[0] + OFF3; c.w3[1] = a.wht[z1] + OFF3;
    const bool wt = t.bt[K_CONV2_FWD] != 8, c1 = ssh_c1(a, t);
    c.src = a.src; c.idx = a.idx; c.from_ring = a.from_ring; c.a1w = a.h_a1; c.w1[0] = a.wht[0] + OFF1; c.w1[1] = a.wht[z1] + OFF1;
    if (c1) {
      if (ns == 2) return wt ? ssh::launch_chain<2, true, true>(c, a.nz, s) : ssh::launch_chain<2, false, true>(c, a.nz, s);
      return wt ? ssh::launch_chain<1, true, true>(c, a.nz, s) : ssh::launch_chain<1, false, true>(c, a.nz, s);
    }
    if (ns == 2) return wt ? ssh::launch_chain<2, true, false>(c, a.nz, s) : ssh::launch_chain<2, false, false>(c, a.nz, s);
    return wt ? ssh::launch_chain<1, true, false>(c, a.nz, s) : ssh::launch_chain<1, false, false>(c, a.nz, s);
  }
  if (!ss_takes(id, a, t)) return hipSuccess;
  *handled = true;
  const bool chain = ss_chains(a, t);
  if (chain && id == K_CONV3_FWD) return hipSuccess;            // (rode in the conv2 launch)
  const int ns = a.nz * a.B > 256 ? 2 : 1;
  ss::Args c2, c3;
  c2.B = c3.B = a.B; c2.G = c3.G = (a.B + ns - 1) / ns; c2.dbg = c3.dbg = 0;
#ifdef SDQN_TIMING
  if (const char* e = getenv("SDQN_SS_DBG")) c2.dbg = c3.dbg = atoi(e);
#endif
  c2.in = a.a1; c2.out = a.a2; c2.w[0] = a.theta[0] + OFF2; c2.w[1] = a.theta[a.nz > 1 ? 1 : 0] + OFF2; c2.wt = t.wt & 1;
  c3.in = a.a2; c3.out = a.a3; c3.w[0] = a.theta[0] + OFF3; c3.w[1] = a.theta[a.nz > 1 ? 1 : 0] + OFF3; c3.wt = (t.wt >> 1) & 1;
  if (chain) {
    ss::ChainArgs cc; cc.l1 = c2; cc.l2 = c3;
    return ns == 2 ? ss::launch_chain<C2S2, C3S2>(cc, a.nz, s) : ss::launch_chain<C2S1, C3S1>(cc, a.nz, s);
  }
  if (id == K_CONV2_FWD) return ns == 2 ? ss::launch<C2S2>(c2, a.nz, s) : ss::launch<C2S1>(c2, a.nz, s);
  return ns == 2 ? ss::launch<C3S2>(c3, a.nz, s) : ss::launch<C3S1>(c3, a.nz, s);
}

#ifdef SDQN_TIMING
hipError_t set_timing_buffer_ss(unsigned long long* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_sdqn_dbg), &p, sizeof p); }
#endif

}  // namespace sdqn
