// sdqn_kernels_ss.hip — the throughput regime's forward convolutions on the SAMPLE-STATIONARY routine (conv_ss.h): own translation
// unit, like every other family of launch variants.
//   conv2_fwd  a1 [2][B][20][20][32] -> a2 [2][B][81][64]     4 x 4 stride 2     (deepqnetwork.py:85)
//   conv3_fwd  a2 [2][B][9][9][64]   -> a3 [2][B][49][64]     3 x 3 stride 1     (deepqnetwork.py:87)
// float32, no batch-norm (the raw-output problems stay on the latency engine), B >= 128.  LaunchTune::bt[id] == 0: this routine where its
// workgroups fill the chip (below), else the block-tile engine's built-in shape; 7: this routine always; other menu entries > 0: the
// block-tile engine's block shapes (sdqn_kernels_bt.hip; entry 6 = its built-in 64 x 64 shape).
#include <stdlib.h>
#include "conv_ss.h"
#include "kernels.h"

namespace sdqn {

// two samples per workgroup when one per workgroup would not fit the chip in one round (nz B > 256 workgroups), else one
// (the last three numbers: pixel / row / sample padding of the LDS image in floats — conflict-free fragment reads, tools/exp/ss_bank_search.py)
typedef ss::Cfg<P1, Q1, K1, 4, 4, ST2, P2, Q2, 2, 4, 20, 56> C2S2;
typedef ss::Cfg<P1, Q1, K1, 4, 4, ST2, P2, Q2, 1, 4, 20, 0> C2S1;
typedef ss::Cfg<P2, Q2, K2, 3, 3, 1, P3, Q3, 2, 8, 48, 16> C3S2;
typedef ss::Cfg<P2, Q2, K2, 3, 3, 1, P3, Q3, 1, 8, 48, 0> C3S1;

hipError_t launch_kernel_ss(int id, const StepArgs& a, const LaunchTune& t, hipStream_t s, bool* handled) {
  *handled = false;
  if (a.B < 128 || a.bn || a.h16) return hipSuccess;
  if (id != K_CONV2_FWD && id != K_CONV3_FWD) return hipSuccess;
  if ((t.bt[id] != 0 && t.bt[id] != 7) || t.nw_override[id] > 0) return hipSuccess;      // (menu entry 7: this routine whatever the batch size)
  // one workgroup per CU, NS whole samples each: the routine pays when its workgroups fill (nearly) whole rounds of the chip's 256 CUs —
  // B = 128 and 256 with both nets, B = 256 alone (predict) — and loses to the block-tile engine's finer blocks in between (measured,
  // conv2 / conv3 forward, us: B = 160: 23.8 / 17.4 against 20.7 / 14.0; B = 256: 26.1 / 18.9 against 28.7 / 22.3): below 80 % it declines
  const int ns = a.nz * a.B > 256 ? 2 : 1;
  const int wgs = a.nz * ((a.B + ns - 1) / ns), rounds = (wgs + 255) / 256;
  if (t.bt[id] == 0 && wgs * 5 < rounds * 256 * 4) return hipSuccess;
  ss::Args c;
  c.B = a.B; c.G = (a.B + ns - 1) / ns; c.dbg = 0;
#ifdef SDQN_TIMING
  if (const char* e = getenv("SDQN_SS_DBG")) c.dbg = atoi(e);
#endif
  *handled = true;
  if (id == K_CONV2_FWD) {
    c.in = a.a1; c.out = a.a2; c.w[0] = a.theta[0] + OFF2; c.w[1] = a.theta[a.nz > 1 ? 1 : 0] + OFF2; c.wt = t.wt & 1;
    return ns == 2 ? ss::launch<C2S2>(c, a.nz, s) : ss::launch<C2S1>(c, a.nz, s);
  }
  c.in = a.a2; c.out = a.a3; c.w[0] = a.theta[0] + OFF3; c.w[1] = a.theta[a.nz > 1 ? 1 : 0] + OFF3; c.wt = (t.wt >> 1) & 1;
  return ns == 2 ? ss::launch<C3S2>(c, a.nz, s) : ss::launch<C3S1>(c, a.nz, s);
}

#ifdef SDQN_TIMING
hipError_t set_timing_buffer_ss(unsigned long long* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_sdqn_dbg), &p, sizeof p); }
#endif

}  // namespace sdqn
