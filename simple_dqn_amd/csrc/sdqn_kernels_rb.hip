// sdqn_kernels_rb.hip — instantiations of the register-blocked tile routine (gemm_engine_rb.h) for the throughput
// regime (B >= 128, BASELINE.json configs[2]).  One menu per kernel id: entry 0 is "use the unblocked routine"
// (sdqn_kernels.hip), entries >= 1 are (RM, RN, waves per tile) choices; the defaults below were picked with
// tools/sweep_rb.py on an MI355X at B = 256 (profiles/README.md), `sdqn_net_set_option "rb:<id>"` overrides them.
#include "gemm_engine.h"
#include "kernels.h"

namespace sdqn {

#define RB_CASE(N, P, RM, RN, NW) case N: return launch_gemm<RB<P, RM, RN>, NW>(a, s)

hipError_t launch_kernel_rb(int id, int menu, const StepArgs& a, hipStream_t s) {
  switch (id) {
    case K_CONV1_FWD:                       // M = 2 B 400, N = 32, K = 256 (8 chunks)
      switch (menu) { RB_CASE(1, Conv1Fwd, 2, 1, 1); RB_CASE(2, Conv1Fwd, 4, 1, 1); RB_CASE(3, Conv1Fwd, 2, 1, 2); RB_CASE(4, Conv1Fwd, 4, 1, 2); default: break; }
      break;
    case K_CONV2_FWD:                       // M = 2 B 81, N = 64, K = 512 (16 chunks)
      switch (menu) { RB_CASE(1, Conv2Fwd, 2, 2, 2); RB_CASE(2, Conv2Fwd, 2, 2, 4); RB_CASE(3, Conv2Fwd, 1, 2, 4); RB_CASE(4, Conv2Fwd, 2, 2, 1); RB_CASE(5, Conv2Fwd, 1, 2, 2); default: break; }
      break;
    case K_CONV3_FWD:                       // M = 2 B 49, N = 64, K = 576 (18 chunks)
      switch (menu) { RB_CASE(1, Conv3Fwd, 2, 2, 2); RB_CASE(2, Conv3Fwd, 2, 2, 4); RB_CASE(3, Conv3Fwd, 1, 2, 4); RB_CASE(4, Conv3Fwd, 2, 2, 3); RB_CASE(5, Conv3Fwd, 1, 2, 2); default: break; }
      break;
    case K_FC4_FWD:                         // M = B per net, N = 512, K = 3136 in S4 = 7 slabs of 14 chunks
      switch (menu) { RB_CASE(1, Fc4Fwd, 2, 2, 2); RB_CASE(2, Fc4Fwd, 2, 2, 4); RB_CASE(3, Fc4Fwd, 1, 2, 2); RB_CASE(4, Fc4Fwd, 2, 2, 7); default: break; }
      break;
    case K_FC4_DGRAD:                       // M = B, N = 3136, K = 512
      switch (menu) { RB_CASE(1, Fc4Dgrad, 2, 2, 4); RB_CASE(2, Fc4Dgrad, 2, 2, 8); RB_CASE(3, Fc4Dgrad, 2, 2, 2); RB_CASE(4, Fc4Dgrad, 1, 2, 4); default: break; }
      break;
    case K_FC4_WGRAD:                       // M = 3136, N = 512, K = B
      switch (menu) { RB_CASE(1, Fc4Wgrad, 2, 2, 1); RB_CASE(2, Fc4Wgrad, 2, 2, 2); RB_CASE(3, Fc4Wgrad, 2, 2, 4); RB_CASE(4, Fc4Wgrad, 1, 2, 2); default: break; }
      break;
    case K_CONV3_DGRAD:                     // M = B 81, N = 64, K = 576
      switch (menu) { RB_CASE(1, Conv3Dgrad, 2, 2, 2); RB_CASE(2, Conv3Dgrad, 2, 2, 4); RB_CASE(3, Conv3Dgrad, 2, 2, 1); RB_CASE(4, Conv3Dgrad, 1, 2, 4); default: break; }
      break;
    case K_CONV3_WGRAD:                     // M = 576, N = 64, K = B 49 in slabs (tps3)
      switch (menu) { RB_CASE(1, Conv3Wgrad, 2, 2, 8); RB_CASE(2, Conv3Wgrad, 2, 2, 4); RB_CASE(3, Conv3Wgrad, 1, 2, 8); RB_CASE(4, Conv3Wgrad, 2, 1, 8); default: break; }
      break;
    case K_CONV2_DGRAD:                     // M = B 100 per parity class (x 4), N = 32, K = 256
      switch (menu) { RB_CASE(1, Conv2Dgrad, 2, 1, 1); RB_CASE(2, Conv2Dgrad, 4, 1, 1); RB_CASE(3, Conv2Dgrad, 2, 1, 2); RB_CASE(4, Conv2Dgrad, 4, 1, 2); default: break; }
      break;
    case K_CONV2_WGRAD:                     // M = 512, N = 64, K = B 81 in slabs (tps2)
      switch (menu) { RB_CASE(1, Conv2Wgrad, 2, 2, 8); RB_CASE(2, Conv2Wgrad, 2, 2, 4); RB_CASE(3, Conv2Wgrad, 1, 2, 8); RB_CASE(4, Conv2Wgrad, 2, 1, 8); default: break; }
      break;
    case K_CONV1_WGRAD:                     // M = 256, N = 32, K = B 400 in slabs (tps1); u8 patches re-gathered from the ring
      switch (menu) { RB_CASE(1, Conv1Wgrad, 2, 1, 8); RB_CASE(2, Conv1Wgrad, 2, 1, 16); RB_CASE(3, Conv1Wgrad, 4, 1, 8); RB_CASE(4, Conv1Wgrad, 1, 1, 8); default: break; }
      break;
    default: break;
  }
  return hipErrorInvalidValue;
}

#ifdef SDQN_TIMING
// every translation unit has its own copy of the stamp-buffer pointer (gemm_engine.h)
hipError_t set_timing_buffer_rb(unsigned long long* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_sdqn_dbg), &p, sizeof p); }
#endif

}  // namespace sdqn
