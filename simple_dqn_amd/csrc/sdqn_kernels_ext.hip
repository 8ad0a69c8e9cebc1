// sdqn_kernels_ext.hip — every launch variant that is not the default fp32 step:
//   * --datatype float16 (problems_h16.h: packed-fp16 MFMA forward / dgrad, LDS-transposed packed-fp16 weight gradients),
//   * what a float16 launch runs at B >= 128 when `bt:<id>` = -1 takes it off the half block-tile routine (the same-box reference of
//     tools/sweep_bt.py): the latency regime's kernels.
// Kept apart from sdqn_kernels.hip on purpose: see the note there.
#include "gemm_engine.h"
#include "problems_h16.h"
#include "kernels.h"

namespace sdqn {

template <class P>
static hipError_t launch_nw(int nw, const StepArgs& a, hipStream_t s) {
  switch (nw) {
    case 1: return launch_gemm<P, 1>(a, s);
    case 2: return launch_gemm<P, 2>(a, s);
    case 4: return launch_gemm<P, 4>(a, s);
    case 8: return launch_gemm<P, 8>(a, s);
    case 16: return launch_gemm<P, 16>(a, s);
    default: return hipErrorInvalidValue;
  }
}



// fp16 mode: forward / dgrad on packed-fp16 MFMA (problems_h16.h), wgrad on the fp32 engine with half operands
static hipError_t launch_kernel_h16(int id, const StepArgs& a, const LaunchTune& t, hipStream_t s) {
  if (id >= 0 && id < 12 && t.nw_override[id] > 0) {          // tuning hook (sdqn_net_set_option "nw:<id>")
    const int nw = t.nw_override[id];
    switch (id) {
      case K_CONV1_FWD: return launch_nw<Conv1FwdH>(nw, a, s);
      case K_CONV2_FWD: return launch_nw<Conv2FwdH>(nw, a, s);
      case K_CONV3_FWD: return launch_nw<Conv3FwdH>(nw, a, s);
      case K_FC4_FWD: return launch_nw<Fc4FwdH>(nw, a, s);
      case K_FC4_DGRAD: return launch_nw<Fc4DgradH>(nw, a, s);
      case K_CONV3_DGRAD: return launch_nw<Conv3DgradH>(nw, a, s);
      case K_CONV2_DGRAD: return launch_nw<Conv2DgradH>(nw, a, s);
      case K_FC4_WGRAD: return launch_nw<Fc4WgradHW>(nw, a, s);
      case K_CONV3_WGRAD: return launch_nw<Conv3WgradHW>(nw, a, s);
      case K_CONV2_WGRAD: return launch_nw<Conv2WgradHW>(nw, a, s);
      case K_CONV1_WGRAD: return launch_nw<Conv1WgradHW>(nw, a, s);
      default: break;
    }
  }
  if (a.h16 == 2) {            // weight gradients on packed-fp16 MFMA too (default); h16 == 1: fp32 MFMA with half operands (round 1)
    switch (id) {
      case K_FC4_WGRAD:
        if (a.B <= 32) return launch_gemm<Fc4WgradHW, 1>(a, s);
        return launch_gemm<Fc4WgradHW, 8>(a, s);
      case K_CONV3_WGRAD: return launch_gemm<Conv3WgradHW, 8>(a, s);
      case K_CONV2_WGRAD: return launch_gemm<Conv2WgradHW, 8>(a, s);
      case K_CONV1_WGRAD: return launch_gemm<Conv1WgradHW, 16>(a, s);
      case K_BWD3:
        if (a.B <= 32) return launch_multi<512, Conv3DgradH, 8, Conv3WgradHW, 8, Fc4WgradHW, 1>(a, true, true, s);
        return launch_multi<512, Conv3DgradH, 8, Conv3WgradHW, 8, Fc4WgradHW, 8>(a, true, true, s);
      case K_BWD2: return launch_multi<512, NoProblem, 2, Conv2DgradH, 8, Conv2WgradHW, 8>(a, true, true, s);
      case K_BWD1: return launch_multi<1024, NoProblem, 2, Conv1WgradHW, 16, NoProblem, 2>(a, true, false, s);
      // round 4, B >= 128: every weight gradient that does not need delta1 in ONE launch, behind the block-tile dgrad chain (sdqn_api_step.hip)
      case K_WGRADS: return launch_multi<512, Fc4WgradHW, 8, Conv3WgradHW, 8, Conv2WgradHW, 8>(a, true, true, s);
      default: break;
    }
  }
  // (B >= 128 comes here only when a launch is taken off the half block-tile routine — `bt:<id>` = -1, the same-box reference of
  //  tools/sweep_bt.py; round 3's register-blocked forward routine for that case left the tree in round 6: tools/exp/gemm_engine_rb.h)
  switch (id) {
    case K_CONV1_FWD: return launch_gemm<Conv1FwdH, 8>(a, s);
    case K_CONV2_FWD: return launch_gemm<Conv2FwdH, 16>(a, s);
    case K_CONV3_FWD: return launch_gemm<Conv3FwdH, 16>(a, s);     // 18 chunks over 16 waves (9 waves x 2 chunks: -0.3 %)
    case K_FC4_FWD: return launch_gemm<Fc4FwdH, 14>(a, s);
    case K_FC4_DGRAD: return launch_gemm<Fc4DgradH, 16>(a, s);
    case K_FC4_WGRAD:
      if (a.B <= 32) return launch_gemm<Fc4WgradH, 1>(a, s);
      return launch_gemm<Fc4WgradH, 8>(a, s);
    case K_CONV3_DGRAD: return launch_gemm<Conv3DgradH, 8>(a, s);
    case K_CONV3_WGRAD: return launch_gemm<Conv3WgradH, 8>(a, s);
    case K_CONV2_DGRAD: return launch_gemm<Conv2DgradH, 8>(a, s);
    case K_CONV2_WGRAD: return launch_gemm<Conv2WgradH, 8>(a, s);
    case K_CONV1_WGRAD: return launch_gemm<Conv1WgradH, 16>(a, s);
    case K_BWD3:
      if (a.B <= 32) return launch_multi<512, Fc4WgradH, 1, Conv3DgradH, 8, Conv3WgradH, 8>(a, true, true, s);
      return launch_multi<512, Fc4WgradH, 8, Conv3DgradH, 8, Conv3WgradH, 8>(a, true, true, s);
    case K_BWD2: return launch_multi<512, NoProblem, 2, Conv2DgradH, 8, Conv2WgradH, 8>(a, true, true, s);
    case K_BWD1: return launch_multi<1024, NoProblem, 2, Conv1WgradH, 16, NoProblem, 2>(a, true, false, s);
    case K_WGRADS: return launch_multi<512, Fc4WgradH, 8, Conv3WgradH, 8, Conv2WgradH, 8>(a, true, true, s);     // (h16_wgrad_mfma = 0: fp32 MFMA on the half operands)
    default: return hipErrorInvalidValue;
  }
}



hipError_t launch_kernel_ext(int id, const StepArgs& a, const LaunchTune& t, hipStream_t s, bool* handled) {
  *handled = true;
  if (a.h16) return launch_kernel_h16(id, a, t, s);
  *handled = false;
  return hipSuccess;
}

#ifdef SDQN_TIMING
// every translation unit has its own copy of the stamp-buffer pointer (gemm_engine.h)
hipError_t set_timing_buffer_rb(unsigned long long* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_sdqn_dbg), &p, sizeof p); }
hipError_t set_wave_timing_buffer_rb(unsigned long long* const* p, const unsigned* nb, hipStream_t s) {
  hipError_t e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_sdqn_wdbg_blocks), nb, sizeof *nb, 0, hipMemcpyHostToDevice, s);
  return e != hipSuccess ? e : hipMemcpyToSymbolAsync(HIP_SYMBOL(g_sdqn_wdbg), p, sizeof *p, 0, hipMemcpyHostToDevice, s);
}
#endif

}  // namespace sdqn
