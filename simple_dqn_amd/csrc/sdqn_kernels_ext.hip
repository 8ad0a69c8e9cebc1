// sdqn_kernels_ext.hip — every launch variant that is not the default fp32 step:
//   * --datatype float16 (problems_h16.h: packed-fp16 MFMA forward / dgrad, LDS-transposed packed-fp16 weight gradients),
//   * option "hoist" (the next step's target-net forward riding in this step's launches; measured slower, off by default),
//   * the register-blocked tile routine for B >= 128 (gemm_engine_rb.h; measured slower, `set_option "rb:<id>"`).
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

#ifdef SDQN_EXPERIMENTS      // the fp32 register-blocked menus, "hoist" and "bwd_order": built, correct, measured slower (tools/exp/README.md)
#define RB_CASE(N, P, RM, RN, NW) case N: return launch_gemm<RB<P, RM, RN>, NW>(a, s)

static hipError_t launch_kernel_rb(int id, int menu, const StepArgs& a, hipStream_t s) {
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
#endif  // SDQN_EXPERIMENTS


// fp16 mode: forward / dgrad on packed-fp16 MFMA (problems_h16.h), wgrad on the fp32 engine with half operands
// register-blocked packed-fp16 forward / dgrad (gemm_tile_hb) for B >= 128: menu per kernel id (tools/sweep_rb.py DATATYPE=float16)
#define HB_CASE(N, P, RM, RN, NW) case N: return launch_gemm<RB<P, RM, RN>, NW>(a, s)
static hipError_t launch_kernel_hb(int id, int menu, const StepArgs& a, hipStream_t s) {
  switch (id) {
    case K_CONV1_FWD: switch (menu) { HB_CASE(1, Conv1FwdH, 2, 1, 1); HB_CASE(2, Conv1FwdH, 4, 1, 1); HB_CASE(3, Conv1FwdH, 2, 1, 2); default: break; } break;
    case K_CONV2_FWD: switch (menu) { HB_CASE(1, Conv2FwdH, 2, 2, 1); HB_CASE(2, Conv2FwdH, 2, 2, 2); HB_CASE(3, Conv2FwdH, 1, 2, 1); HB_CASE(4, Conv2FwdH, 2, 2, 4); HB_CASE(5, Conv2FwdH, 4, 2, 1); default: break; } break;
    case K_CONV3_FWD: switch (menu) { HB_CASE(1, Conv3FwdH, 2, 2, 1); HB_CASE(2, Conv3FwdH, 2, 2, 2); HB_CASE(3, Conv3FwdH, 1, 2, 1); HB_CASE(4, Conv3FwdH, 2, 2, 4); HB_CASE(5, Conv3FwdH, 4, 2, 1); default: break; } break;
    case K_FC4_FWD: switch (menu) { HB_CASE(1, Fc4FwdH, 2, 2, 1); HB_CASE(2, Fc4FwdH, 2, 2, 2); HB_CASE(3, Fc4FwdH, 4, 2, 1); HB_CASE(4, Fc4FwdH, 2, 4, 1); default: break; } break;
    case K_FC4_DGRAD: switch (menu) { HB_CASE(1, Fc4DgradH, 2, 2, 1); HB_CASE(2, Fc4DgradH, 2, 2, 2); HB_CASE(3, Fc4DgradH, 4, 2, 1); HB_CASE(4, Fc4DgradH, 2, 4, 1); default: break; } break;
    case K_CONV3_DGRAD: switch (menu) { HB_CASE(1, Conv3DgradH, 2, 2, 1); HB_CASE(2, Conv3DgradH, 2, 2, 2); HB_CASE(3, Conv3DgradH, 1, 2, 1); HB_CASE(4, Conv3DgradH, 4, 2, 1); default: break; } break;
    case K_CONV2_DGRAD: switch (menu) { HB_CASE(1, Conv2DgradH, 2, 1, 1); HB_CASE(2, Conv2DgradH, 4, 1, 1); HB_CASE(3, Conv2DgradH, 2, 1, 2); default: break; } break;
    default: break;
  }
  return hipErrorInvalidValue;
}

static hipError_t launch_kernel_h16(int id, const StepArgs& a, const LaunchTune& t, hipStream_t s) {
  if (a.B >= 128 && id >= 0 && id < 12 && t.rb[id] > 0) {
    const hipError_t e = launch_kernel_hb(id, t.rb[id], a, s);
    if (e != hipErrorInvalidValue) return e;
  }
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
        if (t.order == 1) {        // round 1's dispatch order (experiment)
          if (a.B <= 32) return launch_multi<512, Fc4WgradHW, 1, Conv3DgradH, 8, Conv3WgradHW, 8>(a, true, true, s);
          return launch_multi<512, Fc4WgradHW, 8, Conv3DgradH, 8, Conv3WgradHW, 8>(a, true, true, s);
        }
        if (a.B <= 32) return launch_multi<512, Conv3DgradH, 8, Conv3WgradHW, 8, Fc4WgradHW, 1>(a, true, true, s);
        return launch_multi<512, Conv3DgradH, 8, Conv3WgradHW, 8, Fc4WgradHW, 8>(a, true, true, s);
      case K_BWD2: return launch_multi<512, NoProblem, 2, Conv2DgradH, 8, Conv2WgradHW, 8>(a, true, true, s);
      case K_BWD1: return launch_multi<1024, NoProblem, 2, Conv1WgradHW, 16, NoProblem, 2>(a, true, false, s);
      // round 4, B >= 128: every weight gradient that does not need delta1 in ONE launch, behind the block-tile dgrad chain (sdqn_api.hip)
      case K_WGRADS: return launch_multi<512, Fc4WgradHW, 8, Conv3WgradHW, 8, Conv2WgradHW, 8>(a, true, true, s);
      default: break;
    }
  }
  if (a.B >= 128) {
    // throughput regime: the forward launches are operand-traffic bound (their time is flat in the number of K-split waves,
    // tools/sweep_nw.py), so they run on the register-blocked routine (gemm_tile_hb: each half8 fragment feeds 2 MFMAs) —
    // DATATYPE=float16 tools/sweep_rb.py at B = 256: conv1_fwd 24.8 -> 20.8 us, conv2_fwd 27.5 -> 20.4, conv3_fwd 19.8 -> 16.5,
    // fc4_fwd 17.3 -> 14.9.  The dgrads (M = B or few tiles per N) lose parallelism when blocked and stay unblocked.
    switch (id) {
      case K_CONV1_FWD: return launch_gemm<RB<Conv1FwdH, 2, 1>, 1>(a, s);
      case K_CONV2_FWD: return launch_gemm<RB<Conv2FwdH, 2, 2>, 4>(a, s);
      case K_CONV3_FWD: return launch_gemm<RB<Conv3FwdH, 2, 2>, 4>(a, s);
      case K_FC4_FWD: return launch_gemm<RB<Fc4FwdH, 2, 2>, 1>(a, s);
      default: break;
    }
  }
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


#ifdef SDQN_EXPERIMENTS
static hipError_t launch_kernel_hoist(int id, const StepArgs& a, const LaunchTune& t, hipStream_t s, bool* handled) {
  *handled = true;
  {
    // hoist (B <= 32): the target-net forward of the NEXT step rides in launches of this step that have room for it in the
    // same round of workgroups (bwd2: 592 of 1024 slots, bwd1: 200 of 512; conv1/conv2 online-only: 400 / 162 workgroups):
    //   K_BWD2(i)   + target conv1(i+1)      K_BWD1(i)   + target conv2(i+1)
    //   K_CONV1(i+1) + target conv3(i+1)     K_CONV2(i+1) + target fc4(i+1)      -> the head of step i+1 finds both slab sets
    // same tiles / waves per tile as the plain launches: bit-identical values.  StepArgs::nz = 1 in the two forward launches.
    switch (id) {
      case K_BWD2:
        if ((t.hoist & 1) && a.f4w_count == 0) return launch_multi<512, Conv1FwdTarget, 8, Conv2Dgrad, 8, Conv2Wgrad, 8>(a, true, true, s);
        break;
      case K_BWD1:
        if ((t.hoist & 1) && a.f4w_count == 0) return launch_multi<1024, TargetOnly<Conv2Fwd>, 16, Conv1Wgrad, 16, NoProblem, 2>(a, true, false, s);
        break;
      case K_CONV1_FWD:
        if (t.hoist & 2) return launch_multi<1024, Conv1Fwd, 8, TargetOnly<Conv3Fwd>, 16, NoProblem, 2>(a, true, false, s);   // (same tiling as the plain conv3_fwd launch)
        break;
      case K_CONV2_FWD:
        if (t.hoist & 2) return launch_multi<1024, Conv2Fwd, 16, Staged<Fc4FwdTarget>, 14, NoProblem, 2>(a, true, false, s);
        break;
      default: break;
    }
  }
  *handled = false;
  return hipSuccess;
}

#endif  // SDQN_EXPERIMENTS

hipError_t launch_kernel_ext(int id, const StepArgs& a, const LaunchTune& t, hipStream_t s, bool* handled) {
  *handled = true;
  if (a.h16) return launch_kernel_h16(id, a, t, s);
#ifdef SDQN_EXPERIMENTS
  if (!a.bn && a.B >= 128 && id >= 0 && id < 12 && t.rb[id] > 0) return launch_kernel_rb(id, t.rb[id], a, s);     // experiments (tools/sweep_rb.py)
  if (a.B <= 32 && t.hoist && !a.bn) return launch_kernel_hoist(id, a, t, s, handled);
  if (a.B >= 128 && t.order == 3 && !a.bn && a.f4w_count > 0 && id == K_BWD3)       // experiment: the new order in the throughput regime
    return launch_multi<512, Staged<Conv3Dgrad>, 8, Conv3Wgrad, 8, Fc4Wgrad, 8>(a, true, true, s);
  if (a.B <= 32 && t.order && !a.bn && a.f4w_count > 0 && id == K_BWD3) {
    // experiment (option "bwd_order"): which problem's workgroups are dispatched first inside bwd3.  Built-in: conv3_dgrad, conv3_wgrad,
    // fc4_wgrad (12 950 steps/s); 1 = round 1's order fc4_wgrad, conv3_dgrad, conv3_wgrad (12 810); 2 = conv3_dgrad, fc4_wgrad, conv3_wgrad (12 900)
    if (t.order == 1) return launch_multi<512, Fc4Wgrad, 1, Staged<Conv3Dgrad>, 8, Conv3Wgrad, 8>(a, true, true, s);
    if (t.order == 2) return launch_multi<512, Staged<Conv3Dgrad>, 8, Fc4Wgrad, 1, Conv3Wgrad, 8>(a, true, true, s);
  }
#endif
  *handled = false;
  return hipSuccess;
}

#ifdef SDQN_TIMING
// every translation unit has its own copy of the stamp-buffer pointer (gemm_engine.h)
hipError_t set_timing_buffer_rb(unsigned long long* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_sdqn_dbg), &p, sizeof p); }
#endif

}  // namespace sdqn
