// sdqn_kernels_bt.hip — the throughput regime (B >= 128, float32) on the block-tile engine (gemm_engine_bt.h): own translation unit,
// like every other family of launch variants (hipcc's schedule of a kernel depends on what is instantiated beside it).
//
//   forward   conv2 / conv3 / fc4          one launch each (conv1 stays on its packed-bf16 kernel, sdqn_kernels_r3.hip)
//   backward  fc4_dgrad                    one launch
//             bwd3 = fc4_wgrad (+ fused RMSProp of W4) || conv3_dgrad || conv3_wgrad          one multi-problem launch
//             bwd2 = conv2_dgrad (4 stride-parity classes) || conv2_wgrad                     one multi-problem launch
//   and every backward problem as a launch of its own (fused_launches = 0 / two_streams): the same block shapes, so fused and
//   unfused steps stay bit-identical.
// LaunchTune::bt[id]: 0 = the built-in block shape, n > 0 = menu entry n (tools/sweep_bt.py), < 0 = this launch on the latency engine.
#include "gemm_engine_bt.h"
#include "problems_wt.h"
#include "kernels.h"

namespace sdqn {

#define BT(P, BM, BN, WM, WN, D) BtCfg<P, BM, BN, WM, WN, D>
#define BTX(P, BM, BN, WM, WN, D, X) BtCfg<P, BM, BN, WM, WN, D, X>
#define BT_CASE(N, P, BM, BN, WM, WN, D) case N: return launch_bt<BT(P, BM, BN, WM, WN, D)>(a, s)

// built-in block shapes (menu entry 0 maps onto these)
typedef BT(Conv2FwdWT, 64, 64, 2, 2, 2) C2F;
typedef BT(Conv3FwdWT, 64, 64, 2, 2, 2) C3F;
typedef BT(Fc4FwdWT, 64, 64, 2, 2, 2) F4F;
typedef BT(Fc4DgradWT, 64, 64, 2, 2, 2) F4D;
typedef BT(Fc4WgradBT, 64, 64, 2, 2, 2) F4W;
typedef BT(Conv3DgradWT, 64, 64, 2, 2, 2) C3D;
typedef BT(Conv3WgradWT, 64, 64, 2, 2, 2) C3W;
typedef BT(Conv2DgradWT, 128, 32, 4, 1, 2) C2D;
typedef BT(Conv2WgradWT, 64, 64, 2, 2, 2) C2W;
typedef BT(NoProblem, 64, 64, 2, 2, 1) NOP;

static hipError_t launch_single(int id, int menu, const StepArgs& a, hipStream_t s) {
  switch (id) {
    case K_CONV2_FWD:                       // M = 2 B 81, N = 64, K = 512
      switch (menu) {
        case 0: return launch_bt<C2F>(a, s);
        BT_CASE(1, Conv2FwdWT, 64, 64, 2, 2, 3); BT_CASE(2, Conv2FwdWT, 128, 64, 2, 2, 2); BT_CASE(3, Conv2FwdWT, 128, 64, 4, 1, 2);
        BT_CASE(4, Conv2FwdWT, 64, 64, 2, 2, 1); BT_CASE(5, Conv2FwdWT, 128, 64, 2, 2, 3);
        default: break;
      }
      break;
    case K_CONV3_FWD:                       // M = 2 B 49, N = 64, K = 576
      switch (menu) {
        case 0: return launch_bt<C3F>(a, s);
        BT_CASE(1, Conv3FwdWT, 64, 64, 2, 2, 3); BT_CASE(2, Conv3FwdWT, 128, 64, 2, 2, 2); BT_CASE(3, Conv3FwdWT, 128, 64, 4, 1, 2);
        BT_CASE(4, Conv3FwdWT, 64, 64, 2, 2, 1); BT_CASE(5, Conv3FwdWT, 128, 64, 2, 2, 3);
        default: break;
      }
      break;
    case K_FC4_FWD:                         // M = B per net, N = 512, K = 3136 in S4 slabs
      switch (menu) {
        case 0: return launch_bt<F4F>(a, s);
        BT_CASE(1, Fc4FwdWT, 64, 64, 2, 2, 3); BT_CASE(2, Fc4FwdWT, 128, 64, 2, 2, 2); BT_CASE(3, Fc4FwdWT, 128, 128, 2, 2, 2);
        BT_CASE(4, Fc4FwdWT, 64, 128, 2, 2, 2); BT_CASE(5, Fc4FwdWT, 128, 128, 2, 2, 3);
        default: break;
      }
      break;
    case K_FC4_DGRAD:                       // M = B, N = 3136, K = 512
      switch (menu) {
        case 0: return launch_bt<F4D>(a, s);
        BT_CASE(1, Fc4DgradWT, 64, 64, 2, 2, 3); BT_CASE(2, Fc4DgradWT, 64, 128, 2, 2, 2); BT_CASE(3, Fc4DgradWT, 32, 128, 1, 4, 2);
        BT_CASE(4, Fc4DgradWT, 128, 64, 2, 2, 2); BT_CASE(5, Fc4DgradWT, 32, 128, 1, 4, 3);
        default: break;
      }
      break;
    case K_FC4_WGRAD: return launch_bt<F4W>(a, s);
    case K_CONV3_DGRAD: return launch_bt<C3D>(a, s);
    case K_CONV3_WGRAD: return launch_bt<C3W>(a, s);
    case K_CONV2_DGRAD: return launch_bt<C2D>(a, s);
    case K_CONV2_WGRAD: return launch_bt<C2W>(a, s);
    default: break;
  }
  return hipErrorInvalidValue;
}

// the built-in block shapes on packed-bf16 MFMA through exact three-way splits of both operands (gemm_engine_bt.h: X = 9 / 6 partial products)
template <int X>
static hipError_t launch_single_x(int id, const StepArgs& a, hipStream_t s) {
  switch (id) {
    case K_CONV2_FWD: return launch_bt<BTX(Conv2FwdWT, 64, 64, 2, 2, 2, X)>(a, s);
    case K_CONV3_FWD: return launch_bt<BTX(Conv3FwdWT, 64, 64, 2, 2, 2, X)>(a, s);
    case K_FC4_FWD: return launch_bt<BTX(Fc4FwdWT, 64, 64, 2, 2, 2, X)>(a, s);
    case K_FC4_DGRAD: return launch_bt<BTX(Fc4DgradWT, 64, 64, 2, 2, 2, X)>(a, s);
    case K_FC4_WGRAD: return launch_bt<BTX(Fc4WgradBT, 64, 64, 2, 2, 2, X)>(a, s);
    case K_CONV3_DGRAD: return launch_bt<BTX(Conv3DgradWT, 64, 64, 2, 2, 2, X)>(a, s);
    case K_CONV3_WGRAD: return launch_bt<BTX(Conv3WgradWT, 64, 64, 2, 2, 2, X)>(a, s);
    case K_CONV2_DGRAD: return launch_bt<BTX(Conv2DgradWT, 128, 32, 4, 1, 2, X)>(a, s);
    case K_CONV2_WGRAD: return launch_bt<BTX(Conv2WgradWT, 64, 64, 2, 2, 2, X)>(a, s);
    default: break;
  }
  return hipErrorInvalidValue;
}
template <int X>
static hipError_t launch_fused_x(int id, const StepArgs& a, hipStream_t s) {
  const bool f4 = a.f4w_count > 0;
  if (id == K_BWD3)
    return launch_bt_multi<BTX(Conv3DgradWT, 64, 64, 2, 2, 2, X), BTX(Conv3WgradWT, 64, 64, 2, 2, 2, X), BTX(Fc4WgradBT, 64, 64, 2, 2, 2, X)>(a, true, true, f4, s);
  if (id == K_BWD2 && !f4)
    return launch_bt_multi<NOP, BTX(Conv2WgradWT, 64, 64, 2, 2, 2, X), BTX(Conv2DgradWT, 128, 32, 4, 1, 2, X)>(a, false, true, true, s);
  return hipErrorInvalidValue;
}

static hipError_t launch_fused(int id, int menu, const StepArgs& a, hipStream_t s) {
  const bool f4 = a.f4w_count > 0;         // (B > 32: all of fc4_wgrad rides in bwd3 or none of it, sdqn_api.hip)
  if (id == K_BWD3) {
    // dispatch order = block-id order: the long conv3 problems first, the short fc4_wgrad tiles (8 chunks + the RMSProp stream) fill in
    switch (menu) {
      case 0: return launch_bt_multi<C3D, C3W, F4W>(a, true, true, f4, s);
      case 1: return launch_bt_multi<BT(Conv3DgradWT, 64, 64, 2, 2, 3), BT(Conv3WgradWT, 64, 64, 2, 2, 3), BT(Fc4WgradBT, 64, 64, 2, 2, 3)>(a, true, true, f4, s);
      case 2: return launch_bt_multi<BT(Conv3DgradWT, 128, 64, 2, 2, 2), BT(Conv3WgradWT, 64, 64, 2, 2, 2), F4W>(a, true, true, f4, s);
      case 3: return launch_bt_multi<F4W, C3D, C3W>(a, f4, true, true, s);
      case 4: return launch_bt_multi<BT(Conv3DgradWT, 128, 64, 2, 2, 2), BT(Conv3WgradWT, 128, 64, 2, 2, 2), BT(Fc4WgradBT, 64, 128, 2, 2, 2)>(a, true, true, f4, s);
      default: break;
    }
  } else if (id == K_BWD2 && !f4) {
    switch (menu) {
      case 0: return launch_bt_multi<NOP, C2W, C2D>(a, false, true, true, s);      // the long 9-chunk wgrad blocks are dispatched first, the 800 short dgrad blocks fill in (49.6 -> 43.8 us)
      case 1: return launch_bt_multi<NOP, BT(Conv2DgradWT, 128, 32, 4, 1, 3), BT(Conv2WgradWT, 64, 64, 2, 2, 3)>(a, false, true, true, s);
      case 2: return launch_bt_multi<NOP, BT(Conv2DgradWT, 256, 32, 4, 1, 2), C2W>(a, false, true, true, s);
      case 3: return launch_bt_multi<NOP, C2D, C2W>(a, false, true, true, s);
      case 4: return launch_bt_multi<NOP, BT(Conv2DgradWT, 128, 32, 4, 1, 2), BT(Conv2WgradWT, 128, 64, 2, 2, 2)>(a, false, true, true, s);
      default: break;
    }
  }
  return hipErrorInvalidValue;
}

// every K range of the launch must be whole chunks for the x-contiguous loaders' zero fill to be the only tail handling — it is
// (the loaders mask any k >= kend), so the routine takes every B >= 128; what it does not take: fp16 mode, batch-norm (raw outputs)
hipError_t launch_kernel_bt(int id, const StepArgs& a, const LaunchTune& t, hipStream_t s, bool* handled) {
  *handled = false;
  if (a.B < 128 || a.h16 || a.bn || t.hoist || t.order) return hipSuccess;
  if (id < 0 || id >= K_COUNT || t.bt[id] < 0) return hipSuccess;
  // fc4 forward / dgrad have 64 / 196 blocks of 64 x 64 — one or two per CU, nothing to overlap their waits with — and measured slower
  // here than on the latency engine (fc4_fwd 23.4 vs 18.7 us, fc4_dgrad 17.0 vs 15.2 at B = 256): block-tile only on request (menu entry > 0)
  if ((id == K_FC4_FWD || id == K_FC4_DGRAD) && t.bt[id] == 0 && t.btx[id] == 0) return hipSuccess;
  if (id < 12 && (t.nw_override[id] > 0 || t.rb[id] > 0)) return hipSuccess;      // explicit latency-engine tuning hooks win
  hipError_t e = hipErrorInvalidValue;
  const int x = t.btx[id];
  if (x == 9 && t.bt[id] == 0) e = (id == K_BWD3 || id == K_BWD2) ? launch_fused_x<9>(id, a, s) : launch_single_x<9>(id, a, s);
  else if (x == 6 && t.bt[id] == 0) e = (id == K_BWD3 || id == K_BWD2) ? launch_fused_x<6>(id, a, s) : launch_single_x<6>(id, a, s);
  else
  if (id == K_BWD3 || id == K_BWD2) e = launch_fused(id, t.bt[id], a, s);
  else if (id == K_CONV2_FWD || id == K_CONV3_FWD || id == K_FC4_FWD || id == K_FC4_DGRAD || id == K_FC4_WGRAD || id == K_CONV3_DGRAD ||
           id == K_CONV3_WGRAD || id == K_CONV2_DGRAD || id == K_CONV2_WGRAD) e = launch_single(id, t.bt[id], a, s);
  if (e == hipErrorInvalidValue) return hipSuccess;       // no such entry: the caller falls through to the latency engine
  *handled = true;
  return e;
}

}  // namespace sdqn
