// sdqn_kernels_r3.hip — round-3 launch variants of the default fp32 step (own translation unit: hipcc's schedule of a kernel
// depends on what else is instantiated beside it, see sdqn_kernels.hip).
//
//   K_FC4_DGRAD with LaunchTune::r3 bit 0 (B <= 32): ONE launch of 1024-thread workgroups =
//       98 x Staged<Fc4DgradSig> tiles (16 waves each, K = 512 split over the waves)          block ids 0..97   (dispatched first)
//     + 98 x 16 Fc4WgradWait tiles (one 32x32 tile of gW4 per wave, K = B, fused RMSProp)      block ids 98..195
//   Same tiles, same K split, same epilogues as the separate launches -> bit-identical results; what changes is WHEN the
//   25.7 MB read-modify-write of W4 and its RMSProp state runs: beside the latency-bound dgrad (98 of 256 CUs busy) instead of
//   inside bwd3.  The in-place update is ordered behind the dgrad's reads of the same W4 rows by one flag word per row block
//   (problems.h: Fc4DgradSig / Fc4WgradWait) — a write-after-read hand-off, no data crosses between the workgroups.
#include "gemm_engine.h"
#include "kernels.h"

namespace sdqn {

hipError_t launch_kernel_r3(int id, const StepArgs& a, const LaunchTune& t, hipStream_t s, bool* handled) {
  *handled = true;
  if (id == K_FC4_DGRAD && (t.r3 & 1) && a.B <= 32 && !a.h16 && a.f4w_count > 0 && a.f4d_flags)
    return launch_multi<1024, Staged<Fc4DgradSig>, 16, Fc4WgradWait, 1, NoProblem, 2>(a, true, false, s);
  *handled = false;
  return hipSuccess;
}

}  // namespace sdqn
