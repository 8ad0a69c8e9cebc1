// sdqn_kernels_r3.hip — round-3 launch variants of the default fp32 step (own translation unit: hipcc's schedule of a kernel
// depends on what else is instantiated beside it, see sdqn_kernels.hip).
//
//   K_CONV3_FWD with LaunchTune::r3 bit 1 (B < 128): gemm36_kernel below.
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

// ---- conv3 forward with 36-deep K-chunks -----------------------------------------------------------------------------------
// K = 576 = 18 chunks of 32: over the 16 waves of a tile that is two waves with TWO chunks and fourteen with one — and the
// direct-load routine has no prefetch across chunks, so the tile's life is two serial (operand round trip + 16 MFMAs) legs:
// s_memtime stamps (tools/phase_timing.py) put conv3_fwd's last operands 9.5 k cycles after issue, 2.4x conv2_fwd's single leg.
// 576 = 16 x 36: every wave owns ONE 36-deep chunk = 18 steps of v_mfma_f32_32x32x2_f32.  k-slot map shared by both operands:
//   step t < 16 : k = kc + 8 (t >> 2) + 4 h + (t & 3)      (the engine's map: four 16-byte loads of the lane's own row)
//   step 16, 17 : k = kc + 32 + 2 h + (t - 16)             (one 8-byte load)
// Same fixed-order LDS combine over the 16 waves and the same epilogue as gemm_tile.  Only the partition of the K sum changes, so
// values differ from the 32-deep routine in the last bits (both are fp32 fmaf chains); every caller of conv3_fwd uses THIS routine
// (train, predict, predict_one), the hoisted / batch-norm / tuning-hook variants keep the old one consistently on both nets.
template <class P>
__global__ void __launch_bounds__(1024) gemm36_kernel(const StepArgs a) {
  static_assert(P::A_K && !P::B_K && P::B_REG, "36-deep routine: k-contiguous A (own-row loads), row-major B");
  __shared__ float smem[16 * PANEL];
  const int gx = gridDim.x, gy = gridDim.y;
  const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const int tl = (a.xcd_map & 1) ? xcd_tile_id(lin, gx * gy * gridDim.z) : lin;
  const int bz = tl / (gx * gy), rr = tl - bz * (gx * gy);
  const int bx = rr % gx, by = rr / gx;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = bx * 32, n0 = by * 32;
  int z, ks, kbeg, kend;
  P::ksplit(a, bz, z, ks, kbeg, kend);                    // (whole K: 0 .. 576)
  const int M = P::M(a), N = P::N(a);
  const int hb = lane >> 5;
  const int mrow = m0 + (lane & 31), ncol = n0 + (lane & 31);
  const typename P::aoff_t arow = P::a_row(a, z, mrow < M ? mrow : M - 1);
  const int bcol = P::b_col(a, z, ncol < N ? ncol : N - 1);
  const float* abase = P::a_ptr(a, z);
  const float* bbase = P::b_ptr(a, z);
  const int kc = kbeg + wave * 36;
  float fa[18], fb[18];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f4 v = P::a_load4(a, z, arow + P::a_col(a, z, kc + 8 * j + 4 * hb));
    fa[4 * j] = v.x; fa[4 * j + 1] = v.y; fa[4 * j + 2] = v.z; fa[4 * j + 3] = v.w;
  }
  { const float2 v = *reinterpret_cast<const float2*>(abase + (arow + P::a_col(a, z, kc + 32 + 2 * hb)));
    fa[16] = v.x; fa[17] = v.y; }
  const uint32_t boff = 4u * ((uint32_t)bcol + (uint32_t)(kc + 4 * hb) * (uint32_t)P::B_LD);       // byte offset of (k = kc + 4 h, column)
#pragma unroll
  for (int t = 0; t < 16; ++t) fb[t] = ld_byte_off(bbase, boff + 4u * (uint32_t)((8 * (t >> 2) + (t & 3)) * P::B_LD));
  const uint32_t boff2 = 4u * ((uint32_t)bcol + (uint32_t)(kc + 32 + 2 * hb) * (uint32_t)P::B_LD);
  fb[16] = ld_byte_off(bbase, boff2); fb[17] = ld_byte_off(bbase, boff2 + 4u * (uint32_t)P::B_LD);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
  for (int t = 0; t < 18; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[t], acc, 0, 0, 0);
  float* cw = smem + wave * PANEL;
#pragma unroll
  for (int r = 0; r < 16; ++r) cw[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 33 + (lane & 31)] = acc[r];
  __syncthreads();
  {
    const int e = threadIdx.x, ml = e >> 5, nl = e & 31;
    float v = smem[ml * 33 + nl];
#pragma unroll
    for (int w = 1; w < 16; ++w) v += smem[w * PANEL + ml * 33 + nl];           // fixed order
    if (m0 + ml < M && n0 + nl < N) P::store(a, z, ks, m0 + ml, n0 + nl, v);
  }
}

hipError_t launch_kernel_r3(int id, const StepArgs& a, const LaunchTune& t, hipStream_t s, bool* handled) {
  *handled = true;
  if (id == K_FC4_DGRAD && (t.r3 & 1) && a.B <= 32 && !a.h16 && a.f4w_count > 0 && a.f4d_flags)
    return launch_multi<1024, Staged<Fc4DgradSig>, 16, Fc4WgradWait, 1, NoProblem, 2>(a, true, false, s);
  if (id == K_CONV3_FWD && (t.r3 & 2) && a.B < 128 && !a.h16 && !a.bn) {
    static_assert(CRS3 == 16 * 36, "conv3's K is 16 chunks of 36");
    const dim3 grid((Conv3Fwd::M(a) + 31) / 32, (Conv3Fwd::N(a) + 31) / 32, Conv3Fwd::nbz(a));
    hipLaunchKernelGGL((gemm36_kernel<Conv3Fwd>), grid, dim3(1024), 0, s, a);
    return hipGetLastError();
  }
  *handled = false;
  return hipSuccess;
}

}  // namespace sdqn
