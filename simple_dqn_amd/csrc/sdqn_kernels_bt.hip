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
#include <stdlib.h>
#include "gemm_engine_bt.h"
#include "problems_wt.h"
#include "problems_h16.h"
#include "kernels.h"

namespace sdqn {

#define BT(P, BM, BN, WM, WN, D) BtCfg<P, BM, BN, WM, WN, D>
#define BTU(P, BM, BN, WM, WN, D) BtCfg<P, BM, BN, WM, WN, D, 0, 1, 1>        // unconditional ring loads (run-time chunk counts)
#define BTX(P, BM, BN, WM, WN, D, X) BtCfg<P, BM, BN, WM, WN, D, X>
#define BT2(P, BM, BN, WM, WN, D) BtCfg<P, BM, BN, WM, WN, D, 0, 2>       // two chunks per barrier interval
#define BT_CASE(N, P, BM, BN, WM, WN, D) case N: return launch_bt<BT(P, BM, BN, WM, WN, D)>(a, s)

// built-in block shapes (menu entry 0 maps onto these)
typedef BT(Conv2FwdWT, 64, 64, 2, 2, 2) C2F;
typedef BT(Conv3FwdWT, 64, 64, 2, 2, 2) C3F;
typedef BT(Fc4FwdWT, 64, 64, 2, 2, 2) F4F;
typedef BT(Fc4DgradWT, 64, 64, 2, 2, 3) F4D;          // (three chunks in flight: W4 streams from memory; 12.4 us at B = 256, the latency engine 14.4)
typedef BTU(Fc4WgradBT, 64, 64, 2, 2, 2) F4W;          // the weight gradients' K slabs are run-time chunk counts: unconditional ring loads
typedef BT(Conv3DgradWT, 64, 64, 2, 2, 2) C3D;
typedef BTU(Conv3WgradWT, 64, 64, 2, 2, 2) C3W;
typedef BT(Conv2DgradWT, 128, 32, 4, 1, 2) C2D;
typedef BTU(Conv2WgradWT, 64, 64, 2, 2, 2) C2W;
typedef BT(NoProblem, 64, 64, 2, 2, 1) NOP;

static hipError_t launch_single(int id, int menu, const StepArgs& a, hipStream_t s) {
  switch (id) {
    case K_CONV2_FWD:                       // M = 2 B 81, N = 64, K = 512
      switch (menu) {
        case 0: return launch_bt<C2F>(a, s);
        BT_CASE(1, Conv2FwdWT, 64, 64, 2, 2, 3); BT_CASE(2, Conv2FwdWT, 128, 64, 2, 2, 2); BT_CASE(3, Conv2FwdWT, 128, 64, 4, 1, 2);
        BT_CASE(4, Conv2FwdWT, 64, 64, 2, 2, 1); BT_CASE(5, Conv2FwdWT, 128, 64, 2, 2, 3);
        case 6: return launch_bt<C2F>(a, s);      // (entry 0 is the sample-stationary routine since round 6: sdqn_kernels_ss.hip)
        default: break;
      }
      break;
    case K_CONV3_FWD:                       // M = 2 B 49, N = 64, K = 576
      switch (menu) {
        case 0: return launch_bt<C3F>(a, s);
        BT_CASE(1, Conv3FwdWT, 64, 64, 2, 2, 3); BT_CASE(2, Conv3FwdWT, 128, 64, 2, 2, 2); BT_CASE(3, Conv3FwdWT, 128, 64, 4, 1, 2);
        BT_CASE(4, Conv3FwdWT, 64, 64, 2, 2, 1); BT_CASE(5, Conv3FwdWT, 128, 64, 2, 2, 3);
        case 6: return launch_bt<C3F>(a, s);
        default: break;
      }
      break;
    case K_FC4_FWD:                         // M = B per net, N = 512, K = 3136 in S4 slabs
      switch (menu) {
        case 0: return launch_bt<F4F>(a, s);
        BT_CASE(1, Fc4FwdWT, 64, 64, 2, 2, 3); BT_CASE(2, Fc4FwdWT, 128, 64, 2, 2, 2); BT_CASE(3, Fc4FwdWT, 128, 128, 2, 2, 2);
        BT_CASE(4, Fc4FwdWT, 64, 128, 2, 2, 2); BT_CASE(5, Fc4FwdWT, 128, 128, 2, 2, 3);
        case 8: return launch_bt<BTU(Fc4FwdWT, 64, 64, 2, 2, 2)>(a, s);        // unconditional ring loads (run-time K split): 21.0 us at S4 = 7, the latency engine 18.1
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


static hipError_t launch_fused(int id, int menu, const StepArgs& a, hipStream_t s) {
  const bool f4 = a.f4w_count > 0;         // (B > 32: all of fc4_wgrad rides in bwd3 or none of it, sdqn_api_step.hip)
  if (id == K_BWD3) {
    // dispatch order = block-id order: the long conv3 problems first, the short fc4_wgrad tiles (8 chunks + the RMSProp stream) fill in
    switch (menu) {
      case 0: return launch_bt_multi<C3D, C3W, F4W>(a, true, true, f4, s);
      case 1: return launch_bt_multi<BT(Conv3DgradWT, 64, 64, 2, 2, 3), BT(Conv3WgradWT, 64, 64, 2, 2, 3), BT(Fc4WgradBT, 64, 64, 2, 2, 3)>(a, true, true, f4, s);
      case 2: return launch_bt_multi<BT(Conv3DgradWT, 128, 64, 2, 2, 2), BT(Conv3WgradWT, 64, 64, 2, 2, 2), F4W>(a, true, true, f4, s);
      case 3: return launch_bt_multi<F4W, C3D, C3W>(a, f4, true, true, s);
      case 4: return launch_bt_multi<BT(Conv3DgradWT, 128, 64, 2, 2, 2), BT(Conv3WgradWT, 128, 64, 2, 2, 2), BT(Fc4WgradBT, 64, 128, 2, 2, 2)>(a, true, true, f4, s);
      case 7: return launch_bt_multi<C3D, BT(Conv3WgradWT, 64, 64, 2, 2, 2), BT(Fc4WgradBT, 64, 64, 2, 2, 2)>(a, true, true, f4, s);     // guarded ring loads (until round 4's second session: 39.2 vs 38.1 us)
      default: break;
    }
  } else if (id == K_BWD2 && !f4) {
    switch (menu) {
      case 0: return launch_bt_multi<NOP, C2W, C2D>(a, false, true, true, s);      // the long 9-chunk wgrad blocks are dispatched first, the 800 short dgrad blocks fill in (49.6 -> 43.8 us)
      case 1: return launch_bt_multi<NOP, BT(Conv2DgradWT, 128, 32, 4, 1, 3), BT(Conv2WgradWT, 64, 64, 2, 2, 3)>(a, false, true, true, s);
      case 2: return launch_bt_multi<NOP, BT(Conv2DgradWT, 256, 32, 4, 1, 2), C2W>(a, false, true, true, s);
      case 3: return launch_bt_multi<NOP, C2D, C2W>(a, false, true, true, s);
      case 4: return launch_bt_multi<NOP, BT(Conv2DgradWT, 128, 32, 4, 1, 2), BT(Conv2WgradWT, 128, 64, 2, 2, 2)>(a, false, true, true, s);
      case 7: return launch_bt_multi<NOP, BT(Conv2WgradWT, 64, 64, 2, 2, 2), C2D>(a, false, true, true, s);      // guarded ring loads
      default: break;
    }
  }
  return hipErrorInvalidValue;
}

// ---- conv1 weight gradient: bytes x (three exact bf16 planes of delta1) on packed-bf16 MFMA, frame rows staged through LDS ------------
// gW1[(c,r,s)][map] = sum over k = (sample, y, x) of frame byte x delta1[k][map] (deepqnetwork.py:162 for the first Convolution layer; the
// 1/255 of :100 applied once to each split-K partial, like conv1_wgrad_bf16_kernel of the latency regime, sdqn_kernels_r3.hip).  A byte is
// exact in bf16 and delta = hi + mid + lo exactly (problems.h: split_bf16x3), so the sum runs as THREE v_mfma_f32_32x32x16_bf16 per 16 k with
// every product exact and fp32 accumulation — 0.19x the matrix time of the fp32 form.
// Every earlier form of this stage (the fp32 engine's A_GROUP4 loader, the latency regime's bf16 kernel, a first block-tile cut of this
// one) fetched a lane's 4 patch bytes with ONE unaligned 16-byte load of its own — and all of them land at ~25 us at B = 256, whatever
// the matrix work: tools/exp/c1w_dbg.py ablations (same bytes every chunk: no change; no MFMAs: no change; the loads forced to 16-byte
// alignment: -11 us) put ~150 cycles of address processing on every such fully divergent load instruction.  So here the frame bytes
// arrive the other way round:
//   * K runs in chunks of 80 output positions = 4 output rows of ONE sample (400 = 5 x 80: a chunk never straddles samples) = pixel rows
//     4 y0 .. 4 y0 + 19 of each of the state's 4 frames: 1 680 CONTIGUOUS, 16-byte aligned bytes per frame, staged in LDS with 420 plain
//     16-byte loads per workgroup; a lane then reads its patch bytes with ds_read_u8 (offsets of the 80 positions are compile-time);
//   * delta1's [80 k][32 maps] chunk (10 KB, contiguous) is fetched, split into three bf16 planes and written to LDS once per workgroup
//     in its own (k-major) layout — 8-byte stores, no transpose; a lane gathers its 8 k of a step with conflict-free 2-byte reads;
//   * one workgroup (4 waves) owns all 256 rows x 32 maps of one K slab: wave w = input frame c of the state, two 32-row sub-tiles
//     (kernel rows r = 0..3 / 4..7, column s = lane & 7); double-buffered LDS stages, the next chunk's loads in flight under the MFMAs.
struct C1wBtArgs { const uint8_t* src; const float* d1; float* slab1; const int64_t* idx; int B, from_ring, tps1, Kt; };
typedef __bf16 c1w_bf16x8 __attribute__((ext_vector_type(8)));
constexpr int C1W_CH = 80;                          // output positions per chunk: 4 output rows of one sample, 5 MFMA steps of 16
constexpr int C1W_REG = 20 * W0;                    // bytes of one frame's staged rows (pixel rows 4 y0 .. 4 y0 + 19): 1680
constexpr int C1W_DPITCH = 3 * K1 + 8;              // ushorts per k row of the delta planes: [plane][32 maps] + pad (208 bytes)
constexpr int C1W_STAGE = C0 * C1W_REG + C1W_CH * C1W_DPITCH * 2;      // bytes per LDS stage: 6720 + 16640
__host__ __device__ constexpr int c1w_pos_off(int p) { return (p / Q1) * (ST1 * W0) + (p % Q1) * ST1; }     // position p of the chunk -> byte offset of its patch origin in the staged rows

__global__ void __launch_bounds__(256) c1w_bt_kernel(const C1wBtArgs c) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * C1W_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = blockIdx.x, Kt = c.Kt, kb = ks * c.tps1 * 32;
  int ke = kb + c.tps1 * 32; if (ke > Kt) ke = Kt;
  const int nch = (ke - kb) / C1W_CH;               // (the host launches this kernel only with slabs of whole chunks)
  // loader items: frame rows = 4 x 105 float4 (items 0..419), delta = 640 float4 (k = item >> 3, maps 4 (item & 7) ..)
  // D chunks in flight (register set ch mod D): a workgroup walks its slab alone on its CU and every chunk touches frame rows and delta
  // rows nobody has read before — with one chunk in flight it waited a cold round trip per chunk (3.5 us per chunk for 0.8 us of work)
  // loader items: frame rows = 4 x 105 float4 (items 0..419: two per thread), delta = 640 float4 (k = item >> 3, maps 4 (item & 7) ..: three)
  struct Stg { uint4 b0, b1; float4 d0, d1, d2; };
  auto ld_b = [&](int64_t fb, int it) { if (it > 419) it = 419; const int fc = it / 105, q = it - fc * 105;
                                        return *reinterpret_cast<const uint4*>(c.src + fb + (int64_t)fc * FRAME + 16 * q); };
  auto ld_d = [&](int k0, int it) { if (it > 639) it = 639; return *reinterpret_cast<const float4*>(c.d1 + (size_t)k0 * K1 + 4 * it); };
  auto gload = [&](int ch, Stg& g) {
    const int k0 = kb + ch * C1W_CH, n = k0 / PIX1, y0 = (k0 - n * PIX1) / Q1;
    const int64_t fb = (c.from_ring ? (c.idx[n] - C0) * (int64_t)FRAME : (int64_t)n * STATE) + (int64_t)y0 * (ST1 * W0);     // problems.h: sbase, z = 0
    g.b0 = ld_b(fb, tid); g.b1 = ld_b(fb, tid + 256);
    g.d0 = ld_d(k0, tid); g.d1 = ld_d(k0, tid + 256); g.d2 = ld_d(k0, tid + 512);
  };
  auto st_b = [&](unsigned char* st, int it, const uint4& v) { if (it < 420) *reinterpret_cast<uint4*>(st + (it / 105) * C1W_REG + 16 * (it % 105)) = v; };
  auto st_d = [&](unsigned short* dp, int it, const float4& d) {
    if (it >= 640) return;
    const int k = it >> 3, n4 = (it & 7) * 4;
    uint16_t a0, a1, a2, b0, b1, b2, c0, c1, c2, e0, e1, e2;
    split_bf16x3(d.x, a0, a1, a2); split_bf16x3(d.y, b0, b1, b2); split_bf16x3(d.z, c0, c1, c2); split_bf16x3(d.w, e0, e1, e2);
    auto pk2 = [](uint16_t lo, uint16_t hi) { return (uint32_t)lo | ((uint32_t)hi << 16); };
    unsigned short* row = dp + k * C1W_DPITCH + n4;
    *reinterpret_cast<uint2*>(row) = make_uint2(pk2(a0, b0), pk2(c0, e0));
    *reinterpret_cast<uint2*>(row + K1) = make_uint2(pk2(a1, b1), pk2(c1, e1));
    *reinterpret_cast<uint2*>(row + 2 * K1) = make_uint2(pk2(a2, b2), pk2(c2, e2));
  };
  auto lds_store = [&](const Stg& g, unsigned char* st) {
    st_b(st, tid, g.b0); st_b(st, tid + 256, g.b1);
    unsigned short* dp = reinterpret_cast<unsigned short*>(st + C0 * C1W_REG);
    st_d(dp, tid, g.d0); st_d(dp, tid + 256, g.d1); st_d(dp, tid + 512, g.d2);
  };
  f32x16 acc[2];
#pragma unroll
  for (int sm = 0; sm < 2; ++sm)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[sm][q] = 0.0f;
  // this lane's patch element: frame c = wave, kernel row r = 4 sm + (i >> 3), column s = i & 7 (problems.h: col1: m = 64 c + 8 r + s)
  const int lane_off = wave * C1W_REG + (i >> 3) * W0 + (i & 7);
  if (nch > 0) {
    Stg g;
    gload(0, g);
    lds_store(g, smem);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
      {
        {
          const unsigned char* st = smem + (ch & 1) * C1W_STAGE;
          if (ch + 1 < nch) gload(ch + 1, g);
          const unsigned char* pa = st + lane_off;
          const unsigned short* pd = reinterpret_cast<const unsigned short*>(st + C0 * C1W_REG) + i;
#pragma unroll
          for (int s5 = 0; s5 < 5; ++s5) {
            // B fragments: k = 16 s5 + 8 h + e of map i, one per plane (2-byte reads, lanes along the maps: conflict-free)
            union { uint32_t u[4]; c1w_bf16x8 v; } B0, B1, B2;
            const unsigned short* pk = pd + (16 * s5 + 8 * h) * C1W_DPITCH;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              B0.u[e >> 1] = (uint32_t)pk[e * C1W_DPITCH] | ((uint32_t)pk[(e + 1) * C1W_DPITCH] << 16);
              B1.u[e >> 1] = (uint32_t)pk[e * C1W_DPITCH + K1] | ((uint32_t)pk[(e + 1) * C1W_DPITCH + K1] << 16);
              B2.u[e >> 1] = (uint32_t)pk[e * C1W_DPITCH + 2 * K1] | ((uint32_t)pk[(e + 1) * C1W_DPITCH + 2 * K1] << 16);
            }
            // A fragments: the 8 positions' patch bytes (exact in bf16: upper half of the float), both sub-tiles (4 kernel rows apart)
            int off[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) off[e] = h ? c1w_pos_off(16 * s5 + 8 + e) : c1w_pos_off(16 * s5 + e);
#pragma unroll
            for (int sm = 0; sm < 2; ++sm) {
              union { uint32_t u[4]; c1w_bf16x8 v; } A;
#pragma unroll
              for (int e = 0; e < 8; e += 2) {
                const uint32_t f0 = __float_as_uint((float)pa[off[e] + sm * (4 * W0)]), f1 = __float_as_uint((float)pa[off[e + 1] + sm * (4 * W0)]);
                A.u[e >> 1] = __builtin_amdgcn_perm(f1, f0, 0x07060302u);
              }
              acc[sm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, B2.v, acc[sm], 0, 0, 0);     // small planes first
              acc[sm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, B1.v, acc[sm], 0, 0, 0);
              acc[sm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, B0.v, acc[sm], 0, 0, 0);
            }
          }
          if (ch + 1 < nch) { lds_store(g, smem + ((ch + 1) & 1) * C1W_STAGE); __syncthreads(); }
        }
      }
    }
  }
  // epilogue: / 255 (deepqnetwork.py:100, once per partial sum), write-through stores of the slab; lanes along the maps: 128-byte rows
  const float r255 = 1.0f / 255.0f;
#pragma unroll
  for (int sm = 0; sm < 2; ++sm)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = 64 * wave + 32 * sm + bt::acc_row(q, h);
      const float sv = acc[sm][q], qv = sv * r255;
      wt_store(c.slab1 + (int64_t)ks * NW1 + m * K1 + i, fmaf(fmaf(-qv, 255.0f, sv), r255, qv));       // s / 255 to within an ulp (as div255, sdqn_kernels_r3.hip)
    }
}

static hipError_t launch_c1w_bt(const StepArgs& a, hipStream_t s) {
  // slabs of whole 80-position chunks only (tps1 a multiple of 5: 160 positions); other slab sizes stay on the latency engine's kernel
  if ((a.tps1 * 32) % C1W_CH != 0) return hipErrorInvalidValue;
  C1wBtArgs c; c.src = a.src; c.d1 = a.d1; c.slab1 = a.slab1; c.idx = a.idx; c.B = a.B; c.from_ring = a.from_ring; c.tps1 = a.tps1; c.Kt = a.B * PIX1;
  SDQN_LAUNCH(c1w_bt_kernel, dim3(Conv1Wgrad::nbz(a)), dim3(256), 0, s, c);
  return hipGetLastError();
}

// ---- float16 mode: conv1's weight gradient, one workgroup per K slab of whole 80-position chunks ------------------------------------------
// Same plan as c1w_bt_kernel (the frame rows a chunk's patches come from are staged ONCE, coalesced; a patch element is never fetched
// from memory per position) with half operands: the staged rows hold half(x / 255) — the forward pass's conv1 input values — converted
// when they are stored, delta1 is the half copy as it lies in memory ([k][32 maps]), and both fragments come out of LDS through the
// transpose read (ds_read_b64_tr_b16: 4 consecutive k of one m / n per read; a lane addresses 4 consecutive kernel columns of ONE
// position's patch row, which are contiguous in the staged frame row).  One v_mfma_f32_32x32x16_f16 per step and sub-tile.
struct C1wHArgs { const uint8_t* src; const half_t* d1; float* slab1; const int64_t* idx; int B, from_ring, tps1, Kt; float inv_loss_scale; };
constexpr int C1H_PITCH = W0 + 4;                   // halves per staged frame row (176 bytes)
constexpr int C1H_FR = 20 * C1H_PITCH;              // halves per frame: pixel rows 4 y0 .. 4 y0 + 19
constexpr int C1H_DPITCH = K1 + 16;                 // halves per k row of delta1 (96 bytes)
constexpr int C1H_STAGE = C0 * C1H_FR + C1W_CH * C1H_DPITCH;      // halves per LDS stage: 7040 + 3840
typedef unsigned int c1h_u32x4 __attribute__((ext_vector_type(4)));

// EXACT (round 6, the default): the frame bytes enter as exact halves (half(1024 + b) = 0x6400 | b, - 1024: two perms + two packed adds per
// dword instead of four cvt / IEEE-divide / cvt chains — the conversion was ~3x the chunk's matrix time) from 16-byte loads, and the 1 / 255 of
// deepqnetwork.py:100 multiplies the split-K partial once, with the loss scale.  !EXACT: the first form (half(b / 255) operands, 4-byte loads).
template <bool EXACT>
__device__ __forceinline__ void c1w_h_body(const C1wHArgs& c, const int ks, half_t* const smem) {      // smem: 2 * C1H_STAGE halves
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Kt = c.Kt, kb = ks * c.tps1 * 32;
  int ke = kb + c.tps1 * 32; if (ke > Kt) ke = Kt;
  const int nch = (ke - kb) / C1W_CH;               // (the host launches this kernel only with slabs of whole chunks)
  // loader items: frame rows = 4 frames x 20 rows x 84 bytes, contiguous and 16-byte aligned per frame: 420 pieces of 16 bytes (EXACT; 2 per
  // thread, the second partly) or 1680 dwords (7 per thread); delta = 320 x 16 bytes (2, partly)
  struct Stg { uint32_t b[EXACT ? 1 : 7]; c1h_u32x4 f[2]; c1h_u32x4 d0, d1; };
  int fsrc[2], fdst[2][4];                          // EXACT: byte offset of piece j in the chunk's frame window / LDS half offsets of its 4 dwords
  if constexpr (EXACT) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int it = tid + 256 * j; if (it > 419) it = 419;
      const int fc = it / 105, q4 = it - fc * 105;
      fsrc[j] = fc * FRAME + 16 * q4;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const int q = 4 * q4 + e, row = q / 21, col = 4 * (q - row * 21); fdst[j][e] = fc * C1H_FR + row * C1H_PITCH + col; }
    }
  }
  auto gload = [&](int ch, Stg& g) {
    const int k0 = kb + ch * C1W_CH, n = k0 / PIX1, y0 = (k0 - n * PIX1) / Q1;
    const int64_t fb = (c.from_ring ? (c.idx[n] - C0) * (int64_t)FRAME : (int64_t)n * STATE) + (int64_t)y0 * (ST1 * W0);     // problems.h: sbase, z = 0
    if constexpr (EXACT) {
#pragma unroll
      for (int j = 0; j < 2; ++j) g.f[j] = *reinterpret_cast<const c1h_u32x4*>(c.src + fb + fsrc[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        int it = tid + 256 * j; if (it > 1679) it = 1679;
        const int fc = it / 420, q = it - fc * 420;                             // 420 dwords = 20 rows x 84 bytes, contiguous in the frame
        g.b[j] = *reinterpret_cast<const uint32_t*>(c.src + fb + (int64_t)fc * FRAME + 4 * q);
      }
    }
    const c1h_u32x4* dp = reinterpret_cast<const c1h_u32x4*>(c.d1 + (size_t)k0 * K1);
    g.d0 = dp[tid]; g.d1 = dp[tid + 256 < 320 ? tid + 256 : 319];
  };
  auto lds_store = [&](const Stg& g, half_t* st) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    if constexpr (EXACT) {
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      const h2 m1024 = {(half_t)-1024.0f, (half_t)-1024.0f};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (tid + 256 * j >= 420) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t w = g.f[j][e];
          union { uint32_t u; h2 v; } lo, hi;
          lo.u = __builtin_amdgcn_perm(0x64646464u, w, 0x04010400u); hi.u = __builtin_amdgcn_perm(0x64646464u, w, 0x04030402u);
          lo.v = lo.v + m1024; hi.v = hi.v + m1024;
          union { uint32_t u[2]; h4 v; } o; o.u[0] = lo.u; o.u[1] = hi.u;
          *reinterpret_cast<h4*>(st + fdst[j][e]) = o.v;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int it = tid + 256 * j;
        if (it < 1680) {
          const int fc = it / 420, q = it - fc * 420, row = q / 21, col = 4 * (q - row * 21);
          h4 v; const uint32_t w = g.b[j];
          v[0] = (half_t)norm_u8(w & 255u); v[1] = (half_t)norm_u8((w >> 8) & 255u); v[2] = (half_t)norm_u8((w >> 16) & 255u); v[3] = (half_t)norm_u8(w >> 24);
          *reinterpret_cast<h4*>(st + fc * C1H_FR + row * C1H_PITCH + col) = v;
        }
      }
    }
    half_t* dl = st + C0 * C1H_FR;
    *reinterpret_cast<c1h_u32x4*>(dl + (tid >> 2) * C1H_DPITCH + 8 * (tid & 3)) = g.d0;
    if (tid + 256 < 320) *reinterpret_cast<c1h_u32x4*>(dl + ((tid + 256) >> 2) * C1H_DPITCH + 8 * (tid & 3)) = g.d1;
  };
  f32x16 acc[2];
#pragma unroll
  for (int sm = 0; sm < 2; ++sm)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[sm][q] = 0.0f;
  // transpose-read geometry (gemm_engine_bt.h: bt_tile_hw): 16-lane group G covers m / n columns 16 (G & 1) .. + 15 and k 8 (G >> 1) .. + 7 of
  // a step; lane t of the group addresses k-row (t >> 2) (+ 4: second read), columns 4 (t & 3) .. + 3.  A: m = 64 c + 8 r + s (problems.h:
  // col1) — the wave is frame c, sub-tile sm holds kernel rows 4 sm .. + 3, a 16-column group two kernel rows, a lane's 4 columns s = 0..3 / 4..7
  const int t = lane & 15, G = lane >> 4;
  const int a_m = wave * C1H_FR + (2 * (G & 1) + ((t & 3) >> 1)) * C1H_PITCH + 4 * (t & 1);
  int a_pos[10];                                    // staged-row offset of position 16 s5 + 8 (G >> 1) + 4 r + (t >> 2) (chunk-invariant)
#pragma unroll
  for (int q = 0; q < 10; ++q) { const int pos = 16 * (q >> 1) + 8 * (G >> 1) + 4 * (q & 1) + (t >> 2); a_pos[q] = (pos / Q1) * (ST1 * C1H_PITCH) + (pos % Q1) * ST1; }
  const int b_lane = (8 * (G >> 1) + (t >> 2)) * C1H_DPITCH + 16 * (G & 1) + 4 * (t & 3);
  if (nch > 0) {
    Stg g;
    gload(0, g);
    lds_store(g, smem);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
      const half_t* st = smem + (ch & 1) * C1H_STAGE;
      if (ch + 1 < nch) gload(ch + 1, g);
      const half_t* pa = st + a_m;
      const half_t* pd = st + C0 * C1H_FR + b_lane;
#pragma unroll
      for (int s5 = 0; s5 < 5; ++s5) {
        const half8 fb = bt_tr_frag(pd + 16 * s5 * C1H_DPITCH, pd + (16 * s5 + 4) * C1H_DPITCH);
#pragma unroll
        for (int sm = 0; sm < 2; ++sm) {
          const half8 fa = bt_tr_frag(pa + a_pos[2 * s5] + sm * (4 * C1H_PITCH), pa + a_pos[2 * s5 + 1] + sm * (4 * C1H_PITCH));
          acc[sm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[sm], 0, 0, 0);
        }
      }
      if (ch + 1 < nch) { lds_store(g, smem + ((ch + 1) & 1) * C1H_STAGE); __syncthreads(); }
    }
  }
  // epilogue: the loss scale divided out (problems_h16.h: Conv1WgradH::store); lanes along the maps: 128-byte rows of the slab
#pragma unroll
  for (int sm = 0; sm < 2; ++sm)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = 64 * wave + 32 * sm + bt::acc_row(q, h);
      c.slab1[(int64_t)ks * NW1 + m * K1 + i] = acc[sm][q] * (EXACT ? c.inv_loss_scale * (1.0f / 255.0f) : c.inv_loss_scale);
    }
}
template <bool EXACT>
__global__ void __launch_bounds__(256) c1w_h_kernel(const C1wHArgs c) {
  __shared__ __attribute__((aligned(16))) half_t smem[2 * C1H_STAGE];
  c1w_h_body<EXACT>(c, (int)blockIdx.x, smem);
}
// conv1's weight gradient INSIDE the float16 weight-gradient launch (round 6): delta1 is complete before that launch starts (both dgrads run
// in front of it), so its K-slab workgroups are a fourth block-id range of the same launch — one launch and one boundary fewer, and the
// 7.8 us of bwd1 packs under fc4_wgrad's RMSProp stream instead of following it.  Same body, same slabs: bit-identical.
template <class C0, class C1, class C2>
__global__ void __launch_bounds__(bt::NT) bt_multi_c1w_kernel(const StepArgs a, const MultiDims d, const C1wHArgs c1, const int nc1, const int c1_first) {
  constexpr int L01 = C0::LDS > C1::LDS ? C0::LDS : C1::LDS, L012 = L01 > C2::LDS ? L01 : C2::LDS, LC = C1H_STAGE, L = L012 > LC ? L012 : LC;     // floats
  __shared__ __attribute__((aligned(16))) float smem[L];
  static_assert(bt::NT == 256, "c1w_h_body is written for 256 threads");
  if constexpr (has_preload_multi<typename C1::P>::value) C1::P::preload_multi(a, d);
  const int nb = d.n[0] + d.n[1] + d.n[2];
  int b = blockIdx.x;
  if (c1_first) { if (b < nc1) { c1w_h_body<true>(c1, b, reinterpret_cast<half_t*>(smem)); return; } b -= nc1; }
  else if (b >= nb) { c1w_h_body<true>(c1, b - nb, reinterpret_cast<half_t*>(smem)); return; }
  const int xm = a.xcd_map;
  if (b < d.n[0]) { const int l = (xm & 1) ? xcd_tile_id_range(b, 0, d.n[0]) : b, pz = d.gx[0] * d.gy[0], bz = l / pz, r = l - bz * pz; bt_run_tile<C0>(a, r % d.gx[0], r / d.gx[0], bz, smem); }
  else if (b < d.n[0] + d.n[1]) { const int l = (xm & 2) ? xcd_tile_id_range(b, d.n[0], d.n[1]) : b - d.n[0], pz = d.gx[1] * d.gy[1], bz = l / pz, r = l - bz * pz; bt_run_tile<C1>(a, r % d.gx[1], r / d.gx[1], bz, smem); }
  else { const int l = (xm & 4) ? xcd_tile_id_range(b, d.n[0] + d.n[1], d.n[2]) : b - d.n[0] - d.n[1], pz = d.gx[2] * d.gy[2], bz = l / pz, r = l - bz * pz; bt_run_tile<C2>(a, r % d.gx[2], r / d.gx[2], bz, smem); }
}
template <class C0, class C1, class C2>
static hipError_t launch_bt_multi_c1w(const StepArgs& a, int c1_first, hipStream_t stream) {
  MultiDims d; memset(&d, 0, sizeof d);
  int gz;
  bt_grid<C0>(a, d.gx[0], d.gy[0], gz); d.n[0] = d.gx[0] * d.gy[0] * gz;
  bt_grid<C1>(a, d.gx[1], d.gy[1], gz); d.n[1] = d.gx[1] * d.gy[1] * gz;
  bt_grid<C2>(a, d.gx[2], d.gy[2], gz); d.n[2] = d.gx[2] * d.gy[2] * gz;
  C1wHArgs c; c.src = a.src; c.d1 = a.h_d1; c.slab1 = a.slab1; c.idx = a.idx; c.B = a.B; c.from_ring = a.from_ring; c.tps1 = a.tps1; c.Kt = a.B * PIX1;
  c.inv_loss_scale = a.inv_loss_scale;
  const int nc1 = Conv1Wgrad::nbz(a);
  SDQN_LAUNCH((bt_multi_c1w_kernel<C0, C1, C2>), dim3(d.n[0] + d.n[1] + d.n[2] + nc1), dim3(bt::NT), 0, stream, a, d, c, nc1, c1_first);
  return hipGetLastError();
}

static hipError_t launch_c1w_h(const StepArgs& a, const LaunchTune& t, hipStream_t s) {
  if ((a.tps1 * 32) % C1W_CH != 0) return hipErrorInvalidValue;       // slabs of whole 80-position chunks only
  C1wHArgs c; c.src = a.src; c.d1 = a.h_d1; c.slab1 = a.slab1; c.idx = a.idx; c.B = a.B; c.from_ring = a.from_ring; c.tps1 = a.tps1; c.Kt = a.B * PIX1;
  c.inv_loss_scale = a.inv_loss_scale;
  if (t.bt[K_BWD1] == 1) SDQN_LAUNCH(c1w_h_kernel<false>, dim3(Conv1Wgrad::nbz(a)), dim3(256), 0, s, c);        // first form (half(b / 255) operands)
  else SDQN_LAUNCH(c1w_h_kernel<true>, dim3(Conv1Wgrad::nbz(a)), dim3(256), 0, s, c);
  return hipGetLastError();
}

// ---- float16 mode: conv1 forward (fused gather + / 255 + conv + Rectlin), one workgroup per (net, sample) --------------------------------
// The generic half routines fetch every patch row (8 bytes) of every output position straight from memory: 16 divergent loads per lane
// and 32 x 32 tile, 17.9 us at B = 256 for 3.4 GFLOP and 27 MB.  Here the sample's four frames (28 KB, contiguous in the ring) are read
// ONCE with coalesced 16-byte loads, converted to half(x / 255) (problems_h16.h: ldh8_u8) into an LDS image [frame][84 rows][88], W1's
// transposed half copy ([32 maps][256 k]) beside it, and all 400 x 32 outputs come from v_mfma_f32_16x16x32_f16 steps whose A fragment
// is two 8-byte LDS reads (the 8 consecutive kernel columns of a patch row are contiguous in the image) and whose B fragment is one
// 16-byte read.  25 row tiles of 16 positions over the 4 waves, both 16-map column tiles per A fragment.
struct Conv1HArgs { const uint8_t* src; const int64_t* idx; const half_t* wht[2]; half_t* h_a1; int B, from_ring; };
constexpr int C1F_PITCH = W0 + 4;                    // halves per image row (176 bytes)
constexpr int C1F_FR = H0 * C1F_PITCH;               // halves per frame
constexpr int C1F_WPITCH = CRS1 + 8;                 // halves per map row of W1 (528 bytes)
constexpr int C1F_LDS = C0 * C1F_FR + K1 * C1F_WPITCH;      // 29 568 + 8 448 halves = 76 032 bytes: two workgroups per CU
typedef _Float16 c1f_h4 __attribute__((ext_vector_type(4)));
typedef float c1f_f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) conv1_h_kernel(const Conv1HArgs c) {
  __shared__ __attribute__((aligned(16))) half_t smem[C1F_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int z = blockIdx.x / c.B, n = blockIdx.x - z * c.B;
  const int64_t fb = c.from_ring ? (c.idx[n] - C0 + z) * (int64_t)FRAME : ((int64_t)z * c.B + n) * (int64_t)STATE;      // problems.h: sbase
  // ---- stage: 1764 x 16 bytes of frames (7 per thread, the last partly), 1024 x 16 bytes of weights (4 per thread) -------------------------
  const c1h_u32x4* fp = reinterpret_cast<const c1h_u32x4*>(c.src + fb);
  const c1h_u32x4* wp = reinterpret_cast<const c1h_u32x4*>(c.wht[z] + OFF1);
  c1h_u32x4 fv[7], wv[4];
#pragma unroll
  for (int j = 0; j < 7; ++j) { const int it = tid + 256 * j; fv[j] = fp[it < STATE / 16 ? it : STATE / 16 - 1]; }
#pragma unroll
  for (int j = 0; j < 4; ++j) wv[j] = wp[tid + 256 * j];
  half_t* wl = smem + C0 * C1F_FR;
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int it = tid + 256 * j, nn = it >> 5, k8 = it & 31; *reinterpret_cast<c1h_u32x4*>(wl + nn * C1F_WPITCH + 8 * k8) = wv[j]; }
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int it = tid + 256 * j;
    if (it < STATE / 16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int q = 4 * it + e;                    // dword of the state: frame q / 1764, row (q % 1764) / 21, columns 4 (q % 21) .. + 3
        const int fc = q / (FRAME / 4), r = q - fc * (FRAME / 4), row = r / (W0 / 4), col = 4 * (r - row * (W0 / 4));
        const uint32_t w = fv[j][e];
        c1f_h4 v;
        v[0] = (half_t)norm_u8(w & 255u); v[1] = (half_t)norm_u8((w >> 8) & 255u); v[2] = (half_t)norm_u8((w >> 16) & 255u); v[3] = (half_t)norm_u8(w >> 24);
        *reinterpret_cast<c1f_h4*>(smem + fc * C1F_FR + row * C1F_PITCH + col) = v;
      }
    }
  }
  __syncthreads();
  // ---- compute: lane = (row r = lane & 15 of the 16-position tile, k-group kg = lane >> 4: kernel row 4 (s & 1) + kg of frame s >> 1) -----------
  const int r = lane & 15, kg = lane >> 4;
  const half_t* wb0 = wl + r * C1F_WPITCH + 8 * kg;                     // B: map r (and 16 + r), k = 32 s + 8 kg ..
  for (int rt = wave; rt < PIX1 / 16; rt += 4) {
    const int pos = 16 * rt + r, p = pos / Q1, q = pos - p * Q1;
    const half_t* pa = smem + (ST1 * p + kg) * C1F_PITCH + ST1 * q;
    c1f_f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const half_t* ap = pa + (st >> 1) * C1F_FR + 4 * (st & 1) * C1F_PITCH;
      const c1f_h4 a_lo = *reinterpret_cast<const c1f_h4*>(ap), a_hi = *reinterpret_cast<const c1f_h4*>(ap + 4);
      half8 fa; fa[0] = a_lo[0]; fa[1] = a_lo[1]; fa[2] = a_lo[2]; fa[3] = a_lo[3]; fa[4] = a_hi[0]; fa[5] = a_hi[1]; fa[6] = a_hi[2]; fa[7] = a_hi[3];
      const half8 fb0 = *reinterpret_cast<const half8*>(wb0 + 32 * st), fb1 = *reinterpret_cast<const half8*>(wb0 + 16 * C1F_WPITCH + 32 * st);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb1, acc1, 0, 0, 0);
    }
    // C/D map of the 16 x 16 shapes: element e of lane l = (row 4 (l >> 4) + e, column l & 15); Rectlin, half store (Conv1FwdH::store)
    half_t* out = c.h_a1 + (((int64_t)z * c.B + n) * PIX1 + 16 * rt + 4 * kg) * K1 + r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { out[e * K1] = (half_t)fmaxf(acc0[e], 0.0f); out[e * K1 + 16] = (half_t)fmaxf(acc1[e], 0.0f); }
  }
}

// Second form (round 6, the default): the frame bytes stay EXACT.  half(1024 + b) is the bit pattern 0x6400 | b, so a dword of four bytes
// becomes four halves with two v_perm_b32 and two v_pk_add_f16 (- 1024: exact) instead of four (cvt, IEEE divide by 255, cvt) chains —
// the conversion was ~1 000 vector instructions per thread in front of the first MFMA; the 1 / 255 of deepqnetwork.py:100 multiplies the
// fp32 sum once per output (the sum of exact byte x half products is CLOSER to the fp32 reference than the first form's half(b / 255)
// operands; the oracle's half mode models the latter, the difference is below the half rounding of the output: test).  The LDS image is
// the state's own byte order widened (row pitch 84 halves: a 16-byte piece of the ring is 32 contiguous LDS bytes, no index arithmetic),
// the weights are the ROW operand (D[map][position]: a lane holds 4 consecutive maps of one position) and a tile's 16 x 32 outputs are
// collected in a wave-private 1 KB LDS tile and leave as whole 128-byte lines, one 16-byte store per lane (the first form: eight 2-byte
// stores per lane into 32-byte pieces of lines).
constexpr int C1G_FR = H0 * W0;                       // halves per frame of the linear image
constexpr int C1G_OPITCH = 40;                        // halves per position of a wave's output tile (80 bytes: 16-byte aligned, conflict-free)
constexpr int C1G_LDS = C0 * C1G_FR + K1 * C1F_WPITCH + 4 * 16 * C1G_OPITCH;      // 28 224 + 8 448 + 2 560 halves = 78 464 bytes: two workgroups per CU
typedef _Float16 c1g_h2 __attribute__((ext_vector_type(2)));

template <bool WT>
__global__ void __launch_bounds__(256) conv1_hb_kernel(const Conv1HArgs c) {
  __shared__ __attribute__((aligned(16))) half_t smem[C1G_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int z = blockIdx.x / c.B, n = blockIdx.x - z * c.B;
  const int64_t fb = c.from_ring ? (c.idx[n] - C0 + z) * (int64_t)FRAME : ((int64_t)z * c.B + n) * (int64_t)STATE;      // problems.h: sbase
  const c1h_u32x4* fp = reinterpret_cast<const c1h_u32x4*>(c.src + fb);
  const c1h_u32x4* wp = reinterpret_cast<const c1h_u32x4*>(c.wht[z] + OFF1);
  c1h_u32x4 fv[7], wv[4];
#pragma unroll
  for (int j = 0; j < 7; ++j) { const int it = tid + 256 * j; fv[j] = fp[it < STATE / 16 ? it : STATE / 16 - 1]; }
#pragma unroll
  for (int j = 0; j < 4; ++j) wv[j] = wp[tid + 256 * j];
  half_t* wl = smem + C0 * C1G_FR;
  half_t* ot = wl + K1 * C1F_WPITCH + wave * (16 * C1G_OPITCH);
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int it = tid + 256 * j, nn = it >> 5, k8 = it & 31; *reinterpret_cast<c1h_u32x4*>(wl + nn * C1F_WPITCH + 8 * k8) = wv[j]; }
  const c1g_h2 m1024 = {(half_t)-1024.0f, (half_t)-1024.0f};
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int it = tid + 256 * j;
    if (it < STATE / 16) {
      c1h_u32x4 o[2];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t w = fv[j][e];
        union { uint32_t u; c1g_h2 h; } lo, hi;
        lo.u = __builtin_amdgcn_perm(0x64646464u, w, 0x04010400u);          // halves 0x6400 | b0, 0x6400 | b1 = 1024 + b
        hi.u = __builtin_amdgcn_perm(0x64646464u, w, 0x04030402u);
        lo.h = lo.h + m1024; hi.h = hi.h + m1024;                             // exact: b < 2048
        o[e >> 1][2 * (e & 1)] = lo.u; o[e >> 1][2 * (e & 1) + 1] = hi.u;
      }
      c1h_u32x4* d = reinterpret_cast<c1h_u32x4*>(smem + 16 * it);
      d[0] = o[0]; d[1] = o[1];
    }
  }
  __syncthreads();
  const int r = lane & 15, kg = lane >> 4;
  const half_t* wb0 = wl + r * C1F_WPITCH + 8 * kg;                     // weights: map r (and 16 + r), k = 32 st + 8 kg ..
  half8 fw[2][8];
#pragma unroll
  for (int st = 0; st < 8; ++st) { fw[0][st] = *reinterpret_cast<const half8*>(wb0 + 32 * st); fw[1][st] = *reinterpret_cast<const half8*>(wb0 + 16 * C1F_WPITCH + 32 * st); }
  half_t* const outs = c.h_a1 + ((int64_t)z * c.B + n) * (PIX1 * K1);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)outs, 0, PIX1 * K1 * 2, 0x00020000);
  for (int rt = wave; rt < PIX1 / 16; rt += 4) {
    const int pos = 16 * rt + r, p = pos / Q1, q = pos - p * Q1;
    const half_t* pa = smem + (ST1 * p + kg) * W0 + ST1 * q;
    c1f_f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const half_t* ap = pa + (st >> 1) * C1G_FR + 4 * (st & 1) * W0;
      const c1f_h4 a_lo = *reinterpret_cast<const c1f_h4*>(ap), a_hi = *reinterpret_cast<const c1f_h4*>(ap + 4);
      half8 fa; fa[0] = a_lo[0]; fa[1] = a_lo[1]; fa[2] = a_lo[2]; fa[3] = a_lo[3]; fa[4] = a_hi[0]; fa[5] = a_hi[1]; fa[6] = a_hi[2]; fa[7] = a_hi[3];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[0][st], fa, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[1][st], fa, acc1, 0, 0, 0);
    }
    // D[map 4 kg + e][position r]: Rectlin(sum / 255) as half, 4 consecutive maps = 8 bytes of the position's 64-byte row
    c1f_h4 h0, h1;
#pragma unroll
    for (int e = 0; e < 4; ++e) { h0[e] = (half_t)fmaxf(acc0[e] * (1.0f / 255.0f), 0.0f); h1[e] = (half_t)fmaxf(acc1[e] * (1.0f / 255.0f), 0.0f); }
    *reinterpret_cast<c1f_h4*>(ot + r * C1G_OPITCH + 4 * kg) = h0;
    *reinterpret_cast<c1f_h4*>(ot + r * C1G_OPITCH + 16 + 4 * kg) = h1;
    __builtin_amdgcn_s_waitcnt(0xc07f);                                  // lgkmcnt(0): the tile is wave-private
    __builtin_amdgcn_wave_barrier();
    const c1h_u32x4 v = *reinterpret_cast<const c1h_u32x4*>(ot + (lane >> 2) * C1G_OPITCH + 8 * (lane & 3));
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, 1024 * rt + 16 * lane, 0, WT ? 16 : 0);
    __builtin_amdgcn_wave_barrier();
  }
}

static hipError_t launch_conv1_h(const StepArgs& a, const LaunchTune& t, hipStream_t s) {
  Conv1HArgs c; c.src = a.src; c.idx = a.idx; c.wht[0] = a.wht[0]; c.wht[1] = a.wht[1]; c.h_a1 = a.h_a1; c.B = a.B; c.from_ring = a.from_ring;
  if (t.bt[K_CONV1_FWD] == 1) SDQN_LAUNCH(conv1_h_kernel, dim3(a.nz * a.B), dim3(256), 0, s, c);           // first form (half(b / 255) operands)
  else if (t.bt[K_CONV1_FWD] == 2) SDQN_LAUNCH(conv1_hb_kernel<false>, dim3(a.nz * a.B), dim3(256), 0, s, c);
  else SDQN_LAUNCH(conv1_hb_kernel<true>, dim3(a.nz * a.B), dim3(256), 0, s, c);
  return hipGetLastError();
}

// ---- float32 mode: conv1's weight gradient with transpose-read fragments (second form of c1w_bt_kernel) --------------------------------------
// c1w_bt_kernel builds a lane's A fragment from 8 single-byte LDS reads + conversions and its three B fragments from 24 two-byte reads per
// step.  Here the staged frame rows hold the bytes as bf16 (exact: a byte is an 8-bit integer) and every fragment is two
// ds_read_b64_tr_b16 (tools/exp/tr16_probe.hip; the geometry of c1w_h_kernel).  Same products (byte x bf16 plane of delta1, exact), the
// same three MFMAs per step and sub-tile in the same order, the same epilogue: bit-identical to c1w_bt_kernel.
constexpr int C1B_STAGE = C0 * C1H_FR + C1W_CH * C1W_DPITCH;        // ushorts per LDS stage: 7040 + 8320
__global__ void __launch_bounds__(256) c1w_bt2_kernel(const C1wBtArgs c) {
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * C1B_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = blockIdx.x, Kt = c.Kt, kb = ks * c.tps1 * 32;
  int ke = kb + c.tps1 * 32; if (ke > Kt) ke = Kt;
  const int nch = (ke - kb) / C1W_CH;
  typedef float c1b_f4 __attribute__((ext_vector_type(4)));
  // frame rows of a chunk: 4 frames x 1 680 contiguous, 16-byte aligned bytes = 420 pieces of 16 bytes (2 per thread, the second partly; round 6 —
  // until then 7 four-byte loads per thread with their index arithmetic: the float16 twin of this loop went from 9.2 to 7.7 us that way)
  struct Stg { c1h_u32x4 f[2]; c1b_f4 d0, d1, d2; };
  int fsrc[2], fdst[2][4];                          // byte offset of piece j in the chunk's frame window / LDS element offsets of its 4 dwords
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int it = tid + 256 * j; if (it > 419) it = 419;
    const int fc = it / 105, q4 = it - fc * 105;
    fsrc[j] = fc * FRAME + 16 * q4;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int q = 4 * q4 + e, row = q / 21, col = 4 * (q - row * 21); fdst[j][e] = fc * C1H_FR + row * C1H_PITCH + col; }
  }
  auto gload = [&](int ch, Stg& g) {
    const int k0 = kb + ch * C1W_CH, n = k0 / PIX1, y0 = (k0 - n * PIX1) / Q1;
    const int64_t fb = (c.from_ring ? (c.idx[n] - C0) * (int64_t)FRAME : (int64_t)n * STATE) + (int64_t)y0 * (ST1 * W0);
#pragma unroll
    for (int j = 0; j < 2; ++j) g.f[j] = *reinterpret_cast<const c1h_u32x4*>(c.src + fb + fsrc[j]);
    const c1b_f4* dp = reinterpret_cast<const c1b_f4*>(c.d1 + (size_t)k0 * K1);
    g.d0 = dp[tid]; g.d1 = dp[tid + 256]; g.d2 = dp[tid + 512 < 640 ? tid + 512 : 639];
  };
  auto st_d = [&](unsigned short* dp, int it, const c1b_f4& d) {
    if (it >= 640) return;
    const int k = it >> 3, n4 = (it & 7) * 4;
    uint16_t a0, a1, a2, b0, b1, b2, c0, c1, c2, e0, e1, e2;
    split_bf16x3(d.x, a0, a1, a2); split_bf16x3(d.y, b0, b1, b2); split_bf16x3(d.z, c0, c1, c2); split_bf16x3(d.w, e0, e1, e2);
    auto pk2 = [](uint16_t lo, uint16_t hi) { return (uint32_t)lo | ((uint32_t)hi << 16); };
    unsigned short* row = dp + k * C1W_DPITCH + n4;
    *reinterpret_cast<uint2*>(row) = make_uint2(pk2(a0, b0), pk2(c0, e0));
    *reinterpret_cast<uint2*>(row + K1) = make_uint2(pk2(a1, b1), pk2(c1, e1));
    *reinterpret_cast<uint2*>(row + 2 * K1) = make_uint2(pk2(a2, b2), pk2(c2, e2));
  };
  auto lds_store = [&](const Stg& g, unsigned short* st) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (tid + 256 * j >= 420) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t w = g.f[j][e];
        const uint32_t f0 = __float_as_uint((float)(w & 255u)), f1 = __float_as_uint((float)((w >> 8) & 255u));
        const uint32_t f2 = __float_as_uint((float)((w >> 16) & 255u)), f3 = __float_as_uint((float)(w >> 24));
        *reinterpret_cast<uint2*>(st + fdst[j][e]) =
            make_uint2(__builtin_amdgcn_perm(f1, f0, 0x07060302u), __builtin_amdgcn_perm(f3, f2, 0x07060302u));      // bf16 = upper half of the float
      }
    }
    unsigned short* dp = st + C0 * C1H_FR;
    st_d(dp, tid, g.d0); st_d(dp, tid + 256, g.d1); st_d(dp, tid + 512, g.d2);
  };
  f32x16 acc[2];
#pragma unroll
  for (int sm = 0; sm < 2; ++sm)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[sm][q] = 0.0f;
  const int t = lane & 15, G = lane >> 4;
  const int a_m = wave * C1H_FR + (2 * (G & 1) + ((t & 3) >> 1)) * C1H_PITCH + 4 * (t & 1);
  int a_pos[10];
#pragma unroll
  for (int q = 0; q < 10; ++q) { const int pos = 16 * (q >> 1) + 8 * (G >> 1) + 4 * (q & 1) + (t >> 2); a_pos[q] = (pos / Q1) * (ST1 * C1H_PITCH) + (pos % Q1) * ST1; }
  const int b_lane = (8 * (G >> 1) + (t >> 2)) * C1W_DPITCH + 16 * (G & 1) + 4 * (t & 3);
  auto frag = [](const unsigned short* p0, const unsigned short* p1) {
    const half8 f = bt_tr_frag(reinterpret_cast<const half_t*>(p0), reinterpret_cast<const half_t*>(p1));
    union { half8 h; c1w_bf16x8 b; } u; u.h = f; return u.b;                   // (bits are bits: the transpose read moves 16-bit elements)
  };
  if (nch > 0) {
    Stg g;
    gload(0, g);
    lds_store(g, smem);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
      const unsigned short* st = smem + (ch & 1) * C1B_STAGE;
      if (ch + 1 < nch) gload(ch + 1, g);
      const unsigned short* pa = st + a_m;
      const unsigned short* pd = st + C0 * C1H_FR + b_lane;
#pragma unroll
      for (int s5 = 0; s5 < 5; ++s5) {
        const unsigned short* pk = pd + 16 * s5 * C1W_DPITCH;
        const c1w_bf16x8 B0 = frag(pk, pk + 4 * C1W_DPITCH), B1 = frag(pk + K1, pk + K1 + 4 * C1W_DPITCH), B2 = frag(pk + 2 * K1, pk + 2 * K1 + 4 * C1W_DPITCH);
#pragma unroll
        for (int sm = 0; sm < 2; ++sm) {
          const c1w_bf16x8 A = frag(pa + a_pos[2 * s5] + sm * (4 * C1H_PITCH), pa + a_pos[2 * s5 + 1] + sm * (4 * C1H_PITCH));
          acc[sm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B2, acc[sm], 0, 0, 0);     // small planes first (as c1w_bt_kernel)
          acc[sm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B1, acc[sm], 0, 0, 0);
          acc[sm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B0, acc[sm], 0, 0, 0);
        }
      }
      if (ch + 1 < nch) { lds_store(g, smem + ((ch + 1) & 1) * C1B_STAGE); __syncthreads(); }
    }
  }
  const float r255 = 1.0f / 255.0f;
#pragma unroll
  for (int sm = 0; sm < 2; ++sm)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = 64 * wave + 32 * sm + bt::acc_row(q, h);
      const float sv = acc[sm][q], qv = sv * r255;
      wt_store(c.slab1 + (int64_t)ks * NW1 + m * K1 + i, fmaf(fmaf(-qv, 255.0f, sv), r255, qv));
    }
}

static hipError_t launch_c1w_bt2(const StepArgs& a, hipStream_t s) {
  if ((a.tps1 * 32) % C1W_CH != 0) return hipErrorInvalidValue;
  C1wBtArgs c; c.src = a.src; c.d1 = a.d1; c.slab1 = a.slab1; c.idx = a.idx; c.B = a.B; c.from_ring = a.from_ring; c.tps1 = a.tps1; c.Kt = a.B * PIX1;
  SDQN_LAUNCH(c1w_bt2_kernel, dim3(Conv1Wgrad::nbz(a)), dim3(256), 0, s, c);
  return hipGetLastError();
}

// ---- plane mode (StepArgs::xp = 9 / 6): conv2 / conv3 forward, the three dgrads on packed-bf16 MFMA with weight planes ------------------
// EXPERIMENTS BUILD ONLY.  Exact (9 partial products) it ran the B = 256 step in 262.7 us against 224.5 on fp32 MFMA, with 6 products in
// 244.9 (one box, alternating runs; tools/exp/README.md): the launches are latency- not MFMA-bound there and the split costs VALU time.
hipError_t launch_refresh_planes(const float*, unsigned short*, unsigned short*, hipStream_t) { return hipErrorInvalidValue; }   // (never called: no planes)

// ---- float16 mode, B >= 128: forward / dgrad launches on the half block-tile routine (menu per kernel id like the fp32 one) ---------------
#define BTH(P, BM, BN, WM, WN, D) BtCfgH<P, BM, BN, WM, WN, D>
#define BTH128_CASE(N, P, BM, BN, WM, WN) case N: return launch_bt_h<BtCfgH<P, BM, BN, WM, WN, 2, 128> >(a, s)      /* 128-deep chunks */
#define BTH_CASE(N, P, BM, BN, WM, WN, D) case N: return launch_bt_h<BTH(P, BM, BN, WM, WN, D)>(a, s)
static hipError_t launch_single_h(int id, int menu, const StepArgs& a, hipStream_t s) {
  switch (id) {
    // (conv1 forward gathers bytes and converts per lane: 19.4 us here against 17.7 on the register-blocked routine -> only on request)
    case K_CONV1_FWD: switch (menu) { BTH_CASE(1, Conv1FwdH, 128, 32, 4, 1, 2); BTH_CASE(2, Conv1FwdH, 256, 32, 4, 1, 2); default: break; } break;
    case K_CONV2_FWD: switch (menu) { BTH_CASE(0, Conv2FwdH, 64, 64, 2, 2, 2); BTH_CASE(6, Conv2FwdH, 64, 64, 2, 2, 2); BTH128_CASE(3, Conv2FwdH, 64, 64, 2, 2); BTH_CASE(1, Conv2FwdH, 128, 64, 2, 2, 2); BTH_CASE(2, Conv2FwdH, 64, 64, 2, 2, 3); default: break; } break;
    case K_CONV3_FWD: switch (menu) { BTH_CASE(0, Conv3FwdH, 64, 64, 2, 2, 2); BTH_CASE(6, Conv3FwdH, 64, 64, 2, 2, 2); BTH128_CASE(3, Conv3FwdH, 64, 64, 2, 2); BTH_CASE(1, Conv3FwdH, 128, 64, 2, 2, 2); BTH_CASE(2, Conv3FwdH, 64, 64, 2, 2, 3); default: break; } break;
    case K_FC4_FWD: switch (menu) { BTH_CASE(0, Fc4FwdH, 64, 64, 2, 2, 2); BTH128_CASE(3, Fc4FwdH, 64, 64, 2, 2); BTH_CASE(1, Fc4FwdH, 128, 128, 2, 2, 2); BTH_CASE(2, Fc4FwdH, 64, 128, 2, 2, 2); default: break; } break;
    case K_FC4_DGRAD: switch (menu) { BTH_CASE(0, Fc4DgradH, 64, 64, 2, 2, 2); BTH128_CASE(3, Fc4DgradH, 64, 64, 2, 2); BTH_CASE(1, Fc4DgradH, 128, 128, 2, 2, 2); BTH_CASE(2, Fc4DgradH, 64, 128, 2, 2, 2); default: break; } break;
    case K_CONV3_DGRAD: switch (menu) { BTH_CASE(0, Conv3DgradH, 64, 64, 2, 2, 2); BTH_CASE(6, Conv3DgradH, 64, 64, 2, 2, 2); BTH128_CASE(3, Conv3DgradH, 64, 64, 2, 2); BTH_CASE(1, Conv3DgradH, 128, 64, 2, 2, 2); BTH_CASE(2, Conv3DgradH, 64, 64, 2, 2, 3); default: break; } break;
    case K_CONV2_DGRAD: switch (menu) { BTH_CASE(0, Conv2DgradH, 128, 32, 4, 1, 2); BTH_CASE(6, Conv2DgradH, 128, 32, 4, 1, 2); BTH128_CASE(3, Conv2DgradH, 128, 32, 4, 1); BTH_CASE(1, Conv2DgradH, 256, 32, 4, 1, 2); BTH_CASE(2, Conv2DgradH, 128, 32, 4, 1, 3); default: break; } break;
    default: break;
  }
  return hipErrorInvalidValue;
}

// every K range of the launch must be whole chunks for the x-contiguous loaders' zero fill to be the only tail handling — it is
// (the loaders mask any k >= kend), so the routine takes every B >= 128; what it does not take: fp16 mode, batch-norm (raw outputs)
hipError_t launch_kernel_bt(int id, const StepArgs& a, const LaunchTune& t, hipStream_t s, bool* handled) {
  *handled = false;
  if (a.bn) return hipSuccess;
  if (id < 0 || id >= K_COUNT || t.bt[id] < 0) return hipSuccess;
  if (a.B < 128) {                         // below the throughput regime: float16's exact-byte conv1 kernels
    // forward: one workgroup per (net, sample) pays from 2 x 48 workgroups up (fused-loop steps/s against the latency engine's tiles:
    // B = 32 18 490 vs 18 755, 48 14 833 vs 14 527, 64 13 943 vs 13 370, 100 11 198 vs 10 453); menu entry 7 = always, 6 = never
    if (a.h16 && id == K_CONV1_FWD && (t.bt[id] == 7 || (t.bt[id] == 0 && a.B >= 48)) && t.nw_override[id] == 0 && a.idx_t == nullptr) { *handled = true; return launch_conv1_h(a, t, s); }
    if (a.h16 == 2 && id == K_BWD1 && t.bt[id] == 7 && a.f4w_count == 0) {
      const hipError_t e1 = launch_c1w_h(a, t, s);
      if (e1 == hipErrorInvalidValue) return hipSuccess;
      *handled = true;
      return e1;
    }
    return hipSuccess;
  }
  if (a.h16) {                             // float16 mode: forward launches, dgrads, and the weight gradients behind them
    if (id == K_WGRADS && a.h16 == 2) {    // fc4_wgrad (+ fused RMSProp) || conv3_wgrad || conv2_wgrad: k-major half panels, transpose reads
      // (2 chunks of loads in flight per thread: 20.5 us at B = 256 against 22.4 / 22.0 with 3 / 4 — the launch is not load-latency bound)
      typedef BtCfgHW<Fc4WgradH, 64, 64, 2, 2, 2> HF4W;
      typedef BtCfgHW<Conv3WgradH, 64, 64, 2, 2, 2> HC3W;
      typedef BtCfgHW<Conv2WgradH, 64, 64, 2, 2, 2> HC2W;
      *handled = true;
      // r3 bit 4 (the step orchestration's decision, sdqn_api_step.hip): conv1's weight gradient rides in this launch and K_BWD1 launches nothing;
      // bit 5: its workgroups first in the block-id order
      if (t.r3 & 16) return launch_bt_multi_c1w<HF4W, HC3W, HC2W>(a, (t.r3 & 32) ? 1 : 0, s);
      if (t.bt[id] == 1) return launch_bt_multi<BtCfgHW<Fc4WgradH, 64, 64, 2, 2, 4>, BtCfgHW<Conv3WgradH, 64, 64, 2, 2, 4>, BtCfgHW<Conv2WgradH, 64, 64, 2, 2, 4>>(a, true, true, true, s);
      if (t.bt[id] == 2) return launch_bt_multi<BtCfgHW<Fc4WgradH, 64, 64, 2, 2, 3>, BtCfgHW<Conv3WgradH, 64, 64, 2, 2, 3>, BtCfgHW<Conv2WgradH, 64, 64, 2, 2, 3>>(a, true, true, true, s);
      return launch_bt_multi<HF4W, HC3W, HC2W>(a, true, true, true, s);
    }
    if (id == K_CONV1_FWD && t.bt[id] >= 0 && t.bt[id] <= 2 && t.nw_override[id] == 0 && a.idx_t == nullptr) {   // one workgroup per (net, sample)
      *handled = true;
      return launch_conv1_h(a, t, s);
    }
    if (id == K_BWD1 && (t.r3 & 16)) { *handled = true; return hipSuccess; }      // (rode in the weight-gradient launch)
    if (id == K_BWD1 && a.h16 == 2 && a.f4w_count == 0) {    // conv1's weight gradient: all 256 x 32 outputs of a K slab per workgroup, A from the bytes
      // (the generic half routine with A from the bytes — BtCfgHW<Conv1WgradH, 256, 32, 4, 1> — fetches 8-byte patch-row pieces straight
      //  from memory: 16 divergent loads per thread and chunk, 18.9 us at B = 256, no better than the wave-tile routine's 18.7)
      const hipError_t e1 = launch_c1w_h(a, t, s);
      if (e1 == hipErrorInvalidValue) return hipSuccess;
      *handled = true;
      return e1;
    }
    if (id >= 12 || t.nw_override[id] > 0) return hipSuccess;
    const hipError_t eh = launch_single_h(id, t.bt[id], a, s);
    if (eh == hipErrorInvalidValue) return hipSuccess;
    *handled = true;
    return eh;
  }
  // fc4 forward has 64 blocks of 64 x 64 per K slab and measured slower here than on the latency engine (21.0 vs 18.1 us at B = 256 with 7
  // slabs and unconditional ring loads): block-tile only on request (menu entry > 0).  fc4_dgrad (196 blocks) moved here in round 4's second
  // session: 12.4 us against 14.4 once its gating activations are fetched before the K loop (17.0 with the dependent loads in the epilogue)
  if (id == K_FC4_FWD && t.bt[id] == 0) return hipSuccess;
  if (id < 12 && t.nw_override[id] > 0) return hipSuccess;      // explicit latency-engine tuning hooks win
  hipError_t e = hipErrorInvalidValue;
  if ((id == K_BWD1 && a.f4w_count == 0) || id == K_CONV1_WGRAD) {             // conv1's weight gradient: bytes x three bf16 planes of delta1
    e = t.bt[id] == 1 ? launch_c1w_bt(a, s) : launch_c1w_bt2(a, s);      // (menu 1: the first form, fragments from single-byte / two-byte LDS reads)
    if (e == hipErrorInvalidValue) return hipSuccess;
    *handled = true;
    return e;
  }
  if (id == K_BWD3 || id == K_BWD2) e = launch_fused(id, t.bt[id], a, s);
  else if (id == K_CONV2_FWD || id == K_CONV3_FWD || id == K_FC4_FWD || id == K_FC4_DGRAD || id == K_FC4_WGRAD || id == K_CONV3_DGRAD ||
           id == K_CONV3_WGRAD || id == K_CONV2_DGRAD || id == K_CONV2_WGRAD) e = launch_single(id, t.bt[id], a, s);
  if (e == hipErrorInvalidValue) return hipSuccess;       // no such entry: the caller falls through to the latency engine
  *handled = true;
  return e;
}

#ifdef SDQN_TIMING
hipError_t set_timing_buffer_bt(unsigned long long* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_sdqn_dbg), &p, sizeof p); }
hipError_t set_wave_timing_buffer_bt(unsigned long long* const* p, const unsigned* nb, hipStream_t s) {
  hipError_t e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_sdqn_wdbg_blocks), nb, sizeof *nb, 0, hipMemcpyHostToDevice, s);
  return e != hipSuccess ? e : hipMemcpyToSymbolAsync(HIP_SYMBOL(g_sdqn_wdbg), p, sizeof *p, 0, hipMemcpyHostToDevice, s);
}
#endif

}  // namespace sdqn
