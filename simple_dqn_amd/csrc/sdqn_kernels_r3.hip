// sdqn_kernels_r3.hip — round-3 launch variants of the default fp32 step (own translation unit: hipcc's schedule of a kernel
// depends on what else is instantiated beside it, see sdqn_kernels.hip).
//
//   K_CONV1_FWD with LaunchTune::r3 bit 2: conv1_bf16_kernel below (bytes x 3-way bf16 split of W1 on packed-bf16 MFMA).
//   K_BWD1 with LaunchTune::r3 bit 3 (no fc4 share in the launch): conv1_wgrad_bf16_kernel (bytes x on-the-fly 3-way bf16 split of delta1).
//   K_CONV3_FWD with LaunchTune::r3 bit 1 (B < 128): gemm36_kernel below.
//   K_FC4_DGRAD with LaunchTune::r3 bit 0 (B <= 32): ONE launch of 1024-thread workgroups =
//       98 x Staged<Fc4DgradSig> tiles (16 waves each, K = 512 split over the waves)          block ids 0..97   (dispatched first)
//     + 98 x 16 Fc4WgradWait tiles (one 32x32 tile of gW4 per wave, K = B, fused RMSProp)      block ids 98..195
//   Same tiles, same K split, same epilogues as the separate launches -> bit-identical results; what changes is WHEN the
//   25.7 MB read-modify-write of W4 and its RMSProp state runs: beside the latency-bound dgrad (98 of 256 CUs busy) instead of
//   inside bwd3.  The in-place update is ordered behind the dgrad's reads of the same W4 rows by one flag word per row block
//   (problems.h: Fc4DgradSig / Fc4WgradWait) — a write-after-read hand-off, no data crosses between the workgroups.
#include "gemm_engine.h"
#include "problems_h16.h"
#include "kernels.h"
#include "update_body.h"
#include "problems_wt.h"      // the write-through (sc1) epilogue variants of the fp32 problems

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
  if constexpr (has_preload<P>::value) P::preload(a, gridDim.x, gridDim.y, gridDim.z);
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

// ---- conv1 forward on packed-bf16 MFMA --------------------------------------------------------------------------------------
// conv1's input is bytes.  An integer 0..255 is EXACT in bf16, and an fp32 weight is exactly the sum of three bf16 numbers
// (problems.h: split_bf16x3), so  sum_k x_k w_k = sum_k x_k hi_k + sum_k x_k mid_k + sum_k x_k lo_k  with every product exact
// (8 x 8 significant bits) on v_mfma_f32_32x32x16_bf16: 3 instructions of 32 cycles per 16 k instead of 8 fp32 MFMAs of 64 —
// 5.3x less matrix-pipe time for the same fp32-accumulated sum.  The 1/255 of deepqnetwork.py:100 is applied once to the sum
// (IEEE division) instead of to every pixel: (sum_k x_k w_k) / 255 vs sum_k fl(x_k / 255) w_k — both within fp32 round-off of the
// exact value, in a different order (the oracle's BLAS has its own); covered by the same per-stage tolerances as before.
//   * one WAVE per 32 x 32 output tile (32 positions x all 32 maps), no K split, no LDS combine: 16 k-steps = two kernel rows (r, r+1)
//     of one input frame each; lane (i, h) feeds row i with the 8 bytes frame[c][4 y + r + h][4 x .. 4 x + 7] (ONE 8-byte load — the
//     32 lanes of a half-wave read one contiguous 132-byte span), converted with v_cvt_f32_ubyte + v_perm (bf16 = upper half of the float)
//   * the three weight planes live in LDS ([plane][map][k], pitch 264: ds_read_b128 of the 16-lane groups is conflict-free), loaded
//     once per workgroup; a wave runs `tpw` tiles back to back at B >= 128
//   * epilogue: / 255 (reciprocal multiply + one fma correction: within an ulp of the IEEE quotient), Rectlin, NHWC store — no barrier
//     anywhere after the plane load
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
constexpr int W1P_PITCH = CRS1 + 8;

// Everything the kernel reads from its arguments fits ONE 64-byte scalar load (the engine's StepArgs is ~450 bytes and hipcc fetched
// it piecemeal: five dependent scalar round trips, 2.5 k of the tile's 11 k cycles — tools/phase_timing.py).
// IDX_IN: the sampled ring indexes ride behind that block in the kernel arguments (B <= 32, ring paths: they are host data at launch
// time); every lane fetches "its" index (lane & 31) with one vector load from the argument segment at wave start, the tile's two
// candidates (a 32-row tile touches at most two samples) come out with v_readlane: no dependent global load, no second scalar trip.
struct Conv1Args {
  const uint8_t* src; float* a1; const unsigned short* w1p[2]; const int64_t* idx;
  int B, nz, from_ring, tiles_per_net, wgs_per_net, tpw, xcd, pad_;
};
struct IdxIn { int64_t v[32]; };

__device__ __forceinline__ float div255(float s) {        // s / 255 to within an ulp: reciprocal multiply + one fma correction
  const float r = 1.0f / 255.0f;
  const float q = s * r;
  return fmaf(fmaf(-q, 255.0f, s), r, q);
}

// FUSED: the body runs inside the "update + next step's conv1" launch (upd_conv1_kernel below).  Target-net workgroups come first in
// block order and need nothing from the update; ONLINE workgroups take W1 straight from theta — written by the update blocks of the
// same launch with write-through stores — after those 64 blocks have counted themselves in (one relaxed poll by one lane, sc1 loads,
// no fence), and split it into the three LDS planes themselves (the same split_bf16x3 as the global planes: same bits).
struct FusedW1 { const float* theta; unsigned* ctr; unsigned target; unsigned* timeout; };

template <bool IDX_IN, bool FUSED>
__device__ __forceinline__ void conv1_bf16_body(const Conv1Args& c, const int64_t my_idx, const int cb, unsigned short* sw, const FusedW1& fw) {
  const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  {   // every argument field in flight NOW, one wait: left alone hipcc fetches each field where it is first used (five dependent scalar trips)
    const uint8_t* f0 = c.src; float* f1 = c.a1; const unsigned short *f2 = c.w1p[0], *f3 = c.w1p[1]; const int64_t* f4 = c.idx;
    int g0 = c.B, g1 = c.from_ring, g2 = c.tiles_per_net, g3 = c.wgs_per_net, g4 = c.tpw;
    asm volatile("" :: "s"(f0), "s"(f1), "s"(f2), "s"(f3), "s"(f4), "s"(g0), "s"(g1), "s"(g2), "s"(g3), "s"(g4));
  }
  // (XCD-contiguous map inside each net: neighbouring tiles share frames and halo rows)
  const int zi = cb / c.wgs_per_net, wg = c.xcd ? xcd_tile_id(cb - zi * c.wgs_per_net, c.wgs_per_net) : cb - zi * c.wgs_per_net;
  const int z = FUSED ? 1 - zi : zi;                                                         // 0 online, 1 target (nz = 1: online only); FUSED: target first
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int M = c.B * PIX1, tiles_per_net = c.tiles_per_net, tpw = c.tpw;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2), aligned(4)));
  const int tile0 = (wg * 4 + wave) * tpw;
  SDQN_STAMP(0);
  // frame bytes of one tile: 16 x 8-byte loads per lane, issued before anything waits
  auto load_tile = [&](int tile, u32x2* raw) {
    const int m0 = tile * 32, mrow = m0 + i, mc = mrow < M ? mrow : M - 1;
    const int n = mc / PIX1, pix = mc - n * PIX1, p = pix / Q1, q = pix - p * Q1;
    int64_t org;
    if constexpr (IDX_IN) {
      const int n_lo = m0 / PIX1, n_hi = n_lo + 1 < c.B ? n_lo + 1 : n_lo;                  // wave-uniform
      const uint32_t lo0 = __builtin_amdgcn_readlane((int)(my_idx & 0xFFFFFFFF), n_lo), lo1 = __builtin_amdgcn_readlane((int)(my_idx >> 32), n_lo);
      const uint32_t hi0 = __builtin_amdgcn_readlane((int)(my_idx & 0xFFFFFFFF), n_hi), hi1 = __builtin_amdgcn_readlane((int)(my_idx >> 32), n_hi);
      const int64_t i_lo = (int64_t)(((uint64_t)lo1 << 32) | lo0), i_hi = (int64_t)(((uint64_t)hi1 << 32) | hi0);
      org = ((n == n_lo ? i_lo : i_hi) - C0 + z) * (int64_t)FRAME;
    } else org = c.from_ring ? (c.idx[n] - C0 + z) * (int64_t)FRAME : ((int64_t)z * c.B + n) * (int64_t)STATE;       // problems.h: sbase
    const uint8_t* src = c.src + org + (int64_t)(p * ST1 + h) * W0 + q * ST1;
#pragma unroll
    for (int t = 0; t < 16; ++t) raw[t] = *reinterpret_cast<const u32x2*>(src + (t >> 2) * FRAME + 2 * (t & 3) * W0);
  };
  u32x2 raw[16];
#ifdef SDQN_TIMING
#pragma unroll
  for (int t = 0; t < 16; ++t) raw[t] = (u32x2){0u, 0u};
#endif
  if (tile0 < tiles_per_net) load_tile(tile0, raw);                                          // in flight under the plane fill below
  SDQN_STAMP(1);
  if (FUSED && z == 0) {
    if (threadIdx.x == 0) {                                   // the 64 W1 blocks of THIS launch have published (bounded: never a hung GPU)
      int spins = 0;
      while ((int)(__hip_atomic_load(fw.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - fw.target) < 0) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > 2000000) { __hip_atomic_store(fw.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    __syncthreads();
    // thread (n = t & 31, k-block = t >> 5 (+ 8 per pass)) takes 8 consecutive k of ONE map: 32 coalesced sc1 dword loads in flight
    // (W1i is [(c,r,s)][map]: a k-row is 128 contiguous bytes over the 32 lanes), then ONE 16-byte LDS store per plane and pass
    // (a float4-per-thread split needs 96 two-byte LDS stores per thread with 4-way bank conflicts: measured 2.6 us per workgroup)
    const uint32_t* th = reinterpret_cast<const uint32_t*>(fw.theta + OFF1);
    const int n = threadIdx.x & 31;
    uint32_t wv[4][8];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        wv[ps][j] = __hip_atomic_load(th + (8 * ((threadIdx.x >> 5) + 8 * ps) + j) * K1 + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: past L1
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int k0 = 8 * ((threadIdx.x >> 5) + 8 * ps);
      union { uint16_t h[8]; uint4 v; } P0, P1, P2;
#pragma unroll
      for (int j = 0; j < 8; ++j) split_bf16x3(__uint_as_float(wv[ps][j]), P0.h[j], P1.h[j], P2.h[j]);
      *reinterpret_cast<uint4*>(sw + n * W1P_PITCH + k0) = P0.v;
      *reinterpret_cast<uint4*>(sw + (K1 + n) * W1P_PITCH + k0) = P1.v;
      *reinterpret_cast<uint4*>(sw + (2 * K1 + n) * W1P_PITCH + k0) = P2.v;
    }
  } else {
    const uint4* wp = reinterpret_cast<const uint4*>(c.w1p[z]);
    static_assert(3 * K1 * (CRS1 / 8) == 12 * 256, "3072 chunks of 8 bf16: 12 per thread");
    uint4 v[12];
#pragma unroll
    for (int u = 0; u < 12; ++u) v[u] = wp[threadIdx.x + 256 * u];                           // all 12 loads in flight before the first LDS store
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      const int cc = threadIdx.x + 256 * u, row = cc >> 5, col = cc & 31;
      *reinterpret_cast<uint4*>(sw + row * W1P_PITCH + col * 8) = v[u];
    }
  }
  SDQN_STAMP(2);
  __syncthreads();
  SDQN_STAMP(3);
#ifdef SDQN_TIMING
  asm volatile("" :: "v"(raw[0].x), "v"(raw[15].y));
  SDQN_STAMP(4);
#endif
  const unsigned short* bw = sw + i * W1P_PITCH + 8 * h;
  auto cvt = [](const u32x2& r, bf16x8_t& out) {                                              // 8 bytes -> 8 exact bf16
    union { uint32_t u[4]; bf16x8_t v; } A;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const uint32_t w = d ? r.y : r.x;
      const uint32_t f0 = __float_as_uint((float)(w & 255u)), f1 = __float_as_uint((float)((w >> 8) & 255u));
      const uint32_t f2 = __float_as_uint((float)((w >> 16) & 255u)), f3 = __float_as_uint((float)(w >> 24));
      A.u[2 * d] = __builtin_amdgcn_perm(f1, f0, 0x07060302u);                               // {hi16(f1), hi16(f0)}: bf16 = upper half of the float
      A.u[2 * d + 1] = __builtin_amdgcn_perm(f3, f2, 0x07060302u);
    }
    out = A.v;
  };
  for (int it = 0; it < tpw; ++it) {
    const int tile = tile0 + it;
    if (tile >= tiles_per_net) break;                                                         // wave-uniform
    const int m0 = tile * 32;
    u32x2 nxt[16];
    const bool more = it + 1 < tpw && tile + 1 < tiles_per_net;
    if (more) load_tile(tile + 1, nxt);                                                      // next tile's bytes fly under this tile's MFMAs
    // one accumulator per weight plane: three independent MFMA chains (a single wave per SIMD has nothing else to hide the
    // dependent-accumulator latency behind); summed hi + mid + lo at the end.  B fragments are read one step ahead.
    f32x16 acc0, acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; acc2[r] = 0.0f; }
    bf16x8_t Bc0 = *reinterpret_cast<const bf16x8_t*>(bw), Bc1 = *reinterpret_cast<const bf16x8_t*>(bw + K1 * W1P_PITCH),
             Bc2 = *reinterpret_cast<const bf16x8_t*>(bw + 2 * K1 * W1P_PITCH), Ac;
    cvt(raw[0], Ac);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      bf16x8_t Bn0 = Bc0, Bn1 = Bc1, Bn2 = Bc2, An = Ac;
      if (t + 1 < 16) {
        Bn0 = *reinterpret_cast<const bf16x8_t*>(bw + 16 * (t + 1));
        Bn1 = *reinterpret_cast<const bf16x8_t*>(bw + K1 * W1P_PITCH + 16 * (t + 1));
        Bn2 = *reinterpret_cast<const bf16x8_t*>(bw + 2 * K1 * W1P_PITCH + 16 * (t + 1));
        cvt(raw[t + 1], An);
      }
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac, Bc0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac, Bc1, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ac, Bc2, acc2, 0, 0, 0);
      Ac = An; Bc0 = Bn0; Bc1 = Bn1; Bc2 = Bn2;
    }
#ifdef SDQN_TIMING
    asm volatile("" :: "v"(acc0[0]), "v"(acc1[7]), "v"(acc2[15]));
    SDQN_STAMP(5);
#endif
    float* out = c.a1 + ((int64_t)z * M + m0 + 4 * h) * K1 + i;
    if (m0 + 32 <= M) {                                                                       // full tile (wave-uniform): no per-row guard
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = fmaxf(div255((acc0[r] + acc1[r]) + acc2[r]), 0.0f);
        if (c.pad_) wt_store(out + ((r & 3) + 8 * (r >> 2)) * K1, v);             // write-through (LaunchTune::wt bit 7)
        else out[((r & 3) + 8 * (r >> 2)) * K1] = v;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = (r & 3) + 8 * (r >> 2);
        if (m0 + 4 * h + ml < M) out[ml * K1] = fmaxf(div255((acc0[r] + acc1[r]) + acc2[r]), 0.0f);
      }
    }
    if (more) {
#pragma unroll
      for (int t = 0; t < 16; ++t) raw[t] = nxt[t];
    }
  }
  SDQN_STAMP(6);
}

template <bool IDX_IN>
__global__ void __launch_bounds__(256) conv1_bf16_kernel(const Conv1Args c, const IdxIn ix) {
  __shared__ __attribute__((aligned(16))) unsigned short sw[3 * K1 * W1P_PITCH];          // 50 688 B
  int64_t my_idx = 0;
  if constexpr (IDX_IN) {                                   // issued first: needs nothing but the argument-segment pointer
    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    my_idx = *reinterpret_cast<const int64_t*>(ka + sizeof(Conv1Args) + 8 * (threadIdx.x & 31));
  }
  (void)ix;
  const FusedW1 fw = {nullptr, nullptr, 0u, nullptr};
  conv1_bf16_body<IDX_IN, false>(c, my_idx, (int)blockIdx.x, sw, fw);
}

// ---- conv1 forward for B >= 128, round 5: persistent workgroups, W1's planes in REGISTERS, the frames streamed through LDS in row chunks ----
// Round 4's staged kernel (one workgroup per (net, sample); tools/exp/conv1_forms_r5.hip.txt) ran one serial phase per workgroup — load 28 KB of frames plus 50 KB of weight planes, barrier, 13 tiles over 4
// waves, stores — and both co-resident workgroups of a CU sit in the same phase at the same time (15.9 us at B = 256: 0.28 of the HBM
// roofline of its 35 MB; 52 % of the wave-cycles in s_waitcnt, matrix pipe 9 % busy); every tile re-reads all 48 KB of planes from LDS and
// re-converts its bytes (each byte 4 x per net).  Here:
//   * ONE workgroup of 10 waves per CU, persistent over the samples of ONE net (workgroup g of net z takes samples g, g + Gz, ...: two at
//     B = 256), so the 48 KB of planes are fetched once per workgroup and live in REGISTERS as MFMA fragments (3 planes x 8 steps x 16
//     bytes per lane for the wave's 16-map half) — no plane traffic in the loop at all;
//   * the unit of work is an ITEM = 4 output rows of one sample = 80 positions = 5 tiles of 16 x both 16-map halves = ONE tile per wave:
//     its input is 20 rows of each of the 4 frames (4 contiguous 1 680-byte pieces of the ring), fetched C1R_D items ahead into registers
//     (one 16-byte load per thread), converted ONCE — bytes -> bf16, exact — into a double-buffered LDS image [frame][20 rows][84] one
//     item before its use; one barrier per item.  HBM latency hides behind C1R_D items of MFMAs, an item's stores drain under the next ones;
//   * v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the row operand: D[map][position], so a lane ends up with 4 consecutive maps of one
//     position = one 16-byte store; a position's fragment is the 8 consecutive bf16 of a patch row = two 8-byte LDS reads (k-step t:
//     frame t >> 1, kernel rows 4 (t & 1) + lane group) at an address that is the same for every item.
// Same exact products (byte x bf16 plane), fp32 accumulation, (hi + mid) + lo, one division by 255 of the sum, Rectlin: the 16 x 16 x 32
// instruction adds its 32 products in its own order, so values differ from the 32 x 32 x 16 kernels in the last bit (same tolerances).
constexpr int C1R_ROWS = 4 * ST1 + 4;                // input rows of an item: 20
constexpr int C1R_SEG = C1R_ROWS * W0;               // bytes (= bf16 elements) per frame piece: 1 680 = 105 x 16
constexpr int C1R_ITEM = C0 * C1R_SEG;               // 6 720 elements: 13 440 bytes of bf16 per item
constexpr int C1R_CHUNKS = C1R_ITEM / 16;            // 420 sixteen-byte loads per item
constexpr int C1R_OPITCH = K1 + 4;                   // floats per position of the output image in LDS (pitch 36: conflict-free 16-byte writes)
constexpr int C1R_OUT = 4 * Q1 * C1R_OPITCH;         // 2 880 floats per item
static_assert(4 * Q1 * K1 * 4 == 640 * 16, "an item's output is exactly one 16-byte store per thread of the 640");
constexpr int C1R_NCH = P1 / 4;                      // items per sample: 5
constexpr int C1R_D = C1R_NCH;                       // items of loads in flight per thread = one sample's worth: the item loop unrolls per
                                                     // sample into straight-line code, register slots and vmcnt waits are static
typedef float c1p_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t c1p_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t c1p_u32x2 __attribute__((ext_vector_type(2)));
static_assert(P1 % 4 == 0 && (4 * Q1) % 16 == 0 && C1R_SEG % 16 == 0 && (4 * ST1 * W0) % 16 == 0, "an item is whole tiles and whole 16-byte pieces");

// ---- conv1 weight gradient on packed-bf16 MFMA ---------------------------------------------------------------------------
// gW1[(c,r,s)][map] = sum over (sample, y, x) of byte(c, 4y + r, 4x + s) / 255 * delta1(sample, y, x, map): the bytes are exact in
// bf16 again, and the fp32 deltas are split into three bf16 (hi + mid + lo == delta exactly, problems.h) ON THE FLY by the lanes that
// load them (v_cvt_pk_bf16_f32 + exact residuals): 3 MFMAs of 32 cycles per 16 k instead of 8 of 64, every product exact, fp32
// accumulation, ONE division by 255 of each split-K partial.  Same tile = (32 rows of (c,r,s), all 32 maps, one K slab), same 16
// waves x one 32-deep chunk each, same slab layout and fixed-order LDS combine as the engine's Conv1Wgrad -> the update kernel reads it
// unchanged.  A: lane (row, h) takes two aligned groups of 4 consecutive output positions with ONE unaligned 16-byte ring load each
// (bytes 0, 4, 8, 12: the engine's A_GROUP4 trick); the groups' (sample, y, x) are wave-uniform per half-wave: scalar index math.
struct C1wArgs { const uint8_t* src; const float* d1; float* slab1; const int64_t* idx; int B, from_ring, tps1, Kt, xcd, pad_; };

template <bool IDX_IN>
__global__ void __launch_bounds__(1024) conv1_wgrad_bf16_kernel(const C1wArgs c, const IdxIn ix) {
  __shared__ float smem[16 * PANEL];
  const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  int64_t my_idx = 0;
  if constexpr (IDX_IN) {
    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    my_idx = *reinterpret_cast<const int64_t*>(ka + sizeof(C1wArgs) + 8 * i);
  }
  (void)ix;
  {
    const uint8_t* f0 = c.src; const float* f1 = c.d1; float* f2 = c.slab1; const int64_t* f4 = c.idx;
    int g0 = c.B, g1 = c.from_ring, g2 = c.tps1, g3 = c.Kt;
    asm volatile("" :: "s"(f0), "s"(f1), "s"(f2), "s"(f4), "s"(g0), "s"(g1), "s"(g2), "s"(g3));
  }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-contiguous tile map (gemm_engine.h: xcd_tile_id): the 8 row tiles of a K slab read the same frames and deltas, so a slab belongs on
  // ONE XCD's L2 (workgroup b runs on XCD b % 8: without the map a slab's 8 tiles land on 8 different L2s, 5.6x the algorithmic traffic)
  const int lin = (int)(blockIdx.x + gridDim.x * blockIdx.z);
  const int tl = c.xcd ? xcd_tile_id(lin, (int)(gridDim.x * gridDim.z)) : lin;
  const int bx = tl & 7, ks = tl >> 3;
  const int m = 32 * bx + i;
  const int colm = (m >> 6) * FRAME + ((m >> 3) & 7) * W0 + (m & 7);                         // problems.h: col1
  const int Kt = c.Kt, kb = ks * c.tps1 * 32;
  int ke = kb + c.tps1 * 32; if (ke > Kt) ke = Kt;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(1)));
  // a 32-deep chunk touches at most two samples: their ring indexes are fetched ONCE per chunk (two readlanes / two scalar loads issued
  // together), the origin of each aligned group of 4 positions is wave-uniform arithmetic on top (the half-waves differ by 8 positions)
  auto load_chunk = [&](int kc, u32x4* ra, float* rb) {
    const int n0 = kc / PIX1, n1 = n0 + 1 < c.B ? n0 + 1 : n0;
    int64_t i0, i1;
    if constexpr (IDX_IN) {
      const uint32_t a0 = __builtin_amdgcn_readlane((int)(my_idx & 0xFFFFFFFF), n0), a1 = __builtin_amdgcn_readlane((int)(my_idx >> 32), n0);
      const uint32_t b0 = __builtin_amdgcn_readlane((int)(my_idx & 0xFFFFFFFF), n1), b1 = __builtin_amdgcn_readlane((int)(my_idx >> 32), n1);
      i0 = (int64_t)(((uint64_t)a1 << 32) | a0); i1 = (int64_t)(((uint64_t)b1 << 32) | b0);
    } else if (c.from_ring) { i0 = c.idx[n0]; i1 = c.idx[n1]; }
    else { i0 = n0; i1 = n1; }
    const int64_t unit = c.from_ring ? (int64_t)FRAME : (int64_t)STATE, off = c.from_ring ? -(int64_t)C0 * FRAME : 0;     // problems.h: sbase, z = 0
    auto group_org = [&](int k4u) -> int64_t {
      const int k4 = k4u < Kt - 4 ? k4u : Kt - 4;
      const int n = k4 / PIX1, pix = k4 - n * PIX1, y = pix / Q1, x = pix - y * Q1;
      return (n == n0 ? i0 : i1) * unit + off + (int64_t)(y * ST1) * W0 + x * ST1;
    };
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int64_t o0 = group_org(kc + 16 * st + 4 * g), o1 = group_org(kc + 16 * st + 8 + 4 * g);
        ra[2 * st + g] = *reinterpret_cast<const u32x4*>(c.src + (h ? o1 : o0) + colm);
      }
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = kc + 16 * st + 8 * h + e, kcl = k < Kt ? k : Kt - 1;
        const float v = c.d1[(uint32_t)(kcl * K1 + i)];
        rb[8 * st + e] = k < ke ? v : 0.0f;
      }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  int kc = kb + wave * 32;
  u32x4 ra[4]; float rb[16];
  if (kc < ke) load_chunk(kc, ra, rb);
  while (kc < ke) {
    u32x4 na[4]; float nb[16];
    const int kn = kc + 16 * 32;
    if (kn < ke) load_chunk(kn, na, nb);                                                    // next chunk's operands fly under this chunk's work
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      union { uint32_t u[4]; bf16x8_t v; } A;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const u32x4 w = ra[2 * st + g];
        const uint32_t f0 = __float_as_uint((float)(w.x & 255u)), f1 = __float_as_uint((float)(w.y & 255u));
        const uint32_t f2 = __float_as_uint((float)(w.z & 255u)), f3 = __float_as_uint((float)(w.w & 255u));
        A.u[2 * g] = __builtin_amdgcn_perm(f1, f0, 0x07060302u);
        A.u[2 * g + 1] = __builtin_amdgcn_perm(f3, f2, 0x07060302u);
      }
      bf16x8_t B0, B1, B2;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = rb[8 * st + e];
        const __bf16 hi = (__bf16)x; const float r1 = x - (float)hi;
        const __bf16 mid = (__bf16)r1; const float r2 = r1 - (float)mid;
        B0[e] = hi; B1[e] = mid; B2[e] = (__bf16)r2;
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, B0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, B1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, B2, acc, 0, 0, 0);
    }
    kc = kn;
    if (kc < ke) {
#pragma unroll
      for (int t = 0; t < 4; ++t) ra[t] = na[t];
#pragma unroll
      for (int t = 0; t < 16; ++t) rb[t] = nb[t];
    }
  }
  float* cw = smem + wave * PANEL;
#pragma unroll
  for (int r = 0; r < 16; ++r) cw[((r & 3) + 8 * (r >> 2) + 4 * h) * 33 + i] = acc[r];
  __syncthreads();
  {
    const int e = threadIdx.x, ml = e >> 5, nl = e & 31;
    float v = smem[ml * 33 + nl];
#pragma unroll
    for (int w = 1; w < 16; ++w) v += smem[w * PANEL + ml * 33 + nl];                       // fixed order
    float* dst = c.slab1 + (int64_t)ks * NW1 + (32 * bx + ml) * K1 + nl;
    if (c.pad_) wt_store(dst, div255(v)); else *dst = div255(v);
  }
}


// the three planes of one net's W1 from its fp32 weights (after set_weights / replica broadcast; the update kernel writes them itself)
__global__ void __launch_bounds__(256) w1_planes_kernel(const float* theta, unsigned short* w1p) {
  const int e = blockIdx.x * 256 + threadIdx.x;                       // e = k * 32 + n (W1i layout [(c,r,s)][map])
  if (e >= NW1) return;
  const int k = e >> 5, n = e & 31;
  uint16_t hi, mid, lo; split_bf16x3(theta[OFF1 + e], hi, mid, lo);
  w1p[n * CRS1 + k] = hi; w1p[W1P_PLANE + n * CRS1 + k] = mid; w1p[2 * W1P_PLANE + n * CRS1 + k] = lo;
}
hipError_t launch_w1_planes(const float* theta, unsigned short* w1p, hipStream_t s) {
  SDQN_LAUNCH(w1_planes_kernel, dim3(NW1 / 256), dim3(256), 0, s, theta, w1p);
  return hipGetLastError();
}


// ---- float16 mode: the same write-through epilogues (half outputs leave with 2-byte sc1 stores), every launch but bwd3 -----------------------------------
__device__ __forceinline__ void wt_store_h(half_t* p, half_t v) {
  union { half_t h; unsigned short u; } c; c.h = v;
  __hip_atomic_store(reinterpret_cast<unsigned short*>(p), c.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct Conv1FwdHWT : Conv1FwdH {
  static constexpr bool PRELOAD = SDQN_PRELOAD != 0;
  __device__ static void preload(const StepArgs& a, unsigned g0, unsigned g1, unsigned g2) { SDQN_TOUCH("s"(a.src), "s"(a.idx), "s"(a.h_a1), "s"(a.wht[0]), "s"(a.wht[1]), "s"(a.B), "s"(a.nz), "s"(a.from_ring), "s"(a.xcd_map), "s"(g0), "s"(g1), "s"(g2)); }

  __device__ static void store(const StepArgs& a, int z, int, int m, int n, float v) { wt_store_h(&a.h_a1[((int64_t)z * M(a) + m) * K1 + n], (half_t)fmaxf(v, 0.0f)); }
};
struct Conv2FwdHWT : Conv2FwdH {
  static constexpr bool PRELOAD = SDQN_PRELOAD != 0;
  __device__ static void preload(const StepArgs& a, unsigned g0, unsigned g1, unsigned g2) { SDQN_TOUCH("s"(a.h_a1), "s"(a.h_a2), "s"(a.wht[0]), "s"(a.wht[1]), "s"(a.B), "s"(a.nz), "s"(a.xcd_map), "s"(g0), "s"(g1), "s"(g2)); }

  __device__ static void store(const StepArgs& a, int z, int, int m, int n, float v) { wt_store_h(&a.h_a2[((int64_t)z * M(a) + m) * K2 + n], (half_t)fmaxf(v, 0.0f)); }
};
struct Conv3FwdHWT : Conv3FwdH {
  static constexpr bool PRELOAD = SDQN_PRELOAD != 0;
  __device__ static void preload(const StepArgs& a, unsigned g0, unsigned g1, unsigned g2) { SDQN_TOUCH("s"(a.h_a2), "s"(a.h_a3), "s"(a.wht[0]), "s"(a.wht[1]), "s"(a.B), "s"(a.nz), "s"(a.xcd_map), "s"(g0), "s"(g1), "s"(g2)); }

  __device__ static void store(const StepArgs& a, int z, int, int m, int n, float v) { wt_store_h(&a.h_a3[((int64_t)z * M(a) + m) * K3 + n], (half_t)fmaxf(v, 0.0f)); }
};
struct Fc4FwdHWT : Fc4FwdH {
  static constexpr bool PRELOAD = SDQN_PRELOAD != 0;
  __device__ static void preload(const StepArgs& a, unsigned g0, unsigned g1, unsigned g2) { SDQN_TOUCH("s"(a.h_a3), "s"(a.slab4), "s"(a.wht[0]), "s"(a.wht[1]), "s"(a.B), "s"(a.nz), "s"(a.S4), "s"(a.xcd_map), "s"(g0), "s"(g1), "s"(g2)); }

  __device__ static void store(const StepArgs& a, int z, int ks, int m, int n, float v) { wt_store(&a.slab4[(((int64_t)ks * 2 + z) * a.B + m) * NFC + n], v); }
};
struct Fc4DgradHWT : Fc4DgradH {
  static constexpr bool PRELOAD = SDQN_PRELOAD != 0;
  __device__ static void preload(const StepArgs& a, unsigned g0, unsigned g1, unsigned g2) { SDQN_TOUCH("s"(a.h_d4), "s"(a.wh[0]), "s"(a.h_a3), "s"(a.h_d3p), "s"(a.h_d3), "s"(a.B), "s"(a.xcd_map), "s"(g0), "s"(g1), "s"(g2)); }

  __device__ static void store(const StepArgs& a, int, int, int m, int n, float v) {
    int pix = n >> 6, f = n & 63, p = pix / Q3, q = pix - p * Q3;
    const half_t dv = (float)a.h_a3[(int64_t)m * NIN4 + n] > 0.0f ? (half_t)v : (half_t)0.0f;
    wt_store_h(&a.h_d3p[((m * PD3 + p + 2) * PD3 + q + 2) * K3 + f], dv);
    wt_store_h(&a.h_d3[(int64_t)m * NIN4 + n], dv);
  }
};
struct Conv2DgradHWT : Conv2DgradH {
  static constexpr bool PRELOAD_MULTI = SDQN_PRELOAD != 0;      // bwd2 of the float16 mode: conv2_dgrad + conv2_wgrad in one statement
  __device__ static void preload_multi(const StepArgs& a, const MultiDims& d) {
    SDQN_TOUCH("s"(a.h_d2p), "s"(a.wh[0]), "s"(a.h_a1), "s"(a.h_d1), "s"(a.h_d2), "s"(a.slab2), "s"(a.tps2), "s"(a.inv_loss_scale), "s"(a.B), "s"(a.xcd_map),
               "s"(d.n[0]), "s"(d.n[1]), "s"(d.gx[1]), "s"(d.gx[2]), "s"(d.gy[1]), "s"(d.gy[2]));
  }

  __device__ static void store(const StepArgs& a, int z, int, int m, int c, float v) {
    int py = z >> 1, px = z & 1;
    int n = m / 100, pix = m - n * 100, i = pix / 10, j = pix - i * 10;
    int o = ((n * P1 + 2 * i + py) * Q1 + 2 * j + px) * K1 + c;
    wt_store_h(&a.h_d1[o], (float)a.h_a1[o] > 0.0f ? (half_t)v : (half_t)0.0f);
  }
};
struct Conv2WgradHWWT : Conv2WgradHW {
  __device__ static void store(const StepArgs& a, int, int ks, int m, int n, float v) { wt_store(&a.slab2[(int64_t)ks * NW2 + m * K2 + n], v * a.inv_loss_scale); }
};
struct Conv1WgradHWWT : Conv1WgradHW {
  static constexpr bool PRELOAD_MULTI = SDQN_PRELOAD != 0;      // bwd1 of the float16 mode
  __device__ static void preload_multi(const StepArgs& a, const MultiDims& d) {
    SDQN_TOUCH("s"(a.src), "s"(a.idx), "s"(a.h_d1), "s"(a.slab1), "s"(a.tps1), "s"(a.inv_loss_scale), "s"(a.B), "s"(a.from_ring), "s"(a.xcd_map),
               "s"(d.n[0]), "s"(d.n[1]), "s"(d.gx[1]), "s"(d.gy[1]));
  }

  __device__ static void store(const StepArgs& a, int, int ks, int m, int n, float v) { wt_store(&a.slab1[(int64_t)ks * NW1 + m * K1 + n], v * a.inv_loss_scale); }
};
// ---- the pipeline with SPECIALISED waves: 10 matrix waves + NLW = 4 staging waves (one per SIMD) ------------------------------------
// In the first form of this kernel (conv1_bf16_rows_kernel, round 5; tools/exp/conv1_forms_r5.hip.txt) every wave did both halves of a trip — staging (last item's output out, next item's bytes converted into
// LDS, the load five items ahead) and matrix work — so its MFMAs start behind a vmcnt wait and the ten waves reach their MFMAs together
// (1.1 us per item where the matrix pipe needs 0.5).  Here the staging waves (10 and up) do ALL the global traffic (2 loads + 3 stores per
// thread and item with four of them) and the conversion; waves 0..9 only read fragments from LDS, run their 24 MFMAs and write the output image: no matrix wave ever
// waits on vmcnt.  One barrier per item, the same double buffers, the same arithmetic in the same order: bit-identical to the 10-wave kernel.
// 15.3 -> 13.5 us at B = 256 (same box; two staging waves: 14.2-14.8 — the conversion on 128 threads is then the critical path).  (Staging TWO items ahead into three image buffers, so that a matrix wave reads the next item's
// fragments under this item's MFMAs: 15.1 us — slower; all eight fragments ahead do not fit 168 registers beside the planes.)
template <int NLW>                                                 // staging waves (2 or 4)
__global__ void __launch_bounds__(640 + 64 * NLW) conv1_bf16_rows2_kernel(const Conv1Args c) {
  constexpr int C1S_LOADERS = 64 * NLW, NT = 640 + 64 * NLW;
  constexpr int C1S_LPT = (C1R_CHUNKS + C1S_LOADERS - 1) / C1S_LOADERS;   // 16-byte input pieces per staging thread and item (420 pieces)
  constexpr int C1S_FPT = (640 + C1S_LOADERS - 1) / C1S_LOADERS;          // 16-byte output pieces per staging thread and item (640 pieces)
  __shared__ __attribute__((aligned(16))) unsigned short img[2 * C1R_ITEM];                   // 26 880 B
  __shared__ __attribute__((aligned(16))) unsigned short sw[3 * K1 * W1P_PITCH];              // 50 688 B (used once, before the loop)
  __shared__ __attribute__((aligned(16))) float outl[2 * C1R_OUT];                           // 23 040 B: an item's output, double-buffered
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Gz = c.wgs_per_net;
  const int z = (int)blockIdx.x / Gz, g = (int)blockIdx.x - z * Gz;
  const int nsamp = (c.B - g + Gz - 1) / Gz, nitems = C1R_NCH * nsamp;
  int64_t my_idx = 0;
  if (c.from_ring) { const int nn = g + Gz * lane; my_idx = c.idx[nn < c.B ? nn : c.B - 1]; }
  // W1's planes: one coalesced copy per workgroup into LDS (all 768 threads, 4 pieces each)
  {
    const c1p_u32x4* wp = reinterpret_cast<const c1p_u32x4*>(c.w1p[z]);
    constexpr int NPC = 3 * K1 * CRS1 / 8, PPT = (NPC + NT - 1) / NT;
    c1p_u32x4 wv[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) { const int cc = tid + NT * j; wv[j] = wp[cc < NPC ? cc : NPC - 1]; }
#pragma unroll
    for (int j = 0; j < PPT; ++j) { const int cc = tid + NT * j, row = cc >> 5, col = cc & 31; if (cc < NPC) *reinterpret_cast<c1p_u32x4*>(sw + row * W1P_PITCH + col * 8) = wv[j]; }
  }
  if (wave >= 10) {
    // ================= staging waves =================
    const int lid = tid - 640;
    auto org_of = [&](int si) -> int64_t {
      if (!c.from_ring) return ((int64_t)z * c.B + g + (int64_t)si * Gz) * (int64_t)STATE;
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_idx, si), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_idx >> 32), si);
      return ((int64_t)(((uint64_t)hi << 32) | lo) - C0 + z) * (int64_t)FRAME;
    };
    int poff[C1S_LPT], pdst[C1S_LPT];                         // piece j of this thread: byte offset inside a sample's item window / LDS element
#pragma unroll
    for (int j = 0; j < C1S_LPT; ++j) {
      const int pc = lid + C1S_LOADERS * j, pcc = pc < C1R_CHUNKS ? pc : C1R_CHUNKS - 1;
      const int fc = pcc / (C1R_SEG / 16), pw = pcc - fc * (C1R_SEG / 16);
      poff[j] = fc * FRAME + 16 * pw; pdst[j] = pc < C1R_CHUNKS ? fc * C1R_SEG + 16 * pw : -1;
    }
    auto gload = [&](int item, c1p_u32x4 (&fv)[C1S_LPT]) {    // (clamped, unconditional: a guarded load costs a vmcnt(0))
      const int ic = item < nitems ? item : nitems - 1;
      const int si = ic / C1R_NCH, ch = ic - si * C1R_NCH;
      const uint8_t* base = c.src + org_of(si) + ch * (4 * ST1 * W0);
#pragma unroll
      for (int j = 0; j < C1S_LPT; ++j) fv[j] = *reinterpret_cast<const c1p_u32x4*>(base + poff[j]);
    };
    auto lstore = [&](const c1p_u32x4 (&fv)[C1S_LPT], unsigned short* dst) {      // 16 bytes -> 16 bf16 (the upper half of the float of an 8-bit integer)
#pragma unroll
      for (int j = 0; j < C1S_LPT; ++j) {
        if (pdst[j] < 0) continue;
        c1p_u32x4 o[2];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t w = fv[j][e];
          const uint32_t f0 = __float_as_uint((float)(w & 255u)), f1 = __float_as_uint((float)((w >> 8) & 255u));
          const uint32_t f2 = __float_as_uint((float)((w >> 16) & 255u)), f3 = __float_as_uint((float)(w >> 24));
          o[e >> 1][2 * (e & 1)] = __builtin_amdgcn_perm(f1, f0, 0x07060302u);
          o[e >> 1][2 * (e & 1) + 1] = __builtin_amdgcn_perm(f3, f2, 0x07060302u);
        }
        c1p_u32x4* d = reinterpret_cast<c1p_u32x4*>(dst + pdst[j]);
        d[0] = o[0]; d[1] = o[1];
      }
    };
    auto flush = [&](int item) {                              // 640 sixteen-byte pieces of the item's output: 5 per staging thread, whole lines in order
      const int si = item / C1R_NCH, ch = item - si * C1R_NCH;
      float* outs = c.a1 + (((int64_t)z * c.B + g + (int64_t)si * Gz) * PIX1 + ch * (4 * Q1)) * K1;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)outs, 0, 4 * Q1 * K1 * 4, 0x00020000);
#pragma unroll
      for (int u = 0; u < C1S_FPT; ++u) {
        const int pc = lid + C1S_LOADERS * u;
        if (pc >= 640) continue;
        const c1p_f32x4 v = *reinterpret_cast<const c1p_f32x4*>(outl + (item & 1) * C1R_OUT + (pc >> 3) * C1R_OPITCH + 4 * (pc & 7));
        c1p_u32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(v[e]);
        if (c.pad_) __builtin_amdgcn_raw_buffer_store_b128(w, rs, 16 * pc, 0, 16); else __builtin_amdgcn_raw_buffer_store_b128(w, rs, 16 * pc, 0, 0);
      }
    };
    c1p_u32x4 fv[C1R_D][C1S_LPT];
#pragma unroll
    for (int u = 0; u < C1R_D; ++u) gload(u, fv[u]);
    lstore(fv[0], img);
    gload(C1R_D, fv[0]);
    __syncthreads();                                          // planes + item 0 in LDS
    for (int i0 = 0; i0 < nitems; i0 += C1R_D) {
#pragma unroll
      for (int u = 0; u < C1R_D; ++u) {
        const int item = i0 + u;
        if (item > 0) flush(item - 1);
        lstore(fv[(u + 1) % C1R_D], img + ((item + 1) & 1) * C1R_ITEM);
        gload(item + 1 + C1R_D, fv[(u + 1) % C1R_D]);
        __syncthreads();
      }
    }
    if (nitems > 0) flush(nitems - 1);
    return;
  }
  // ================= matrix waves =================
  const int tj = wave >> 1, nh = wave & 1;
  __syncthreads();                                            // planes + item 0 in LDS
  bf16x8_t Bf[3][8];
  {
    const unsigned short* bw = sw + (nh * 16 + m) * W1P_PITCH + 8 * kg;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int t = 0; t < 8; ++t) Bf[p][t] = *reinterpret_cast<const bf16x8_t*>(bw + p * K1 * W1P_PITCH + 32 * t);
  }
  const int pos = 16 * tj + m, pl = pos / Q1, q = pos - pl * Q1;
  const int a_off = (ST1 * pl + kg) * W0 + ST1 * q;
  for (int item = 0; item < nitems; ++item) {
    const unsigned short* cur = img + (item & 1) * C1R_ITEM + a_off;
    bf16x8_t A[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const unsigned short* ap = cur + (t >> 1) * C1R_SEG + 4 * (t & 1) * W0;
      union { c1p_u32x2 u[2]; bf16x8_t v; } X;
      X.u[0] = *reinterpret_cast<const c1p_u32x2*>(ap); X.u[1] = *reinterpret_cast<const c1p_u32x2*>(ap + 4);
      A[t] = X.v;
    }
    c1p_f32x4 acc[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) acc[p] = c1p_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bf[p][t], A[t], acc[p], 0, 0, 0);
    c1p_f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(div255((acc[0][e] + acc[1][e]) + acc[2][e]), 0.0f);
    *reinterpret_cast<c1p_f32x4*>(outl + (item & 1) * C1R_OUT + pos * C1R_OPITCH + nh * 16 + 4 * kg) = v;
    __syncthreads();
  }
}

hipError_t launch_kernel_r3(int id, const StepArgs& a, const LaunchTune& t, hipStream_t s, bool* handled) {
  *handled = true;
  if (id == K_CONV1_FWD && (t.r3 & 4) && !a.h16 && !a.bn && a.w1p[0] && a.w1p[a.nz > 1 ? 1 : 0]) {
    const int tiles = (a.B * PIX1 + 31) / 32, tpw = a.B >= 128 ? 4 : 1, wgs = (tiles + 4 * tpw - 1) / (4 * tpw);
    Conv1Args c; c.src = a.src; c.a1 = a.a1; c.w1p[0] = a.w1p[0]; c.w1p[1] = a.w1p[1]; c.idx = a.idx;
    c.B = a.B; c.nz = a.nz; c.from_ring = a.from_ring; c.tiles_per_net = tiles; c.wgs_per_net = wgs; c.tpw = tpw; c.xcd = t.r3_xcd & 1; c.pad_ = (t.wt >> 7) & 1;
    static_assert(sizeof(Conv1Args) == 72, "the index block follows 8-byte aligned at byte 72");
    if (a.B >= 128 && t.bt[K_CONV1_FWD] >= 0) {           // throughput regime  (option bt:0 = -1: the per-tile kernel, the test reference)
      // round 5: persistent workgroups (one per CU), planes in registers, frames streamed in 4-row items; every workgroup of a net the same
      // number of samples where that is possible: Gz = ceil(B / ceil(B / (256 / nz)))
      // (rounds 4-5's other forms — one workgroup per sample with the frames staged, and the 10-wave pipeline without specialised
      //  waves, 15.9 / 14.5 us against 12.4-13.5 at B = 256 — left the tree in round 6: tools/exp/conv1_forms_r5.hip.txt)
      const int cap = 256 / (a.nz > 1 ? 2 : 1), per = (a.B + cap - 1) / cap;
      c.wgs_per_net = (a.B + per - 1) / per;
      SDQN_LAUNCH(conv1_bf16_rows2_kernel<4>, dim3(a.nz * c.wgs_per_net), dim3(896), 0, s, c);      // (staging waves 2 / 4 / 6: 14.8 / 13.5 / 13.4 us)
      return hipGetLastError();
    }
    IdxIn ix;
    if (t.host_idx && a.from_ring && a.B <= 32) {
      memset(ix.v, 0, sizeof ix.v);
      memcpy(ix.v, t.host_idx, (size_t)a.B * sizeof(int64_t));
      SDQN_LAUNCH(conv1_bf16_kernel<true>, dim3(a.nz * wgs), dim3(256), 0, s, c, ix);
    } else {
      memset(ix.v, 0, sizeof ix.v);
      SDQN_LAUNCH(conv1_bf16_kernel<false>, dim3(a.nz * wgs), dim3(256), 0, s, c, ix);
    }
    return hipGetLastError();
  }
  if ((id == K_BWD1 || id == K_CONV1_WGRAD) && (t.r3 & 8) && (id == K_CONV1_WGRAD || a.f4w_count == 0) && !a.h16 && !a.bn) {
    C1wArgs c; c.src = a.src; c.d1 = a.d1; c.slab1 = a.slab1; c.idx = a.idx; c.B = a.B; c.from_ring = a.from_ring; c.tps1 = a.tps1; c.Kt = a.B * PIX1; c.xcd = (t.r3_xcd >> 1) & 1; c.pad_ = (t.wt >> 6) & 1;
    static_assert(sizeof(C1wArgs) == 56, "the index block follows at byte 56 of the argument segment");
    const dim3 grid(CRS1 / 32, 1, Conv1Wgrad::nbz(a));
    IdxIn ix; memset(ix.v, 0, sizeof ix.v);
    if (t.host_idx && a.from_ring && a.B <= 32) {
      memcpy(ix.v, t.host_idx, (size_t)a.B * sizeof(int64_t));
      SDQN_LAUNCH(conv1_wgrad_bf16_kernel<true>, grid, dim3(1024), 0, s, c, ix);
    } else SDQN_LAUNCH(conv1_wgrad_bf16_kernel<false>, grid, dim3(1024), 0, s, c, ix);
    return hipGetLastError();
  }
  // conv2 / conv3 forward with ONE workgroup per 32 x 64 output block (N = 64 = two 32-wide tiles): the register-blocked routine with
  // 1 x 2 accumulators per wave loads the gathered A rows once for both tiles (the gather is the expensive operand: 64 cache lines per
  // load instruction).  Same k order per accumulator as the unblocked tile: bit-identical.
  if (a.B <= 32 && a.h16 == 2 && !a.bn && t.wt && !(id >= 0 && id < 12 && t.nw_override[id] > 0)) {      // float16 mode, default launch forms
    if (id == K_CONV1_FWD && (t.wt & 128)) return launch_gemm<Conv1FwdHWT, 8>(a, s);
    if (id == K_CONV2_FWD && (t.wt & 1)) return launch_gemm<Conv2FwdHWT, 16>(a, s);
    if (id == K_CONV3_FWD && (t.wt & 2)) return launch_gemm<Conv3FwdHWT, 16>(a, s);
    if (id == K_FC4_FWD && (t.wt & 4)) return launch_gemm<Fc4FwdHWT, 14>(a, s);
    if (id == K_FC4_DGRAD && (t.wt & 8)) return launch_gemm<Fc4DgradHWT, 16>(a, s);
    // (bwd3 stays on plain stores in this mode: 16 404 vs 16 445 steps/s alone, 16 775 vs 16 809 with the others — tools/exp/README.md)
    if (id == K_BWD2 && (t.wt & 32)) return launch_multi<512, NoProblem, 2, Conv2DgradHWT, 8, Conv2WgradHWWT, 8>(a, true, true, s);
    if (id == K_BWD1 && (t.wt & 64)) return launch_multi<1024, NoProblem, 2, Conv1WgradHWWT, 16, NoProblem, 2>(a, true, false, s);
  }
  if (a.B <= 32 && !a.h16 && !a.bn && t.wt && !(id >= 0 && id < 12 && t.nw_override[id] > 0)) {       // write-through epilogues: the default launch forms with the *WT problems
    if (id == K_CONV2_FWD && (t.wt & 1) && !(t.r3 & 16)) return launch_gemm<Conv2FwdWT, 16>(a, s);
    if (id == K_FC4_FWD && (t.wt & 4)) return launch_gemm<Staged<Fc4FwdWT>, 14>(a, s);
    if (id == K_FC4_DGRAD && (t.wt & 8) && !(t.r3 & 1)) return launch_gemm<Staged<Fc4DgradWT>, 16>(a, s);
    if (id == K_BWD3 && (t.wt & 16) && a.f4w_count > 0) return launch_multi<512, Staged<Conv3DgradWT>, 8, Conv3WgradWT, 8, Fc4WgradWT, 1>(a, true, true, s);
    if (id == K_BWD2 && (t.wt & 32) && a.f4w_count == 0) return launch_multi<512, NoProblem, 2, Conv2DgradWT, 8, Conv2WgradWT, 8>(a, true, true, s);
  }
  if (a.B >= 128 && !a.h16 && !a.bn && t.wt && !(id >= 0 && id < 12 && t.nw_override[id] > 0)) {
    // throughput regime: the same launch forms as sdqn_kernels.hip, write-through epilogues
    if (id == K_CONV2_FWD && (t.wt & 1)) return launch_gemm<Staged<Conv2FwdWT>, 8>(a, s);
    if (id == K_CONV3_FWD && (t.wt & 2)) return launch_gemm<Staged<Conv3FwdWT>, 8>(a, s);
    if (id == K_FC4_FWD && (t.wt & 4)) return launch_gemm<Staged<Fc4FwdWT>, 8>(a, s);
    if (id == K_FC4_DGRAD && (t.wt & 8)) return launch_gemm<Staged<Fc4DgradWT>, 4>(a, s);
    if (id == K_BWD3 && (t.wt & 16) && a.f4w_count > 0) return launch_multi<512, Fc4WgradWT, 8, Staged<Conv3DgradWT>, 8, Conv3WgradWT, 8>(a, true, true, s);
    if (id == K_BWD2 && (t.wt & 32) && a.f4w_count == 0) return launch_multi<512, NoProblem, 2, Staged<Conv2DgradWT>, 8, Conv2WgradWT, 8>(a, true, true, s);
    if (id == K_BWD1 && (t.wt & 64) && a.f4w_count == 0 && !(t.r3 & 8)) return launch_multi<1024, NoProblem, 2, Conv1WgradWT, 16, NoProblem, 2>(a, true, false, s);
  }
  if (id == K_CONV3_FWD && (t.r3 & 2) && a.B < 128 && !a.h16 && !a.bn) {
    static_assert(CRS3 == 16 * 36, "conv3's K is 16 chunks of 36");
    const dim3 grid((Conv3Fwd::M(a) + 31) / 32, (Conv3Fwd::N(a) + 31) / 32, Conv3Fwd::nbz(a));
    if ((t.wt & 2) && a.B <= 32) SDQN_LAUNCH((gemm36_kernel<Conv3FwdWT>), grid, dim3(1024), 0, s, a);
    else SDQN_LAUNCH((gemm36_kernel<Conv3Fwd>), grid, dim3(1024), 0, s, a);
    return hipGetLastError();
  }
  *handled = false;
  return hipSuccess;
}

#ifdef SDQN_TIMING
hipError_t set_timing_buffer_r3(unsigned long long* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_sdqn_dbg), &p, sizeof p); }
hipError_t set_wave_timing_buffer_r3(unsigned long long* const* p, const unsigned* nb, hipStream_t s) {
  hipError_t e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_sdqn_wdbg_blocks), nb, sizeof *nb, 0, hipMemcpyHostToDevice, s);
  return e != hipSuccess ? e : hipMemcpyToSymbolAsync(HIP_SYMBOL(g_sdqn_wdbg), p, sizeof *p, 0, hipMemcpyHostToDevice, s);
}
#endif

}  // namespace sdqn
