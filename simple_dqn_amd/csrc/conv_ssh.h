// conv_ssh.h — float16 mode: the convolution layers as sample-stationary CHAINS, one launch each:
//   forward   conv1 -> conv2 -> conv3 (deepqnetwork.py:83-87 with the replay gather and the / 255 of :94-100; the online and the target net of
//             :119-130 together), at every batch size                                                      conv_ssh_chain_kernel
//   backward  conv3_dgrad -> conv2_dgrad (deepqnetwork.py:162), B >= 128                                    conv_ssh_dgrad_chain_kernel
//
// The packed-fp16 block-tile routines run these two layers in 9.1 + 7.4 us at B = 256 for 1.1 + 0.8 us of matrix time: they are data
// movement (every 64 x 64 block re-fetches its patches, expanded 4x / 9x by im2col, and the whole weight panel) plus two launch boundaries.
// Here one workgroup per CU owns NS whole samples of one net (conv1 in front: template flag C1, described at its code):
//   * its input maps ([20][20][32] halves per sample, contiguous), W2^T ([64 maps][512 k] halves) and — into REGISTERS, for later — W3^T
//     ([64][576]) are requested once, up front, with 16-byte loads; image and W2 go to LDS (pixel pitch 40 halves, weight row pitch 592:
//     a tile's 16 positions / maps are 160 bytes apart modulo 256, conflict-free ds_read_b128);
//   * conv2 = v_mfma_f32_16x16x32_f16 with im2col at ds_read time: one MFMA step is one kernel tap (r, s) x 32 channels; lane (m, kq) reads
//     the 8 channels 8 kq .. + 7 of pixel (2 p + r, 2 q + s) with one ds_read_b128; the weights are the ROW operand (D[map][position]: a lane
//     holds 4 consecutive maps of a position).  Eight waves: wave = (map pair, position group): 2 map tiles x 3 position tiles = 6
//     accumulators, 5 fragment reads per 6 MFMAs on all eight waves (all 64 maps x 3 tiles on four waves — 27 fragment reads per step
//     instead of 38 — is no faster at B = 256 and 10 % slower with one sample per workgroup: two waves per SIMD hide each other's LDS latency);
//   * behind a barrier the Rectlin'ed half outputs are written into the conv3 image (pixel pitch 80 halves) over the dead conv2 image, W3
//     comes out of the registers over W2, and a2 leaves for the backward pass as whole lines copied from the LDS image;
//   * conv3 the same way (two MFMA steps per tap: 64 channels), a3 collected in LDS and stored as whole lines.
// a2 is never read back from memory by the forward pass.  Arithmetic: half operands, fp32 accumulation, k ascending in the order (r, s, c)
// in ONE accumulator per output — the same sums as the block-tile routine's, in another order of the same fp32 additions.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "launch.h"

namespace sdqn {
namespace ssh {

typedef _Float16 h_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NT_ = 512;                       // threads: eight waves, all of them load, compute and store
constexpr int NO = 64;                         // output maps of both layers
constexpr int P1 = 40;                         // halves per pixel of the conv2 image (32 + 8)
constexpr int P2 = 80;                         // halves per pixel of the conv3 image / the a3 collection (64 + 16)
constexpr int KP = 592;                        // halves per map row of a staged weight matrix (512 / 576 + padding: 1184 bytes = 160 mod 256)
constexpr int PX1 = 400, PX2 = 81, PX3 = 49;   // pixels per sample: a1, a2, a3
constexpr int KK2 = 512, KK3 = 576;            // k per output: conv2, conv3

struct Args {
  const h_t* a1;            // [nz][B][400][32]
  const h_t* w2[2];         // per net: W2^T [64][512]   (k = (r, s, c))
  const h_t* w3[2];         // per net: W3^T [64][576]
  h_t* a2;                  // [nz][B][81][64]
  h_t* a3;                  // [nz][B][49][64]
  int B, G;                 // G = workgroups per net = ceil(B / NS)
  // C1 (conv1 rides in front: the workgroup computes its own a1 from the frames): the replay ring / the staged states, the sampled
  // indexes, W1^T [32 maps][256 k] per net, and where a1 goes for the backward pass
  const uint8_t* src; const int64_t* idx; const h_t* w1[2]; h_t* a1w; int from_ring;
};

template <int NS> struct Lds {
  static constexpr int R1 = NS * PX1 * P1;                   // conv2 image; later: conv3 image (NS * 81 * 80) + a3 collection (NS * 49 * 80)
  static_assert(NS * PX2 * P2 + NS * PX3 * P2 <= R1, "conv3 image and a3 collection fit the dead conv2 image");
  static constexpr int RW = NO * KP;
  static constexpr int TOTAL = R1 + RW;                      // NS = 2: 32 000 + 37 888 halves = 139 776 bytes
  static_assert(TOTAL * 2 <= 160 * 1024, "LDS budget");
  static_assert(4 * 84 * 84 + 32 * 264 <= RW, "conv1's frames (as halves) + W1 stage in the weight region before W2 arrives");
};

// one layer's matrix work for this wave: NPT position tiles x 2 map tiles; A fragments from `img` (lane base apos[t], + the tap's immediate),
// weight fragments from `wl`.  STEPS k-steps of 32; step -> (tap, channel half) -> immediate offsets.
template <int NPT, int STEPS, int CI, int S, int WI, int PITCH>
__device__ __forceinline__ void mm(const h_t* img, const int (&apos)[3], const h_t* wb, f32x4 (&acc)[2][3]) {
#if defined(SSH_ABL) && SSH_ABL == 1                         // (tools/exp/ssh_bench.hip: the launch without its matrix work: 9.4 -> 5.1 us at B = 256)
  return;
#endif
  constexpr int SPT = CI / 32;                               // MFMA steps per tap
#pragma unroll
  for (int st = 0; st < STEPS; ++st) {
    const int tap = st / SPT, hf = st - tap * SPT, r = tap / S, s = tap - r * S;
    const int aoff = (r * WI + s) * PITCH + 32 * hf;
    const h8 w0 = *reinterpret_cast<const h8*>(wb + 32 * st), w1 = *reinterpret_cast<const h8*>(wb + 16 * KP + 32 * st);
#pragma unroll
    for (int t = 0; t < NPT; ++t) {
      const h8 x = *reinterpret_cast<const h8*>(img + apos[t] + aoff);
      acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, x, acc[0][t], 0, 0, 0);
      acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, x, acc[1][t], 0, 0, 0);
    }
  }
}

template <int NS, bool WT, bool C1>
__global__ void __launch_bounds__(NT_) conv_ssh_chain_kernel(const Args c) {
  typedef Lds<NS> L;
  __shared__ __attribute__((aligned(16))) h_t smem[L::TOTAL];
  h_t* const r1 = smem;
  h_t* const rw = smem + L::R1;
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int z = (int)blockIdx.x / c.G, g = (int)blockIdx.x - z * c.G;
  const int n0 = g * NS;
  const int nvalid = c.B - n0 < NS ? c.B - n0 : NS;          // (the last workgroup of an odd batch has one sample)

  // ---- every global read of the launch, up front: image, W2 (-> LDS now), W3 (-> registers, into LDS behind conv2) ----
  constexpr int IPC = NS * PX1 * 4, IPT = (IPC + NT_ - 1) / NT_;            // 16-byte pieces of the image / per thread
  constexpr int W2P = NO * KK2 / 8 / NT_, W3P = NO * KK3 / 8 / NT_;         // 8, 9 pieces per thread
  static_assert(W2P * NT_ * 8 == NO * KK2 && W3P * NT_ * 8 == NO * KK3, "whole weight pieces per thread");
  const u32x4* const w2p = reinterpret_cast<const u32x4*>(c.w2[z]);
  const u32x4* const w3p = reinterpret_cast<const u32x4*>(c.w3[z]);
  u32x4 w2v[W2P], w3v[W3P];
  if constexpr (C1) {
    // ---- conv1 in front (deepqnetwork.py:83 + the gather and the / 255 of :94-100): per sample, the state's four frames (28 224 contiguous
    // bytes of the ring) -> exact halves (0x6400 | b = half(1024 + b), - 1024) in the weight region, W1^T beside them; 25 tiles of 16
    // positions over the eight waves, both 16-map tiles per fragment, v_mfma_f32_16x16x32_f16 (one step = 4 kernel rows x 8 columns of one
    // frame); Rectlin(sum / 255) as half straight into the conv2 image.  W2 / W3 wait in registers meanwhile.
    constexpr int FPC = 4 * 84 * 84 / 16, FPT = (FPC + NT_ - 1) / NT_;      // 1 764 pieces of 16 bytes: 4 per thread, the last partly
    h_t* const ci = rw;                                                      // [frame][84][84] halves (the state's own byte order widened)
    h_t* const cw = rw + 4 * 84 * 84;                                        // W1^T [32][256 + 8]
    auto frames = [&](int sI, u32x4 (&fv)[FPT]) {
      const int nn = n0 + (sI < nvalid ? sI : nvalid - 1);
      const int64_t fb = c.from_ring ? (c.idx[nn] - 4 + z) * (int64_t)(84 * 84) : ((int64_t)z * c.B + nn) * (int64_t)(4 * 84 * 84);      // problems.h: sbase
      const u32x4* const fp = reinterpret_cast<const u32x4*>(c.src + fb);
#pragma unroll
      for (int j = 0; j < FPT; ++j) { const int it = tid + NT_ * j; fv[j] = fp[it < FPC ? it : FPC - 1]; }
    };
    u32x4 fv[2][FPT], w1v[2];
    frames(0, fv[0]);
#pragma unroll
    for (int j = 0; j < 2; ++j) w1v[j] = reinterpret_cast<const u32x4*>(c.w1[z])[tid + NT_ * j];
    if constexpr (NS > 1) frames(1, fv[1]);
#pragma unroll
    for (int j = 0; j < W2P; ++j) w2v[j] = w2p[tid + NT_ * j];
#pragma unroll
    for (int j = 0; j < W3P; ++j) w3v[j] = w3p[tid + NT_ * j];
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int it = tid + NT_ * j; *reinterpret_cast<u32x4*>(cw + (it >> 5) * 264 + 8 * (it & 31)) = w1v[j]; }
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 m1024 = {(h_t)-1024.0f, (h_t)-1024.0f};
    const h_t* const wb1 = cw + m * 264 + 8 * kq;                            // weights: map m (and 16 + m), k = 32 st + 8 kq ..
#pragma unroll
    for (int sI = 0; sI < NS; ++sI) {
      if (sI > 0) __syncthreads();                                           // the previous sample's fragments are read
#pragma unroll
      for (int j = 0; j < FPT; ++j) {
        const int it = tid + NT_ * j;
        if (it < FPC) {
          u32x4 o[2];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned w = fv[sI][j][e];
            union { unsigned u; h2 h; } lo, hi;
            lo.u = __builtin_amdgcn_perm(0x64646464u, w, 0x04010400u); hi.u = __builtin_amdgcn_perm(0x64646464u, w, 0x04030402u);
            lo.h = lo.h + m1024; hi.h = hi.h + m1024;
            o[e >> 1][2 * (e & 1)] = lo.u; o[e >> 1][2 * (e & 1) + 1] = hi.u;
          }
          u32x4* const d = reinterpret_cast<u32x4*>(ci + 16 * it);
          d[0] = o[0]; d[1] = o[1];
        }
      }
      __syncthreads();
      h8 fw[2][8];
#pragma unroll
      for (int st = 0; st < 8; ++st) { fw[0][st] = *reinterpret_cast<const h8*>(wb1 + 32 * st); fw[1][st] = *reinterpret_cast<const h8*>(wb1 + 16 * 264 + 32 * st); }
      for (int rt = wave; rt < PX1 / 16; rt += 8) {
        const int pos = 16 * rt + m, p = pos / 20, q = pos - p * 20;
        const h_t* const pa = ci + (4 * p + kq) * 84 + 4 * q;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 8; ++st) {
          const h_t* const ap = pa + (st >> 1) * (84 * 84) + 4 * (st & 1) * 84;
          const h4 xl = *reinterpret_cast<const h4*>(ap), xh = *reinterpret_cast<const h4*>(ap + 4);
          h8 x; x[0] = xl[0]; x[1] = xl[1]; x[2] = xl[2]; x[3] = xl[3]; x[4] = xh[0]; x[5] = xh[1]; x[6] = xh[2]; x[7] = xh[3];
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[0][st], x, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[1][st], x, a1, 0, 0, 0);
        }
        h4 v0, v1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v0[e] = (h_t)fmaxf(a0[e] * (1.0f / 255.0f), 0.0f); v1[e] = (h_t)fmaxf(a1[e] * (1.0f / 255.0f), 0.0f); }
        h_t* const o = r1 + (sI * PX1 + pos) * P1 + 4 * kq;
        *reinterpret_cast<h4*>(o) = v0;
        *reinterpret_cast<h4*>(o + 16) = v1;
      }
    }
    __syncthreads();                                                         // a1 complete in the conv2 image; conv1's staging is dead
#pragma unroll
    for (int j = 0; j < W2P; ++j) { const int pc = tid + NT_ * j; *reinterpret_cast<u32x4*>(rw + (pc >> 6) * KP + 8 * (pc & 63)) = w2v[j]; }
    {                                                                        // a1 leaves for the backward pass: nvalid x 400 rows of 64 bytes, whole lines
      h_t* const o1 = c.a1w + ((int64_t)z * c.B + n0) * (PX1 * 32);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)o1, 0, nvalid * PX1 * 32 * 2, 0x00020000);
#pragma unroll
      for (int j = 0; j < IPT; ++j) {
        const int pc = tid + NT_ * j;
        if (pc < IPC) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(r1 + (pc >> 2) * P1 + 8 * (pc & 3));
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, 16 * pc, 0, WT ? 16 : 0);
        }
      }
    }
    __syncthreads();
  } else {
  const u32x4* const ip = reinterpret_cast<const u32x4*>(c.a1 + ((int64_t)z * c.B + n0) * (PX1 * 32));
  u32x4 iv[IPT];
  const int ipmax = nvalid * PX1 * 4 - 1;
#pragma unroll
  for (int j = 0; j < W2P; ++j) w2v[j] = w2p[tid + NT_ * j];
#pragma unroll
  for (int j = 0; j < IPT; ++j) { const int pc = tid + NT_ * j; iv[j] = ip[pc < ipmax ? pc : ipmax]; }
#pragma unroll
  for (int j = 0; j < W3P; ++j) w3v[j] = w3p[tid + NT_ * j];
#pragma unroll
  for (int j = 0; j < W2P; ++j) { const int pc = tid + NT_ * j; *reinterpret_cast<u32x4*>(rw + (pc >> 6) * KP + 8 * (pc & 63)) = w2v[j]; }
#pragma unroll
  for (int j = 0; j < IPT; ++j) { const int pc = tid + NT_ * j; if (pc < IPC) *reinterpret_cast<u32x4*>(r1 + (pc >> 2) * P1 + 8 * (pc & 3)) = iv[j]; }
  __syncthreads();
  }

  // ---- conv2: 16 taps x 32 channels.  wave = (map pair mp, position group pg): maps 32 mp .. + 31, position tiles TPG pg .. + TPG - 1 ----
  const int mp = wave & 1, pg = wave >> 1;
  const h_t* const wb = rw + (32 * mp + m) * KP + 8 * kq;
  f32x4 acc[2][3];
  int apos[3];
  {
    constexpr int NPOS = NS * PX2, NTL = (NPOS + 15) / 16, TPG = (NTL + 3) / 4;    // 162 positions: 11 tiles, 3 per group (NS = 1: 81, 6 tiles, 2)
    static_assert(TPG <= 3, "accumulators");
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
      int P = 16 * (TPG * pg + t) + m; if (P > NPOS - 1) P = NPOS - 1;      // (padding rows of the last tile: a valid pixel nobody stores)
      const int sp = P / PX2, rem = P - sp * PX2, p = rem / 9, q = rem - p * 9;
      apos[t] = (sp * PX1 + (2 * p) * 20 + 2 * q) * P1 + 8 * kq;
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    {
      const int nt = NTL - TPG * pg;                                         // this wave's tiles (wave-uniform)
      if (nt >= 3 && TPG >= 3) mm<3, 16, 32, 4, 20, P1>(r1, apos, wb, acc);
      else if (nt >= 2 && TPG >= 2) mm<2, 16, 32, 4, 20, P1>(r1, apos, wb, acc);
      else if (nt >= 1) mm<1, 16, 32, 4, 20, P1>(r1, apos, wb, acc);
    }
    __syncthreads();                                                         // every read of the conv2 image and of W2 is done
    // a2 = Rectlin(.) as half into the conv3 image (over the conv2 image): D[map 4 kq + e][position m]
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
      const int P = 16 * (TPG * pg + t) + m;
      if (P < NPOS) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          h4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (h_t)fmaxf(acc[u][t][e], 0.0f);
          *reinterpret_cast<h4*>(r1 + P * P2 + 32 * mp + 16 * u + 4 * kq) = v;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < W3P; ++j) { const int pc = tid + NT_ * j, n = pc / 72, k8 = pc - n * 72; *reinterpret_cast<u32x4*>(rw + n * KP + 8 * k8) = w3v[j]; }
    __syncthreads();
    // a2 leaves for the backward pass: nvalid x 81 rows of 128 bytes, whole lines from the LDS image
    {
      h_t* const o2 = c.a2 + ((int64_t)z * c.B + n0) * (PX2 * NO);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)o2, 0, nvalid * PX2 * NO * 2, 0x00020000);
      constexpr int OPC = NPOS * 8;                                          // 16-byte pieces
#pragma unroll
      for (int j = 0; j < (OPC + NT_ - 1) / NT_; ++j) {
        const int pc = tid + NT_ * j;
        if (pc < OPC) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(r1 + (pc >> 3) * P2 + 8 * (pc & 7));
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, 16 * pc, 0, WT ? 16 : 0);      // (rows past the batch: dropped by the range check)
        }
      }
    }
  }
  // ---- conv3: 9 taps x 64 channels (two steps per tap).  Position tiles 2 pg, 2 pg + 1 of 7 (NS = 1: 4) ----
  {
    constexpr int NPOS = NS * PX3, NTL = (NPOS + 15) / 16, TPG = (NTL + 3) / 4;    // 98 positions: 7 tiles, 2 per group (NS = 1: 49, 4 tiles, 1)
    static_assert(TPG <= 3, "accumulators");
    h_t* const o3l = r1 + NS * PX2 * P2;                                     // a3 collection, behind the conv3 image
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
      int P = 16 * (TPG * pg + t) + m; if (P > NPOS - 1) P = NPOS - 1;
      const int sp = P / PX3, rem = P - sp * PX3, p = rem / 7, q = rem - p * 7;
      apos[t] = (sp * PX2 + p * 9 + q) * P2 + 8 * kq;
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    {
      const int nt = NTL - TPG * pg;
      if (nt >= 3 && TPG >= 3) mm<3, 18, 64, 3, 9, P2>(r1, apos, wb, acc);
      else if (nt >= 2 && TPG >= 2) mm<2, 18, 64, 3, 9, P2>(r1, apos, wb, acc);
      else if (nt >= 1) mm<1, 18, 64, 3, 9, P2>(r1, apos, wb, acc);
    }
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
      const int P = 16 * (TPG * pg + t) + m;
      if (P < NPOS) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          h4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (h_t)fmaxf(acc[u][t][e], 0.0f);
          *reinterpret_cast<h4*>(o3l + P * P2 + 32 * mp + 16 * u + 4 * kq) = v;
        }
      }
    }
    __syncthreads();
    h_t* const o3 = c.a3 + ((int64_t)z * c.B + n0) * (PX3 * NO);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)o3, 0, nvalid * PX3 * NO * 2, 0x00020000);
    constexpr int OPC = NPOS * 8;
#pragma unroll
    for (int j = 0; j < (OPC + NT_ - 1) / NT_; ++j) {
      const int pc = tid + NT_ * j;
      if (pc < OPC) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(o3l + (pc >> 3) * P2 + 8 * (pc & 7));
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, 16 * pc, 0, WT ? 16 : 0);
      }
    }
  }
}

// ---- the backward data path of the same two layers, float16, B >= 128: conv3_dgrad -> conv2_dgrad as ONE launch ------------------------------
// (deepqnetwork.py:162, model.bprop through the third and second Convolution layers; online net only.)  One workgroup per sample:
//   * delta3's padded plane ([11][11][64] halves, borders zero), W3 in its master layout ([(r, s, c)][64 f]: the row of output map c at tap
//     (r, s) is k-contiguous), a2 (the Rectlin gate) are fetched once into LDS; W2 (master layout) and a1 (the second gate) into registers;
//   * delta2 = full correlation: position (y, x), tap (r, s) reads pixel (y + 2 - r, x + 2 - s) of the padded plane — im2col at ds_read
//     time again, k = (r, s, f) ascending, v_mfma_f32_16x16x32_f16 with the weights as the row operand; 6 position tiles x 4 map tiles over
//     six waves (2 x 2 each: 4 fragment reads per 4 MFMAs);
//   * gated by a2 > 0 it is written as half into a second padded plane in LDS (borders zeroed at entry) — conv2_dgrad's input, never
//     re-read from memory — and leaves as the dense delta2 of conv2's weight gradient; W2 replaces W3 in LDS;
//   * delta1: the stride-2 transposed convolution as four parity classes (py, px), each a 2 x 2 correlation over the padded delta2 plane
//     (class position (i, j), tap (aa, bb) reads pixel (i + 1 - aa, j + 1 - bb); weights row ((py + 2 aa) 4 + px + 2 bb) 32 + c): wave =
//     (class, half of its 7 position tiles), both 16-map tiles; the outputs are collected as the dense [20][20][32] plane in LDS and leave
//     as whole lines, gated by a1 > 0 from the registers on the way out.
// Same half operands and the same k order in one fp32 accumulator per output as Conv3DgradH / Conv2DgradH on the block-tile routine.
struct DArgs {
  const h_t* d3p;           // [B][11][11][64], loss-scaled, borders zero
  const h_t* w3;            // master layout [(r, s, c)][64 f]  (9 x 64 rows)
  const h_t* w2;            // master layout [(r, s, c)][64 f]  (16 x 32 rows)
  const h_t* a2;            // [B][81][64]  (online net)
  const h_t* a1;            // [B][400][32]
  h_t* d2;                  // [B][81][64]
  h_t* d1;                  // [B][400][32]
  int B;
};
constexpr int DP = 80;                                   // halves per pixel of a padded delta plane / per weight row in LDS (64 + 16)
constexpr int D_A = 0;                                   // delta3 plane: 121 x 80
constexpr int D_C = D_A + 121 * DP;                      // a2 gate: 81 x 64
constexpr int D_D = D_C + PX2 * NO;                      // delta2 plane: 121 x 80
constexpr int D_B = D_D + 121 * DP;                      // weights: 576 rows x 80 (W3), then 512 rows (W2)
constexpr int D_E = 0;                                   // delta1 collection [400][32] over the dead delta3 plane + gate
constexpr int D_TOTAL = D_B + KK3 * DP;                  // 70 624 halves = 141 248 bytes
static_assert(PX1 * 32 <= D_D && D_TOTAL * 2 <= 160 * 1024, "LDS budget");

// NPT position tiles x 2 map tiles, STEPS k-steps of 32; AOFF(st) / WOFF(st): the step's offsets (compile-time after unrolling)
template <int NPT, int STEPS, class AOFF, class WOFF>
__device__ __forceinline__ void mmg(const h_t* img, const int* apos, const h_t* wb, f32x4 (*acc)[4], AOFF aoff, WOFF woff) {
#pragma unroll
  for (int st = 0; st < STEPS; ++st) {
    const h8 w0 = *reinterpret_cast<const h8*>(wb + woff(st)), w1 = *reinterpret_cast<const h8*>(wb + 16 * DP + woff(st));
#pragma unroll
    for (int t = 0; t < NPT; ++t) {
      const h8 x = *reinterpret_cast<const h8*>(img + apos[t] + aoff(st));
      acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, x, acc[0][t], 0, 0, 0);
      acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, x, acc[1][t], 0, 0, 0);
    }
  }
}

template <bool WT>
__global__ void __launch_bounds__(NT_) conv_ssh_dgrad_chain_kernel(const DArgs c) {
  __shared__ __attribute__((aligned(16))) h_t smem[D_TOTAL];
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.x;
  const u32x4* const p3 = reinterpret_cast<const u32x4*>(c.d3p + (int64_t)n * (121 * NO));
  const u32x4* const pw3 = reinterpret_cast<const u32x4*>(c.w3);
  const u32x4* const pa2 = reinterpret_cast<const u32x4*>(c.a2 + (int64_t)n * (PX2 * NO));
  const u32x4* const pw2 = reinterpret_cast<const u32x4*>(c.w2);
  const u32x4* const pa1 = reinterpret_cast<const u32x4*>(c.a1 + (int64_t)n * (PX1 * 32));
  constexpr int N3 = 121 * 8, NA2 = PX2 * 8, NA1 = PX1 * 4;                 // 16-byte pieces: 968, 648, 1600
  u32x4 v3[2], vw3[9], va2[2], vw2[8], va1[4];
#pragma unroll
  for (int j = 0; j < 9; ++j) vw3[j] = pw3[tid + NT_ * j];
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int pc = tid + NT_ * j; v3[j] = p3[pc < N3 ? pc : N3 - 1]; va2[j] = pa2[pc < NA2 ? pc : NA2 - 1]; }
  {                                                                          // the delta2 plane's borders (all of it: the interior is overwritten)
    const u32x4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 3; ++j) { const int pc = tid + NT_ * j; if (pc < 121 * DP / 8) *reinterpret_cast<u32x4*>(smem + D_D + 8 * pc) = z4; }
  }
#pragma unroll
  for (int j = 0; j < 9; ++j) { const int pc = tid + NT_ * j; *reinterpret_cast<u32x4*>(smem + D_B + (pc >> 3) * DP + 8 * (pc & 7)) = vw3[j]; }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int pc = tid + NT_ * j;
    if (pc < N3) *reinterpret_cast<u32x4*>(smem + D_A + (pc >> 3) * DP + 8 * (pc & 7)) = v3[j];
    if (pc < NA2) *reinterpret_cast<u32x4*>(smem + D_C + 8 * pc) = va2[j];
  }
  // (needed behind conv3_dgrad only: in flight under it)
#pragma unroll
  for (int j = 0; j < 8; ++j) vw2[j] = pw2[tid + NT_ * j];
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int pc = tid + NT_ * j; va1[j] = pa1[pc < NA1 ? pc : NA1 - 1]; }
  __syncthreads();

  f32x4 acc[2][4];
  int apos[4];
  // ---- conv3_dgrad: wave = (map pair mp, position group pg of two 16-position tiles; 6 tiles: pg = 3 idles) ----
  {
    const int mp = wave & 1, pg = wave >> 1;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      int P = 16 * (2 * pg + t) + m; if (P > PX2 - 1) P = PX2 - 1;
      const int y = P / 9, x = P - y * 9;
      apos[t] = D_A + (y * 11 + x) * DP + 8 * kq;
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const h_t* const wb = smem + D_B + (32 * mp + m) * DP + 8 * kq;
    auto aoff = [](int st) { const int rs = st >> 1, r = rs / 3, s = rs - r * 3; return ((2 - r) * 11 + (2 - s)) * DP + 32 * (st & 1); };
    auto woff = [](int st) { return (st >> 1) * (NO * DP) + 32 * (st & 1); };
    if (pg < 3) mmg<2, 18>(smem, apos, wb, acc, aoff, woff);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int P = 16 * (2 * pg + t) + m;
      if (pg < 3 && P < PX2) {
        const int y = P / 9, x = P - y * 9;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int c0 = 32 * mp + 16 * u + 4 * kq;
          const h4 g = *reinterpret_cast<const h4*>(smem + D_C + P * NO + c0);
          h4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (float)g[e] > 0.0f ? (h_t)acc[u][t][e] : (h_t)0.0f;
          *reinterpret_cast<h4*>(smem + D_D + ((y + 1) * 11 + x + 1) * DP + c0) = v;
        }
      }
    }
  }
  __syncthreads();                                                           // delta2 plane complete; every read of W3 is done
#pragma unroll
  for (int j = 0; j < 8; ++j) { const int pc = tid + NT_ * j; *reinterpret_cast<u32x4*>(smem + D_B + (pc >> 3) * DP + 8 * (pc & 7)) = vw2[j]; }
  {                                                                          // the dense delta2 (conv2's weight gradient reads it): 81 rows of 128 bytes
    h_t* const o2 = c.d2 + (int64_t)n * (PX2 * NO);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)o2, 0, PX2 * NO * 2, 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pc = tid + NT_ * j;
      if (pc < NA2) {
        const int P = pc >> 3, y = P / 9, x = P - y * 9;
        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + D_D + ((y + 1) * 11 + x + 1) * DP + 8 * (pc & 7));
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, 16 * pc, 0, WT ? 16 : 0);
      }
    }
  }
  __syncthreads();
  // ---- conv2_dgrad: wave = (parity class z = (py, px), half h of the class's 7 position tiles: 4 + 3), both 16-map tiles ----
  {
    const int z = wave >> 1, h = wave & 1, py = z >> 1, px = z & 1;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      int P = 16 * (4 * h + t) + m; if (P > 99) P = 99;
      const int i = P / 10, j = P - i * 10;
      apos[t] = D_D + (i * 11 + j) * DP + 8 * kq;
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const h_t* const wb = smem + D_B + ((py * 4 + px) * 32 + m) * DP + 8 * kq;
    auto aoff = [](int st) { const int ab = st >> 1, aa = ab >> 1, bb = ab & 1; return ((1 - aa) * 11 + (1 - bb)) * DP + 32 * (st & 1); };
    auto woff = [](int st) { const int ab = st >> 1, aa = ab >> 1, bb = ab & 1; return (8 * aa + 2 * bb) * (32 * DP) + 32 * (st & 1); };
    if (h == 0) mmg<4, 8>(smem, apos, wb, acc, aoff, woff); else mmg<3, 8>(smem, apos, wb, acc, aoff, woff);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int P = 16 * (4 * h + t) + m;
      if (4 * h + t < 7 && P < 100) {
        const int i = P / 10, j = P - i * 10, pix = (2 * i + py) * 20 + 2 * j + px;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          h4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (h_t)acc[u][t][e];
          *reinterpret_cast<h4*>(smem + D_E + pix * 32 + 16 * u + 4 * kq) = v;
        }
      }
    }
  }
  __syncthreads();
  {                                                                          // delta1 = collected plane gated by a1 > 0: 400 rows of 64 bytes, whole lines
    h_t* const o1 = c.d1 + (int64_t)n * (PX1 * 32);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)o1, 0, PX1 * 32 * 2, 0x00020000);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pc = tid + NT_ * j;
      if (pc < NA1) {
        union { u32x4 u; h8 h; } g, v, o;
        g.u = va1[j];
        v.u = *reinterpret_cast<const u32x4*>(smem + D_E + 8 * pc);
#pragma unroll
        for (int e = 0; e < 8; ++e) o.h[e] = (float)g.h[e] > 0.0f ? v.h[e] : (h_t)0.0f;
        __builtin_amdgcn_raw_buffer_store_b128(o.u, rs, 16 * pc, 0, WT ? 16 : 0);
      }
    }
  }
}

template <bool WT>
inline hipError_t launch_dgrad_chain(const DArgs& c, hipStream_t s) {
  if (c.B <= 0) return hipSuccess;
  SDQN_LAUNCH((conv_ssh_dgrad_chain_kernel<WT>), dim3(c.B), dim3(NT_), 0, s, c);
  return hipGetLastError();
}

template <int NS, bool WT, bool C1>
inline hipError_t launch_chain(const Args& c, int nz, hipStream_t s) {
  if (c.B <= 0) return hipSuccess;
  SDQN_LAUNCH((conv_ssh_chain_kernel<NS, WT, C1>), dim3(nz * c.G), dim3(NT_), 0, s, c);
  return hipGetLastError();
}

}  // namespace ssh
}  // namespace sdqn
