// bn_kernels.hip — the --batch_norm variant of the network (deepqnetwork.py:26,83-89): Neon BatchNorm between the
// linear part and the Rectlin of conv1..3 and fc4 [neon-recalled: neon/layers/layer.py BatchNorm, rho 0.9, eps 1e-3;
// restated in oracle/dqn_bn_numpy.py].  A non-default learner option: kept OUT of the tile
// engine — the forward GEMM stages run as *Raw problems (no Rectlin) and these elementwise passes run between them:
//
//   forward  (per layer): bn_partial (training, z = 0: per-feature sum / sum of squares, fixed-order, in double)
//                         bn_apply   (every workgroup finalises the statistics it needs from the partials;
//                                     y = relu(gamma * (x - mean) * rstd + beta); target / predict use running stats)
//   backward (per layer): bn_bwd_partial (sum err, sum err * xhat)  ->  bn_bwd_apply (dx in place, dense + padded)
//
// Activations are NHWC, so "feature" is the contiguous dimension and a row is one (sample, pixel).
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace sdqn {

constexpr float BN_RHO = 0.9f, BN_ONE_MINUS_RHO = (float)(1.0 - 0.9), BN_EPS = 1e-3f;
constexpr int BN_ROWS_PER_PART = 256;

__device__ inline float bn_x(const BnArgs& b, int z, int r, int c) {
  if (b.S4 > 0) {                                   // fc4: the linear output still lives in the split-K slabs
    float v = 0.0f;
    for (int s = 0; s < b.S4; ++s) v += b.x[(int64_t)s * 2 * b.B * NFC + ((int64_t)z * b.B + r) * NFC + c];     // fixed order
    return v;
  }
  return b.x[((int64_t)z * b.rows + r) * b.C + c];
}

// grid (RB, C/32), 256 threads: lane c = t & 31 owns one feature, 8 row lanes; partial[rb][c] = {sum, sumsq}
__global__ void __launch_bounds__(256) bn_partial_kernel(const BnArgs b) {
  __shared__ double sh[8][32][2];
  const int t = threadIdx.x, cl = t & 31, rl = t >> 5;
  const int c = blockIdx.y * 32 + cl, rb = blockIdx.x;
  const int r0 = rb * BN_ROWS_PER_PART, r1 = min(r0 + BN_ROWS_PER_PART, b.rows);
  double s = 0.0, q = 0.0;
#pragma unroll 4
  for (int r = r0 + rl; r < r1; r += 8) { const double v = (double)bn_x(b, 0, r, c); s += v; q += v * v; }
  sh[rl][cl][0] = s; sh[rl][cl][1] = q;
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < 8; ++k) { s += sh[k][cl][0]; q += sh[k][cl][1]; }                    // fixed order
    b.partial[((int64_t)rb * b.C + c) * 2] = s; b.partial[((int64_t)rb * b.C + c) * 2 + 1] = q;
  }
}

// Totals of the two partial sums of every feature, by the whole 256-thread workgroup: G = 256 / min(C, 256) thread groups
// each add every G-th row block, then group 0 adds the G group sums in order.  Every workgroup of every kernel that
// needs the totals runs exactly this (same order -> same bits everywhere).  tot[c] = {sum0, sum1}; sh: 256 x 2 doubles.
__device__ inline void bn_totals(const BnArgs& b, double (*tot)[2], double (*sh)[2]) {
  const int t = threadIdx.x;
  const int RB = (b.rows + BN_ROWS_PER_PART - 1) / BN_ROWS_PER_PART;
  const int Cc = b.C < 256 ? b.C : 256, G = 256 / Cc;
  const int cl = t % Cc, grp = t / Cc;
  for (int cb = 0; cb < b.C; cb += Cc) {
    const int c = cb + cl;
    double s = 0.0, q = 0.0;
#pragma unroll 4
    for (int rb = grp; rb < RB; rb += G) { s += b.partial[((int64_t)rb * b.C + c) * 2]; q += b.partial[((int64_t)rb * b.C + c) * 2 + 1]; }
    sh[t][0] = s; sh[t][1] = q;
    __syncthreads();
    if (grp == 0) {
      for (int k = 1; k < G; ++k) { s += sh[k * Cc + cl][0]; q += sh[k * Cc + cl][1]; }
      tot[c][0] = s; tot[c][1] = q;
    }
    __syncthreads();
  }
}

// grid = ceil(nz * rows * C / 4 / 256) workgroups; every one first builds scale/shift of ALL C features of both nets
// in LDS (C <= 512), workgroup 0 also publishes mean / rstd for the backward pass and updates the running statistics
__global__ void __launch_bounds__(256) bn_apply_kernel(const BnArgs b) {
  __shared__ float sc[2][512], sf[2][512];
  __shared__ double tot[512][2], sh[256][2];
  const int t = threadIdx.x;
  const int off = bn_off(b.layer);
  if (b.train) bn_totals(b, tot, sh);
  for (int z = 0; z < b.nz; ++z) {
    const float* th = b.theta[z] + b.off_bn;
    for (int c = t; c < b.C; c += 256) {
      float mean, var;
      if (z == 0 && b.train) {
        const double m = tot[c][0] / (double)b.rows;
        const double v = tot[c][1] / (double)b.rows - m * m;                                     // biased variance (be.var)
        mean = (float)m; var = (float)(v > 0.0 ? v : 0.0);
        if (blockIdx.x == 0) {
          float* run = b.theta[0] + b.off_bn + BN_PARAMS;
          run[off + c] = run[off + c] * BN_RHO + BN_ONE_MINUS_RHO * mean;                          // gmean
          run[off + b.C + c] = run[off + b.C + c] * BN_RHO + BN_ONE_MINUS_RHO * var;              // gvar
        }
      } else {
        mean = th[BN_PARAMS + off + c]; var = th[BN_PARAMS + off + b.C + c];
      }
      const float rstd = 1.0f / sqrtf(var + BN_EPS);
      if (z == 0 && b.train && blockIdx.x == 0) { b.mean[c] = mean; b.rstd[c] = rstd; }
      sc[z][c] = rstd; sf[z][c] = mean;
    }
  }
  __syncthreads();
  const int C4 = b.C / 4;
  const int64_t n4 = (int64_t)b.nz * b.rows * C4;
  const int64_t i4 = (int64_t)blockIdx.x * 256 + t;
  if (i4 >= n4) return;
  const int c0 = (int)(i4 % C4) * 4;
  const int64_t row = i4 / C4;
  const int z = (int)(row / b.rows), r = (int)(row - (int64_t)z * b.rows);
  const float* th = b.theta[z] + b.off_bn + off;
  float4 o;
  float* op = &o.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + k;
    const float xh = (bn_x(b, z, r, c) - sf[z][c]) * sc[z][c];                                    // xhat
    op[k] = fmaxf(xh * th[b.C + c] + th[c], 0.0f);                                                // gamma, beta; Rectlin
  }
  *reinterpret_cast<float4*>(b.a + (row * b.C + c0)) = o;
}

// backward, z = 0 only.  err = delta at the BatchNorm output (already masked by the Rectlin), dense [rows][C]
__global__ void __launch_bounds__(256) bn_bwd_partial_kernel(const BnArgs b) {
  __shared__ double sh[8][32][2];
  const int t = threadIdx.x, cl = t & 31, rl = t >> 5;
  const int c = blockIdx.y * 32 + cl, rb = blockIdx.x;
  const int r0 = rb * BN_ROWS_PER_PART, r1 = min(r0 + BN_ROWS_PER_PART, b.rows);
  const float mean = b.mean[c], rstd = b.rstd[c];
  double s = 0.0, q = 0.0;
#pragma unroll 4
  for (int r = r0 + rl; r < r1; r += 8) {
    const float e = b.d[(int64_t)r * b.C + c];
    const float xh = (bn_x(b, 0, r, c) - mean) * rstd;
    s += (double)e; q += (double)(xh * e);
  }
  sh[rl][cl][0] = s; sh[rl][cl][1] = q;
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < 8; ++k) { s += sh[k][cl][0]; q += sh[k][cl][1]; }
    b.partial[((int64_t)rb * b.C + c) * 2] = s; b.partial[((int64_t)rb * b.C + c) * 2 + 1] = q;
  }
}

__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const BnArgs b) {
  __shared__ float gb[512], gg[512];
  __shared__ double tot[512][2], sh[256][2];
  const int t = threadIdx.x;
  const int off = bn_off(b.layer);
  bn_totals(b, tot, sh);
  for (int c = t; c < b.C; c += 256) {
    gb[c] = (float)tot[c][0]; gg[c] = (float)tot[c][1];
    if (blockIdx.x == 0) { b.g[b.off_bn + off + c] = gb[c]; b.g[b.off_bn + off + b.C + c] = gg[c]; }    // grad_beta, grad_gamma (sums: A9 divides)
  }
  __syncthreads();
  const int C4 = b.C / 4;
  const int64_t n4 = (int64_t)b.rows * C4;
  const int64_t i4 = (int64_t)blockIdx.x * 256 + t;
  if (i4 >= n4) return;
  const int c0 = (int)(i4 % C4) * 4;
  const int r = (int)(i4 / C4);
  const float* th = b.theta[0] + b.off_bn + off;
  const float m = (float)b.rows;
  const float4 e4 = *reinterpret_cast<const float4*>(b.d + ((int64_t)r * b.C + c0));
  const float* ep = &e4.x;
  float4 o; float* op = &o.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + k;
    const float rstd = b.rstd[c];
    const float xh = (bn_x(b, 0, r, c) - b.mean[c]) * rstd;
    op[k] = th[b.C + c] * (ep[k] - (xh * gg[c] + gb[c]) / m) * rstd;                              // gamma * (err - (xhat*gg + gb)/m) * rstd
  }
  *reinterpret_cast<float4*>(b.d + ((int64_t)r * b.C + c0)) = o;
  if (b.dpad) {                                                                                   // padded plane used by the dgrad gather
    const int n = r / b.PQ, pix = r - n * b.PQ, p = pix / b.Qw, qx = pix - p * b.Qw;
    *reinterpret_cast<float4*>(b.dpad + ((((int64_t)n * b.PD + p + b.pad) * b.PD + qx + b.pad) * b.C + c0)) = o;
  }
}

// beta / gamma of the four layers: the same optimizer arithmetic as every other parameter (grad / be.bsz first, A9)
__global__ void __launch_bounds__(256) bn_update_kernel(const UpdateArgs u) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < BN_PARAMS; i += gridDim.x * 256) {
    const int64_t e = u.bn_first + i;
    float w = u.theta[e], s1 = u.state[e], s2 = u.opt != 0 ? u.state2[e] : 0.0f;
    w = opt_apply(w, s1, s2, u.g[e], u);
    u.theta[e] = w; u.state[e] = s1; if (u.opt != 0) u.state2[e] = s2;
  }
}
hipError_t launch_bn_update(const UpdateArgs& u, hipStream_t s) {
  hipLaunchKernelGGL(bn_update_kernel, dim3((BN_PARAMS + 255) / 256), dim3(256), 0, s, u);
  return hipGetLastError();
}

hipError_t launch_bn_forward(const BnArgs& b, hipStream_t s) {
  const int RB = (b.rows + BN_ROWS_PER_PART - 1) / BN_ROWS_PER_PART;
  if (b.train) hipLaunchKernelGGL(bn_partial_kernel, dim3(RB, b.C / 32), dim3(256), 0, s, b);
  const int64_t n4 = (int64_t)b.nz * b.rows * (b.C / 4);
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, b);
  return hipGetLastError();
}
hipError_t launch_bn_backward(const BnArgs& b, hipStream_t s) {
  const int RB = (b.rows + BN_ROWS_PER_PART - 1) / BN_ROWS_PER_PART;
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(RB, b.C / 32), dim3(256), 0, s, b);
  const int64_t n4 = (int64_t)b.rows * (b.C / 4);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, b);
  return hipGetLastError();
}

}  // namespace sdqn
