// generic_net.hip — see generic_net.h.  The reference's train step (src/deepqnetwork.py:107-172) and predict (:174-186)
// for any input geometry and for float32 / float64 arithmetic: every Convolution / Affine layer (deepqnetwork.py:83-91)
// is im2col + GEMM in Neon's own layouts —
//     conv   : cols (N*P*Q, C*R*S) @ W (C*R*S, K)      column index c*R*S + r*S + s   
//     affine : x (N, nin) @ W(nout, nin)^T             fc4's nin flattens conv3's output in (K, P, Q) order
// which is also how Neon's CPU backend computes them [neon-recalled], so no layout conversion happens at the C ABI.
// Activations are NHWC ([n][y][x][c] = the GEMM output as it falls); fc4 reads them through the same im2col with a
// kernel as large as the map (R = P3, S = Q3), which yields exactly the (K, P, Q) flatten.  Backward (A8): weight
// gradients cols^T @ delta (split-K slabs reduced in fixed order), input gradients delta @ W^T followed by col2im in
// the oracle's (r, s) order and the Rectlin mask.  TD targets in double like the reference's host arithmetic (:136-143).
// Optimizers: Neon's RMSProp / Adam / Adadelta with `grad / be.bsz` first (A9, A10), one rounding per operation.
//
// Throughput is not the point of this path (the 84 x 84 x 4 float32 / float16 configurations run on the tuned kernels); it is a
// plain 64 x 64 LDS-tiled FMA GEMM, ~40 launches per step.
#include "generic_net.h"
#include "problems.h"          // MetaRec
#include <vector>
#include <algorithm>
#include <math.h>
#include <string.h>

namespace sdqn {
namespace {

template <typename T> __device__ inline T fma_t(T a, T b, T c);
template <> __device__ inline float fma_t<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <> __device__ inline double fma_t<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename T> __device__ inline T sqrt_t(T x);
template <> __device__ inline float sqrt_t<float>(float x) { return sqrtf(x); }
template <> __device__ inline double sqrt_t<double>(double x) { return sqrt(x); }

struct ConvGeom {                       // one Convolution layer, or fc4 seen as a convolution over the whole map
  int C, H, W, R, S, st, K, P, Q;
  __host__ __device__ int crs() const { return C * R * S; }
};

// ---- im2col ---------------------------------------------------------------------------------------------
// col[m][j], m = (n*P + p)*Q + q, j = (c*R + r)*S + s
template <typename T>
__global__ void __launch_bounds__(256) im2col_u8_kernel(const uint8_t* __restrict__ x, T* __restrict__ col, int64_t total, ConvGeom g) {
  const int crs = g.crs();
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int j = (int)(i % crs); const int64_t m = i / crs;
    const int s = j % g.S, r = (j / g.S) % g.R, c = j / (g.R * g.S);
    const int q = (int)(m % g.Q), p = (int)((m / g.Q) % g.P); const int64_t n = m / ((int64_t)g.P * g.Q);
    const uint8_t b = x[((n * g.C + c) * g.H + (p * g.st + r)) * (int64_t)g.W + (q * g.st + s)];     // states are [n][hist][H][W]
    col[i] = (T)b / (T)255;                                                                          // _setInput, deepqnetwork.py:94-100
  }
}
template <typename T>
__global__ void __launch_bounds__(256) im2col_nhwc_kernel(const T* __restrict__ a, T* __restrict__ col, int64_t total, ConvGeom g) {
  const int crs = g.crs();
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int j = (int)(i % crs); const int64_t m = i / crs;
    const int s = j % g.S, r = (j / g.S) % g.R, c = j / (g.R * g.S);
    const int q = (int)(m % g.Q), p = (int)((m / g.Q) % g.P); const int64_t n = m / ((int64_t)g.P * g.Q);
    col[i] = a[((n * g.H + (p * g.st + r)) * (int64_t)g.W + (q * g.st + s)) * g.C + c];
  }
}
// adjoint of im2col + Rectlin mask of the layer's input: dx[n][y][x][c] = (sum over (r, s) ascending of dcol[...]) * (act > 0)
template <typename T>
__global__ void __launch_bounds__(256) col2im_kernel(const T* __restrict__ dcol, const T* __restrict__ act, T* __restrict__ dx, int64_t total, ConvGeom g) {
  const int crs = g.crs();
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % g.C); const int x = (int)((i / g.C) % g.W); const int y = (int)((i / ((int64_t)g.C * g.W)) % g.H);
    const int64_t n = i / ((int64_t)g.C * g.W * g.H);
    T acc = (T)0;
    for (int r = 0; r < g.R; ++r) {
      const int yy = y - r;
      if (yy < 0 || yy % g.st) continue;
      const int p = yy / g.st; if (p >= g.P) continue;
      for (int s = 0; s < g.S; ++s) {
        const int xx = x - s;
        if (xx < 0 || xx % g.st) continue;
        const int q = xx / g.st; if (q >= g.Q) continue;
        acc += dcol[((n * g.P + p) * g.Q + q) * crs + (c * g.R + r) * g.S + s];
      }
    }
    dx[i] = act[i] > (T)0 ? acc : (T)0;
  }
}

// ---- GEMM: C[M][N] = sum_k A(m,k) B(k,n), A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn] -------------------------------
template <typename T> struct GemmArgs {
  const T* A; int64_t sam, sak; const T* B; int64_t sbk, sbn; T* C; const T* mask;
  int M, N, K, kchunk, relu;
};
constexpr int BM = 64, BN = 64, BK = 16;
template <typename T>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmArgs<T> g) {
  __shared__ T As[BK][BM + 1];
  __shared__ T Bs[BK][BN + 1];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int k_begin = blockIdx.z * g.kchunk, k_end = min(g.K, k_begin + g.kchunk);
  T acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (T)0;
  for (int k0 = k_begin; k0 < k_end; k0 += BK) {
#pragma unroll
    for (int e = tid; e < BM * BK; e += 256) {
      int m, k;
      if (g.sak == 1) { m = e / BK; k = e % BK; } else { k = e / BM; m = e % BM; }
      const int64_t gm = m0 + m, gk = k0 + k;
      As[k][m] = (gm < g.M && gk < k_end) ? g.A[gm * g.sam + gk * g.sak] : (T)0;
    }
#pragma unroll
    for (int e = tid; e < BK * BN; e += 256) {
      int n, k;
      if (g.sbn == 1) { k = e / BN; n = e % BN; } else { n = e / BK; k = e % BK; }
      const int64_t gn = n0 + n, gk = k0 + k;
      Bs[k][n] = (gn < g.N && gk < k_end) ? g.B[gk * g.sbk + gn * g.sbn] : (T)0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      T a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fma_t<T>(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
  const bool split = gridDim.z > 1;
  T* C = g.C + (split ? (int64_t)blockIdx.z * g.M * g.N : 0);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
      if (m < g.M && n < g.N) {
        T v = acc[i][j];
        if (!split) {
          if (g.relu) v = v > (T)0 ? v : (T)0;
          if (g.mask) v = g.mask[(int64_t)m * g.N + n] > (T)0 ? v : (T)0;
        }
        C[(int64_t)m * g.N + n] = v;
      }
    }
}
template <typename T>
__global__ void __launch_bounds__(256) reduce_slabs_kernel(const T* __restrict__ slabs, T* __restrict__ C, const T* __restrict__ mask, int64_t mn, int nz, int relu) {
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < mn; i += (int64_t)gridDim.x * 256) {
    T v = slabs[i];
    for (int z = 1; z < nz; ++z) v += slabs[(int64_t)z * mn + i];         // fixed order: slab 0, 1, 2, ...
    if (relu) v = v > (T)0 ? v : (T)0;
    if (mask) v = mask[i] > (T)0 ? v : (T)0;
    C[i] = v;
  }
}

// ---- TD target, error, cost, clip (deepqnetwork.py:133-159) ----------------------------------------------------------------------
template <typename T>
__global__ void head_kernel(const T* __restrict__ q_on, const T* __restrict__ q_tg, const uint8_t* __restrict__ act,
                            const int64_t* __restrict__ rew, const uint8_t* __restrict__ term, T* __restrict__ dq,
                            T* __restrict__ cost_terms, T* __restrict__ maxq, int N, int A, double discount, double minr, double maxr, T clip) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  T m = q_tg[(int64_t)n * A];
  for (int a = 1; a < A; ++a) { const T v = q_tg[(int64_t)n * A + a]; m = v > m ? v : m; }     // be.max(postq, axis=0) :124
  maxq[n] = m;
  double r = (double)rew[n];
  r = r < minr ? minr : (r > maxr ? maxr : r);                                // np.clip(rewards, min_reward, max_reward) :136
  const double y = term[n] ? r : r + discount * (double)m;                    // :139-143, python float arithmetic
  const T target = (T)y;                                                      // stored into the backend dtype
  const int at = act[n];
  T d = (T)0;
  for (int a = 0; a < A; ++a) {
    T e = (a == at) ? q_on[(int64_t)n * A + a] - target : (T)0;               // deltas = preq - targets: 0 off the taken action
    if (a == at) d = e;
    if (clip != (T)0) e = e < -clip ? -clip : (e > clip ? clip : e);          // :158-159
    dq[(int64_t)n * A + a] = e;
  }
  cost_terms[n] = (T)0.5 * (d * d);                                           // SumSquared before the clip (A7) :154
}
template <typename T>
__global__ void cost_kernel(const T* __restrict__ cost_terms, T* __restrict__ cost, double* __restrict__ cost_sum, int N) {
  if (blockIdx.x || threadIdx.x) return;
  double s = 0;
  for (int n = 0; n < N; ++n) s += (double)cost_terms[n];
  const T c = (T)(s / N);                                                     // mean over the batch (A7)
  cost[0] = c; cost_sum[0] += (double)c;
}

// ---- optimizers (deepqnetwork.py:50-59,165; Neon semantics A9, A10) -------------------------------------------------------------------
template <typename T> struct OptArgs {
  T* w; T* s1; T* s2; const T* g; int64_t n;
  T bsz, rho, omr, lr, eps, b1, omb1, b2, omb2, lr_t; int opt;
};
template <typename T>
__global__ void __launch_bounds__(256) update_kernel(const OptArgs<T> u) {
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < u.n; i += (int64_t)gridDim.x * 256) {
    const T gr = u.g[i] / u.bsz;                                              // grad / be.bsz
    T w = u.w[i], a = u.s1[i];
    if (u.opt == 0) {                                                         // RMSProp
      a = u.rho * a + (gr * gr) * u.omr;
      w = w - (gr * u.lr) / (sqrt_t<T>(a + u.eps) + u.eps);
    } else if (u.opt == 1) {                                                  // Adam
      T v = u.s2[i];
      a = a * u.b1 + u.omb1 * gr;
      v = v * u.b2 + (u.omb2 * gr) * gr;
      w = w - (u.lr_t * a) / (sqrt_t<T>(v) + u.eps);
      u.s2[i] = v;
    } else {                                                                  // Adadelta
      T v = u.s2[i];
      a = a * u.rho + (u.omr * gr) * gr;
      const T upd = sqrt_t<T>((v + u.eps) / (a + u.eps)) * gr;
      v = v * u.rho + (u.omr * upd) * upd;
      w = w - upd;
      u.s2[i] = v;
    }
    u.s1[i] = a; u.w[i] = w;
  }
}

inline unsigned grid_for(int64_t n) { int64_t b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

// ---- the network ---------------------------------------------------------------------------------------------------------------
#define GCHK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) return e__; } while (0)

template <typename T>
class GenericNetT : public GenericNet {
 public:
  sdqn_net_cfg cfg; hipStream_t st = nullptr;
  int B = 0, A = 0, hist = 0, H = 0, W = 0;
  ConvGeom cv[4];                              // conv1..3, fc4 (as a convolution over the whole conv3 map)
  int64_t off[6] = {0}, rows[5] = {0}, cols[5] = {0}, NP = 0;
  T *theta = nullptr, *theta_t = nullptr, *s1 = nullptr, *s2 = nullptr, *g = nullptr;
  T* col[4] = {nullptr}; T* act[4] = {nullptr}; T* dact[3] = {nullptr};
  T *q = nullptr, *dq = nullptr, *d4 = nullptr, *dcol = nullptr, *cost_terms = nullptr, *cost = nullptr, *maxq = nullptr, *slab = nullptr;
  double* cost_sum = nullptr; int64_t slab_cap = 0;
  uint8_t* st_states = nullptr; uint8_t* st_small = nullptr;
  std::vector<void*> allocs; std::vector<uint8_t> small_host;

  ~GenericNetT() override { if (st) hipStreamSynchronize(st); for (void* p : allocs) hipFree(p); }
  bool is_f64() const override { return sizeof(T) == 8; }
  int64_t layer_size(int l) const override { return (l >= 0 && l < 5) ? rows[l] * cols[l] : -1; }
  int64_t param_count() const override { return NP; }
  size_t state_bytes() const override { return (size_t)hist * H * W; }

  template <typename U> hipError_t dalloc(U** p, int64_t n) {
    hipError_t e = hipMalloc((void**)p, (size_t)(n > 0 ? n : 1) * sizeof(U));
    if (e == hipSuccess) { allocs.push_back(*p); e = hipMemsetAsync(*p, 0, (size_t)(n > 0 ? n : 1) * sizeof(U), st); }
    return e;
  }
  int64_t mrows(int l, int n) const { return (int64_t)n * cv[l].P * cv[l].Q; }

  hipError_t init(std::string* err) {
    // deepqnetwork.py:83-91: Conv(8,8,32,s4), Conv(4,4,64,s2), Conv(3,3,64,s1), Affine(512), Affine(A); output (H-R)//s+1 (A1)
    const int RS[3][4] = {{8, 8, 32, 4}, {4, 4, 64, 2}, {3, 3, 64, 1}};
    int C = hist, h = H, w = W;
    for (int l = 0; l < 3; ++l) {
      ConvGeom& g_ = cv[l];
      g_.C = C; g_.H = h; g_.W = w; g_.R = RS[l][0]; g_.S = RS[l][1]; g_.K = RS[l][2]; g_.st = RS[l][3];
      if (h < g_.R || w < g_.S) { *err = "screen too small for the layer stack of deepqnetwork.py:83-87"; return hipErrorInvalidValue; }
      g_.P = (h - g_.R) / g_.st + 1; g_.Q = (w - g_.S) / g_.st + 1;
      rows[l] = g_.crs(); cols[l] = g_.K;
      C = g_.K; h = g_.P; w = g_.Q;
    }
    cv[3].C = C; cv[3].H = h; cv[3].W = w; cv[3].R = h; cv[3].S = w; cv[3].st = 1; cv[3].K = 512; cv[3].P = 1; cv[3].Q = 1;
    rows[3] = 512; cols[3] = cv[3].crs();            // Neon Linear: (nout, nin)
    rows[4] = A; cols[4] = 512;
    for (int l = 0; l < 5; ++l) off[l + 1] = off[l] + rows[l] * cols[l];
    NP = off[5];
    GCHK(dalloc(&theta, NP)); GCHK(dalloc(&s1, NP)); GCHK(dalloc(&s2, NP)); GCHK(dalloc(&g, NP));
    if (cfg.target_enabled) GCHK(dalloc(&theta_t, NP)); else theta_t = theta;        // deepqnetwork.py:64-73
    int64_t dcol_n = 0;
    for (int l = 0; l < 4; ++l) {
      GCHK(dalloc(&col[l], mrows(l, B) * cv[l].crs()));
      GCHK(dalloc(&act[l], mrows(l, B) * cv[l].K));
      if (l >= 1) dcol_n = std::max(dcol_n, mrows(l, B) * cv[l].crs());
      if (l < 3) GCHK(dalloc(&dact[l], mrows(l, B) * cv[l].K));
    }
    GCHK(dalloc(&dcol, dcol_n));
    GCHK(dalloc(&q, (int64_t)2 * B * A)); GCHK(dalloc(&dq, (int64_t)B * A)); GCHK(dalloc(&d4, (int64_t)B * 512));
    GCHK(dalloc(&cost_terms, B)); GCHK(dalloc(&cost, 1)); GCHK(dalloc(&maxq, B)); GCHK(dalloc(&cost_sum, 1));
    slab_cap = std::max((int64_t)8 << 20, 2 * NP);        // split-K slabs (gemm() never asks for more than fits)
    GCHK(dalloc(&slab, slab_cap));
    GCHK(dalloc(&st_states, (int64_t)2 * B * (int64_t)state_bytes()));
    GCHK(dalloc(&st_small, (int64_t)B * 10));
    small_host.resize((size_t)B * 10);
    return hipStreamSynchronize(st);
  }

  // C = A B (+ Rectlin / mask), K split into fixed slabs when the output has too few tiles to fill the chip
  hipError_t gemm(GemmArgs<T> a) {
    const int tm = (a.M + BM - 1) / BM, tn = (a.N + BN - 1) / BN;
    const int64_t tiles = (int64_t)tm * tn;
    int splits = 1;
    if (tiles < 256 && a.K >= 4 * BK) {
      int64_t want = 1024 / tiles; const int64_t by_k = a.K / (2 * BK);
      if (want > by_k) want = by_k;
      if (want > 64) want = 64;
      while (want > 1 && want * (int64_t)a.M * a.N > slab_cap) --want;
      if (want > 1) splits = (int)want;
    }
    int kchunk = (a.K + splits - 1) / splits; kchunk = (kchunk + BK - 1) / BK * BK;
    splits = (a.K + kchunk - 1) / kchunk;
    a.kchunk = kchunk;
    T* out = a.C;
    if (splits > 1) a.C = slab;
    hipLaunchKernelGGL(gemm_kernel<T>, dim3(tn, tm, splits), dim3(256), 0, st, a);
    if (splits > 1) {
      const int64_t mn = (int64_t)a.M * a.N;
      hipLaunchKernelGGL(reduce_slabs_kernel<T>, dim3(grid_for(mn)), dim3(256), 0, st, (const T*)slab, out, a.mask, mn, splits, a.relu);
    }
    return hipGetLastError();
  }
  static GemmArgs<T> ga(const T* A, int64_t sam, int64_t sak, const T* Bm, int64_t sbk, int64_t sbn, T* C, int M, int N, int K,
                        int relu = 0, const T* mask = nullptr) {
    GemmArgs<T> a; a.A = A; a.sam = sam; a.sak = sak; a.B = Bm; a.sbk = sbk; a.sbn = sbn; a.C = C; a.mask = mask;
    a.M = M; a.N = N; a.K = K; a.kchunk = K; a.relu = relu; return a;
  }

  // Q(states; W) of n states -> q + z*B*A; leaves cols / activations of this pass in col[] / act[]  (deepqnetwork.py:119-130,178-180)
  hipError_t forward(int z, const uint8_t* states_dev, int n) {
    const T* Wt = z ? theta_t : theta;
    for (int l = 0; l < 4; ++l) {
      const ConvGeom& c = cv[l];
      const int64_t m = mrows(l, n), total = m * c.crs();
      if (l == 0) hipLaunchKernelGGL(im2col_u8_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, states_dev, col[0], total, c);
      else hipLaunchKernelGGL(im2col_nhwc_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, (const T*)act[l - 1], col[l], total, c);
      if (l < 3) GCHK(gemm(ga(col[l], c.crs(), 1, Wt + off[l], c.K, 1, act[l], (int)m, c.K, c.crs(), 1)));     // cols @ W, Rectlin
      else GCHK(gemm(ga(col[3], c.crs(), 1, Wt + off[3], 1, c.crs(), act[3], n, 512, c.crs(), 1)));            // x @ W4^T, Rectlin
    }
    return gemm(ga(act[3], 512, 1, Wt + off[4], 1, 512, q + (int64_t)z * B * A, n, A, 512, 0));                // a4 @ W5^T
  }

  hipError_t train_dev(const uint8_t* pre, const uint8_t* post, const uint8_t* actions, const int64_t* rew, const uint8_t* term, int epoch) override {
    GCHK(forward(1, post, B));                                                 // target net on the poststates :119-125
    GCHK(forward(0, pre, B));                                                  // online net on the prestates, tensors kept :128-130
    hipLaunchKernelGGL(head_kernel<T>, dim3((B + 63) / 64), dim3(64), 0, st, (const T*)q, (const T*)(q + (int64_t)B * A), actions, rew, term,
                       dq, cost_terms, maxq, B, A, cfg.discount_rate, cfg.min_reward, cfg.max_reward, (T)cfg.clip_error);
    hipLaunchKernelGGL(cost_kernel<T>, dim3(1), dim3(64), 0, st, (const T*)cost_terms, cost, cost_sum, B);
    // ---- bprop (A8) :162
    GCHK(gemm(ga(dq, 1, A, act[3], 512, 1, g + off[4], A, 512, B)));                                   // gW5 = dq^T @ a4
    GCHK(gemm(ga(dq, A, 1, theta + off[4], 512, 1, d4, B, 512, A, 0, act[3])));                        // d4 = (dq @ W5) * (a4 > 0)
    GCHK(gemm(ga(d4, 1, 512, col[3], cv[3].crs(), 1, g + off[3], 512, cv[3].crs(), B)));               // gW4 = d4^T @ x4
    GCHK(gemm(ga(d4, 512, 1, theta + off[3], cv[3].crs(), 1, dcol, B, cv[3].crs(), 512)));             // dx4 = d4 @ W4
    { const int64_t total = mrows(2, B) * cv[2].K;
      hipLaunchKernelGGL(col2im_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, (const T*)dcol, (const T*)act[2], dact[2], total, cv[3]); }
    for (int l = 2; l >= 0; --l) {
      const ConvGeom& c = cv[l];
      const int64_t m = mrows(l, B);
      GCHK(gemm(ga(col[l], 1, c.crs(), dact[l], c.K, 1, g + off[l], c.crs(), c.K, (int)m)));           // gW = cols^T @ delta (sum over the batch)
      if (l > 0) {                                                                                     // conv1 computes no input gradient
        GCHK(gemm(ga(dact[l], c.K, 1, theta + off[l], 1, c.K, dcol, (int)m, c.crs(), c.K)));           // dcols = delta @ W^T
        const int64_t total = mrows(l - 1, B) * cv[l - 1].K;
        hipLaunchKernelGGL(col2im_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, (const T*)dcol, (const T*)act[l - 1], dact[l - 1], total, c);
      }
    }
    // ---- optimizer :165
    OptArgs<T> u; u.w = theta; u.s1 = s1; u.s2 = s2; u.g = g; u.n = NP; u.opt = cfg.optimizer;
    u.bsz = (T)B; u.rho = (T)cfg.decay_rate; u.omr = (T)(1.0 - cfg.decay_rate); u.lr = (T)cfg.learning_rate; u.eps = (T)cfg.epsilon;
    u.b1 = (T)cfg.beta_1; u.omb1 = (T)(1.0 - cfg.beta_1); u.b2 = (T)cfg.beta_2; u.omb2 = (T)(1.0 - cfg.beta_2);
    const double tt = (double)epoch + 1.0;
    u.lr_t = (T)(cfg.learning_rate * sqrt(1.0 - pow(cfg.beta_2, tt)) / (1.0 - pow(cfg.beta_1, tt)));
    hipLaunchKernelGGL(update_kernel<T>, dim3(grid_for(NP)), dim3(256), 0, st, u);
    return hipGetLastError();
  }
  hipError_t upload_meta(const uint8_t* actions, const int64_t* rew, const uint8_t* term) {
    memcpy(small_host.data(), rew, (size_t)B * 8); memcpy(small_host.data() + (size_t)B * 8, actions, B); memcpy(small_host.data() + (size_t)B * 9, term, B);
    GCHK(hipMemcpyAsync(st_small, small_host.data(), (size_t)B * 10, hipMemcpyHostToDevice, st));
    return hipStreamSynchronize(st);                 // the caller's arrays (and small_host) are free once this returns, like the reference's
  }
  hipError_t train_host(const uint8_t* pre, const uint8_t* actions, const int64_t* rew, const uint8_t* post, const uint8_t* term, int epoch) override {
    const size_t sb = (size_t)B * state_bytes();
    GCHK(hipMemcpyAsync(st_states, pre, sb, hipMemcpyHostToDevice, st));
    GCHK(hipMemcpyAsync(st_states + sb, post, sb, hipMemcpyHostToDevice, st));
    GCHK(upload_meta(actions, rew, term));
    return train_dev(st_states, st_states + sb, st_small + (size_t)B * 8, reinterpret_cast<const int64_t*>(st_small), st_small + (size_t)B * 9, epoch);
  }
  hipError_t train_dev_host_meta(const uint8_t* pre_dev, const uint8_t* post_dev, const uint8_t* actions, const int64_t* rew, const uint8_t* term, int epoch) override {
    GCHK(upload_meta(actions, rew, term));
    return train_dev(pre_dev, post_dev, st_small + (size_t)B * 8, reinterpret_cast<const int64_t*>(st_small), st_small + (size_t)B * 9, epoch);
  }

  template <typename U> static void conv_out(const std::vector<T>& v, void* host) { U* o = (U*)host; for (size_t i = 0; i < v.size(); ++i) o[i] = (U)v[i]; }
  hipError_t fetch(const T* dev, int64_t n, void* host, bool f64) {
    std::vector<T> tmp((size_t)n);
    GCHK(hipMemcpyAsync(tmp.data(), dev, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, st));
    GCHK(hipStreamSynchronize(st));
    if (f64) conv_out<double>(tmp, host); else conv_out<float>(tmp, host);
    return hipSuccess;
  }
  T* which_buf(int which) { switch (which) { case 0: return theta; case 1: return theta_t; case 2: return s1; case 3: return g; case 4: return s2; default: return nullptr; } }
  hipError_t set_param(int which, int layer, const void* host, bool f64) override {
    T* b = which_buf(which); if (!b || which == 3 || layer < 0 || layer > 4) return hipErrorInvalidValue;
    const int64_t n = rows[layer] * cols[layer];
    std::vector<T> tmp((size_t)n);
    if (f64) { const double* s = (const double*)host; for (int64_t i = 0; i < n; ++i) tmp[(size_t)i] = (T)s[i]; }
    else { const float* s = (const float*)host; for (int64_t i = 0; i < n; ++i) tmp[(size_t)i] = (T)s[i]; }
    GCHK(hipStreamSynchronize(st));
    GCHK(hipMemcpy(b + off[layer], tmp.data(), (size_t)n * sizeof(T), hipMemcpyHostToDevice));
    return hipSuccess;
  }
  hipError_t get_param(int which, int layer, void* host, bool f64) override {
    T* b = which_buf(which); if (!b || layer < 0 || layer > 4) return hipErrorInvalidValue;
    return fetch(b + off[layer], rows[layer] * cols[layer], host, f64);
  }
  hipError_t predict_dev(const uint8_t* states_dev, int n, void* q_host, bool f64) override {
    if (n < 1 || n > B) return hipErrorInvalidValue;
    GCHK(forward(0, states_dev, n));
    return fetch(q, (int64_t)n * A, q_host, f64);                              // (n, A): deepqnetwork.py:186 qvalues.T
  }
  hipError_t predict_host(const uint8_t* states_host, int n, void* q_host, bool f64) override {
    if (n < 1 || n > B) return hipErrorInvalidValue;
    GCHK(hipMemcpyAsync(st_states, states_host, (size_t)n * state_bytes(), hipMemcpyHostToDevice, st));
    return predict_dev(st_states, n, q_host, f64);
  }
  hipError_t read_cost(double* c) override { T v; GCHK(hipMemcpyAsync(&v, cost, sizeof(T), hipMemcpyDeviceToHost, st)); GCHK(hipStreamSynchronize(st)); *c = (double)v; return hipSuccess; }
  hipError_t reset_cost_sum() override { return hipMemsetAsync(cost_sum, 0, 8, st); }
  hipError_t read_cost_sum(double* s) override { GCHK(hipMemcpyAsync(s, cost_sum, 8, hipMemcpyDeviceToHost, st)); return hipStreamSynchronize(st); }
  hipError_t last_q(void* preq, void* maxpostq, bool f64) override {
    if (preq) GCHK(fetch(q, (int64_t)B * A, preq, f64));
    if (maxpostq) GCHK(fetch(maxq, B, maxpostq, f64));
    return hipSuccess;
  }
  hipError_t update_target() override {
    if (theta_t != theta) return hipMemcpyAsync(theta_t, theta, (size_t)NP * sizeof(T), hipMemcpyDeviceToDevice, st);
    return hipSuccess;
  }
};

template <typename T>
GenericNet* make_t(const sdqn_net_cfg& c, hipStream_t s, std::string* err) {
  GenericNetT<T>* n = new GenericNetT<T>();
  n->cfg = c; n->st = s; n->B = c.batch_size; n->A = c.num_actions; n->hist = c.history_length; n->H = c.screen_height; n->W = c.screen_width;
  hipError_t e = n->init(err);
  if (e != hipSuccess) { if (err->empty()) *err = std::string("generic network: ") + hipGetErrorString(e); delete n; return nullptr; }
  return n;
}

__global__ void __launch_bounds__(256) gather_generic_kernel(const GatherGenericArgs g) {
  const int k = blockIdx.y, which = blockIdx.x / g.hist, j = blockIdx.x % g.hist;
  const int64_t index = g.idx[k];
  const uint8_t* src = g.ring + (index - g.hist + j + which) * g.frame;          // pre: frames idx-hist .. idx-1, post: idx-hist+1 .. idx
  uint8_t* dst = (which ? g.post : g.pre) + ((int64_t)k * g.hist + j) * g.frame;
  if ((g.frame & 3) == 0 && ((reinterpret_cast<uintptr_t>(g.ring) | reinterpret_cast<uintptr_t>(dst)) & 3) == 0) {
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src); uint32_t* d4 = reinterpret_cast<uint32_t*>(dst);
    for (int64_t i = threadIdx.x; i < g.frame / 4; i += 256) d4[i] = s4[i];
  } else {
    for (int64_t i = threadIdx.x; i < g.frame; i += 256) dst[i] = src[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {                                     // replay_memory.py:76-78
    const MetaRec rec = reinterpret_cast<const MetaRec*>(g.meta)[index];
    g.actions[k] = rec.action; g.rewards[k] = rec.reward; g.terminals[k] = rec.terminal;
  }
}

}  // namespace

GenericNet* make_generic_net(const sdqn_net_cfg& c, hipStream_t s, std::string* err) {
  err->clear();
  if (c.datatype == 2) return make_t<double>(c, s, err);
  return make_t<float>(c, s, err);
}

hipError_t launch_gather_generic(const GatherGenericArgs& g, hipStream_t s) {
  hipLaunchKernelGGL(gather_generic_kernel, dim3(2 * g.hist, g.B), dim3(256), 0, s, g);
  return hipGetLastError();
}

}  // namespace sdqn
