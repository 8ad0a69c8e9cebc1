// emul.cpp — TEST-ONLY host executor of the problem definitions in simple_dqn_amd/csrc/problems.h.
// It runs every GEMM-shaped stage of the step with naive loops through the very same
// a_row/a_col/b_row/b_col/store index functions the HIP tile engine uses, so the index math and the
// Neon<->internal layout converters can be validated against the oracle on a machine without a GPU.
// It is NOT part of the product: nothing under simple_dqn_amd/ links or loads it.
#include <stdint.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../../simple_dqn_amd/csrc/problems.h"

using namespace sdqn;

template <class P>
static void run(const StepArgs& a) {
  const int M = P::M(a), N = P::N(a);
  for (int bz = 0; bz < P::nbz(a); ++bz) {
    int z, ks, kb, ke; P::ksplit(a, bz, z, ks, kb, ke);
    std::vector<typename P::aoff_t> ar(M), ac(std::max(ke - kb, 0));
    std::vector<int> br(std::max(ke - kb, 0)), bc(N);
    for (int m = 0; m < M; ++m) ar[m] = P::a_row(a, z, m);
    for (int k = kb; k < ke; ++k) { ac[k - kb] = P::a_col(a, z, k); br[k - kb] = P::b_row(a, z, k); }
    for (int n = 0; n < N; ++n) bc[n] = P::b_col(a, z, n);
    std::vector<float> arow(std::max(ke - kb, 0));
    for (int m = 0; m < M; ++m) {
      for (int k = kb; k < ke; ++k) arow[k - kb] = P::a_load(a, z, ar[m] + ac[k - kb]);
      for (int n = 0; n < N; ++n) {
        float acc = 0.f;
        for (int k = kb; k < ke; ++k) acc += arow[k - kb] * P::b_load(a, z, br[k - kb] + bc[n]);
        P::store(a, z, ks, m, n, acc);
      }
    }
  }
}

extern "C" void emul_div_bsz(const float* x, float bsz, float* out, int n) { for (int i = 0; i < n; ++i) out[i] = div_bsz(x[i], bsz); }
extern "C" void emul_norm_u8(float* out256) { for (int i = 0; i < 256; ++i) out256[i] = norm_u8((uint32_t)i); }

extern "C" int emul_step(int B, int A, const float* const* w_online /*5, Neon*/, const float* const* w_target,
                         const uint8_t* pre, const uint8_t* act, const int64_t* rew, const uint8_t* post,
                         const uint8_t* term, double discount, double clip, double minr, double maxr,
                         float* q_out /*[2][B][A]*/, float* const* g_out /*5, Neon*/, float* cost_out) {
  const int64_t NP = OFF5 + (int64_t)A * NFC;
  std::vector<float> th(NP), tt(NP), g(NP, 0.f);
  for (int l = 0; l < 5; ++l) {
    int64_t rows, cols, off; layer_dims(l, A, rows, cols, off);
    for (int64_t r = 0; r < rows; ++r) for (int64_t c = 0; c < cols; ++c) {
      th[off + neon_to_internal(l, r, c)] = w_online[l][r * cols + c];
      tt[off + neon_to_internal(l, r, c)] = w_target[l][r * cols + c];
    }
  }
  std::vector<uint8_t> st((size_t)2 * B * STATE);
  memcpy(st.data(), pre, (size_t)B * STATE); memcpy(st.data() + (size_t)B * STATE, post, (size_t)B * STATE);
  StepArgs a; memset(&a, 0, sizeof a);
  a.src = st.data(); a.from_ring = 0; a.B = B; a.A = A; a.nz = 2; a.theta[0] = th.data(); a.theta[1] = tt.data();
  a.S4 = 7; a.tps1 = 3; a.tps2 = 2; a.tps3 = 2;
  std::vector<float> a1((size_t)2 * B * PIX1 * K1), a2((size_t)2 * B * PIX2 * K2), a3((size_t)2 * B * PIX3 * K3),
      slab4((size_t)a.S4 * 2 * B * NFC), a4((size_t)2 * B * NFC), d4((size_t)B * NFC), d3p((size_t)B * PD3 * PD3 * K3, 0.f),
      d2p((size_t)B * PD2 * PD2 * K2, 0.f), d1((size_t)B * PIX1 * K1), d3((size_t)B * PIX3 * K3), d2((size_t)B * PIX2 * K2);
  const int ns1 = Conv1Wgrad::nbz(a), ns2 = Conv2Wgrad::nbz(a), ns3 = Conv3Wgrad::nbz(a);
  std::vector<float> s1((size_t)ns1 * NW1), s2((size_t)ns2 * NW2), s3((size_t)ns3 * NW3);
  a.a1 = a1.data(); a.a2 = a2.data(); a.a3 = a3.data(); a.slab4 = slab4.data(); a.a4 = a4.data(); a.d4 = d4.data();
  a.d3p = d3p.data(); a.d2p = d2p.data(); a.d3 = d3.data(); a.d2 = d2.data(); a.d1 = d1.data(); a.g = g.data(); a.slab1 = s1.data(); a.slab2 = s2.data(); a.slab3 = s3.data();
  run<Conv1Fwd>(a); run<Conv2Fwd>(a); run<Conv3Fwd>(a); run<Fc4Fwd>(a);
  // head (mirrors head_kernel in sdqn_kernels.hip)
  std::vector<float> dq((size_t)B * A, 0.f);
  double cost = 0;
  for (int n = 0; n < B; ++n) {
    float q[2][MAX_ACTIONS];
    for (int z = 0; z < 2; ++z) {
      for (int j = 0; j < NFC; ++j) {
        float v = 0; for (int s = 0; s < a.S4; ++s) v += slab4[(((size_t)s * 2 + z) * B + n) * NFC + j];
        a4[((size_t)z * B + n) * NFC + j] = v > 0 ? v : 0;
      }
      for (int k = 0; k < A; ++k) {
        float s = 0; for (int j = 0; j < NFC; ++j) s += a.theta[z][OFF5 + k * NFC + j] * a4[((size_t)z * B + n) * NFC + j];
        q[z][k] = s; q_out[((size_t)z * B + n) * A + k] = s;
      }
    }
    float m = q[1][0]; for (int k = 1; k < A; ++k) m = std::max(m, q[1][k]);
    double rr = (double)rew[n]; rr = rr < minr ? minr : (rr > maxr ? maxr : rr);
    double y = term[n] ? rr : rr + discount * (double)m;
    float d = q[0][act[n]] - (float)y;
    cost += 0.5f * d * d;
    float dc = d; if (clip != 0) dc = std::min(std::max(d, (float)-clip), (float)clip);
    dq[(size_t)n * A + act[n]] = dc;
    for (int j = 0; j < NFC; ++j) d4[(size_t)n * NFC + j] = a4[(size_t)n * NFC + j] > 0 ? a.theta[0][OFF5 + act[n] * NFC + j] * dc : 0.f;
  }
  *cost_out = (float)(cost / B);
  run<Fc4Dgrad>(a); run<Fc4Wgrad>(a); run<Conv3Dgrad>(a); run<Conv3Wgrad>(a);
  run<Conv2Dgrad>(a); run<Conv2Wgrad>(a); run<Conv1Wgrad>(a);
  for (int i = 0; i < NW1; ++i) { float s = 0; for (int k = 0; k < ns1; ++k) s += s1[(size_t)k * NW1 + i]; g[OFF1 + i] = s; }
  for (int i = 0; i < NW2; ++i) { float s = 0; for (int k = 0; k < ns2; ++k) s += s2[(size_t)k * NW2 + i]; g[OFF2 + i] = s; }
  for (int i = 0; i < NW3; ++i) { float s = 0; for (int k = 0; k < ns3; ++k) s += s3[(size_t)k * NW3 + i]; g[OFF3 + i] = s; }
  for (int k = 0; k < A; ++k) for (int j = 0; j < NFC; ++j) {
    float s = 0; for (int n = 0; n < B; ++n) s += dq[(size_t)n * A + k] * a4[(size_t)n * NFC + j];
    g[OFF5 + k * NFC + j] = s;
  }
  for (int l = 0; l < 5; ++l) {
    int64_t rows, cols, off; layer_dims(l, A, rows, cols, off);
    for (int64_t r = 0; r < rows; ++r) for (int64_t c = 0; c < cols; ++c) g_out[l][r * cols + c] = g[off + neon_to_internal(l, r, c)];
  }
  return 0;
}
