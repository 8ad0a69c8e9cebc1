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
#include "../../simple_dqn_amd/csrc/bt_map.h"

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

// ---- the block-tile engine's maps (bt_map.h) executed on the host ------------------------------------------------------------
// Mirrors gemm_engine_bt.h's bt_tile step by step with a simulated workgroup: 256 "threads" fill the two LDS panels through the
// loader item maps, 4 waves x 64 lanes fetch their MFMA fragments through frag_off(), the MFMA is the documented lane semantics
// (lane l feeds A[i = l & 31][k-slot l >> 5] and B[k-slot l >> 5][j = l & 31]; register r of lane l is C[acc_row(r, l >> 5)][l & 31]),
// the epilogue goes through P::store with the engine's sub-tile -> (m, n) map.  Any hole in the loader coverage, any disagreement of
// the A and B k-slot maps, any wrong panel offset shows up as a wrong gradient against the oracle.
template <class P, int BM, int BN, int WM, int WN>
static void run_bt(const StepArgs& a) {
  using namespace sdqn::bt;
  constexpr bool AK = P::A_K, BKC = P::B_K;
  constexpr int SM = BM / (32 * WM), SN = BN / (32 * WN), PA = passes(BM), PB = passes(BN);
  const int M = P::M(a), N = P::N(a);
  const int gx = (M + BM - 1) / BM, gy = (N + BN - 1) / BN;
  std::vector<float> As(panel_floats(AK, BM)), Bs(panel_floats(BKC, BN));
  for (int bz = 0; bz < P::nbz(a); ++bz) {
    int z, ks, kbeg, kend; P::ksplit(a, bz, z, ks, kbeg, kend);
    const int nch = (kend - kbeg + BK - 1) / BK;
    for (int by = 0; by < gy; ++by) for (int bx = 0; bx < gx; ++bx) {
      const int m0 = bx * BM, n0 = by * BN;
      std::vector<float> acc((size_t)4 * SM * SN * 64 * 16, 0.f);          // [wave][sm][sn][lane][r]
      for (int c = 0; c < nch; ++c) {
        const int kc = kbeg + c * BK;
        std::fill(As.begin(), As.end(), -1e30f); std::fill(Bs.begin(), Bs.end(), -1e30f);     // a missed slot poisons the result
        for (int tid = 0; tid < NT; ++tid) {
          for (int p = 0; p < PA; ++p) {
            f4 v; int off;
            if (AK) {
              const int m = m0 + km_item_row(tid, p), k = kc + km_item_k(tid);
              const typename P::aoff_t ar = P::a_row(a, z, m < M ? m : M - 1);
              v = P::a_load4(a, z, ar + P::a_col(a, z, k < kend ? k : kbeg));
              if (k >= kend) v.x = v.y = v.z = v.w = 0.f;
              off = km_off(km_item_row(tid, p), km_item_k(tid));
            } else {
              const int m = m0 + mk_item_x(BM, tid), k = kc + mk_item_k(BM, tid, p);
              const typename P::aoff_t ar = P::a_row(a, z, m + 4 <= M ? m : M - 4);
              v = P::a_load4(a, z, ar + P::a_col(a, z, k < kend ? k : kbeg));
              if (k >= kend) v.x = v.y = v.z = v.w = 0.f;
              off = mk_off(BM, mk_item_k(BM, tid, p), mk_item_x(BM, tid));
            }
            As[off] = v.x; As[off + 1] = v.y; As[off + 2] = v.z; As[off + 3] = v.w;
          }
          for (int p = 0; p < PB; ++p) {
            f4 v; int off;
            if (BKC) {
              const int n = n0 + km_item_row(tid, p), k = kc + km_item_k(tid);
              const int bc = P::b_col(a, z, n < N ? n : N - 1);
              v = P::b_load4(a, z, bc + P::b_row(a, z, k < kend ? k : kbeg));
              if (k >= kend) v.x = v.y = v.z = v.w = 0.f;
              off = km_off(km_item_row(tid, p), km_item_k(tid));
            } else {
              const int n = n0 + mk_item_x(BN, tid), k = kc + mk_item_k(BN, tid, p);
              const int bc = P::b_col(a, z, n + 4 <= N ? n : N - 4);
              v = P::b_load4(a, z, bc + P::b_row(a, z, k < kend ? k : kbeg));
              if (k >= kend) v.x = v.y = v.z = v.w = 0.f;
              off = mk_off(BN, mk_item_k(BN, tid, p), mk_item_x(BN, tid));
            }
            Bs[off] = v.x; Bs[off + 1] = v.y; Bs[off + 2] = v.z; Bs[off + 3] = v.w;
          }
        }
        for (int wave = 0; wave < 4; ++wave) {
          const int wm = wave / WN, wn = wave % WN;
          for (int sm = 0; sm < SM; ++sm) for (int sn = 0; sn < SN; ++sn) {
            float* ac = &acc[((((size_t)wave * SM + sm) * SN + sn) * 64) * 16];
            for (int t = 0; t < 16; ++t) {
              float fa[64], fb[64];
              for (int l = 0; l < 64; ++l) {
                const int i = l & 31, h = l >> 5;
                fa[l] = As[frag_off(AK, BM, (wm * SM + sm) * 32 + i, t, h)];
                fb[l] = Bs[frag_off(BKC, BN, (wn * SN + sn) * 32 + i, t, h)];
              }
              for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
                const int row = acc_row(r, l >> 5), col = l & 31;
                float d = ac[l * 16 + r];
                d += fa[row] * fb[col];                 // k-slot 0
                d += fa[row + 32] * fb[col + 32];       // k-slot 1
                ac[l * 16 + r] = d;
              }
            }
          }
        }
      }
      for (int wave = 0; wave < 4; ++wave) {
        const int wm = wave / WN, wn = wave % WN;
        for (int sm = 0; sm < SM; ++sm) for (int sn = 0; sn < SN; ++sn) {
          const int ms = m0 + (wm * SM + sm) * 32, ns = n0 + (wn * SN + sn) * 32;
          const float* ac = &acc[((((size_t)wave * SM + sm) * SN + sn) * 64) * 16];
          for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
            const int m = ms + acc_row(r, l >> 5), n = ns + (l & 31);
            if (m < M && n < N) P::store(a, z, ks, m, n, ac[l * 16 + r]);
          }
        }
      }
    }
  }
}

static int g_bt_variant = 0;       // 0: naive loops; 1 / 2: the block-tile maps at the built-in / the alternative block shapes
extern "C" void emul_set_bt(int v) { g_bt_variant = v; }
template <class P, int BM1, int BN1, int WM1, int WN1, int BM2, int BN2, int WM2, int WN2>
static void run_any(const StepArgs& a) {
  if (g_bt_variant == 1) run_bt<P, BM1, BN1, WM1, WN1>(a);
  else if (g_bt_variant == 2) run_bt<P, BM2, BN2, WM2, WN2>(a);
  else run<P>(a);
}

// XCD-contiguous workgroup -> tile maps (problems.h): both must be bijections for every grid size / every sub-range of a multi-problem launch
extern "C" int emul_xcd_tile_id(int b, int nwg) { return xcd_tile_id(b, nwg); }
extern "C" int emul_xcd_tile_id_range(int b, int s, int n) { return xcd_tile_id_range(b, s, n); }
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
  run<Conv1Fwd>(a);
  run_any<Conv2Fwd, 64, 64, 2, 2, 128, 64, 4, 1>(a); run_any<Conv3Fwd, 64, 64, 2, 2, 128, 64, 2, 2>(a); run_any<Fc4Fwd, 64, 64, 2, 2, 128, 128, 2, 2>(a);
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
  run_any<Fc4Dgrad, 64, 64, 2, 2, 32, 128, 1, 4>(a); run_any<Fc4Wgrad, 64, 64, 2, 2, 64, 128, 2, 2>(a);
  run_any<Conv3Dgrad, 64, 64, 2, 2, 128, 64, 2, 2>(a); run_any<Conv3Wgrad, 64, 64, 2, 2, 128, 64, 2, 2>(a);
  run_any<Conv2Dgrad, 128, 32, 4, 1, 256, 32, 4, 1>(a); run_any<Conv2Wgrad, 64, 64, 2, 2, 128, 64, 2, 2>(a); run<Conv1Wgrad>(a);
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
