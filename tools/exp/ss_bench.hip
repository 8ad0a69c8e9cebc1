// Experiment (not product): conv_ss.h's kernels standalone on synthetic buffers, with the timing build's stamps and compile-time ablations.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DSDQN_TIMING [-DSS_ABL=n] -I simple_dqn_amd/csrc -o ss_bench tools/exp/ss_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "conv_ss.h"
namespace sdqn { LaunchEvents& launch_events() { static thread_local LaunchEvents e; return e; } }
using namespace sdqn;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class C>
int bench(const char* name, int B) {
  const int nz = 2, G = (B + C::NS - 1) / C::NS;
  const size_t nin = (size_t)nz * B * C::HI * C::WI * C::CI, nout = (size_t)nz * B * C::NPOS * ss::NO, nw = (size_t)C::NCH * C::CI * ss::NO;
  float *in, *out, *w; unsigned long long* dbg;
  CHK(hipMalloc(&in, nin * 4)); CHK(hipMalloc(&out, nout * 4)); CHK(hipMalloc(&w, 2 * nw * 4)); CHK(hipMalloc(&dbg, (size_t)nz * G * 64));
  std::vector<float> h(nin); for (size_t i = 0; i < nin; ++i) h[i] = (float)((i * 2654435761u) >> 20 & 1023) / 1024.0f;
  CHK(hipMemcpy(in, h.data(), nin * 4, hipMemcpyHostToDevice));
  std::vector<float> hw(2 * nw); for (size_t i = 0; i < 2 * nw; ++i) hw[i] = (float)((int)((i * 40503u) >> 8 & 255) - 128) / 2048.0f;
  CHK(hipMemcpy(w, hw.data(), 2 * nw * 4, hipMemcpyHostToDevice));
  CHK(hipMemset(dbg, 0, (size_t)nz * G * 64));
  CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_sdqn_dbg), &dbg, sizeof dbg));
  ss::Args c; c.in = in; c.out = out; c.w[0] = w; c.w[1] = w + nw; c.B = B; c.G = G; c.wt = getenv("SS_WT") ? atoi(getenv("SS_WT")) : 1; c.dbg = getenv("SS_DBG") ? atoi(getenv("SS_DBG")) : 0;
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int r = 0; r < 3; ++r) CHK(ss::launch<C>(c, nz, 0));
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  for (int r = 0; r < 20; ++r) CHK(ss::launch<C>(c, nz, 0));
  CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> s((size_t)nz * G * 8);
  CHK(hipMemcpy(s.data(), dbg, s.size() * 8, hipMemcpyDeviceToHost));
  auto med = [&](int a, int b) { std::vector<long long> v; for (int i = 0; i < nz * G; ++i) v.push_back((long long)(s[i * 8 + b] - s[i * 8 + a])); std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  const int mf = 4 * C::NT * C::GR;                   // MFMAs per chunk
  printf("%-10s B %d abl %d: %6.2f us/launch | barrier0 %lld, chunk0 %lld, chunks 1-4 %lld each (%.1f / MFMA), chunks 5..%d %lld each (%.1f / MFMA), final %lld (ideal %d), stg end +%lld\n", name, B,
#ifdef SS_ABL
         SS_ABL,
#else
         0,
#endif
         ms * 1e3 / 20, med(0, 1), med(1, 2), med(2, 3) / 4, med(2, 3) / 4.0 / mf, C::KO - 1, C::KO > 5 ? med(3, 4) / (C::KO - 5) : 0, C::KO > 5 ? med(3, 4) / (double)(C::KO - 5) / mf : 0.0,
         med(4, 5), 32 * 4 * ss::FC * C::GR * C::NT, med(5, 7));
#if defined(SS_ABL) && SS_ABL == 9
  printf("   final phase: K-outer end -> step 0 %lld, pair 0 %lld, pair 1 %lld, pair 2 %lld, rest %lld\n", med(4, 2), med(2, 3), med(3, 6), med(6, 7), med(7, 5));
#endif
  hipFree(in); hipFree(out); hipFree(w); hipFree(dbg);
  return 0;
}

int main() {
  typedef ss::Cfg<P1, Q1, K1, 4, 4, ST2, P2, Q2, 2, 4, 20, 56> C2S2;
  typedef ss::Cfg<P2, Q2, K2, 3, 3, 1, P3, Q3, 2, 8, 48, 16> C3S2;
  if (bench<C2S2>("conv2_fwd", 256)) return 1;
  if (bench<C3S2>("conv3_fwd", 256)) return 1;
  return 0;
}
