// Experiment (not product): conv_ssh.h's forward chain standalone on synthetic buffers; -DSSH_ABL=1: no matrix work (loads, LDS traffic of the
// staging, stores only).   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off [-DSSH_ABL=1] -I simple_dqn_amd/csrc -o ssh_bench tools/exp/ssh_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "conv_ssh.h"
namespace sdqn { LaunchEvents& launch_events() { static thread_local LaunchEvents e; return e; } }
using namespace sdqn;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NS> int bench(int B) {
  const int nz = 2;
  ssh::h_t *a1, *a2, *a3, *w;
  const size_t n1 = (size_t)nz * B * 400 * 32, n2 = (size_t)nz * B * 81 * 64, n3 = (size_t)nz * B * 49 * 64, nw = 64 * 512 + 64 * 576;
  CHK(hipMalloc(&a1, n1 * 2)); CHK(hipMalloc(&a2, n2 * 2)); CHK(hipMalloc(&a3, n3 * 2)); CHK(hipMalloc(&w, 2 * nw * 2));
  std::vector<ssh::h_t> h(n1); for (size_t i = 0; i < n1; ++i) h[i] = (ssh::h_t)((float)((i * 2654435761u) >> 20 & 1023) / 1024.0f);
  CHK(hipMemcpy(a1, h.data(), n1 * 2, hipMemcpyHostToDevice));
  std::vector<ssh::h_t> hw(2 * nw); for (size_t i = 0; i < 2 * nw; ++i) hw[i] = (ssh::h_t)((float)((int)((i * 40503u) >> 8 & 255) - 128) / 2048.0f);
  CHK(hipMemcpy(w, hw.data(), 2 * nw * 2, hipMemcpyHostToDevice));
  ssh::Args c; c.a1 = a1; c.a2 = a2; c.a3 = a3; c.B = B; c.G = (B + NS - 1) / NS;
  c.w2[0] = w; c.w3[0] = w + 64 * 512; c.w2[1] = w + nw; c.w3[1] = w + nw + 64 * 512;
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int r = 0; r < 3; ++r) CHK((ssh::launch_chain<NS, true, false>(c, nz, 0)));
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  for (int r = 0; r < 50; ++r) CHK((ssh::launch_chain<NS, true, false>(c, nz, 0)));
  CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  printf("forward chain B %d NS %d abl %d: %6.2f us/launch (back to back)\n", B, NS,
#ifdef SSH_ABL
         SSH_ABL,
#else
         0,
#endif
         ms * 1e3 / 50);
  hipFree(a1); hipFree(a2); hipFree(a3); hipFree(w);
  return 0;
}
int main() { return bench<2>(256) || bench<1>(32) || bench<1>(128); }
