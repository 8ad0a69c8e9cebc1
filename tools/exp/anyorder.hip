// Experiment: does hipExtAnyOrderLaunch (AQL packet without the barrier bit) let a kernel of the SAME stream start while its
// predecessor is still running on gfx950 / ROCm 7?  A spins ~100 us, B (any-order) ~5 us, C (ordinary) ~5 us.
//   hipcc --offload-arch=gfx950 -O3 -o anyorder anyorder.hip && ./anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
__global__ void spin(unsigned long long* out, int slot, long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while ((long long)(wall_clock64() - t0) < ticks) {}
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[2 * slot] = t0; out[2 * slot + 1] = wall_clock64(); }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64 * 8); hipMemset(d, 0, 64 * 8);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);      // kHz
  const long long us = rate / 1000;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s, d, 0, 100 * us);
    hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d, 1, 5 * us);
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, 2, 5 * us);
    hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d, 3, 5 * us);
    hipStreamSynchronize(s);
  }
  unsigned long long h[8]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const double k = 1000.0 / rate;
  printf("wall clock %d kHz\nA start 0.0 end %.1f us\nB(any-order) start %.1f end %.1f\nC(ordered) start %.1f end %.1f\nD(any-order after C) start %.1f end %.1f\n",
         rate, (h[1] - h[0]) * k, (double)(long long)(h[2] - h[0]) * k, (double)(long long)(h[3] - h[0]) * k, (double)(long long)(h[4] - h[0]) * k,
         (double)(long long)(h[5] - h[0]) * k, (double)(long long)(h[6] - h[0]) * k, (double)(long long)(h[7] - h[0]) * k);
  printf("%s\n", h[2] < h[1] ? "ANY-ORDER WORKS: B started while A was running" : "any-order launch is serialised like an ordinary one");
  return 0;
}
