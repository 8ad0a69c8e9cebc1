// Experiment: what does a cross-stream dependency cost on the MAIN stream when it is (almost always) already satisfied?
// Main stream: STEPS x 10 kernels of ~5 us.  Side stream: STEPS x 4 kernels of ~5 us (independent work, e.g. a hoisted
// target-net forward).  Variants: (a) main only; (b) both streams, no dependencies; (c) main waits once per step for the
// side chain of the PREVIOUS step (event recorded on the side stream); (d) as (c) + the side stream waits once per step
// for an event of the main stream (so it cannot run more than one step ahead).
//   hipcc --offload-arch=gfx950 -O3 -o two_stream_cost two_stream_cost.hip && ./two_stream_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
__global__ void spin(long long ticks) { const unsigned long long t0 = wall_clock64(); while ((long long)(wall_clock64() - t0) < ticks) {} }
int main() {
  hipStream_t m, s; hipStreamCreateWithFlags(&m, hipStreamNonBlocking); hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  const long long us = rate / 1000; const int STEPS = 2000;
  std::vector<hipEvent_t> es(STEPS), em(STEPS);
  for (auto& e : es) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  for (auto& e : em) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  for (int variant = 0; variant < 4; ++variant) {
    for (int rep = 0; rep < 2; ++rep) {
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < STEPS; ++i) {
        if (variant >= 1) {
          if (variant == 3 && i > 0) hipStreamWaitEvent(s, em[i - 1], 0);
          for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(spin, dim3(100), dim3(256), 0, s, 5 * us);
          hipEventRecord(es[i], s);
        }
        for (int k = 0; k < 10; ++k) {
          if (k == 4 && variant >= 2 && i > 0) hipStreamWaitEvent(m, es[i - 1], 0);
          hipLaunchKernelGGL(spin, dim3(156), dim3(256), 0, m, 5 * us);
          if (k == 4 && variant == 3) hipEventRecord(em[i], m);
        }
      }
      hipStreamSynchronize(m); hipStreamSynchronize(s);
      double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (rep == 1) printf("variant %d: %.2f us per step of the main stream (10 x 5 us kernels%s)\n", variant, el / STEPS * 1e6,
                           variant == 0 ? "" : variant == 1 ? " + 4 side kernels, no deps" : variant == 2 ? " + side, main waits once" : " + side, both wait once");
    }
  }
  return 0;
}
