// Experiment (not product): how does the duration of a (nearly) empty dependent launch grow with the number of
// workgroups / waves / static LDS?  (wave-launch cost is part of every stage's ramp at B = 32)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int LDS>
__global__ void tiny(float* p) {
  __shared__ float sh[LDS > 0 ? LDS : 1];
  if (LDS > 0) sh[threadIdx.x % LDS] = 1.0f;
  if (threadIdx.x == 0 && blockIdx.x == 0xFFFFFFF) p[0] = sh[0];
}

template <int LDS>
float run(int nb, int nt, float* d, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  float ms = 0;
  for (int w = 0; w < 2; ++w) {
    hipEventRecord(e0, st);
    for (int r = 0; r < 2000; ++r) hipLaunchKernelGGL(tiny<LDS>, dim3(nb), dim3(nt), 0, st, d);
    hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  return ms * 1000.f / 2000;
}

int main() {
  float* d; CK(hipMalloc(&d, 4096));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%8s %8s %8s | %10s %14s %14s\n", "blocks", "threads", "waves", "no LDS us", "32KB LDS us", "64KB LDS us");
  for (int nt : {64, 256, 512, 1024})
    for (int nb : {32, 256, 512, 1024, 2048, 4096, 8192}) {
      if ((long)nb * nt > 8192L * 512) continue;
      printf("%8d %8d %8d | %10.2f %14.2f %14.2f\n", nb, nt, nb * nt / 64, run<0>(nb, nt, d, st, e0, e1), run<8192>(nb, nt, d, st, e0, e1), run<16384>(nb, nt, d, st, e0, e1));
    }
  return 0;
}
