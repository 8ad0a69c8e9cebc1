// icache_chain.hip — does a dependent chain of DIFFERENT kernels cost more per launch than the same kernel repeated?
// (tools/phase_timing.py: the first launch of a kernel after other kernels reaches its first load ~2 k cycles later than a repeated
// launch.)  K distinct instantiations of one template, each NOPS straight-line dependent VALU instructions (unique code, executed once by
// every wave), launched round-robin as ONE hipGraph of 200 dependent launches; 256 workgroups x 64 threads.
//   hipcc --offload-arch=gfx950 -O3 -o icache_chain icache_chain.hip && ./icache_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ID, int NOPS>
__global__ void __launch_bounds__(64) body(float* out, float seed) {
  float v = seed + threadIdx.x;
#pragma unroll
  for (int i = 0; i < NOPS; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v) : "v"(seed + (float)(ID * 131 + i)));   // 8-byte encodings (literal operand differs per op)
  if (v == 12345.678f) out[blockIdx.x] = v;
}
typedef void (*kfn)(float*, float);
template <int NOPS, int... IDs> static std::vector<kfn> table(std::integer_sequence<int, IDs...>) { return {body<IDs, NOPS>...}; }

template <int NOPS>
static int run(hipStream_t s, float* d, int K, const char* what) {
  static std::vector<kfn> fns = table<NOPS>(std::make_integer_sequence<int, 48>());
  const int L = 240;
  hipGraph_t g; hipGraphExec_t ge;
  CHK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < L; ++i) hipLaunchKernelGGL(fns[i % K], dim3(256), dim3(64), 0, s, d, 1.0f);
  CHK(hipStreamEndCapture(s, &g));
  CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) CHK(hipGraphLaunch(ge, s));
  CHK(hipEventRecord(e0, s));
  const int R = 20;
  for (int r = 0; r < R; ++r) CHK(hipGraphLaunch(ge, s));
  CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-28s %2d distinct kernels x %5d B of code each (%4d KB in the cycle): %.3f us per launch\n", what, K, NOPS * 8, K * NOPS * 8 / 1024, ms * 1e3 / (R * L));
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return 0;
}
int main() {
  hipStream_t s; CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float* d; CHK(hipMalloc(&d, 4096));
  for (int K : {1, 2, 4, 8, 12, 16, 24, 48}) if (run<256>(s, d, K, "2 KB bodies")) return 1;
  for (int K : {1, 2, 4, 8, 12, 16, 24, 48}) if (run<768>(s, d, K, "6 KB bodies")) return 1;
  for (int K : {1, 2, 4, 8, 12, 16}) if (run<2048>(s, d, K, "16 KB bodies")) return 1;
  return 0;
}
