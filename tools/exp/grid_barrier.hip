// Experiment (not product): cost of a grid-wide barrier inside a persistent kernel vs. back-to-back dependent launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ inline void grid_barrier(unsigned* cnt, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { if (++spins > 20000000) break; }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// each "stage": every block reads a slice written by a different block in the previous stage, adds 1, writes its own slice
__global__ void __launch_bounds__(512) persistent(float* buf0, float* buf1, unsigned* cnt, int stages, int per_block) {
  const int nb = gridDim.x;
  float* src = buf0; float* dst = buf1;
  for (int s = 0; s < stages; ++s) {
    const int from = (blockIdx.x * 37 + 11 + s) % nb;
    for (int i = threadIdx.x; i < per_block; i += blockDim.x) dst[(size_t)blockIdx.x * per_block + i] = src[(size_t)from * per_block + i] + 1.0f;
    grid_barrier(cnt, (unsigned)nb * (s + 1));
    float* t = src; src = dst; dst = t;
  }
}
__global__ void __launch_bounds__(512) stage(const float* src, float* dst, int s, int per_block) {
  const int nb = gridDim.x;
  const int from = (blockIdx.x * 37 + 11 + s) % nb;
  for (int i = threadIdx.x; i < per_block; i += blockDim.x) dst[(size_t)blockIdx.x * per_block + i] = src[(size_t)from * per_block + i] + 1.0f;
}

int main() {
  const int per_block = 4096;
  for (int stages : {1, 10, 40}) for (int nb : {256, 512}) {
    float *b0, *b1; unsigned* cnt;
    CK(hipMalloc(&b0, (size_t)nb * per_block * 4)); CK(hipMalloc(&b1, (size_t)nb * per_block * 4)); CK(hipMalloc(&cnt, 4));
    CK(hipMemset(b0, 0, (size_t)nb * per_block * 4)); CK(hipMemset(b1, 0, (size_t)nb * per_block * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t st; CK(hipStreamCreate(&st));
    const int reps = 200;
    float ms;
    // A: separate launches
    for (int w = 0; w < 2; ++w) {
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) for (int s = 0; s < stages; ++s) hipLaunchKernelGGL(stage, dim3(nb), dim3(512), 0, st, (s & 1) ? b1 : b0, (s & 1) ? b0 : b1, s, per_block);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const float us_launch = ms * 1000.f / reps / stages;
    // A2: the same dependent launches captured once into a hipGraph and replayed
    float us_graph = -1.f;
    {
      hipGraph_t graph; hipGraphExec_t exec;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int s = 0; s < stages; ++s) hipLaunchKernelGGL(stage, dim3(nb), dim3(512), 0, st, (s & 1) ? b1 : b0, (s & 1) ? b0 : b1, s, per_block);
      CK(hipStreamEndCapture(st, &graph));
      CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      for (int w = 0; w < 2; ++w) {
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(exec, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      }
      us_graph = ms * 1000.f / reps / stages;
      CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    }
    CK(hipMemset(b0, 0, (size_t)nb * per_block * 4));
    // B: persistent kernel with grid barriers (cooperative launch guarantees co-residency)
    float us_pers = -1.f;
    {
      int st_ = stages, pb = per_block;
      void* args[] = {&b0, &b1, &cnt, &st_, &pb};
      bool ok = true;
      for (int w = 0; w < 2 && ok; ++w) {
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) {
          CK(hipMemsetAsync(cnt, 0, 4, st));
          hipLaunchKernelGGL(persistent, dim3(nb), dim3(512), 0, st, b0, b1, cnt, st_, pb); hipError_t e = hipGetLastError(); (void)args;
          if (e != hipSuccess) { printf("nb=%d cooperative launch refused: %s\n", nb, hipGetErrorString(e)); ok = false; (void)hipGetLastError(); break; }
        }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      }
      if (ok) us_pers = ms * 1000.f / reps / stages;
    }
    std::vector<float> h(4); CK(hipMemcpy(h.data(), b0, 16, hipMemcpyDeviceToHost));
    printf("stages %2d blocks %4d x512 thr, %d KB/block/stage: separate launches %.2f us/stage | hipGraph replay %.2f us/stage | persistent+grid barrier %.2f us/stage (incl. 1/%d of a launch+memset) check %.0f\n",
           stages, nb, per_block * 4 / 1024, us_launch, us_graph, us_pers, stages, h[0]);
    CK(hipFree(b0)); CK(hipFree(b1)); CK(hipFree(cnt));
  }
  return 0;
}
