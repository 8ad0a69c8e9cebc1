// Experiment (not product): a DATAFLOW chain inside ONE launch — stage s's workgroups wait on per-producer flags of stage s-1
// (one flag word per producer workgroup, no read-modify-write on shared addresses, unlike grid_barrier.hip's single counter) —
// against the same chain as S dependent launches.  Workgroup b belongs to stage b / G; the dispatcher hands workgroups out in
// index order per XCD, so every producer is resident or finished before any of its consumers starts (and every poll loop is
// bounded + abortable anyway: a wrong assumption ends the run with "ABORT", not a hung GPU).
//   hipcc --offload-arch=gfx950 -O3 -o flag_chain flag_chain.hip && timeout 60 ./flag_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int D>
__global__ void __launch_bounds__(512) chain(float* buf0, float* buf1, unsigned* flags, unsigned* abort_flag, int G, int per_block, unsigned epoch) {
  const int s = blockIdx.x / G, b = blockIdx.x % G;
  const float* src = (s & 1) ? buf1 : buf0;
  float* dst = (s & 1) ? buf0 : buf1;
  const int from0 = (b * 37 + 11 + s) % G;
  if (s > 0) {
    if (threadIdx.x < D) {
      const unsigned* f = flags + (size_t)(s - 1) * G + (from0 + threadIdx.x) % G;
      long spins = 0;
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 255) == 0 && (spins > 400000 || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
        }
      }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  const int part = per_block / D;
  for (int i = threadIdx.x; i < per_block; i += blockDim.x) {
    const int from = (from0 + i / part) % G;
    dst[(size_t)b * per_block + i] = src[(size_t)from * per_block + i] + 1.0f;
  }
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flags + (size_t)s * G + b, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void __launch_bounds__(512) stage(const float* src, float* dst, int s, int per_block, int D) {
  const int G = gridDim.x, b = blockIdx.x;
  const int from0 = (b * 37 + 11 + s) % G, part = per_block / D;
  for (int i = threadIdx.x; i < per_block; i += blockDim.x) {
    const int from = (from0 + i / part) % G;
    dst[(size_t)b * per_block + i] = src[(size_t)from * per_block + i] + 1.0f;
  }
}

template <int D>
int run(int S, int G, int per_block) {
  float *b0, *b1; unsigned *flags, *ab;
  CK(hipMalloc(&b0, (size_t)G * per_block * 4)); CK(hipMalloc(&b1, (size_t)G * per_block * 4));
  CK(hipMalloc(&flags, (size_t)S * G * 4)); CK(hipMalloc(&ab, 4));
  CK(hipMemset(flags, 0, (size_t)S * G * 4)); CK(hipMemset(ab, 0, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipStream_t st; CK(hipStreamCreate(&st));
  const int reps = 200; float ms = 0;
  for (int w = 0; w < 2; ++w) {
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) for (int s = 0; s < S; ++s) hipLaunchKernelGGL(stage, dim3(G), dim3(512), 0, st, (s & 1) ? b1 : b0, (s & 1) ? b0 : b1, s, per_block, D);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  const float us_launch = ms * 1000.f / reps / S;
  std::vector<float> ref((size_t)G * per_block), got((size_t)G * per_block);
  CK(hipMemset(b0, 0, (size_t)G * per_block * 4)); CK(hipMemset(b1, 0, (size_t)G * per_block * 4));
  for (int s = 0; s < S; ++s) hipLaunchKernelGGL(stage, dim3(G), dim3(512), 0, st, (s & 1) ? b1 : b0, (s & 1) ? b0 : b1, s, per_block, D);
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(ref.data(), (S & 1) ? b1 : b0, ref.size() * 4, hipMemcpyDeviceToHost));
  unsigned epoch = 0;
  for (int w = 0; w < 2; ++w) {
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(chain<D>, dim3(S * G), dim3(512), 0, st, b0, b1, flags, ab, G, per_block, ++epoch);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  const float us_chain = ms * 1000.f / reps / S;
  CK(hipMemset(b0, 0, (size_t)G * per_block * 4)); CK(hipMemset(b1, 0, (size_t)G * per_block * 4));
  hipLaunchKernelGGL(chain<D>, dim3(S * G), dim3(512), 0, st, b0, b1, flags, ab, G, per_block, ++epoch);
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(got.data(), (S & 1) ? b1 : b0, got.size() * 4, hipMemcpyDeviceToHost));
  size_t bad = 0; for (size_t i = 0; i < ref.size(); ++i) bad += ref[i] != got[i];
  unsigned habort = 0; CK(hipMemcpy(&habort, ab, 4, hipMemcpyDeviceToHost));
  printf("S %2d stages x G %4d workgroups x512 thr, %2d KB/workgroup, %2d producers each: dependent launches %.2f us/stage | one launch, flag dataflow %.2f us/stage  (mismatches %zu, %s)\n",
         S, G, per_block * 4 / 1024, D, us_launch, us_chain, bad, habort ? "ABORT" : "ok");
  fflush(stdout);
  CK(hipFree(b0)); CK(hipFree(b1)); CK(hipFree(flags)); CK(hipFree(ab));
  return habort ? 2 : 0;
}
int main() {
  for (int G : {256, 512}) {
    if (run<1>(10, G, 4096)) return 1;
    if (run<16>(10, G, 4096)) return 1;
  }
  if (run<16>(3, 512, 4096)) return 1;
  if (run<16>(10, 512, 1024)) return 1;
  return 0;
}
