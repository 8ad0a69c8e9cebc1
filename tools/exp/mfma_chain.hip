// mfma_chain.hip — one-off hardware experiment (not product): what does a DEPENDENT chain of v_mfma_f32_32x32x2_f32 cost per instruction,
// against 2 / 4 independent accumulators in the same wave?  (MI355X guide: 64 cycles per instruction "from one wave per SIMD with 4
// accumulators"; every tile routine of this repo accumulates a chunk's 16 steps into ONE accumulator.)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip && ./mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int SHAPE>
__global__ void k(float* out, long long* cyc, int iters) {
  f32x16 acc[NACC]; f32x4 acc4[NACC];
  for (int j = 0; j < NACC; ++j) { for (int q = 0; q < 16; ++q) acc[j][q] = 0.f; for (int q = 0; q < 4; ++q) acc4[j][q] = 0.f; }
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16 / NACC; ++s)
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        if constexpr (SHAPE == 32) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        else acc4[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[j], 0, 0, 0);
      }
  }
  float sum = 0.f;
  for (int j = 0; j < NACC; ++j) { for (int q = 0; q < 16; ++q) sum += acc[j][q]; for (int q = 0; q < 4; ++q) sum += acc4[j][q]; }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC, int SHAPE> void run(const char* nm, int threads, int blocks = 1) {
  float* out; long long* cyc; hipMalloc(&out, 4 * 1024 * 1024); hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NACC, SHAPE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  // s_memtime counts at 100 MHz on this part; report both raw ticks and the time per instruction through an event-timed full launch
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL((k<NACC, SHAPE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters * 10); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 10 * 16;
  printf("%-44s blocks %4d waves/SIMD %d: %8.2f ns per MFMA per wave (%lld clock64 ticks / %d x 16)\n", nm, blocks, threads / 256 ? threads / 256 : 1, ms * 1e6 / n, c, iters);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<1, 32>("32x32x2 f32, 1 accumulator (dependent chain)", 64);
  run<2, 32>("32x32x2 f32, 2 accumulators", 64);
  run<4, 32>("32x32x2 f32, 4 accumulators", 64);
  run<1, 16>("16x16x4 f32, 1 accumulator (dependent chain)", 64);
  run<2, 16>("16x16x4 f32, 2 accumulators", 64);
  run<4, 16>("16x16x4 f32, 4 accumulators", 64);
  run<1, 32>("32x32x2 f32, 1 accumulator, 2 waves per SIMD", 512);
  run<2, 32>("32x32x2 f32, 2 accumulators, 2 waves per SIMD", 512);
  run<1, 32>("32x32x2 f32, 1 accumulator, 1 wave per SIMD x4", 256);
  run<1, 32>("32x32x2 f32, 1 acc, whole chip 1 wave/SIMD", 256, 256);
  run<4, 32>("32x32x2 f32, 4 acc, whole chip 1 wave/SIMD", 256, 256);
  run<1, 32>("32x32x2 f32, 1 acc, whole chip 2 waves/SIMD", 512, 256);
  run<1, 32>("32x32x2 f32, 1 acc, 196 blocks", 256, 196);
  run<1, 32>("32x32x2 f32, 1 acc, 64 blocks", 256, 64);
  return 0;
}
