// Box probe: measured peaks of this MI355X box next to the spec peaks the rooflines use.
//   hipcc --offload-arch=gfx950 -O3 -o box_probe box_probe.hip && ./box_probe > box.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) triad(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 x = a[i], y = b[i];
    c[i] = make_float4(x.x + 2.f * y.x, x.y + 2.f * y.y, x.z + 2.f * y.z, x.w + 2.f * y.w);
  }
}
__global__ void __launch_bounds__(256) readsum(const float4* __restrict__ a, float* out, size_t n4) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 x = a[i]; s += x.x + x.y + x.z + x.w; }
  if (s == 123.456f) out[0] = s;
}
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) mfma_f32(float* out, int iters) {
  f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
  const float a = threadIdx.x * 1e-3f, b = 1.0f;
  for (int i = 0; i < iters; ++i) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < 16; ++k) s += acc0[k] + acc1[k] + acc2[k] + acc3[k];
  if (s == 123.456f) out[0] = s;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const size_t N = (size_t)1 << 30;                        // 1 GiB per array
  float4 *a, *b, *c; float* out; CK(hipMalloc(&a, N)); CK(hipMalloc(&b, N)); CK(hipMalloc(&c, N)); CK(hipMalloc(&out, 64));
  CK(hipMemset(a, 1, N)); CK(hipMemset(b, 1, N));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  auto timeit = [&](auto fn, int reps) { fn(); hipEventRecord(e0); for (int r = 0; r < reps; ++r) fn(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); return ms / reps; };
  const int blocks = p.multiProcessorCount * 16;
  const double t_copy = timeit([&] { (void)hipMemcpyAsync(c, a, N, hipMemcpyDeviceToDevice, 0); }, 10);
  const double t_triad = timeit([&] { hipLaunchKernelGGL(triad, dim3(blocks), dim3(256), 0, 0, a, b, c, N / 16); }, 10);
  const double t_read = timeit([&] { hipLaunchKernelGGL(readsum, dim3(blocks), dim3(256), 0, 0, a, out, N / 16); }, 10);
  // pinned host memory read by a kernel over PCIe / the host link (zero-copy ring mode)
  float4* h; CK(hipHostMalloc((void**)&h, (size_t)256 << 20, hipHostMallocMapped)); float4* hd; CK(hipHostGetDevicePointer((void**)&hd, h, 0));
  const double t_zc = timeit([&] { hipLaunchKernelGGL(readsum, dim3(blocks), dim3(256), 0, 0, hd, out, ((size_t)256 << 20) / 16); }, 5);
  const double t_h2d = timeit([&] { (void)hipMemcpyAsync(c, h, (size_t)256 << 20, hipMemcpyHostToDevice, 0); }, 5);
  const int iters = 20000;
  const double t_mfma = timeit([&] { hipLaunchKernelGGL(mfma_f32, dim3(p.multiProcessorCount * 8), dim3(256), 0, 0, out, iters); }, 3);
  const double flops = (double)p.multiProcessorCount * 8 * 4 /*waves*/ * iters * 4.0 * (2.0 * 32 * 32 * 2);
  printf("{\n \"device\": \"%s\", \"gcn_arch\": \"%s\", \"compute_units\": %d, \"clock_mhz\": %d, \"lds_per_workgroup_kb\": %zu, \"hbm_gb\": %.1f,\n",
         p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000, p.sharedMemPerBlock / 1024, p.totalGlobalMem / 1e9);
  printf(" \"measured\": {\"d2d_copy_GBps\": %.0f, \"triad_GBps\": %.0f, \"read_GBps\": %.0f, \"zero_copy_host_read_GBps\": %.1f, \"h2d_pinned_GBps\": %.1f, \"fp32_mfma_32x32x2_TFLOPs\": %.1f},\n",
         2.0 * N / t_copy / 1e6, 3.0 * N / t_triad / 1e6, 1.0 * N / t_read / 1e6, (double)(256 << 20) / t_zc / 1e6, (double)(256 << 20) / t_h2d / 1e6, flops / t_mfma / 1e9);
  printf(" \"spec_used_by_rooflines\": {\"hbm_GBps\": 8000, \"fp32_mfma_TFLOPs\": 157.3},\n \"note\": \"1 GiB arrays, HIP events, mean of 5-10 runs after one warm-up; d2d counts read+write bytes\"\n}\n");
  return 0;
}
