// Experiment (not product): what bounds a wave's v_mfma_f32_16x16x4_f32 stream in conv_ss.h's K-outer loop?
//   hipcc --offload-arch=gfx950 -O3 -o mfma16_rate mfma16_rate.hip && ./mfma16_rate
// One workgroup per CU (LDS-limited), NT accumulators per wave, groups of 4 NT MFMAs; variants:
//   V0 registers only (operands never change)              V1 + one ds_read_b128 per 3 MFMAs, conflict-free
//   V2 the same with 2-way bank conflicts                   V3 V1 with 4 extra waves parked at a barrier
//   V4 V1 + ds_read2_b32 B fragments                        V5 V0 with 32x32x2 (reference: 64 cycles each)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int V, int NT, int NTHR>
__global__ void __launch_bounds__(NTHR) k(float* out, unsigned long long* cyc, int groups) {
  __shared__ __attribute__((aligned(16))) float smem[36000];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 36000; i += NTHR) smem[i] = 1.0f + (i & 7);
  __syncthreads();
  if (wave >= 4) { __syncthreads(); return; }
  f32x4 acc[NT];
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0, 0, 0, 0};
  f32x4 av[2][NT]; float bv[2][4];
  const int m = lane & 15, kq = lane >> 4;
  int base[NT];
  for (int t = 0; t < NT; ++t) base[t] = ((16 * t + m) * (V == 2 ? 64 : 36) + 4 * kq) % 30000;
  for (int t = 0; t < NT; ++t) { av[0][t] = f32x4{1.f, 2.f, 3.f, 4.f}; av[1][t] = f32x4{2.f, 1.f, 0.5f, 4.f}; }
  for (int j = 0; j < 4; ++j) { bv[0][j] = 1.0f + j; bv[1][j] = 2.0f + j; }
  if (V == 6) {                 // random operands in every lane (power: does the clock hold?)
    unsigned x = 123456789u * (tid + 1) + blockIdx.x * 2654435761u;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return (float)(int)(x >> 8) * (1.0f / 8388608.0f) - 1.0f; };
    for (int t = 0; t < NT; ++t) for (int e = 0; e < 4; ++e) { av[0][t][e] = rnd(); av[1][t][e] = rnd(); }
    for (int j = 0; j < 4; ++j) { bv[0][j] = rnd() * 0.01f; bv[1][j] = rnd() * 0.01f; }
  }
  const unsigned long long t0 = clock64();
#pragma unroll 1
  for (int g = 0; g < groups; g += 2) {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      if constexpr (V >= 1 && V <= 4) {
        const float* src = smem + ((g + gq) & 3) * 1024;
#pragma unroll
        for (int t = 0; t < NT; ++t) av[(gq + 1) & 1][t] = *reinterpret_cast<const f32x4*>(src + base[t]);
        if constexpr (V == 4) {
#pragma unroll
          for (int j = 0; j < 4; ++j) bv[(gq + 1) & 1][j] = src[lane + 68 * j + 5000];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[gq & 1][t][j], bv[gq & 1][j], acc[t], 0, 0, 0);
      if constexpr (V >= 1 && V <= 4) {
#pragma unroll
        for (int q = 0; q < NT + (V == 4 ? 2 : 0); ++q) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = clock64();
  float s = 0;
  for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
  if (NTHR > 256) __syncthreads();
}

template <int NT>
__global__ void __launch_bounds__(256) k32(float* out, unsigned long long* cyc, int groups) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x16 acc[NT];
  for (int t = 0; t < NT; ++t) for (int q = 0; q < 16; ++q) acc[t][q] = 0;
  float a = 1.0f + lane, b = 2.0f;
  const unsigned long long t0 = clock64();
#pragma unroll 1
  for (int g = 0; g < groups; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  const unsigned long long t1 = clock64();
  float s = 0;
  for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][5];
  out[blockIdx.x * 256 + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <class F>
int run(const char* name, F launch, int grid, int groups, int nt, int cyc_per) {
  float* out; unsigned long long* cyc;
  CHK(hipMalloc(&out, 256 * 1024 * 4)); CHK(hipMalloc(&cyc, 1024 * 4 * 8));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  launch(grid, out, cyc, groups); CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  for (int r = 0; r < 10; ++r) launch(grid, out, cyc, groups);
  CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(grid * 4);
  CHK(hipMemcpy(h.data(), cyc, grid * 4 * 8, hipMemcpyDeviceToHost));
  double sum = 0; for (auto v : h) sum += (double)v;
  const double per = sum / h.size() / ((double)groups * 4 * nt);
  const double us = ms * 1e3 / 10;
  printf("%-44s grid %4d: %6.1f ticks per MFMA (ideal %d), %8.1f us per launch = %5.1f ns per MFMA  -> ticks/ns %.3f\n", name, grid, per, cyc_per, us,
         us * 1e3 / ((double)groups * 4 * nt), per / (us * 1e3 / ((double)groups * 4 * nt)));
  hipFree(out); hipFree(cyc);
  return 0;
}

int main() {
  const int G = 2000;
#define RUN(V, NT, NTHR, grid, label) run(label, [](int g, float* o, unsigned long long* c, int gr) { hipLaunchKernelGGL((k<V, NT, NTHR>), dim3(g), dim3(NTHR), 0, 0, o, c, gr); }, grid, G, NT, 32)
  for (int grid : {256}) {
    RUN(0, 11, 256, grid, "V0 regs only, NT 11");
    RUN(0, 4, 256, grid, "V0 regs only, NT 4");
    RUN(0, 3, 256, grid, "V0 regs only, NT 3");
    RUN(0, 2, 256, grid, "V0 regs only, NT 2");
    RUN(0, 1, 256, grid, "V0 regs only, NT 1");
    RUN(1, 11, 256, grid, "V1 + ds_read_b128 / 3 MFMA, no conflicts");
    RUN(2, 11, 256, grid, "V2 + ds_read_b128 / 3 MFMA, 2-way+ conflicts");
    RUN(1, 11, 512, grid, "V3 = V1 + 4 parked waves");
    RUN(4, 11, 256, grid, "V4 = V1 + B fragments (ds_read_b32)");
    RUN(6, 11, 256, grid, "V6 = V0 with random operands");
    run("V5 32x32x2, NT 4", [](int g, float* o, unsigned long long* c, int gr) { hipLaunchKernelGGL((k32<4>), dim3(g), dim3(256), 0, 0, o, c, gr); }, grid, G, 4, 64);
  }
  return 0;
}
