// mfma_lds.hip — one-off hardware experiment (not product): the block-tile loop's matrix utilisation stays at ~60 % per SIMD whatever the number
// of waves (tools/exp/README.md).  What does the chunk body cost WITHOUT global memory?  One workgroup of 256 threads per CU, per "chunk":
// 16 dependent v_mfma_f32_32x32x2_f32 fed by 8 ds_read_b128 (the KM-panel fragment reads), optionally 4 ds_write_b128 + a barrier.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_lds mfma_lds.hip && ./mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>     // 0: MFMA only, 1: + fragment reads, 2: + LDS writes + barrier per chunk
__global__ void __launch_bounds__(256) k(float* out, int chunks) {
  __shared__ __attribute__((aligned(16))) float smem[2 * (64 + 64) * 36];
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  for (int e = tid; e < 2 * 128 * 36; e += 256) smem[e] = 1e-3f * (e & 63);
  __syncthreads();
  f32x16 acc;
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  float4 w = make_float4(tid * 1e-4f, 1.f, 2.f, 3.f);
  for (int c = 0; c < chunks; ++c) {
    const float* As = smem + (c & 1) * (128 * 36);
    const float* Bs = As + 64 * 36;
    float fa[16], fb[16];
    if (MODE >= 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 va = *reinterpret_cast<const float4*>(As + (wm * 32 + i) * 36 + 8 * j + 4 * h);
        const float4 vb = *reinterpret_cast<const float4*>(Bs + (wn * 32 + i) * 36 + 8 * j + 4 * h);
        fa[4 * j] = va.x; fa[4 * j + 1] = va.y; fa[4 * j + 2] = va.z; fa[4 * j + 3] = va.w;
        fb[4 * j] = vb.x; fb[4 * j + 1] = vb.y; fb[4 * j + 2] = vb.z; fb[4 * j + 3] = vb.w;
      }
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t) { fa[t] = w.x + t; fb[t] = w.y + t; }
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[t], acc, 0, 0, 0);
    if (MODE >= 2) {
      float* Ws = smem + ((c + 1) & 1) * (128 * 36);
      *reinterpret_cast<float4*>(Ws + (tid >> 3) * 36 + (tid & 7) * 4) = w;
      *reinterpret_cast<float4*>(Ws + (32 + (tid >> 3)) * 36 + (tid & 7) * 4) = w;
      *reinterpret_cast<float4*>(Ws + (64 + (tid >> 3)) * 36 + (tid & 7) * 4) = w;
      *reinterpret_cast<float4*>(Ws + (96 + (tid >> 3)) * 36 + (tid & 7) * 4) = w;
      __syncthreads();
    }
  }
  float s = 0.f;
  for (int q = 0; q < 16; ++q) s += acc[q];
  out[blockIdx.x * 256 + tid] = s;
}
template <int MODE> void run(const char* nm, int blocks) {
  float* out; hipMalloc(&out, 4 * 256 * 4096);
  const int chunks = 4000;
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, 200);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, chunks); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // blocks <= 256: one workgroup per CU; 512 / 768: two / three per CU (waves per SIMD)
  const double per_chunk_ns = ms * 1e6 / chunks / ((blocks + 255) / 256);
  printf("%-58s blocks %4d: %7.1f ns per chunk per workgroup-slot (16 MFMAs = 426.7 ns at 2.4 GHz) -> matrix utilisation %.0f %%\n", nm, blocks, per_chunk_ns, 100.0 * 426.7 / per_chunk_ns);
  hipFree(out);
}
int main() {
  run<0>("MFMA chain only", 256);
  run<1>("+ 8 ds_read_b128 fragment reads per chunk", 256);
  run<2>("+ 4 ds_write_b128 and a barrier per chunk", 256);
  run<2>("the same, two workgroups per CU", 512);
  run<2>("the same, three workgroups per CU", 768);
  run<1>("fragment reads only, two workgroups per CU", 512);
  return 0;
}
