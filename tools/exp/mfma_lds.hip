// mfma_lds.hip — one-off hardware experiment (not product): the block-tile loop's matrix utilisation stays at ~60 % per SIMD whatever the number
// of waves (tools/exp/README.md).  What does the chunk body cost WITHOUT global memory?  One workgroup of 256 threads per CU, per "chunk":
// 16 dependent v_mfma_f32_32x32x2_f32 fed by 8 ds_read_b128 (the KM-panel fragment reads), optionally 4 ds_write_b128 + a barrier.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_lds mfma_lds.hip && ./mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>     // 0: MFMA only, 1: + fragment reads, 2: + LDS writes + barrier per chunk, 3 / 4: + the chunk's 16 KB fetched from global memory
                        // through a two-deep register ring (3: a 64 KB region per workgroup, L2-resident; 4: a fresh 16 KB per chunk, streamed)
__global__ void __launch_bounds__(256) k(float* out, int chunks, const float4* __restrict__ src) {
  __shared__ __attribute__((aligned(16))) float smem[2 * (64 + 64) * 36];
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  for (int e = tid; e < 2 * 128 * 36; e += 256) smem[e] = 1e-3f * (e & 63);
  __syncthreads();
  f32x16 acc;
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  float4 w = make_float4(tid * 1e-4f, 1.f, 2.f, 3.f);
  float4 r0[4], r1[4];
  const size_t wg_base = MODE == 3 ? (size_t)blockIdx.x * 4096 : (size_t)blockIdx.x * 1024 * 1024;      // float4 units: 64 KB / 16 MB per workgroup
  auto gl = [&](int c, float4* r) {
    const size_t o = wg_base + (MODE == 3 ? (size_t)(c & 3) * 1024 : ((size_t)c * 1024) % (1024 * 1024));
#pragma unroll
    for (int p = 0; p < 4; ++p) r[p] = src[o + p * 256 + tid];
  };
  if (MODE >= 3) { gl(0, r0); gl(1, r1); }
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
      float4 s0 = w, s1 = w, s2 = w, s3 = w;
      if (MODE >= 3) { float4* r = (c & 1) ? r1 : r0; s0 = r[0]; s1 = r[1]; s2 = r[2]; s3 = r[3]; }
      *reinterpret_cast<float4*>(Ws + (tid >> 3) * 36 + (tid & 7) * 4) = s0;
      *reinterpret_cast<float4*>(Ws + (32 + (tid >> 3)) * 36 + (tid & 7) * 4) = s1;
      *reinterpret_cast<float4*>(Ws + (64 + (tid >> 3)) * 36 + (tid & 7) * 4) = s2;
      *reinterpret_cast<float4*>(Ws + (96 + (tid >> 3)) * 36 + (tid & 7) * 4) = s3;
      if (MODE >= 3) { if (c & 1) gl(c + 2, r1); else gl(c + 2, r0); }
      __syncthreads();
    }
  }
  float s = 0.f;
  for (int q = 0; q < 16; ++q) s += acc[q];
  out[blockIdx.x * 256 + tid] = s;
}
template <int MODE> void run(const char* nm, int blocks) {
  float* out; hipMalloc(&out, 4 * 256 * 4096);
  static float4* src = nullptr; if (!src) { hipMalloc(&src, (size_t)768 * 16 * 1024 * 1024 + (1 << 20)); hipMemset(src, 0, (size_t)768 * 16 * 1024 * 1024); }
  const int chunks = 4000;
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, 200, src);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, chunks, src); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // blocks <= 256: one workgroup per CU; 512 / 768: two / three per CU (waves per SIMD)
  const double per_chunk_ns = ms * 1e6 / chunks / ((blocks + 255) / 256);
  printf("%-58s blocks %4d: %7.1f ns per chunk per workgroup-slot (16 MFMAs = 426.7 ns at 2.4 GHz) -> matrix utilisation %.0f %%\n", nm, blocks, per_chunk_ns, 100.0 * 426.7 / per_chunk_ns);
  hipFree(out);
}

// the same chunk body with the operand panels fetched by global_load_lds_dwordx4 (no staging registers, no ds_write): three LDS stages,
// chunk c + 2 is issued at the top of chunk c, `s_waitcnt vmcnt(4)` in front of the barrier leaves exactly those four loads in flight.
// (the LDS image is lane-linear: the fragment reads below see meaningless values — timing only)
template <int STREAM>
__global__ void __launch_bounds__(256) kg(float* out, int chunks, const float4* __restrict__ src) {
  __shared__ __attribute__((aligned(16))) float smem[3 * 128 * 36];
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  for (int e = tid; e < 3 * 128 * 36; e += 256) smem[e] = 1e-3f * (e & 63);
  __syncthreads();
  f32x16 acc;
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  const size_t wg_base = STREAM ? (size_t)blockIdx.x * 1024 * 1024 : (size_t)blockIdx.x * 4096;
  auto glds = [&](int c) {
    const size_t o = wg_base + (STREAM ? ((size_t)c * 1024) % (1024 * 1024) : (size_t)(c & 3) * 1024);
    float* st = smem + (c % 3) * (128 * 36);
    // inline asm: with the builtin hipcc puts `s_waitcnt vmcnt(0)` in front of the next LDS read (it cannot tell the reads from the landing data)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float4* gp = src + o + p * 256 + tid;
      const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(st + p * 1024 + wave * 256);
      const unsigned ldsu = __builtin_amdgcn_readfirstlane(lds);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gp), "s"(ldsu) : "memory", "m0");
    }
  };
  glds(0); glds(1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __syncthreads();
  for (int c = 0; c < chunks; ++c) {
    glds(c + 2);
    const float* As = smem + (c % 3) * (128 * 36);
    const float* Bs = As + 64 * 36;
    float fa[16], fb[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 va = *reinterpret_cast<const float4*>(As + (wm * 32 + i) * 36 + 8 * j + 4 * h);
      const float4 vb = *reinterpret_cast<const float4*>(Bs + (wn * 32 + i) * 36 + 8 * j + 4 * h);
      fa[4 * j] = va.x; fa[4 * j + 1] = va.y; fa[4 * j + 2] = va.z; fa[4 * j + 3] = va.w;
      fb[4 * j] = vb.x; fb[4 * j + 1] = vb.y; fb[4 * j + 2] = vb.z; fb[4 * j + 3] = vb.w;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[t], acc, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");          // chunk c + 1 has landed; chunk c + 2 stays in flight
    __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
  for (int q = 0; q < 16; ++q) s += acc[q];
  out[blockIdx.x * 256 + tid] = s;
}
template <int STREAM> void rung(const char* nm, int blocks) {
  float* out; hipMalloc(&out, 4 * 256 * 4096);
  static float4* src = nullptr; if (!src) { hipMalloc(&src, (size_t)768 * 16 * 1024 * 1024 + (1 << 20)); hipMemset(src, 0, (size_t)768 * 16 * 1024 * 1024); }
  const int chunks = 4000;
  hipLaunchKernelGGL((kg<STREAM>), dim3(blocks), dim3(256), 0, 0, out, 200, src);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL((kg<STREAM>), dim3(blocks), dim3(256), 0, 0, out, chunks, src); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
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
  run<3>("+ the chunk's 16 KB from an L2-resident region, 1 wg / CU", 256);
  run<3>("the same, two workgroups per CU", 512);
  run<3>("the same, three workgroups per CU", 768);
  run<4>("+ the chunk's 16 KB streamed from HBM, 1 wg / CU", 256);
  run<4>("the same, two workgroups per CU", 512);
  run<4>("the same, three workgroups per CU", 768);
  rung<0>("global_load_lds, 16 KB per chunk, L2-resident, 1 wg / CU", 256);
  rung<0>("the same, two workgroups per CU", 512);
  rung<0>("the same, three workgroups per CU", 768);
  rung<1>("global_load_lds, 16 KB per chunk, streamed, 1 wg / CU", 256);
  rung<1>("the same, two workgroups per CU", 512);
  rung<1>("the same, three workgroups per CU", 768);
  return 0;
}
