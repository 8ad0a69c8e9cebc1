// Experiment: the lane <-> element map of gfx950's ds_read_b64_tr_b16 (LDS transpose read), needed by the half weight-gradient block tiles.
//   hipcc --offload-arch=gfx950 -O2 tools/exp/tr16_probe.hip -o tools/exp/tr16_probe && tools/exp/tr16_probe
// LDS halfs hold their own index; lane l supplies the address of halfs [4 l, 4 l + 4).  Prints what every lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void probe(float* out) {
  __shared__ _Float16 lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (_Float16)(float)i;
  __syncthreads();
  fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(lds + 4 * threadIdx.x));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (float)v[j];
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 4); hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  float h[256]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) {
      printf(" %4.0f", h[l * 4 + j]);
      const int g = l >> 4, t = l & 15, src_lane = 16 * g + 4 * j + (t >> 2), want = 4 * src_lane + (t & 3);   // hypothesis: out[t][j] = in[4 j + (t >> 2)][t & 3]
      bad += (int)h[l * 4 + j] != want;
    }
    printf("\n");
  }
  printf("hypothesis out[16g + t][j] = in[16g + 4j + (t >> 2)][t & 3]: %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
  return 0;
}
