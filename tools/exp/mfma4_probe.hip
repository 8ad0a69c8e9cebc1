// Experiment (not product): operand map of v_mfma_f32_4x4x1_16b_f32 on gfx950 (conv_ss.h computes the left-over positions with it):
// A lane = i + 4 block, B lane = j + 4 block, D[vgpr i][lane j + 4 block] — confirmed, 0 mismatches.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* o) {
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) o[threadIdx.x * 4 + i] = acc[i];
}
int main() {
  float *a, *b, *o; hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&o, 1024);
  float ha[64], hb[64], ho[256];
  for (int l = 0; l < 64; ++l) { ha[l] = 1 + (l % 4) + 10 * (l / 4); hb[l] = 100 * (1 + l % 4) + 1000 * (l / 4); }
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o); hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
  // expected (assumed map): D[block=l/4][i=v][j=l%4] = A[block][i] * B[block][j], A lane = i + 4 block, B lane = j + 4 block
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) { const int blk = l / 4, j = l % 4; const float e = ha[v + 4 * blk] * hb[j + 4 * blk]; if (ho[l * 4 + v] != e) { if (bad < 5) printf("lane %d v %d got %g expected %g\n", l, v, ho[l * 4 + v], e); ++bad; } }
  printf("4x4x1 map check: %d mismatches\n", bad);
  return 0;
}
