// Experiment (not product): does what a kernel leaves in an XCD's L2 survive the boundary to the next DEPENDENT launch?
// Kernel A (256 workgroups) reads a 128 KB "weight" buffer (every workgroup all of it: each XCD's L2 then holds it); kernel B, launched
// behind it on the same stream, has one wave per workgroup time a dependent chain of 16 loads of that buffer with s_memtime.
// Variants: A touches the buffer / A touches something else (cold: B's loads come from MALL / HBM) / B alone after a 20 ms pause.
// Also: the buffer WRITTEN by A with plain stores and with write-through (sc1) stores, then read by B on every XCD.
//   hipcc --offload-arch=gfx950 -O3 -o l2_survive tools/exp/l2_survive.hip && ./l2_survive
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int N = 131072;  // floats: 512 KB = 4096 lines of 128 bytes, 16 per probing workgroup (no line is shared by two probes)
__global__ void touch(const float* __restrict__ w, float* sink) {
  float s = 0.f;
  for (int i = threadIdx.x; i < N; i += blockDim.x) s += w[i];
  if (s == 123.456f) sink[blockIdx.x] = s;
}
__global__ void writer(float* w, int sc1) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    if (sc1) __hip_atomic_store(w + i, (float)(i & 1023), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else w[i] = (float)(i & 1023);
  }
}
__global__ void probe(const float* w, unsigned long long* out, float* sink) {
  if (threadIdx.x >= 64) return;
  // a dependent chain: the next index comes from the loaded value (the buffer holds small integers)
  unsigned long long t0 = __builtin_readcyclecounter();
  // 16 DEPENDENT loads of the workgroup's own 16 lines (lane l reads dword l & 31 of the line: one line per load instruction)
  int line = blockIdx.x * 16;
  float v = w[line * 32 + (threadIdx.x & 31)];
  for (int i = 1; i < 16; ++i) {
    const int dep = __builtin_amdgcn_readfirstlane((int)v);            // (data-dependent: the next load cannot issue before this one has landed)
    line = blockIdx.x * 16 + ((i * 7 + dep) & 15);
    v = w[line * 32 + (threadIdx.x & 31)];
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (v == -1.f) sink[0] = v;
  if (threadIdx.x == 0) out[blockIdx.x] = (t1 - t0) / 16;
}
int main() {
  float *w, *other, *sink; unsigned long long* out;
  CHK(hipMalloc(&w, N * 4)); CHK(hipMalloc(&other, N * 4)); CHK(hipMalloc(&sink, 4096)); CHK(hipMalloc(&out, 256 * 8));
  std::vector<float> h(N); for (int i = 0; i < N; ++i) h[i] = (float)(i & 1023);
  CHK(hipMemcpy(w, h.data(), N * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(other, h.data(), N * 4, hipMemcpyHostToDevice));
  hipStream_t s; CHK(hipStreamCreate(&s));
  auto run = [&](const char* name, int mode) -> int {
    std::vector<unsigned long long> best(256, ~0ull), r(256);
    std::vector<double> med;
    for (int rep = 0; rep < 20; ++rep) {
      if (mode == 0) touch<<<256, 256, 0, s>>>(w, sink);
      if (mode == 1) touch<<<256, 256, 0, s>>>(other, sink);
      if (mode == 3) writer<<<256, 256, 0, s>>>(w, 0);
      if (mode == 4) writer<<<256, 256, 0, s>>>(w, 1);
      if (mode == 2) { hipStreamSynchronize(s); touch<<<256, 256, 0, s>>>(other, sink); hipStreamSynchronize(s); }
      probe<<<256, 64, 0, s>>>(w, out, sink);
      CHK(hipStreamSynchronize(s));
      CHK(hipMemcpy(r.data(), out, 256 * 8, hipMemcpyDeviceToHost));
      std::vector<unsigned long long> q = r; std::sort(q.begin(), q.end());
      med.push_back((double)q[128]);
    }
    std::sort(med.begin(), med.end());
    printf("%-78s median cycles per dependent load (median over workgroups, median of 20 launches): %.0f\n", name, med[10]);
    return 0;
  };
  run("previous launch READ the buffer on every XCD", 0);
  run("previous launch read ANOTHER buffer (this one last touched two launches ago)", 1);
  run("synchronised, another buffer read in between", 2);
  run("previous launch WROTE the buffer with plain stores (each line by one XCD)", 3);
  run("previous launch WROTE the buffer with write-through (sc1) stores", 4);
  run("previous launch READ the buffer on every XCD (again)", 0);
  return 0;
}
