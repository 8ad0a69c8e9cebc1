// Experiment (not product), round 3: VERDICT r2 items 1, 3, 5 — the in-launch hand-off protocols the MI355X guide prescribes
// (write-through `sc1` stores + `sc1` loads, no per-workgroup fences; XCD-hierarchical grid barrier keyed by HW_REG_XCC_ID)
// measured on this stack next to the forms round 1/2 measured (single counter, release/acquire fences per workgroup), and the
// dependent-dispatch floor measured GPU-side (graph replay / pre-filled queue) instead of by a host-bound eager burst.
//
//   hipcc --offload-arch=gfx950 -O3 -o handoff_r3 handoff_r3.hip -ldl && timeout 300 ./handoff_r3 [all|boundary|barrier|chain|stores] [lib.so]
//
// Every spin is bounded and abortable: a wrong residency / ordering assumption ends a section with "ABORT", never a hung GPU.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// ============================================================ A. dependent-dispatch floor ==================================
struct Big { char b[640]; };                                  // a kernel-argument block the size of StepArgs + MultiDims
template <int LDS> __global__ void k_empty(float* p) {
  __shared__ float sh[LDS > 0 ? LDS : 1];
  if (LDS > 0) sh[threadIdx.x % (LDS > 0 ? LDS : 1)] = 1.0f;
  if (threadIdx.x == 0 && blockIdx.x == 0xFFFFFFF) p[0] = sh[0];
}
__global__ void k_empty_big(float* p, Big b) { if (threadIdx.x == 0 && blockIdx.x == 0xFFFFFFF) p[0] = b.b[5]; }
__global__ void k_touch(const float* src, float* dst, int s, int per_block) {      // 16 KB in, 16 KB out per workgroup
  const int nb = gridDim.x, from = (blockIdx.x * 37 + 11 + s) % nb;
  for (int i = threadIdx.x; i < per_block; i += blockDim.x) dst[(size_t)blockIdx.x * per_block + i] = src[(size_t)from * per_block + i] + 1.0f;
}
__global__ void k_spin(long long cycles, float* p) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (p && threadIdx.x == 0xFFFF) p[0] = 1.0f;
}


template <class F>
static int boundary_row(const char* what, hipStream_t st, F launch) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 1000; float ms = 0;
  // (1) eager burst: N launches back to back (host issue rate may be the limit)
  float eager = 0;
  for (int w = 0; w < 2; ++w) {
    CK(hipEventRecord(e0, st)); for (int i = 0; i < N; ++i) launch(st, i); CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); eager = ms * 1000.f / N;
  }
  // (2) pre-filled queue: a 6 ms spin kernel blocks the stream while the host enqueues all N launches; GPU-side cost per launch
  //     = (time from spin start to the end of the chain - spin alone) / N
  float spin_us = 0, filled = 0;
  {
    const long long cyc = 600000;                               // wall_clock64 ticks at 100 MHz -> 6 ms
    CK(hipEventRecord(e0, st)); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, cyc, (float*)nullptr); CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); spin_us = ms * 1000.f;
    CK(hipEventRecord(e0, st)); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, cyc, (float*)nullptr);
    for (int i = 0; i < N; ++i) launch(st, i);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    filled = (ms * 1000.f - spin_us) / N;
  }
  // (3) the same chain as a hipGraph (250 nodes), replayed 4x
  float graph_us = -1;
  {
    hipStream_t cs; CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 250; ++i) launch(cs, i);
    CK(hipStreamEndCapture(cs, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ex, cs)); CK(hipStreamSynchronize(cs));
    CK(hipEventRecord(e0, cs)); for (int r = 0; r < 4; ++r) CK(hipGraphLaunch(ex, cs)); CK(hipEventRecord(e1, cs));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); graph_us = ms * 1000.f / 1000;
    CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(cs));
  }
  printf("  %-58s eager burst %5.2f | pre-filled queue %5.2f | hipGraph %5.2f  us/launch\n", what, eager, filled, graph_us);
  fflush(stdout);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return 0;
}

static int section_boundary(const char* libpath) {
  printf("== A. dependent-dispatch floor (us per dependent launch; 'pre-filled' and 'hipGraph' are GPU-side, 'eager burst' can be host-bound)\n");
  printf("   HIP_FORCE_DEV_KERNARG=%s, library code object %s\n", getenv("HIP_FORCE_DEV_KERNARG") ? getenv("HIP_FORCE_DEV_KERNARG") : "(unset)",
         libpath ? "LOADED" : "not loaded");
  if (libpath) { void* h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL); if (!h) { printf("dlopen failed: %s\n", dlerror()); return 1; }
                 int (*cnt)(int*) = (int (*)(int*))dlsym(h, "sdqn_device_count"); int n = 0; if (cnt) cnt(&n); }
  float *d, *b0, *b1; CK(hipMalloc(&d, 4096));
  const int per_block = 4096;
  CK(hipMalloc(&b0, (size_t)1024 * per_block * 4)); CK(hipMalloc(&b1, (size_t)1024 * per_block * 4));
  CK(hipMemset(b0, 0, (size_t)1024 * per_block * 4)); CK(hipMemset(b1, 0, (size_t)1024 * per_block * 4));
  hipStream_t s_def, s_nb; CK(hipStreamCreate(&s_def)); CK(hipStreamCreateWithFlags(&s_nb, hipStreamNonBlocking));
  Big big; memset(&big, 1, sizeof big);
  struct { const char* n; hipStream_t s; } streams[] = {{"created stream", s_def}, {"non-blocking stream", s_nb}, {"NULL stream", 0}};
  for (auto& S : streams) {
    printf(" -- %s\n", S.n);
    char nm[128];
    for (int nt : {64, 512, 1024}) {
      snprintf(nm, sizeof nm, "empty kernel, 256 WG x %4d thr, 8 B args, no LDS", nt);
      if (boundary_row(nm, S.s, [&](hipStream_t st, int) { hipLaunchKernelGGL(k_empty<0>, dim3(256), dim3(nt), 0, st, d); })) return 1;
    }
    if (S.s != s_def) continue;
    if (boundary_row("empty kernel, 256 WG x  512 thr, 8 B args, 64 KB static LDS", S.s, [&](hipStream_t st, int) { hipLaunchKernelGGL(k_empty<16384>, dim3(256), dim3(512), 0, st, d); })) return 1;
    if (boundary_row("empty kernel, 256 WG x  512 thr, 648 B args", S.s, [&](hipStream_t st, int) { hipLaunchKernelGGL(k_empty_big, dim3(256), dim3(512), 0, st, d, big); })) return 1;
    if (boundary_row("empty kernel, 800 WG x  512 thr, 648 B args", S.s, [&](hipStream_t st, int) { hipLaunchKernelGGL(k_empty_big, dim3(800), dim3(512), 0, st, d, big); })) return 1;
    for (int nb : {256, 512}) {
      snprintf(nm, sizeof nm, "16 KB in + 16 KB out per WG, %d WG x 512 thr (ping-pong)", nb);
      if (boundary_row(nm, S.s, [&](hipStream_t st, int i) { hipLaunchKernelGGL(k_touch, dim3(nb), dim3(512), 0, st, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1, i, per_block); })) return 1;
    }
  }
  CK(hipFree(d)); CK(hipFree(b0)); CK(hipFree(b1));
  return 0;
}

// ============================================================ B. grid barriers in a persistent kernel ========================
struct BarState {
  unsigned census[8];        // workgroups per XCC (counted at kernel start)
  unsigned pad0[8];
  unsigned xcnt[8 * 32];     // per-XCC arrival counter, one 128-B line each
  unsigned xgen[8 * 32];     // per-XCC generation word
  unsigned top[32];          // top-level counter (XCC leaders)
  unsigned flat[32];         // single flat counter (round-1 form)
  unsigned start[32];        // start-up barrier (census complete)
  unsigned abort_flag[32];
  unsigned nxcc_used[32];
};
__device__ __forceinline__ unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 7; }
__device__ __forceinline__ bool spin_until_ge(unsigned* w, unsigned target, unsigned* abort_flag) {
  long spins = 0;
  while ((int)(__hip_atomic_load(w, RLX_AGENT) - target) < 0) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 1023) == 0 && (spins > 4000000 || __hip_atomic_load(abort_flag, RLX_AGENT))) { __hip_atomic_store(abort_flag, 1u, RLX_AGENT); return false; }
  }
  return true;
}
enum { BAR_FLAT_FENCE = 0, BAR_XCD_FENCE = 1, BAR_XCD_NOFENCE = 2, BAR_FLAT_NOFENCE = 3, BAR_XCD_LOCAL = 4 };   // LOCAL: the workgroups of ONE XCC only (no top level)

// gen = barrier index (0, 1, ...); all counters monotonic within the launch (zeroed by a memset before it)
template <int KIND>
__device__ __forceinline__ void grid_barrier(BarState* S, unsigned gen, unsigned xcc, unsigned nwg) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its (write-through) stores
  __syncthreads();
  if (threadIdx.x == 0) {
    if (KIND == BAR_FLAT_FENCE || KIND == BAR_FLAT_NOFENCE) {
      if (KIND == BAR_FLAT_FENCE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      __hip_atomic_fetch_add(S->flat, 1u, RLX_AGENT);
      spin_until_ge(S->flat, nwg * (gen + 1), S->abort_flag);
      if (KIND == BAR_FLAT_FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    } else {
      const unsigned nx = __hip_atomic_load(&S->census[xcc], RLX_AGENT);
      const unsigned old = __hip_atomic_fetch_add(&S->xcnt[xcc * 32], 1u, RLX_AGENT);
      if (old == nx * (gen + 1) - 1) {                         // last arriver of this XCC = its leader for this generation
        if (KIND == BAR_XCD_FENCE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if (KIND != BAR_XCD_LOCAL) {
          __hip_atomic_fetch_add(S->top, 1u, RLX_AGENT);
          spin_until_ge(S->top, __hip_atomic_load(S->nxcc_used, RLX_AGENT) * (gen + 1), S->abort_flag);
        }
        if (KIND == BAR_XCD_FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(&S->xgen[xcc * 32], gen + 1, RLX_AGENT);
      } else {
        spin_until_ge(&S->xgen[xcc * 32], gen + 1, S->abort_flag);
        if (KIND == BAR_XCD_FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    }
  }
  __syncthreads();
}

// phases of: write my slice (per_block floats), barrier, read another workgroup's slice from this phase, add into a register sum.
// SC1: payload stores / loads are 16-byte write-through (sc1) buffer ops -> valid with the NOFENCE barriers.
template <int KIND, bool SC1, bool SC1_LD = SC1>
__global__ void __launch_bounds__(512) persistent(float* buf0, float* buf1, BarState* S, int phases, int per_block, float* out) {
  const unsigned nwg = gridDim.x, b = blockIdx.x;
  const unsigned xcc = xcc_id();
  // start-up: census of workgroups per XCC, then one flat barrier with fences (once per launch; not part of the timed loop's slope)
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&S->census[xcc], 1u, RLX_AGENT);
    __hip_atomic_fetch_add(S->start, 1u, RLX_AGENT);
    spin_until_ge(S->start, nwg, S->abort_flag);
    if (b == 0) { unsigned used = 0; for (int x = 0; x < 8; ++x) used += __hip_atomic_load(&S->census[x], RLX_AGENT) != 0; __hip_atomic_store(S->nxcc_used, used, RLX_AGENT); }
    __hip_atomic_fetch_add(S->start + 1, 1u, RLX_AGENT);
    spin_until_ge(S->start + 1, nwg, S->abort_flag);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  const size_t bytes = (size_t)nwg * per_block * 4;
  float acc = 0.0f;
  for (int s = 0; s < phases; ++s) {
    float* dst = (s & 1) ? buf1 : buf0;
    // XCD-local mode reads a slice of a workgroup on the SAME XCC under the observed b % 8 placement (speed AND, for this mode
    // only, correctness: plain stores stay in that XCC's L2, sc1 loads bypass the reader's L1)
    const unsigned from = KIND == BAR_XCD_LOCAL ? (b + 8u * (1u + (unsigned)s % 3u)) % nwg : (b * 37u + 11u + (unsigned)s) % nwg;
    if (per_block > 0) {
      if (SC1) {
        auto rd = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, (int)bytes, 0x00020000);
        for (int i = threadIdx.x * 4; i < per_block; i += 512 * 4) {
          u4 v; v.x = v.y = v.z = v.w = __float_as_uint((float)(s + 1) + (float)b * 0.001f);
          __builtin_amdgcn_raw_buffer_store_b128(v, rd, (int)(((size_t)b * per_block + i) * 4), 0, 16);
        }
      } else {
        for (int i = threadIdx.x * 4; i < per_block; i += 512 * 4) {
          const float f = (float)(s + 1) + (float)b * 0.001f;
          *reinterpret_cast<float4*>(dst + (size_t)b * per_block + i) = make_float4(f, f, f, f);
        }
      }
    }
    grid_barrier<KIND>(S, (unsigned)s, xcc, nwg);
    if (per_block > 0) {
      float got = 0.0f;
      if (SC1_LD) {
        auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, (int)bytes, 0x00020000);
        for (int i = threadIdx.x * 4; i < per_block; i += 512 * 4) {
          const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((size_t)from * per_block + i) * 4), 0, 16);
          got += __uint_as_float(v.x) + __uint_as_float(v.w);
        }
      } else {
        for (int i = threadIdx.x * 4; i < per_block; i += 512 * 4) {
          const float4 v = *reinterpret_cast<const float4*>(dst + (size_t)from * per_block + i);
          got += v.x + v.w;
        }
      }
      const float want = 2.0f * ((float)(s + 1) + (float)from * 0.001f) * (float)((per_block / 4 + 511 - (int)threadIdx.x) / 512);
      acc += (got == want) ? 0.0f : 1.0f;                      // stale data = a mismatch count
    }
    // (a slice is rewritten two phases later: every reader of phase s has passed barrier s+1 by then)
  }
  if (acc != 0.0f) atomicAdd(out, acc);
}

template <int KIND, bool SC1, bool SC1_LD = SC1>
static int barrier_row(const char* what, int nwg, int per_block) {
  float *b0, *b1, *out; BarState* S;
  const size_t n = (size_t)nwg * (per_block > 0 ? per_block : 4);
  CK(hipMalloc(&b0, n * 4)); CK(hipMalloc(&b1, n * 4)); CK(hipMalloc(&out, 4)); CK(hipMalloc(&S, sizeof(BarState)));
  CK(hipMemset(b0, 0, n * 4)); CK(hipMemset(b1, 0, n * 4)); CK(hipMemset(out, 0, 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float t[2] = {0, 0}; const int P[2] = {20, 60};
  for (int k = 0; k < 2; ++k)
    for (int w = 0; w < 3; ++w) {
      CK(hipMemsetAsync(S, 0, sizeof(BarState), st));
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL((persistent<KIND, SC1, SC1_LD>), dim3(nwg), dim3(512), 0, st, b0, b1, S, P[k], per_block, out);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t[k] = ms * 1000.f;
    }
  unsigned ab = 0; CK(hipMemcpy(&ab, (char*)S + offsetof(BarState, abort_flag), 4, hipMemcpyDeviceToHost));
  unsigned census[8]; CK(hipMemcpy(census, S, sizeof census, hipMemcpyDeviceToHost));
  float bad = 0; CK(hipMemcpy(&bad, out, 4, hipMemcpyDeviceToHost));
  printf("  %-46s %4d WG, %2d KB/WG/phase: %6.2f us/phase (slope of 20 vs 60 phases; 60-phase launch %7.1f us)  stale %.0f %s  census %u %u %u %u %u %u %u %u\n",
         what, nwg, per_block * 4 / 1024, (t[1] - t[0]) / (P[1] - P[0]), t[1], bad, ab ? "ABORT" : "ok",
         census[0], census[1], census[2], census[3], census[4], census[5], census[6], census[7]);
  fflush(stdout);
  CK(hipFree(b0)); CK(hipFree(b1)); CK(hipFree(out)); CK(hipFree(S)); CK(hipStreamDestroy(st));
  return ab ? 2 : 0;
}

static int section_barrier() {
  printf("== B. grid barrier inside ONE persistent launch (512-thread workgroups; phase = write slice, barrier, read another workgroup's slice)\n");
  for (int nwg : {256, 512}) {
    for (int pb : {0, 4096, 16384}) {
      if (barrier_row<BAR_FLAT_FENCE, false>("flat counter, release+acquire fences (r1 form)", nwg, pb)) return 1;
      if (barrier_row<BAR_XCD_FENCE, false>("XCD-hierarchical, fences, plain payload", nwg, pb)) return 1;
      if (barrier_row<BAR_FLAT_NOFENCE, true>("flat counter, NO fences, sc1 payload", nwg, pb)) return 1;
      if (barrier_row<BAR_XCD_NOFENCE, true>("XCD-hierarchical, NO fences, sc1 payload", nwg, pb)) return 1;
      if (barrier_row<BAR_XCD_LOCAL, false, true>("XCC-LOCAL barrier, plain stores + sc1 loads", nwg, pb)) return 1;
    }
  }
  return 0;
}

// ============================================================ C. dataflow chain inside one launch ===========================
// workgroup b of the grid is stage s = b / G, slot i = b % G; it reads D producer slices of stage s-1 (per_block / D floats each),
// adds 1, writes its own slice of stage s.  Per-stage buffers (no WAR hazard).  Flags: one word per producer workgroup.
//   PROTO 0: plain stores, release store of the flag | relaxed polls, acquire fence, plain loads          (round-2 form)
//   PROTO 1: sc1 stores, vmcnt(0) by every wave, barrier, relaxed flag store | relaxed polls, sc1 loads  (guide R1, no fences)
template <int D, int PROTO>
__global__ void __launch_bounds__(512) chain(float* buf, unsigned* flags, unsigned* abort_flag, int G, int per_block, unsigned epoch) {
  const int s = blockIdx.x / G, b = blockIdx.x % G;
  const size_t stage_elems = (size_t)G * per_block;
  const float* src = buf + (size_t)(s > 0 ? s - 1 : 0) * stage_elems;
  float* dst = buf + (size_t)s * stage_elems;
  const int from0 = (b * 37 + 11 + s) % G;
  if (s > 0) {
    if (threadIdx.x < D) {
      unsigned* f = flags + (size_t)(s - 1) * G + (from0 + threadIdx.x) % G;
      long spins = 0;
      while (__hip_atomic_load(f, RLX_AGENT) != epoch) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255) == 0 && (spins > 2000000 || __hip_atomic_load(abort_flag, RLX_AGENT))) { __hip_atomic_store(abort_flag, 1u, RLX_AGENT); break; }
      }
    }
    __syncthreads();
    if (PROTO == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  const int part = per_block / D;
  if (PROTO == 1) {
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(stage_elems * 4), 0x00020000);
    auto rd = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, (int)(stage_elems * 4), 0x00020000);
    for (int i = threadIdx.x * 4; i < per_block; i += 512 * 4) {
      const int from = (from0 + i / part) % G;
      u4 v;
      if (s > 0) v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((size_t)from * per_block + i) * 4), 0, 16);
      else { v.x = v.y = v.z = v.w = 0u; }
      v.x = __float_as_uint(__uint_as_float(v.x) + 1.0f); v.y = __float_as_uint(__uint_as_float(v.y) + 1.0f);
      v.z = __float_as_uint(__uint_as_float(v.z) + 1.0f); v.w = __float_as_uint(__uint_as_float(v.w) + 1.0f);
      __builtin_amdgcn_raw_buffer_store_b128(v, rd, (int)(((size_t)b * per_block + i) * 4), 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flags + (size_t)s * G + b, epoch, RLX_AGENT);
  } else {
    for (int i = threadIdx.x * 4; i < per_block; i += 512 * 4) {
      const int from = (from0 + i / part) % G;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (s > 0) v = *reinterpret_cast<const float4*>(src + (size_t)from * per_block + i);
      v.x += 1.0f; v.y += 1.0f; v.z += 1.0f; v.w += 1.0f;
      *reinterpret_cast<float4*>(dst + (size_t)b * per_block + i) = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flags + (size_t)s * G + b, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ void __launch_bounds__(512) stage_k(float* buf, int s, int G, int per_block, int D) {
  const int b = blockIdx.x;
  const size_t stage_elems = (size_t)G * per_block;
  const float* src = buf + (size_t)(s > 0 ? s - 1 : 0) * stage_elems;
  float* dst = buf + (size_t)s * stage_elems;
  const int from0 = (b * 37 + 11 + s) % G, part = per_block / D;
  for (int i = threadIdx.x * 4; i < per_block; i += 512 * 4) {
    const int from = (from0 + i / part) % G;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s > 0) v = *reinterpret_cast<const float4*>(src + (size_t)from * per_block + i);
    v.x += 1.0f; v.y += 1.0f; v.z += 1.0f; v.w += 1.0f;
    *reinterpret_cast<float4*>(dst + (size_t)b * per_block + i) = v;
  }
}

template <int D>
static int chain_row(int S, int G, int per_block) {
  float* buf; unsigned *flags, *ab;
  const size_t elems = (size_t)S * G * per_block;
  CK(hipMalloc(&buf, elems * 4)); CK(hipMalloc(&flags, (size_t)S * G * 4)); CK(hipMalloc(&ab, 4));
  CK(hipMemset(flags, 0, (size_t)S * G * 4)); CK(hipMemset(ab, 0, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipStream_t st; CK(hipStreamCreate(&st));
  const int reps = 100; float ms = 0;
  // dependent launches, GPU-side: as a hipGraph (one graph = reps x S nodes)
  float us_launch = 0;
  {
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int r = 0; r < 20; ++r) for (int s = 0; s < S; ++s) hipLaunchKernelGGL(stage_k, dim3(G), dim3(512), 0, st, buf, s, G, per_block, D);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ex, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st)); for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ex, st)); CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); us_launch = ms * 1000.f / (100 * S);
    CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
  }
  std::vector<float> ref((size_t)G * per_block), got((size_t)G * per_block);
  CK(hipMemcpy(ref.data(), buf + (size_t)(S - 1) * G * per_block, ref.size() * 4, hipMemcpyDeviceToHost));
  float us_chain[2] = {0, 0}; size_t bad[2] = {0, 0}; unsigned habort = 0;
  unsigned epoch = 0;
  for (int proto = 0; proto < 2 && !habort; ++proto) {
    for (int w = 0; w < 2; ++w) {
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) {
        ++epoch;
        if (proto == 0) hipLaunchKernelGGL((chain<D, 0>), dim3(S * G), dim3(512), 0, st, buf, flags, ab, G, per_block, epoch);
        else hipLaunchKernelGGL((chain<D, 1>), dim3(S * G), dim3(512), 0, st, buf, flags, ab, G, per_block, epoch);
      }
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    us_chain[proto] = ms * 1000.f / reps / S;
    CK(hipMemset(buf, 0, elems * 4));
    ++epoch;
    if (proto == 0) hipLaunchKernelGGL((chain<D, 0>), dim3(S * G), dim3(512), 0, st, buf, flags, ab, G, per_block, epoch);
    else hipLaunchKernelGGL((chain<D, 1>), dim3(S * G), dim3(512), 0, st, buf, flags, ab, G, per_block, epoch);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(got.data(), buf + (size_t)(S - 1) * G * per_block, got.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < ref.size(); ++i) bad[proto] += ref[i] != got[i];
    CK(hipMemcpy(&habort, ab, 4, hipMemcpyDeviceToHost));
  }
  printf("  S %2d x G %4d WG x512 thr, %2d KB/WG, %2d producers each: dependent launches (graph) %5.2f us/stage | dataflow, release/acquire fences %6.2f (bad %zu) | dataflow, sc1 stores+loads, no fences %6.2f (bad %zu) %s\n",
         S, G, per_block * 4 / 1024, D, us_launch, us_chain[0], bad[0], us_chain[1], bad[1], habort ? "ABORT" : "ok");
  fflush(stdout);
  CK(hipFree(buf)); CK(hipFree(flags)); CK(hipFree(ab)); CK(hipStreamDestroy(st));
  return habort ? 2 : 0;
}
static int section_chain() {
  printf("== C. dataflow chain in ONE launch (stage s's workgroups wait on per-producer flags of stage s-1) vs S dependent launches\n");
  for (int G : {256, 512}) {
    if (chain_row<1>(10, G, 4096)) return 1;
    if (chain_row<16>(10, G, 4096)) return 1;
  }
  if (chain_row<16>(5, 256, 4096)) return 1;
  if (chain_row<16>(10, 256, 16384)) return 1;
  if (chain_row<4>(10, 800, 4096)) return 1;
  return 0;
}

// ============================================================ D. store flavour vs the end-of-kernel write-back ================
// A kernel boundary writes back every dirty L2 line the predecessor left ("+ B / 6 TB/s", guide row `boundary`).  Do write-through
// stores (issued while the kernel still computes) make the boundary cheaper than plain stores whose lines are flushed at the end?
// Stage = every workgroup reads 16 KB of the previous stage (plain 16-B loads), spins on MFMA-free ALU work for `work` iterations,
// and writes `out_floats` floats with store flavour F: 0 plain dwordx4, 1 sc1 dwordx4, 2 nt dwordx4, 3 sc0 sc1 dwordx4,
// 4 sc1 DWORD stores (lanes along the row, what a lane-per-column epilogue issues), 5 plain dword stores.
template <int F>
__global__ void __launch_bounds__(512) store_stage(const float* src, float* dst, int s, int in_floats, int out_floats, int work) {
  const int G = gridDim.x, b = blockIdx.x;
  const int from = (b * 37 + 11 + s) % G;
  float acc = 0.0f;
  for (int i = threadIdx.x * 4; i < in_floats; i += 512 * 4) {
    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)from * out_floats + i);
    acc += v.x + v.y + v.z + v.w;
  }
  for (int k = 0; k < work; ++k) acc = acc * 1.000001f + 0.5f;          // a dependent ALU chain: the "compute" the stores could hide under
  auto rd = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, (int)((size_t)G * out_floats * 4), 0x00020000);
  if (F <= 3) {
    for (int i = threadIdx.x * 4; i < out_floats; i += 512 * 4) {
      u4 v; v.x = v.y = v.z = v.w = __float_as_uint(acc + (float)i);
      const int off = (int)(((size_t)b * out_floats + i) * 4);
      if (F == 0) *reinterpret_cast<u4*>(dst + (size_t)b * out_floats + i) = v;
      else __builtin_amdgcn_raw_buffer_store_b128(v, rd, off, 0, F == 1 ? 16 : (F == 2 ? 2 : 17));
    }
  } else {
    for (int i = threadIdx.x; i < out_floats; i += 512) {
      const float v = acc + (float)i;
      if (F == 4) __hip_atomic_store(dst + (size_t)b * out_floats + i, v, RLX_AGENT);
      else dst[(size_t)b * out_floats + i] = v;
    }
  }
}
template <int F>
static int store_row(const char* what, int G, int out_kb, int work) {
  const int out_floats = out_kb * 256, in_floats = 4096;
  float *b0, *b1; CK(hipMalloc(&b0, (size_t)G * out_floats * 4)); CK(hipMalloc(&b1, (size_t)G * out_floats * 4));
  CK(hipMemset(b0, 0, (size_t)G * out_floats * 4)); CK(hipMemset(b1, 0, (size_t)G * out_floats * 4));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipGraph_t g; hipGraphExec_t ex; float ms = 0;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((store_stage<F>), dim3(G), dim3(512), 0, st, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1, i, in_floats, out_floats, work);
  CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ex, st)); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st)); for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ex, st)); CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  printf("  %-34s %4d WG x512 thr, 16 KB in, %3d KB out per WG (%5.1f MB per stage), ALU work %5d: %6.2f us/stage\n",
         what, G, out_kb, G * out_kb / 1024.0, work, ms * 1000.f / 500);
  fflush(stdout);
  CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g)); CK(hipFree(b0)); CK(hipFree(b1)); CK(hipStreamDestroy(st));
  return 0;
}
static int section_stores() {
  printf("== D. store flavour of a stage's OUTPUT vs the dirty-line write-back at the kernel boundary (dependent launches, hipGraph)\n");
  for (int work : {0, 4000}) for (int G : {256, 512}) for (int kb : {16, 64, 128}) {
    if (store_row<0>("plain dwordx4", G, kb, work)) return 1;
    if (store_row<1>("sc1 dwordx4 (write-through)", G, kb, work)) return 1;
    if (store_row<2>("nt dwordx4", G, kb, work)) return 1;
    if (store_row<3>("sc0 sc1 dwordx4", G, kb, work)) return 1;
    if (store_row<5>("plain dword (lane per column)", G, kb, work)) return 1;
    if (store_row<4>("sc1 dword (lane per column)", G, kb, work)) return 1;
  }
  return 0;
}

int main(int argc, char** argv) {
  const char* what = argc > 1 ? argv[1] : "all";
  const char* lib = argc > 2 ? argv[2] : nullptr;
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s, %d CUs\n", p.gcnArchName, p.multiProcessorCount);
  if (!strcmp(what, "all") || !strcmp(what, "boundary")) if (section_boundary(lib)) return 1;
  if (!strcmp(what, "all") || !strcmp(what, "barrier")) if (section_barrier()) return 2;
  if (!strcmp(what, "all") || !strcmp(what, "chain")) if (section_chain()) return 3;
  if (!strcmp(what, "all") || !strcmp(what, "stores")) if (section_stores()) return 4;
  return 0;
}
