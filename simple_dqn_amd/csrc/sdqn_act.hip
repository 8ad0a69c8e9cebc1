// sdqn_act.hip — the acting forward (agent.py:48-59 -> deepqnetwork.py:175-184, batch of ONE state) as ONE launch.
//
// Round 3 ran it as the train step's five forward launches with B = 1: ~33 us per call, almost all of it five dependent kernel
// boundaries + cold starts + five host-side launches for 19 MFLOP of arithmetic.  Here the whole forward is one persistent launch
// built on the only in-launch hand-off that beat a kernel boundary in the round-3 measurements (tools/exp/handoff_r3.hip, table B):
// the XCC-LOCAL one — producer and consumer share an L2, plain stores + `sc1` loads, no fences.
//
//   * every XCC (32 CUs, one L2) computes conv1 -> conv2 -> conv3 of the state REDUNDANTLY into its own scratch copy: nothing of the
//     conv chain ever crosses an XCC (the chip is otherwise idle: redundancy is free, a cross-XCC exchange is not);
//   * work is claimed through per-XCC ticket counters keyed by HW_REG_XCC_ID, in topological order (all conv1 items, then conv2, then
//     conv3), and a consumer item waits until the producer PHASE has signalled all its items (`done` counters with KNOWN item
//     counts) — so correctness does not depend on how many workgroups an XCC received or where they run: a ticket's producers were
//     all claimed earlier by workgroups that are running, never blocked behind it;
//   * fc4 + fc5 stay XCC-local too: the 512 hidden units are 8 STRIPES of 64; an XCC takes a stripe (claimed from one global counter when
//     its tickets reach a new stripe slot: 8 XCCs take one each, fewer XCCs take several), its workgroups compute the stripe's 32 K-chunks
//     (98 rows x 64 units: the W4 loads are issued BEFORE the wait for conv3 — they do not depend on it), the workgroup that signals
//     the stripe's last chunk reduces the chunks in fixed order, applies Rectlin and multiplies by the stripe's rows of W5: a PARTIAL
//     Q-vector per stripe, written to slot [stripe][A] of the destination (system scope when that is mapped host memory).  The HOST adds
//     the 8 stripe partials in stripe order: no cross-XCC reduction, fence or write-through inside the launch at all (the first form of
//     this kernel reduced across XCCs through the fabric: 3.4 us of its 17).
//
// Arithmetic: fp32 MFMA 16x16x4 (exact fp32 products, fp32 accumulation), K split over the 4 waves of a workgroup and combined in
// fixed order through LDS; conv1 reads the bytes directly and divides the sum by 255 (deepqnetwork.py:100 scales the input instead:
// same value to fp32 round-off).  Deterministic: every sum has one fixed order, whatever the claim order was.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "launch.h"

namespace sdqn {
namespace act {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int NT = 256;
constexpr int I1 = PIX1 / 16;                        // 25 conv1 items per XCC: 16 positions x 32 maps
constexpr int I2 = ((PIX2 + 15) / 16) * (K2 / 16);   // 24 conv2 items: 16 positions x 16 maps
constexpr int I3 = ((PIX3 + 15) / 16) * (K3 / 16);   // 16 conv3 items
constexpr int IX = I1 + I2 + I3;
constexpr int KCH = ACT_KCH, ROWS = NIN4 / KCH, NST = NFC / 64;                      // fc4: 8 stripes of 64 units x 32 K-chunks of 98 rows
static_assert(PIX1 % 16 == 0 && NIN4 % KCH == 0 && ROWS <= 7 * 16, "item shapes");
constexpr int A1O = 0, A2O = PIX1 * K1, A3O = A2O + PIX2 * K2, XPO = 21248;          // one XCC's scratch (floats): a1 | a2 | a3 | fc4 partials [stripe][chunk][64]
static_assert(A3O + NIN4 <= XPO && XPO + NST * KCH * 64 <= ACT_XCC_FLOATS, "scratch copy");
// control block of one launch (32-bit words; every counter in its own 64-byte line)
constexpr int C_TICK = 0, C_DONE = 8 * 16, C_XS = C_DONE + 24 * 16, C_XD = C_XS + 8 * 16, C_GSTRIPE = C_XD + 8 * 16, C_ABORT = C_GSTRIPE + 16;
// (C_XS + 16 x + j: stripe id + 1 of XCC x's j-th stripe slot; C_XD + 16 x + j: chunks of that slot signalled)
static_assert(C_ABORT + 16 <= ACT_CTL_WORDS, "control block");
constexpr long SPIN_LIMIT = 1 << 17;                 // ~0.1 s of polling: a lost hand-off ends the launch (host falls back), never hangs it

__device__ __forceinline__ unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 7; }
__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// 16-byte `sc1` load (bypasses this CU's vector L1: data another CU of the same XCC stored a moment ago is read from the shared L2)
__device__ __forceinline__ f32x4 ld4_sc1(__amdgpu_buffer_rsrc_t rs, int float_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, float_off * 4, 0, 16);
  f32x4 f; f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y); f.z = __uint_as_float(v.z); f.w = __uint_as_float(v.w); return f;
}
__device__ __forceinline__ float ld_sc1(const float* p) { return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), RLX_AGENT)); }

struct Wg {
  int tid, lane, w, r, kq;
  unsigned* ctl;
  float* red;                 // LDS
  unsigned* bcast;            // LDS
  unsigned long long* stamps; int nst;
};
template <bool ST>
__device__ __forceinline__ void stamp(Wg& g, unsigned kind) {
  if constexpr (ST) {
    if (g.tid == 0 && g.nst < ACT_STAMPS) { g.stamps[2 * g.nst] = kind; g.stamps[2 * g.nst + 1] = clock64(); ++g.nst; }
  }
}
// thread 0 polls; false = aborted (a peer gave up, or this poll ran out of patience)
__device__ __forceinline__ bool wait_ge(Wg& g, int word, unsigned target) {
  if (g.tid == 0) {
    long spins = 0; unsigned ok = 1;
    while ((int)(__hip_atomic_load(g.ctl + word, RLX_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 255) == 0 && (spins > SPIN_LIMIT || __hip_atomic_load(g.ctl + C_ABORT, RLX_AGENT))) { __hip_atomic_store(g.ctl + C_ABORT, 1u, RLX_AGENT); ok = 0; break; }
    }
    g.bcast[1] = ok;
  }
  __syncthreads();
  const bool ok = g.bcast[1] != 0;
  __syncthreads();
  return ok;
}
// thread 0 polls a word until it is non-zero; returns it (0 = aborted)
__device__ __forceinline__ unsigned wait_nonzero(Wg& g, int word) {
  if (g.tid == 0) {
    long spins = 0; unsigned v;
    while ((v = __hip_atomic_load(g.ctl + word, RLX_AGENT)) == 0u) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 255) == 0 && (spins > SPIN_LIMIT || __hip_atomic_load(g.ctl + C_ABORT, RLX_AGENT))) { __hip_atomic_store(g.ctl + C_ABORT, 1u, RLX_AGENT); break; }
    }
    g.bcast[1] = v;
  }
  __syncthreads();
  const unsigned v = g.bcast[1];
  __syncthreads();
  return v;
}
// every wave has drained its stores (they are in the XCC's L2 / written through), then ONE relaxed increment
__device__ __forceinline__ unsigned signal(Wg& g, int word) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned old = 0;
  if (g.tid == 0) { old = __hip_atomic_fetch_add(g.ctl + word, 1u, RLX_AGENT); g.bcast[2] = old; }
  __syncthreads();
  old = g.bcast[2];
  __syncthreads();
  return old;
}

// the same without the old value: nobody waits for the increment's round trip
__device__ __forceinline__ void signal_nr(Wg& g, int word) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (g.tid == 0) (void)__hip_atomic_fetch_add(g.ctl + word, 1u, RLX_AGENT);
}

// fixed-order combine of the 4 waves' partial 16x16 blocks: red[(w * NB + jb) * 256 + lane * 4 + i]; thread t owns element t of each block
template <int NB>
__device__ __forceinline__ void put_partial(Wg& g, int jb, const f32x4& acc) {
  *reinterpret_cast<f32x4*>(g.red + ((g.w * NB + jb) * 256 + g.lane * 4)) = acc;
}
template <int NB>
__device__ __forceinline__ float combined(const Wg& g, int jb) {
  float s = g.red[(0 * NB + jb) * 256 + g.tid];
#pragma unroll
  for (int w = 1; w < 4; ++w) s += g.red[(w * NB + jb) * 256 + g.tid];
  return s;
}
// element t of a 16x16 accumulator block: (row, col) of the MFMA C/D map (col = lane & 15, row = 4 * (lane >> 4) + i)
__device__ __forceinline__ void elem_rc(int t, int& row, int& col) { const int ln = t >> 2, i = t & 3; row = 4 * (ln >> 4) + i; col = ln & 15; }

// conv1 (deepqnetwork.py:83; k = c*64 + ky*8 + kx): rows 16*rt .. +15 (output positions), all 32 maps; wave w = input channel w
template <bool ST>
__device__ __forceinline__ void conv1_item(const ActArgs& a, Wg& g, float* sx, int rt) {
  const int m = 16 * rt + g.r, p = m / Q1, q = m - p * Q1;
  const uint8_t* src = a.state + g.w * FRAME + (ST1 * p + (g.kq >> 1)) * W0 + ST1 * q + 4 * (g.kq & 1);
  uint32_t ab[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) ab[j] = *reinterpret_cast<const uint32_t*>(src + 2 * j * W0);       // 4 consecutive kx of patch row 2j + (kq >> 1)
  const float* wb = a.theta + OFF1 + (64 * g.w + 4 * g.kq) * K1 + 2 * g.r;                        // (lane & 15 = column pair here)
  f32x2 b[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s) b[j][s] = *reinterpret_cast<const f32x2*>(wb + (16 * j + s) * K1);
  if constexpr (ST) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp<ST>(g, (11u << 16) | rt); }
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float av = (float)((ab[j] >> (8 * s)) & 255u);
      acc0 = mfma(av, b[j][s].x, acc0); acc1 = mfma(av, b[j][s].y, acc1);
    }
  stamp<ST>(g, (12u << 16) | rt);
  put_partial<2>(g, 0, acc0); put_partial<2>(g, 1, acc1);
  __syncthreads();
  int row, col; elem_rc(g.tid, row, col);
  f32x2 o; o.x = fmaxf(combined<2>(g, 0) / 255.0f, 0.0f); o.y = fmaxf(combined<2>(g, 1) / 255.0f, 0.0f);      // :100 (scale), Rectlin
  *reinterpret_cast<f32x2*>(sx + A1O + (16 * rt + row) * K1 + 2 * col) = o;
}

// conv2 (:85; k = ky*128 + kx*32 + c: one patch row = 128 contiguous floats of a1): 16 positions x 16 maps; wave w = patch row ky
// (the weight operand does not depend on the producer phase: its loads are in flight while `ready()` waits for conv1 to complete)
template <bool ST, class Ready>
__device__ __forceinline__ bool conv2_item(const ActArgs& a, Wg& g, float* sx, __amdgpu_buffer_rsrc_t rs, int it, Ready ready) {
  const int rt = it >> 2, cb = it & 3;
  const int m = min(16 * rt + g.r, PIX2 - 1), p = m / Q2, q = m - p * Q2;
  const int abase = A1O + ((ST2 * p + g.w) * Q1 + ST2 * q) * K1 + 4 * g.kq;
  const float* wb = a.theta + OFF2 + (128 * g.w + 4 * g.kq) * K2 + 16 * cb + g.r;
  f32x4 av[8]; float b[8][4];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s) b[j][s] = wb[(16 * j + s) * K2];
  if (!ready()) return false;
  stamp<ST>(g, (3u << 16) | it);
#pragma unroll
  for (int j = 0; j < 8; ++j) av[j] = ld4_sc1(rs, abase + 16 * j);
  if constexpr (ST) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp<ST>(g, (13u << 16) | it); }
  f32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    acc = mfma(av[j].x, b[j][0], acc); acc = mfma(av[j].y, b[j][1], acc); acc = mfma(av[j].z, b[j][2], acc); acc = mfma(av[j].w, b[j][3], acc);
  }
  stamp<ST>(g, (14u << 16) | it);
  put_partial<1>(g, 0, acc);
  __syncthreads();
  int row, col; elem_rc(g.tid, row, col);
  const int mo = 16 * rt + row;
  if (mo < PIX2) sx[A2O + mo * K2 + 16 * cb + col] = fmaxf(combined<1>(g, 0), 0.0f);
  return true;
}

// conv3 (:87; k = (r*3+s)*64 + c): 16 positions x 16 maps; wave w = k in [144 w, 144 w + 144)
template <bool ST, class Ready>
__device__ __forceinline__ bool conv3_item(const ActArgs& a, Wg& g, float* sx, __amdgpu_buffer_rsrc_t rs, int it, Ready ready) {
  const int rt = it >> 2, cb = it & 3;
  const int m = min(16 * rt + g.r, PIX3 - 1), p = m / Q3, q = m - p * Q3;
  const float* wb = a.theta + OFF3 + 16 * cb + g.r;
  f32x4 av[9]; float b[9][4];
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s) b[j][s] = wb[(144 * g.w + 16 * j + 4 * g.kq + s) * K3];
  if (!ready()) return false;
  stamp<ST>(g, (5u << 16) | it);
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const int k = 144 * g.w + 16 * j + 4 * g.kq, rs_ = k >> 6, c = k & 63, rr = rs_ / 3, ss = rs_ - 3 * rr;
    av[j] = ld4_sc1(rs, A2O + ((p + rr) * Q2 + q + ss) * K2 + c);
  }
  if constexpr (ST) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp<ST>(g, (15u << 16) | it); }
  f32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    acc = mfma(av[j].x, b[j][0], acc); acc = mfma(av[j].y, b[j][1], acc); acc = mfma(av[j].z, b[j][2], acc); acc = mfma(av[j].w, b[j][3], acc);
  }
  stamp<ST>(g, (16u << 16) | it);
  put_partial<1>(g, 0, acc);
  __syncthreads();
  int row, col; elem_rc(g.tid, row, col);
  const int mo = 16 * rt + row;
  if (mo < PIX3) sx[A3O + mo * K3 + 16 * cb + col] = fmaxf(combined<1>(g, 0), 0.0f);
  return true;
}

template <bool QSYS, bool ST>
__global__ void __launch_bounds__(NT) act_kernel(const ActArgs a) {
  __shared__ float red[4 * 2 * 256];
  __shared__ unsigned bcast[4];
  Wg g;
  g.tid = threadIdx.x; g.lane = g.tid & 63; g.w = g.tid >> 6; g.r = g.lane & 15; g.kq = g.lane >> 4;
  g.ctl = a.ctl + (size_t)(a.seq & 3u) * ACT_CTL_WORDS; g.red = red; g.bcast = bcast;
  g.stamps = ST ? a.stamps + (size_t)blockIdx.x * 2 * ACT_STAMPS : nullptr; g.nst = 0;
  stamp<ST>(g, 0);
  if (blockIdx.x == 0) {              // the control block of the launch after next (the stream runs acting launches in order)
    unsigned* nx = a.ctl + (size_t)((a.seq + 2u) & 3u) * ACT_CTL_WORDS;
    for (int i = g.tid; i < ACT_CTL_WORDS; i += NT) nx[i] = 0u;
  }
  const unsigned x = xcc_id();
  float* sx = a.scratch + (size_t)x * ACT_XCC_FLOATS;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)sx, 0, ACT_XCC_FLOATS * 4, 0x00020000);
  if constexpr (ST) { if (g.tid == 0) a.stamps[(size_t)blockIdx.x * 2 * ACT_STAMPS + 2 * ACT_STAMPS - 1] = x; }

  // ---- the conv chain of this XCC: tickets in topological order ------------------------------------------------------------------------
  // The NEXT ticket is claimed before the current item starts (its round trip hides behind the item).  Still deadlock-free: the smallest
  // unfinished ticket is always its holder's CURRENT one (a held-ahead ticket is larger than its holder's current one), and all its
  // producers are smaller, hence finished.
  int have = 0;                                             // phases known complete on this XCC
  unsigned* tick = g.ctl + C_TICK + 16 * x;
  if (g.tid == 0) bcast[0] = __hip_atomic_fetch_add(tick, 1u, RLX_AGENT);
  __syncthreads();
  int t = (int)bcast[0];
  const int u4 = g.tid & 15, kg = g.tid >> 4;
  while (t < IX + NST * KCH) {
    unsigned tn = 0;
    if (g.tid == 0) tn = __hip_atomic_fetch_add(tick, 1u, RLX_AGENT);
    if (t < I1) {
      stamp<ST>(g, (1u << 16) | t);
      conv1_item<ST>(a, g, sx, t);
      signal_nr(g, C_DONE + 16 * (3 * x + 0));
      stamp<ST>(g, (2u << 16) | t);
    } else if (t < I1 + I2) {
      if (!conv2_item<ST>(a, g, sx, rs, t - I1, [&]() { if (have < 1) { if (!wait_ge(g, C_DONE + 16 * (3 * x + 0), I1)) return false; have = 1; } return true; })) return;
      signal_nr(g, C_DONE + 16 * (3 * x + 1));
      stamp<ST>(g, (4u << 16) | (t - I1));
    } else if (t < IX) {
      if (!conv3_item<ST>(a, g, sx, rs, t - I1 - I2, [&]() { if (have < 2) { if (!wait_ge(g, C_DONE + 16 * (3 * x + 1), I2)) return false; have = 2; } return true; })) return;
      signal_nr(g, C_DONE + 16 * (3 * x + 2));
      stamp<ST>(g, (6u << 16) | (t - I1 - I2));
    } else {
      // ---- fc4 (:89) + fc5 (:91): chunk c of this XCC's j-th stripe slot ---------------------------------------------------------------
      const int f = t - IX, j = f / KCH, c = f - j * KCH;
      unsigned sid1;
      if (c == 0) {
        // the slot's head ticket takes the next unclaimed stripe for this XCC — in slot order (it first sees slot j - 1 published), so
        // that on every XCC "slot j found no stripe" implies the same for all later slots
        if (g.tid == 0) {
          unsigned v = (unsigned)NST + 1u, prev = 1u;
          if (j > 0) {
            long spins = 0;
            while ((prev = __hip_atomic_load(g.ctl + C_XS + 16 * x + j - 1, RLX_AGENT)) == 0u) {
              __builtin_amdgcn_s_sleep(1);
              if ((++spins & 255) == 0 && (spins > SPIN_LIMIT || __hip_atomic_load(g.ctl + C_ABORT, RLX_AGENT))) { __hip_atomic_store(g.ctl + C_ABORT, 1u, RLX_AGENT); break; }
            }
          }
          if (prev == 0u) v = 0u;                                                                   // (aborted)
          else if (prev <= (unsigned)NST) v = __hip_atomic_fetch_add(g.ctl + C_GSTRIPE, 1u, RLX_AGENT) + 1u;
          if (v) __hip_atomic_store(g.ctl + C_XS + 16 * x + j, v, RLX_AGENT);
          bcast[1] = v;
        }
        __syncthreads();
        sid1 = bcast[1];
        __syncthreads();
      } else {
        sid1 = wait_nonzero(g, C_XS + 16 * x + j);
      }
      if (sid1 == 0u) return;
      if (sid1 > (unsigned)NST) {
        // every stripe has an owner: this and all later slots of the XCC are empty.  The ticket held ahead is abandoned — unless it is a
        // slot's head, whose publication others may be waiting for
        if (g.tid == 0 && (int)tn >= IX && (int)tn < IX + NST * KCH && ((int)tn - IX) % KCH == 0)
          __hip_atomic_store(g.ctl + C_XS + 16 * x + ((int)tn - IX) / KCH, (unsigned)NST + 1u, RLX_AGENT);
        return;
      }
      const int sid = (int)sid1 - 1;
      stamp<ST>(g, (7u << 16) | (sid * KCH + c));
      const float* wp = a.theta + OFF4 + (size_t)(ROWS * c + kg) * NFC + 64 * sid + 4 * u4;
      f32x4 wv[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const bool ok = kg + 16 * i < ROWS;
        wv[i] = ok ? *reinterpret_cast<const f32x4*>(wp + (size_t)16 * i * NFC) : f32x4{0, 0, 0, 0};
      }
      if (have < 3) { if (!wait_ge(g, C_DONE + 16 * (3 * x + 2), I3)) return; have = 3; }
      stamp<ST>(g, (8u << 16) | (sid * KCH + c));
      f32x4 acc = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const bool ok = kg + 16 * i < ROWS;
        const float xv = ok ? ld_sc1(sx + A3O + ROWS * c + kg + 16 * i) : 0.0f;
        acc.x = fmaf(xv, wv[i].x, acc.x); acc.y = fmaf(xv, wv[i].y, acc.y); acc.z = fmaf(xv, wv[i].z, acc.z); acc.w = fmaf(xv, wv[i].w, acc.w);
      }
      *reinterpret_cast<f32x4*>(red + kg * 64 + 4 * u4) = acc;
      __syncthreads();
      float* xp = sx + XPO + sid * (KCH * 64);
      if (g.tid < 64) {
        float sum = red[g.tid];
#pragma unroll
        for (int k = 1; k < 16; ++k) sum += red[k * 64 + g.tid];
        xp[c * 64 + g.tid] = sum;                           // plain store: the reducer is on this XCC
      }
      const unsigned old = signal(g, C_XD + 16 * x + j);
      stamp<ST>(g, (9u << 16) | (sid * KCH + c));
      if (old == (unsigned)(KCH - 1)) {
        // the stripe's last chunk: K-chunk reduce in fixed order + Rectlin, then the stripe's share of every Q-value
        const int u = g.tid & 63, pq = g.tid >> 6;           // thread = (unit, quarter of the chunks)
        float w5v[(MAX_ACTIONS + 3) / 4];
#pragma unroll
        for (int i = 0; i < (MAX_ACTIONS + 3) / 4; ++i) { const int ac = g.w + 4 * i; w5v[i] = ac < a.A ? a.theta[OFF5 + ac * NFC + 64 * sid + u] : 0.0f; }
        float pv[KCH / 4];
#pragma unroll
        for (int k = 0; k < KCH / 4; ++k) pv[k] = ld_sc1(xp + (pq * (KCH / 4) + k) * 64 + u);
        float sq = pv[0];
#pragma unroll
        for (int k = 1; k < KCH / 4; ++k) sq += pv[k];
        red[pq * 64 + u] = sq;
        __syncthreads();
        const float hu = fmaxf(((red[u] + red[64 + u]) + red[128 + u]) + red[192 + u], 0.0f);      // every wave holds all 64 units of the stripe
#pragma unroll
        for (int i = 0; i < (MAX_ACTIONS + 3) / 4; ++i) {
          const int ac = g.w + 4 * i;                        // wave w: actions w, w + 4, ...
          if (ac < a.A) {
            float v = hu * w5v[i];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
            if (g.lane == 0) {
              if constexpr (QSYS) __hip_atomic_store(a.q + sid * ACT_Q_STRIDE + ac, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              else a.q[sid * ACT_Q_STRIDE + ac] = v;
            }
          }
        }
        stamp<ST>(g, (10u << 16) | sid);
      }
    }
    __syncthreads();                                        // (everybody has read bcast[0] / the combine buffer)
    if (g.tid == 0) bcast[0] = tn;
    __syncthreads();
    t = (int)bcast[0];
  }
}


}  // namespace act

hipError_t launch_act(const ActArgs& a, bool q_system_scope, hipStream_t s) {
  const dim3 grid(ACT_GRID), block(act::NT);
  if (a.stamps) {
    if (q_system_scope) SDQN_LAUNCH((act::act_kernel<true, true>), grid, block, 0, s, a);
    else SDQN_LAUNCH((act::act_kernel<false, true>), grid, block, 0, s, a);
  } else {
    if (q_system_scope) SDQN_LAUNCH((act::act_kernel<true, false>), grid, block, 0, s, a);
    else SDQN_LAUNCH((act::act_kernel<false, false>), grid, block, 0, s, a);
  }
  return hipGetLastError();
}


}  // namespace sdqn
