// update_body.h — the optimizer pass (split-K slab reduction + fc5 wgrad + RMSProp / Adam / Adadelta, deepqnetwork.py:165) as a
// device function, so that both the plain update launch (sdqn_kernels.hip) and round 3's fused "update + next step's conv1" launch
// (sdqn_kernels_r3.hip) run the very same code.
#pragma once
#include <cstddef>
#include "kernels.h"

namespace sdqn {

// FUSED (round 3, update + next step's conv1 in one launch): W1's new values leave with write-through (sc1) 16-byte stores, so the
// conv1 workgroups of the SAME launch can read them with sc1 loads once the W1 blocks have signalled (no fence anywhere)
template <bool FUSED = false>
__device__ inline void opt_apply4(float* __restrict__ theta, float* __restrict__ st1, float* __restrict__ st2,
                                  int64_t e, const float4& gs, const UpdateArgs& u, const float4* pre = nullptr) {
  // pre: theta[e] and state[e] already fetched by the caller (issued together with the slab loads: one memory round trip less)
  float4 w = pre ? pre[0] : *reinterpret_cast<float4*>(theta + e);
  float4 a = pre ? pre[1] : *reinterpret_cast<float4*>(st1 + e);
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  if (u.opt != 0) b = *reinterpret_cast<float4*>(st2 + e);
  w.x = opt_apply(w.x, a.x, b.x, gs.x, u); w.y = opt_apply(w.y, a.y, b.y, gs.y, u);
  w.z = opt_apply(w.z, a.z, b.z, gs.z, u); w.w = opt_apply(w.w, a.w, b.w, gs.w, u);
  if (FUSED && e < OFF2) {
    typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
    u4_t v; v.x = __float_as_uint(w.x); v.y = __float_as_uint(w.y); v.z = __float_as_uint(w.z); v.w = __float_as_uint(w.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, __builtin_amdgcn_make_buffer_rsrc((void*)theta, 0, NW1 * 4, 0x00020000), (int)(e * 4), 0, 16);
  } else if (u.wt) {                                  // write-through (nothing left to flush at the kernel boundary: sdqn_kernels_r3.hip)
    typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
    u4_t v; v.x = __float_as_uint(w.x); v.y = __float_as_uint(w.y); v.z = __float_as_uint(w.z); v.w = __float_as_uint(w.w);
    constexpr int FLAT_BYTES = (OFF5 + MAX_ACTIONS * NFC) * 4;                // (wave-uniform descriptors: base of the flat buffer + per-lane offset)
    __builtin_amdgcn_raw_buffer_store_b128(v, __builtin_amdgcn_make_buffer_rsrc((void*)theta, 0, FLAT_BYTES, 0x00020000), (int)(e * 4), 0, 16);
    v.x = __float_as_uint(a.x); v.y = __float_as_uint(a.y); v.z = __float_as_uint(a.z); v.w = __float_as_uint(a.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, __builtin_amdgcn_make_buffer_rsrc((void*)st1, 0, FLAT_BYTES, 0x00020000), (int)(e * 4), 0, 16);
  } else { *reinterpret_cast<float4*>(theta + e) = w; *reinterpret_cast<float4*>(st1 + e) = a; }
  if (FUSED && e < OFF2) *reinterpret_cast<float4*>(st1 + e) = a;
  if (u.opt != 0) *reinterpret_cast<float4*>(st2 + e) = b;
  if (u.w1p && e < OFF2) {                        // conv1's bf16 planes follow W1 (e = k * 32 + n: 4 consecutive maps of one k)
    const int k = (int)(e >> 5), n = (int)(e & 31);
    const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint16_t hi, mid, lo; split_bf16x3(wv[i], hi, mid, lo);
      u.w1p[(n + i) * CRS1 + k] = hi; u.w1p[W1P_PLANE + (n + i) * CRS1 + k] = mid; u.w1p[2 * W1P_PLANE + (n + i) * CRS1 + k] = lo;
    }
  }
  if (u.wh && e < OFF5) {                         // fp16 mode: refresh both half copies of these 4 weights
    const int L = e < OFF2 ? 0 : (e < OFF3 ? 1 : (e < OFF4 ? 2 : 3));
    const int off = L == 0 ? OFF1 : (L == 1 ? OFF2 : (L == 2 ? OFF3 : OFF4));
    const int K = L == 0 ? CRS1 : (L == 1 ? CRS2 : (L == 2 ? CRS3 : NIN4));
    const int N = L == 0 ? K1 : (L == 1 ? K2 : (L == 2 ? K3 : NFC));
    const int64_t r = e - off; const int k = (int)(r / N), n = (int)(r - (int64_t)k * N);
    const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { u.wh[e + i] = (half_t)wv[i]; u.wht[off + (int64_t)(n + i) * K + k] = (half_t)wv[i]; }
  }
}

// Two kinds of workgroups in one launch:
//   bid <  CONV_BLOCKS : conv parameters (77824 floats): split-K slab reduction, slab-parallel —
//                               32 float4 columns x 8 slab groups per workgroup, fixed-order LDS combine
//   bid >= CONV_BLOCKS : fc4 (g written by fc4_wgrad) and fc5 (wgrad computed here), elementwise
constexpr int CONV_F4 = OFF4 / 4;                 // 19456 float4 of conv parameters
constexpr int CONV_BLOCKS = CONV_F4 / 32;         // 608
constexpr int FC5_BLOCKS_PER_ACTION = NFC / 4 / 32;   // 4 workgroups of 32 float4 columns per action row

// OVF (fp16 data parallel only; compiled out of the default kernel): a half overflow in the all-reduced gradient skips the
// whole apply step on every rank
template <bool OVF, bool FUSED = false>
__device__ __forceinline__ void update_body(const UpdateArgs& u, const int bid, const int nblocks, float4 (*part)[32], float* cost_sh) {
  const int t = threadIdx.x;
#ifdef SDQN_TIMING
  SDQN_STAMP(0);
  struct StampAtExit { __device__ ~StampAtExit() { SDQN_STAMP(7); } } stamp_at_exit_;
#endif
  bool skip_apply = false;
  if constexpr (OVF) {
    skip_apply = u.ovf_flag[0] != 0;
    if (bid == 0 && t == 0) {                                 // (every block has read the flag and the scale by its own first lines;
      int* st = const_cast<int*>(u.ovf_flag);                       //  the two half passes of the NEXT step are later launches)
      if (skip_apply) { u.ovf_count[0] += 1; st[2] = 0; if (u.ovf_dynamic && st[1] > 0) st[1] -= 1; }
      else if (u.ovf_dynamic && ++st[2] >= 200) { st[2] = 0; if (st[1] < 15) st[1] += 1; }
    }
  }
  if (u.only_fc4 && bid < CONV_BLOCKS + u.A * FC5_BLOCKS_PER_ACTION) return;
  if (bid < CONV_BLOCKS) {
    const int c4 = t & 31, sg = t >> 5;
    const int64_t e = ((int64_t)bid * 32 + c4) * 4;
    float4 gs = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pre[2] = {gs, gs};
    const bool applies = sg == 0 && u.mode != 1 && !skip_apply;
    if (applies) { pre[0] = *reinterpret_cast<const float4*>(u.theta + e); pre[1] = *reinterpret_cast<const float4*>(u.state + e); }
    if (u.mode == 2) {
      if (sg == 0) gs = *reinterpret_cast<const float4*>(u.g + e);
    } else {
      const int L = e < OFF2 ? 0 : (e < OFF3 ? 1 : 2);                       // uniform per workgroup (128-float groups)
      const int64_t off = e - (L == 0 ? OFF1 : (L == 1 ? OFF2 : OFF3));
      const int64_t nw = L == 0 ? NW1 : (L == 1 ? NW2 : NW3);
      const float* sp = u.slab[L] + off;
      const int ns = u.ns[L];
      // slabs sg, sg + 8, sg + 16, ... in increasing order (the order IS the result: fixed).  Four loads are issued together whatever ns
      // (clamped index + select instead of a data-dependent loop: a 25-slab reduction was three dependent memory round trips for seven
      // of the eight slab groups)
      if (ns > 32) {
        // many slabs (B >= 128: conv1 has 320 at B = 256): SIXTEEN loads in flight per round trip — with four, a slab group walked its 40
        // slabs in ten dependent round trips and the whole launch took 10 us.  Same ascending order sg, sg + 8, ...: same bits.
        for (int s = sg; s < ns; s += 128) {
          float4 v[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) { const int sj = s + 8 * j; v[j] = *reinterpret_cast<const float4*>(sp + (int64_t)(sj < ns ? sj : ns - 1) * nw); }
#pragma unroll
          for (int j = 0; j < 16; ++j)
            if (s + 8 * j < ns) { gs.x += v[j].x; gs.y += v[j].y; gs.z += v[j].z; gs.w += v[j].w; }
        }
      } else
      for (int s = sg; s < ns; s += 32) {
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int sj = s + 8 * j; v[j] = *reinterpret_cast<const float4*>(sp + (int64_t)(sj < ns ? sj : ns - 1) * nw); }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (s + 8 * j < ns) { gs.x += v[j].x; gs.y += v[j].y; gs.z += v[j].z; gs.w += v[j].w; }
      }
      part[sg][c4] = gs;
      __syncthreads();
      if (sg == 0) {
#pragma unroll
        for (int k = 1; k < 8; ++k) { const float4 v = part[k][c4]; gs.x += v.x; gs.y += v.y; gs.z += v.z; gs.w += v.w; }  // fixed order
        *reinterpret_cast<float4*>(u.g + e) = gs;
      }
    }
    if (applies) opt_apply4<FUSED>(u.theta, u.state, u.state2, e, gs, u, pre);
    if (FUSED && bid < NW1 / 128) {                 // this block held 128 floats of W1: its write-through stores are out -> count it in
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) __hip_atomic_fetch_add(u.w1_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  const int fc5_blocks = u.A * FC5_BLOCKS_PER_ACTION;
  if (bid < CONV_BLOCKS + fc5_blocks) {
    // fc5 wgrad (delta . a4^T, A x 512) + its update: the batch plays the role of the slabs — 8 sample groups
    // per workgroup, fixed-order LDS combine (deterministic)
    const int fb = bid - CONV_BLOCKS, c4 = t & 31, sg = t >> 5;
    const int act = fb / FC5_BLOCKS_PER_ACTION, j0 = ((fb - act * FC5_BLOCKS_PER_ACTION) * 32 + c4) * 4;
    const int64_t e = OFF5 + (int64_t)act * NFC + j0;
    float4 gs = make_float4(0.f, 0.f, 0.f, 0.f);
    if (u.mode == 2) {
      if (sg == 0) gs = *reinterpret_cast<const float4*>(u.g + e);
    } else {
      int n = sg;
      for (; n + 24 < u.B; n += 32) {
        const float d0 = u.dq[(int64_t)n * u.A + act], d1 = u.dq[(int64_t)(n + 8) * u.A + act];
        const float d2 = u.dq[(int64_t)(n + 16) * u.A + act], d3 = u.dq[(int64_t)(n + 24) * u.A + act];
        const float4 v0 = *reinterpret_cast<const float4*>(u.a4 + (int64_t)n * NFC + j0);
        const float4 v1 = *reinterpret_cast<const float4*>(u.a4 + (int64_t)(n + 8) * NFC + j0);
        const float4 v2 = *reinterpret_cast<const float4*>(u.a4 + (int64_t)(n + 16) * NFC + j0);
        const float4 v3 = *reinterpret_cast<const float4*>(u.a4 + (int64_t)(n + 24) * NFC + j0);
        gs.x += d0 * v0.x; gs.y += d0 * v0.y; gs.z += d0 * v0.z; gs.w += d0 * v0.w;
        gs.x += d1 * v1.x; gs.y += d1 * v1.y; gs.z += d1 * v1.z; gs.w += d1 * v1.w;
        gs.x += d2 * v2.x; gs.y += d2 * v2.y; gs.z += d2 * v2.z; gs.w += d2 * v2.w;
        gs.x += d3 * v3.x; gs.y += d3 * v3.y; gs.z += d3 * v3.z; gs.w += d3 * v3.w;
      }
      for (; n < u.B; n += 8) {
        const float d = u.dq[(int64_t)n * u.A + act];
        const float4 v = *reinterpret_cast<const float4*>(u.a4 + (int64_t)n * NFC + j0);
        gs.x += d * v.x; gs.y += d * v.y; gs.z += d * v.z; gs.w += d * v.w;
      }
      part[sg][c4] = gs;
      __syncthreads();
      if (sg == 0) {
#pragma unroll
        for (int k = 1; k < 8; ++k) { const float4 v = part[k][c4]; gs.x += v.x; gs.y += v.y; gs.z += v.z; gs.w += v.w; }
        *reinterpret_cast<float4*>(u.g + e) = gs;
      }
    }
    if (sg == 0 && u.mode != 1 && !skip_apply) opt_apply4<FUSED>(u.theta, u.state, u.state2, e, gs, u);
    return;
  }
  // fc4: g written by fc4_wgrad (or all-reduced), elementwise; skipped when fused into fc4_wgrad's epilogue
  const int first_dense = CONV_BLOCKS + fc5_blocks;
  const int nb = nblocks - first_dense;
  if (!u.skip_fc4) {
    for (int64_t i4 = CONV_F4 + (int64_t)(bid - first_dense) * 256 + t; i4 < OFF5 / 4; i4 += (int64_t)nb * 256) {
      const int64_t e = i4 * 4;
      const float4 gs = *reinterpret_cast<const float4*>(u.g + e);
      if (u.mode != 1 && !skip_apply) opt_apply4<FUSED>(u.theta, u.state, u.state2, e, gs, u);
    }
  }
  if (u.next.B > 0 && bid == first_dense) {             // next step's prep rides along (every reader of idx is done)
    // (UpdateArgs is the first kernel parameter of both launches that run this body: its offset in the argument segment is 0)
    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(UpdateArgs, next) + offsetof(PrepArgs, idx_in);
    for (int n = t; n < u.next.B; n += 256) {
      const int64_t i = u.next.idx_in_valid ? *reinterpret_cast<const int64_t*>(ka + 8 * (n & 31)) : u.next.idx_pinned[n];
      u.next.idx[n] = i;
      const MetaRec rec = u.next.meta[i];
      u.next.actions[n] = rec.action; u.next.rewards[n] = rec.reward; u.next.terminals[n] = rec.terminal;
    }
  }
  if (u.mode != 2 && bid == first_dense + (nb > 1 ? 1 : 0)) {             // get_cost: mean over the batch, :154
    // all threads fetch (one memory latency for any B), ONE thread adds in index order: same bits as a serial loop,
    // which at B = 256 was the longest chain of the whole launch (256 dependent L2 round trips = 15 us)
    for (int n = t; n < u.B; n += 256) cost_sh[n] = u.cost_terms[n];
    __syncthreads();
    if (t == 0) {
      float c = 0.0f;
      for (int n = 0; n < u.B; ++n) c += cost_sh[n];
      c = c / (float)u.B;
      u.cost_out[0] = c;
      u.cost_accum[0] += (double)c;
    }
  }
}


}  // namespace sdqn
