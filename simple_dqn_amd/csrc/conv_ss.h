// conv_ss.h — SAMPLE-STATIONARY convolution routine of the throughput regime (B >= 128, float32): conv2 / conv3 forward
// (deepqnetwork.py:85-87; the online and the target net of :119-130 in one launch).
//
// The block-tile engine (gemm_engine_bt.h) re-fetches every operand element once per 64 x 64 block: conv2 forward moved 166 MB
// through the CUs' L1s for 26 MB of activations and 0.13 MB of weights (im2col expansion x 4, the same 128 KB weight panel fetched
// by each of 648 blocks; profiles/r05_pmc_tcp_b256.txt) and a CU's L1 waited on outstanding L2 requests a third of the launch.
// Here an input element crosses the L2 -> CU path exactly ONCE:
//   * one workgroup per CU owns NS whole samples of one net (NS = 2 at B = 256): their input maps ([HI][WI][CI] fp32, contiguous
//     in memory) are copied ONCE into an LDS image (pixel pitch CI + 4 floats) as whole lines, and im2col happens at ds_read time:
//     lane (m, kq) of a 16-position tile reads the 4 consecutive channels c = 16 g + 4 kq .. + 3 of pixel (ST p + r, ST q + s) with
//     one ds_read_b128 — four v_mfma_f32_16x16x4_f32 steps (k-slot kq <-> channel 16 g + 4 kq + j in step j, for A and B alike);
//   * the weights stream through a ring of NR = 4 LDS buffers, one (r, s) chunk of CI k-rows at a time ([k][64 + 4]), in memory order;
//   * OUTPUT-stationary: wave w of the four matrix waves owns output maps 16 w .. 16 w + 15 of ALL of the workgroup's positions
//     (NT tiles of 16 positions: one f32x4 accumulator each); for the first KO chunks K is the outer loop — a chunk's B fragments are read once and
//     used by all NT tiles, and only the input rows of kernel row r = 0 and ONE weight chunk have to be in LDS before the first MFMA;
//     the last FC = 3 chunks run tile by tile (B fragments in registers), so that finished tiles leave while the others still compute;
//   * waves are SPECIALISED (conv1_bf16_rows2_kernel's structure): four staging waves do every global access — the image rows, the
//     weight chunks four chunks ahead of their use, and the output, collected per tile in LDS and stored as whole 256-byte rows with
//     write-through 16-byte stores — so no matrix wave ever waits on vmcnt; one barrier per chunk.
// Arithmetic: exact fp32 products, one accumulator per output element, k ascending in the fixed order (r, s, g, j, kq) —
// run-to-run deterministic; differs from the block-tile engine's sum in the last bits only (another order of the same fp32 sum).
#pragma once
#include <hip/hip_runtime.h>
#include "gemm_engine.h"      // (SDQN_STAMP / g_sdqn_dbg of the timing build)

namespace sdqn {
namespace ss {

#ifdef SDQN_TIMING
// (the buffer pointer is fetched ONCE per wave — SS_STAMP_INIT — a load of it inside the K loop costs a memory round trip per chunk)
#define SS_STAMP_INIT unsigned long long* const ss_dbg_ = g_sdqn_dbg
#define SS_STAMP_T(T, ph) do { if (ss_dbg_ && threadIdx.x == (T)) ss_dbg_[(size_t)blockIdx.x * 8 + (ph)] = clock64(); } while (0)
#else
#define SS_STAMP_INIT do {} while (0)
#define SS_STAMP_T(T, ph) do {} while (0)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NO = 64;            // output maps (conv2 and conv3 alike): four matrix waves x 16
constexpr int WPITCH = NO + 4;    // floats per k-row of a staged weight chunk: rows k and k + 4 are 16 banks apart (conflict-free B reads)
constexpr int NR = 4;             // weight-chunk ring
constexpr int FC = 3;             // chunks of the tile-by-tile final phase
constexpr int NSTG = 256;         // staging threads (waves 4..7)

template <int HI_, int WI_, int CI_, int R_, int S_, int ST_, int PO_, int QO_, int NS_>
struct Cfg {
  static constexpr int HI = HI_, WI = WI_, CI = CI_, R = R_, S = S_, ST = ST_, PO = PO_, QO = QO_, NS = NS_;
  static constexpr int PITCH = CI + 4;                  // floats per pixel of the LDS image
  static constexpr int IMG = HI * WI * PITCH;           // floats per sample image
  static constexpr int NPOS = PO * QO;
  static constexpr int NT = (NS * NPOS + 15) / 16;      // tiles of 16 output positions per workgroup
  static constexpr int NCH = R * S, KO = NCH - FC;      // weight chunks = kernel taps (r, s); the first KO with K as the outer loop
  static constexpr int GR = CI / 16;                    // 16-channel groups per chunk (4 MFMA steps each)
  static constexpr int WCH = CI * WPITCH;               // floats per staged weight chunk
  static constexpr int OUTT = 16 * WPITCH;              // floats per collected output tile
  static constexpr int LDS = NS * IMG + NR * WCH + 2 * OUTT;
  static_assert(LDS * 4 <= 160 * 1024, "LDS budget");
  static_assert(CI % 16 == 0 && (CI / 16) % 2 == 0 && KO >= S && S >= 3 && NCH > NR, "chunk schedule");
  // staging geometry: 16-byte pieces
  static constexpr int PPX = CI / 4;                    // pieces per pixel
  static constexpr int ROWP = WI * PPX;                 // pieces per image row
  static constexpr int WP = CI * NO / 4 / NSTG;         // pieces of a weight chunk per staging thread
  static_assert(WP * NSTG * 4 == CI * NO, "whole weight pieces per thread");
  // image rows: kernel row r = 0 needs rows ST p (p < PO) — loaded before the first MFMA; the other rows ("rest") arrive under the
  // first S - 2 chunks (they are first needed by chunk S = kernel row 1)
  static constexpr bool row_is_first(int h) { return h % ST == 0 && h / ST < PO; }
  static constexpr int n_rest() { int n = 0; for (int h = 0; h < HI; ++h) if (!row_is_first(h)) ++n; return n; }
  static constexpr int rest_row(int i) { int n = 0; for (int h = 0; h < HI; ++h) if (!row_is_first(h)) { if (n == i) return h; ++n; } return HI - 1; }
  static constexpr int NREST = n_rest();
  // a staging pass moves ONE image row: lane lid < ROWP its piece lid (uniform row base + a per-lane constant: no address arithmetic
  // per piece — the piece-linear map p = lid + 256 j cost ~50 VALU instructions per piece in divisions, or a serial walk, in front of
  // the first load)
  static_assert(ROWP <= NSTG, "one row per pass");
  static constexpr int P0 = NS * PO;                                                              // first-row passes
  static constexpr int RPARTS = S - 2;                                                            // intervals that carry rest rows (visible at barrier #(S - 1): the matrix waves read chunk S's first fragments one barrier early)
  static constexpr int PR = NREST > 0 ? (NS * NREST + RPARTS - 1) / RPARTS : 0;                  // rest-row passes per interval
};

struct Args {
  const float* in;          // [nz][B][HI][WI][CI]
  const float* w[2];        // per net: [(r, s, c)][64]
  float* out;               // [nz][B][NPOS][64], Rectlin applied
  int B, G;                 // G = workgroups per net = ceil(B / NS)
  int wt;                   // 1: write-through (sc1) output stores
  int dbg;                  // timing build only: ablation bits (tools/exp/ss_stamps.py); 0 in the product
};

template <class C>
__global__ void __launch_bounds__(512) conv_ss_kernel(const Args c) {
  __shared__ __attribute__((aligned(16))) float smem[C::LDS];
  float* const img = smem;
  float* const wr = smem + C::NS * C::IMG;
  float* const outl = wr + NR * C::WCH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  SS_STAMP_INIT;
  const int z = (int)blockIdx.x / c.G, g = (int)blockIdx.x - z * c.G;
  const int n0 = g * C::NS;
  const int nvalid = c.B - n0 < C::NS ? c.B - n0 : C::NS;               // samples of this workgroup (the last one of an odd batch has one)

  if (wave >= 4) {
    // ================= staging waves: every global access of the workgroup =================
    const int lid = tid - 256;
    // the staging waves' barrier: their LDS stores complete (lgkmcnt), their global loads and stores stay IN FLIGHT — __syncthreads()'s
    // fence also drains vmcnt, i.e. every interval would wait a memory round trip for the weight chunk it has just requested
    // (measured: 3 640-4 470 cycles per conv2 chunk for 2 816 of matrix time, tools/exp/ss_stamps.py)
    auto stg_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    const float* const wsrc = c.w[z];
    const float* const isrc = c.in + ((int64_t)z * c.B + n0) * (C::HI * C::WI * C::CI);
    f32x4 wv[NR][C::WP];                                                 // weight chunks in flight (prologue: four; later one)
    f32x4 r0[C::P0];                                                     // first rows
    f32x4 rr[C::PR > 0 ? C::PR : 1];                                     // one interval's share of the rest rows
    auto w_issue = [&](int ch, f32x4* q) {
#pragma unroll
      for (int j = 0; j < C::WP; ++j) q[j] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)ch * (C::CI * NO) + 4 * (lid + NSTG * j));
    };
    auto w_commit = [&](int ch, const f32x4* q) {
      float* dst = wr + (ch % NR) * C::WCH;
#pragma unroll
      for (int j = 0; j < C::WP; ++j) { const int p = lid + NSTG * j, k = p >> 4, col = p & 15; *reinterpret_cast<f32x4*>(dst + k * WPITCH + 4 * col) = q[j]; }
    };
    // image rows, one per pass: lane lid < ROWP moves 16-byte piece lid of the row
    const bool rlane = lid < C::ROWP;
    const int rcp = rlane ? lid : 0;
    const float* const rsrc = isrc + 4 * rcp;
    float* const rdst = img + (rcp / C::PPX) * C::PITCH + 4 * (rcp % C::PPX);
    auto row_issue = [&](int s_, int row) {
      const int se = s_ < nvalid ? s_ : nvalid - 1;                      // (an odd batch's missing sample: a duplicate nobody stores)
      return *reinterpret_cast<const f32x4*>(rsrc + (se * C::HI + row) * (C::WI * C::CI));
    };
    auto row_commit = [&](int s_, int row, const f32x4& v) { if (rlane) *reinterpret_cast<f32x4*>(rdst + s_ * C::IMG + row * (C::WI * C::PITCH)) = v; };
    // ---- prologue: chunk 0 and the first rows are what the first MFMA waits for; chunks 1..3 fly behind them ----
    w_issue(0, wv[0]);
#pragma unroll
    for (int j = 0; j < C::P0; ++j) r0[j] = row_issue(j / C::PO, C::ST * (j % C::PO));
#pragma unroll
    for (int d = 1; d < NR; ++d) w_issue(d, wv[d]);
    w_commit(0, wv[0]);
#pragma unroll
    for (int j = 0; j < C::P0; ++j) row_commit(j / C::PO, C::ST * (j % C::PO), r0[j]);
    w_commit(1, wv[1]);                                                  // (the matrix waves prefetch chunk 1's first fragments before barrier #1)
    SS_STAMP_T(256, 6);
    stg_barrier();                                                     // barrier #0
    // ---- K-outer intervals: commit what the previous interval issued, issue chunk i + 4 and a share of the rest rows ----
#pragma unroll
    for (int i = 0; i < C::KO; ++i) {
#ifdef SDQN_TIMING
      if ((c.dbg & 1) && i > 0) { stg_barrier(); continue; }
#endif
      if (i == 0) {
#pragma unroll
        for (int d = 2; d < NR; ++d) w_commit(d, wv[d]);
      } else {
        w_commit(i + 3, wv[0]);
        if constexpr (C::PR > 0) {
          if (i - 1 < C::RPARTS) {
#pragma unroll
            for (int j = 0; j < C::PR; ++j) { const int q = (i - 1) * C::PR + j; if (q < C::NS * C::NREST) row_commit(q / C::NREST, C::rest_row(q % C::NREST), rr[j]); }
          }
        }
      }
#ifdef SDQN_TIMING
      if (c.dbg & 2) continue;
#endif
      if (i + 4 < C::NCH) w_issue(i + 4, wv[0]);
      if constexpr (C::PR > 0) {
        if (i < C::RPARTS) {
#pragma unroll
          for (int j = 0; j < C::PR; ++j) { const int q = i * C::PR + j; if (q < C::NS * C::NREST) rr[j] = row_issue(q / C::NREST, C::rest_row(q % C::NREST)); }
        }
      }
      stg_barrier();                                                   // barrier #(i + 1)
    }
    // ---- final phase: finished tiles, two per round, leave as whole 256-byte rows ----
    float* const obase = c.out + ((int64_t)z * c.B + n0) * (C::NPOS * NO);
    const int nrows = nvalid * C::NPOS;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)obase, 0, nrows * NO * 4, 0x00020000);
    const int orow = lid >> 4, ocol = 4 * (lid & 15);
#pragma unroll 1
    for (int t0 = 0; t0 < C::NT; t0 += 2) {
      stg_barrier();                                                   // B: the matrix waves may overwrite the collection buffers
      stg_barrier();                                                   // A: both tiles are collected
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int P = 16 * (t0 + u) + orow;
        const f32x4 v = *reinterpret_cast<const f32x4*>(outl + u * C::OUTT + orow * WPITCH + ocol);
        u32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(v[e]);
        const int off = (P < nrows ? P * NO + ocol : nrows * NO) * 4;     // (past the end: dropped by the buffer's range check)
        if (c.wt) __builtin_amdgcn_raw_buffer_store_b128(w, rs, off, 0, 16); else __builtin_amdgcn_raw_buffer_store_b128(w, rs, off, 0, 0);
      }
    }
    SS_STAMP_T(256, 7);
    return;
  }

  // ================= matrix waves: wave w owns output maps 16 w .. 16 w + 15 of every position =================
  const int m = lane & 15, kq = lane >> 4;
  int abase[C::NT];                                                      // float index of this lane's patch origin per tile (+ its 4 channels)
#pragma unroll
  for (int t = 0; t < C::NT; ++t) {
    int P = 16 * t + m; if (P > C::NS * C::NPOS - 1) P = C::NS * C::NPOS - 1;
    const int s = P / C::NPOS, pos = P - s * C::NPOS, p = pos / C::QO, q = pos - p * C::QO;
    abase[t] = s * C::IMG + ((C::ST * p) * C::WI + C::ST * q) * C::PITCH + 4 * kq;
  }
  const int wlane = (4 * kq) * WPITCH + 16 * wave + m;                   // B fragment: k-row 4 kq (+ 16 g + j), map 16 w + m
  f32x4 acc[C::NT];
#pragma unroll
  for (int t = 0; t < C::NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  SS_STAMP_T(0, 0);
  __syncthreads();                                                       // barrier #0
  SS_STAMP_T(0, 1);
  // K-outer phase, software-pipelined by hand: the fragments of group (i, gq + 1) — or of (i + 1, 0) — are read from LDS while the MFMAs
  // of group (i, gq) run (left alone hipcc reads a whole chunk's fragments at the top of the iteration and waits for them).  Reading
  // chunk i + 1's first group BEFORE barrier #(i + 1) is safe: weight chunk i + 1 and the image rows it needs are visible since
  // barrier #i at the latest (the staging waves' schedule above)
  f32x4 av[2][C::NT];
  float bv[2][4];
  auto load_group = [&](f32x4* a_, float* b_, int i, int gq) {
#if defined(SS_ABL) && SS_ABL == 2
    return;
#endif
#if defined(SS_ABL) && (SS_ABL == 3 || SS_ABL == 1)
    i = 0;
#endif
    const int r = i / C::S, s = i - r * C::S;
    const float* const ai = img + (r * C::WI + s) * C::PITCH + 16 * gq;
    const float* const wi = wr + (i % NR) * C::WCH + wlane + (16 * gq) * WPITCH;
#pragma unroll
    for (int j = 0; j < 4; ++j) b_[j] = wi[j * WPITCH];
#pragma unroll
    for (int t = 0; t < C::NT; ++t) a_[t] = *reinterpret_cast<const f32x4*>(ai + abase[t]);
#if defined(SS_ABL) && SS_ABL == 1
#pragma unroll
    for (int t = 0; t < C::NT; ++t) a_[t] = *reinterpret_cast<const f32x4*>(img + (16 * t + m) * 36 + 4 * kq + 16 * gq);
#endif
  };
  load_group(av[0], bv[0], 0, 0);
#pragma unroll 1
  for (int i = 0; i < C::KO; ++i) {
#pragma unroll
    for (int gq = 0; gq < C::GR; ++gq) {
      if (gq + 1 < C::GR) load_group(av[(gq + 1) & 1], bv[(gq + 1) & 1], i, gq + 1);
      else load_group(av[(gq + 1) & 1], bv[(gq + 1) & 1], i + 1 < C::KO ? i + 1 : i, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < C::NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[gq & 1][t][j], bv[gq & 1][j], acc[t], 0, 0, 0);
      // the next group's reads, ONE between every three MFMAs: issued as a burst at the top of the group (all four matrix waves at once,
      // right behind the barrier) they fill the LDS queue and the in-order wave cannot issue its next MFMA for 300-400 cycles per group
      // (tools/exp/ss_stamps.py: 3 750-4 500 cycles per conv2 chunk against 2 816 of matrix time)
#pragma unroll
      for (int q = 0; q < C::NT + 2; ++q) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); }
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * C::NT - 3 * (C::NT + 2), 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // barrier #(i + 1), WITHOUT __syncthreads()'s fence: that would drain lgkmcnt and expose the reads just issued for chunk i + 1.  This
    // wave wrote nothing; the operand makes the wait for chunk i's last B fragments (and, LDS reads returning in order, for every read of
    // ring slot i % NR) precede the barrier, behind which the staging waves refill that slot; the reads still in flight touch other slots
#ifdef SDQN_TIMING
    if (!(c.dbg & 2))
#endif
    asm volatile("s_barrier" :: "v"(bv[(C::GR - 1) & 1][3]) : "memory");
#ifdef SDQN_TIMING
    if (i == 0) SS_STAMP_T(0, 2);
    if (i == 4) SS_STAMP_T(0, 3);
#endif
  }
  SS_STAMP_T(0, 4);
  // ---- final phase: the last FC chunks tile by tile, B fragments in registers ----
  float bf[FC][C::GR][4];
#pragma unroll
  for (int f = 0; f < FC; ++f)
#pragma unroll
    for (int gq = 0; gq < C::GR; ++gq)
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[f][gq][j] = wr[((C::KO + f) % NR) * C::WCH + wlane + (16 * gq + j) * WPITCH];
#pragma unroll
  for (int t0 = 0; t0 < C::NT; t0 += 2) {
    constexpr int NT = C::NT;
    const bool two = t0 + 1 < NT;
#pragma unroll
    for (int f = 0; f < FC; ++f) {
      const int r = (C::KO + f) / C::S, s = (C::KO + f) - r * C::S;
      const float* const ai = img + (r * C::WI + s) * C::PITCH;
#pragma unroll
      for (int gq = 0; gq < C::GR; ++gq) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(ai + abase[t0] + 16 * gq);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(ai + abase[two ? t0 + 1 : t0] + 16 * gq);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[t0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], bf[f][gq][j], acc[t0], 0, 0, 0);
          if (two) acc[t0 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], bf[f][gq][j], acc[two ? t0 + 1 : t0], 0, 0, 0);
        }
      }
    }
    __syncthreads();                                                     // B: the staging waves have read the previous pair
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      outl[(4 * kq + e) * WPITCH + 16 * wave + m] = fmaxf(acc[t0][e], 0.0f);
      if (two) outl[C::OUTT + (4 * kq + e) * WPITCH + 16 * wave + m] = fmaxf(acc[two ? t0 + 1 : t0][e], 0.0f);
    }
    __syncthreads();                                                     // A
  }
  SS_STAMP_T(0, 5);
}

template <class C>
inline hipError_t launch(const Args& c, int nz, hipStream_t s) {
  SDQN_LAUNCH((conv_ss_kernel<C>), dim3(nz * c.G), dim3(512), 0, s, c);
  return hipGetLastError();
}

}  // namespace ss
}  // namespace sdqn
