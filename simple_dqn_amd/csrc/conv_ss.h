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
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NO = 64;            // output maps (conv2 and conv3 alike): four matrix waves x 16
constexpr int WPITCH = NO + 4;    // floats per k-row of a staged weight chunk: rows k and k + 4 are 16 banks apart (conflict-free B reads)
constexpr int NR = 4;             // weight-chunk ring
#ifndef SS_FC
#define SS_FC 3
#endif
constexpr int FC = SS_FC;         // chunks of the tile-by-tile final phase (-DSS_FC=2: correct, the chained launch 40.3 -> 42.1 us, 5 351-5 366 -> 5 304-5 321 steps/s)
constexpr int NSTG = 256;         // staging threads (waves 4..7)

template <int HI_, int WI_, int CI_, int R_, int S_, int ST_, int PO_, int QO_, int NS_, int PPAD_, int RPAD_, int SPAD_>
struct Cfg {
  static constexpr int HI = HI_, WI = WI_, CI = CI_, R = R_, S = S_, ST = ST_, PO = PO_, QO = QO_, NS = NS_;
  // LDS image of a sample: pixel pitch CI + PPAD floats, row pitch WI PITCH + RPAD, sample pitch HI RPITCH + SPAD.  The paddings are
  // searched (tools/exp/ss_bank_search.py) so that the 16 lanes of every ds_read_b128 group — tile rows {0-3, 12-15} at k-slot kq and
  // rows {4-11} at kq + 1 — hit 16 different 16-byte bank groups in EVERY tile, output-row wraps and the sample seam included: with the
  // plain CI + 4 pitch a third of the reads were 2- or 3-way conflicts and a conv2 chunk took 3 340 cycles instead of 3 070
  static constexpr int PITCH = CI + PPAD_;              // floats per pixel of the LDS image
  static constexpr int RPITCH = WI * PITCH + RPAD_;     // floats per image row
  static constexpr int IMG = HI * RPITCH + SPAD_;       // floats per sample image
  static_assert(PPAD_ % 4 == 0 && RPAD_ % 4 == 0 && SPAD_ % 4 == 0, "16-byte aligned pieces");
  static constexpr int NPOS = PO * QO;
  // tiles of 16 output positions per workgroup.  81 = 5 x 16 + 1 and 49 = 3 x 16 + 1: a padded tile for the 1-2 positions left over
  // would be a tenth (conv2) / a seventh (conv3) of the matrix work.  They run on v_mfma_f32_4x4x1_16b_f32 instead — 16 blocks of 4 x 4,
  // K = 1, 8 cycles: block b = output maps 4 b .. 4 b + 3 (lane l: W[k][map l]), the 4 rows = up to 4 positions (the same in every
  // block), so one instruction is one k for LV positions x all 64 maps; every matrix wave takes a quarter of each chunk's channels
  // (8 such MFMAs per conv2 chunk = 64 cycles against 256 for the padded tile) and the four partial sums meet in LDS at the end.
  // (First form: the staging waves' vector ALU — fp32 MFMA runs on the same multipliers, and every vector instruction of a staging wave
  // cost the matrix wave on its SIMD ~10 cycles: 160-390 cycles per chunk, whatever the instruction mix.)
  static constexpr int LV = (NS * NPOS) % 16 <= 4 ? (NS * NPOS) % 16 : 0;
  static constexpr int NT = LV ? (NS * NPOS) / 16 : (NS * NPOS + 15) / 16;
  static constexpr int NCH = R * S, KO = NCH - FC;      // weight chunks = kernel taps (r, s); the first KO with K as the outer loop
  static constexpr int GR = CI / 16;                    // 16-channel groups per chunk (4 MFMA steps each)
  static constexpr int WCH = CI * WPITCH;               // floats per staged weight chunk
  static constexpr int OUTT = 16 * WPITCH;              // floats per collected output tile
  static constexpr int LDS = NS * IMG + NR * WCH + 2 * OUTT;
  static_assert(LDS * 4 <= 160 * 1024, "LDS budget");
  static_assert(4 * LV * NO <= 2 * OUTT && CI % 16 == 0, "the left-over positions' four partial sums fit a collection buffer");
  static_assert(2 * OUTT <= WCH, "the second collection buffer is the ring slot the final phase leaves free");
  static_assert(FC * (CI / 16) >= 4 && FC * (CI / 16) > FC, "a pair's collection rides in the next pair's first four steps, the left-over positions' final chunks in the last pair's first FC + 1");
  static_assert(CI % 16 == 0 && (CI / 16) % 2 == 0 && KO >= S && S >= 3 && NCH > NR && (KO % S == 0 || KO % S + FC == S), "chunk schedule (the final phase stays inside ONE kernel row or starts one)");
  // staging geometry: 16-byte pieces
  static constexpr int PPX = CI / 4;                    // pieces per pixel
  static constexpr int ROWP = WI * PPX;                 // pieces per image row
  static constexpr int WP = CI * NO / 4 / NSTG;         // pieces of a weight chunk per staging thread
  static_assert(WP * NSTG * 4 == CI * NO, "whole weight pieces per thread");
  // ORDER of the kernel rows in the K loop: the rows congruent to 0 modulo the stride first (conv2, stride 2: r = 0, 2, 1, 3).  Kernel
  // row r needs image rows ST p + r: r = 0 and r = 2 share all but one of theirs, so the other parity — half of the input — is first
  // needed by chunk 8 of 16 instead of chunk 4, and its loads spread over five intervals instead of two (every CU asks for its rows at
  // the same time: 14 MB in two intervals were HBM-bandwidth-bound and the staging waves reached the barriers late).  The weight chunks
  // stream in the same order (logical chunk i = physical tap (rord(i / S), i % S)); the sum order is the logical one
  static constexpr int rord(int ri) { int n = 0; for (int res = 0; res < ST; ++res) for (int r = res; r < R; r += ST) { if (n == ri) return r; ++n; } return R - 1; }
  static constexpr int need_ri(int h) { for (int ri = 0; ri < R; ++ri) { const int d = h - rord(ri); if (d >= 0 && d % ST == 0 && d / ST < PO) return ri; } return R; }
  // image rows: the first kernel row's (need_ri == 0) are loaded before the first MFMA; the others ("rest", in the order they are
  // needed) PR (sample, row) entries per interval, committed LAG intervals after their loads were issued
  static constexpr int n_rest() { int n = 0; for (int h = 0; h < HI; ++h) if (need_ri(h) >= 1 && need_ri(h) < R) ++n; return n; }
  static constexpr int rest_row(int i) { int n = 0; for (int ri = 1; ri < R; ++ri) for (int h = 0; h < HI; ++h) if (need_ri(h) == ri) { if (n == i) return h; ++n; } return HI - 1; }
  static constexpr int first_row(int i) { int n = 0; for (int h = 0; h < HI; ++h) if (need_ri(h) == 0) { if (n == i) return h; ++n; } return HI - 1; }
  static constexpr int NREST = n_rest();
  static constexpr int NQ = NS * NREST;                 // rest entries: entry q = (row rest_row(q / NS), sample q % NS)
  // entry q, issued in interval q / pr and committed lag intervals later, is visible at barrier #(q / pr + lag + 1); the matrix waves
  // read the first fragments of chunk need_ri S one barrier early, i.e. behind barrier #(need_ri S - 1)
  static constexpr bool sched_ok(int pr, int lag) { for (int q = 0; q < NQ; ++q) if (q / pr + lag + 1 > need_ri(rest_row(q / NS)) * S - 1) return false; return true; }
  static constexpr int pick_pr(int lag) { for (int pr = 1; pr <= NQ; ++pr) if (sched_ok(pr, lag)) return pr; return 0; }
  static constexpr int LAG = NQ == 0 ? 1 : (pick_pr(2) ? 2 : 1);
  static constexpr int PR = NQ == 0 ? 0 : pick_pr(LAG);                                           // rest entries per interval
  static constexpr int NI = PR ? (NQ + PR - 1) / PR : 0;                                          // intervals that issue rest rows
  static_assert(NQ == 0 || (PR > 0 && NI - 1 + LAG < KO), "the rest rows arrive in time");
  // a staging pass moves ONE image row: lane lid < ROWP its piece lid (uniform row base + a per-lane constant: no address arithmetic
  // per piece — the piece-linear map p = lid + 256 j cost ~50 VALU instructions per piece in divisions, or a serial walk, in front of
  // the first load)
  static_assert(ROWP <= NSTG, "one row per pass");
  static constexpr int P0 = NS * PO;                                                              // first-row passes
};

struct Args {
  const float* in;          // [nz][B][HI][WI][CI]
  const float* w[2];        // per net: [(r, s, c)][64]
  float* out;               // [nz][B][NPOS][64], Rectlin applied
  int B, G;                 // G = workgroups per net = ceil(B / NS)
  int wt;                   // 1: write-through (sc1) output stores
  int dbg;                  // timing build only: ablation bits (tools/exp/ss_stamps.py); 0 in the product
};

// what a staging thread carries from layer 1 to layer 2 of a chained launch (conv_ss_chain_kernel), in registers: the Rectlin'ed output
// pieces it stored (its 16 bytes of two tiles per collection round), its left-over value, and layer 2's first NR weight chunks, whose
// loads it issued when layer 1's final phase began
template <int NPIECE, int NWREG>
struct Keep { f32x4 v[NPIECE > 0 ? NPIECE : 1]; float left; f32x4 w[NWREG > 0 ? NWREG : 1]; };
struct NoNext { static constexpr int WP = 0, CI = 0, S = 1; static constexpr int rord(int) { return 0; } };

// one layer for this workgroup's samples (every wave of the workgroup calls it; waves 0-3 = matrix, 4-7 = staging).
// MODE bit 0: layer 1 of a chain (keep the output in `keep`, prefetch CN's first weight chunks from cn_w); bit 1: layer 2 (the image and
// the first weight chunks come out of `keep`: no image row is loaded)
template <class C, int MODE = 0, class CN = NoNext, class KeepT = Keep<0, 0> >
__device__ __forceinline__ void conv_ss_body(const Args& c, float* const smem, KeepT& keep, const float* const* cn_w = nullptr) {
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
    // Every vector instruction of a staging wave costs the matrix wave on its SIMD ~8-10 cycles of its MFMA stream (fp32 MFMA runs on
    // the vector ALU's multipliers; measured with the left-over positions' FMAs switched on and off) — so the staging code is written to
    // issue as few as possible: buffer loads / stores whose per-piece part of the address is a SCALAR offset (one per-lane offset
    // register for all pieces), LDS addresses = per-lane constant + immediate, the Rectlin and the left-over FMAs on packed-fp32
    // instructions over NATURAL register pairs (no packing moves)
    const float* const isrc = c.in + ((int64_t)z * c.B + n0) * (C::HI * C::WI * C::CI);
    // (global loads in their scalar-base form — global_load_dwordx4 v, v_lane_offset, s[base] — the base moved per piece on the scalar
    //  unit; buffer loads with a scalar offset would do as well, but hipcc waits for them with vmcnt(0) whatever their order)
    const char* const wsrc = reinterpret_cast<const char*>(c.w[z]);
    auto ldb = [&](const char* base, unsigned voff, int soff) { return *reinterpret_cast<const f32x4*>(base + soff + voff); };
    f32x4 wv[NR][C::WP];                                                 // weight chunks in flight (prologue: four; later one)
    f32x4 r0[C::P0];                                                     // first rows
    f32x4 rr[C::LAG][C::PR > 0 ? C::PR : 1];                             // the rest rows in flight: LAG intervals' worth
    const unsigned wvo = 16 * lid;                                            // weight piece lid + 256 j: byte offset 16 lid (+ 4096 j, + the chunk: scalar)
    float* const wdst = wr + (lid >> 4) * WPITCH + 4 * (lid & 15);       // ... -> LDS k-row (lid >> 4) + 16 j, floats 4 (lid & 15) ..
    auto w_issue = [&](int ch, f32x4* q) {
      const int pc = C::rord(ch / C::S) * C::S + ch % C::S;              // logical chunk ch = physical tap (rord(ch / S), ch % S)
#pragma unroll
      for (int j = 0; j < C::WP; ++j) q[j] = ldb(wsrc, wvo, pc * (C::CI * NO * 4) + 4096 * j);
    };
    auto w_commit = [&](int ch, const f32x4* q) {
#pragma unroll
      for (int j = 0; j < C::WP; ++j) *reinterpret_cast<f32x4*>(wdst + (ch % NR) * C::WCH + 16 * j * WPITCH) = q[j];
    };
    // image rows, one per pass: lane lid < ROWP moves 16-byte piece lid of the row
    const bool rlane = lid < C::ROWP;
    const int rcp = rlane ? lid : 0;
    const unsigned rvo = 16 * rcp;
    int rdst[C::NS];                                                     // float index in img (one base per sample: the second sample's rows lie beyond the 16-bit offset field;
#pragma unroll                                                           //  an INDEX, opaque to hipcc: an asm on the pointer itself turns every access through it into a flat one)
    for (int q = 0; q < C::NS; ++q) { rdst[q] = q * C::IMG + (rcp / C::PPX) * C::PITCH + 4 * (rcp % C::PPX); asm volatile("" : "+v"(rdst[q])); }
    auto row_issue = [&](int s_, int row) {
      const int se = s_ < nvalid ? s_ : nvalid - 1;                      // (an odd batch's missing sample: a duplicate nobody stores; wave-uniform)
      return ldb(reinterpret_cast<const char*>(isrc), rvo, (se * C::HI + row) * (C::WI * C::CI * 4));
    };
#if defined(SS_ABL) && SS_ABL == 7
    auto row_commit = [&](int s_, int row, const f32x4& v) { *reinterpret_cast<f32x4*>(img + rdst[s_] + row * C::RPITCH) = v; };      // (timing experiment: no lane mask)
#else
    auto row_commit = [&](int s_, int row, const f32x4& v) { if (rlane) *reinterpret_cast<f32x4*>(img + rdst[s_] + row * C::RPITCH) = v; };
#endif
    // ---- prologue: chunk 0 and the first rows are what the first MFMA waits for; chunks 1..3 fly behind them ----
    if constexpr (MODE & 2) {
      // layer 2 of a chain: the whole image out of the previous layer's registers (piece j of this thread = its 16 bytes of output row
      // P = 16 j + (lid >> 4) = pixel P % (HI WI) of sample P / (HI WI)); the first NR weight chunks were requested long ago
      constexpr int NPX = C::HI * C::WI;
#pragma unroll
      for (int j = 0; j < (int)(sizeof(keep.v) / sizeof(keep.v[0])); ++j) {
        const int P = 16 * j + (lid >> 4);
        if (P < C::NS * NPX) {
          const int sp = P / NPX, px = P - sp * NPX, y = px / C::WI, x = px - y * C::WI;
          *reinterpret_cast<f32x4*>(img + sp * C::IMG + y * C::RPITCH + x * C::PITCH + 4 * (lid & 15)) = keep.v[j];
        }
      }
      {                                                                  // the previous layer's left-over positions: one value per lane (position, map)
        constexpr int NL = (C::NS * NPX) % 16;
        if (lid < NL * NO) {
          const int P = C::NS * NPX - NL + (lid >> 6), sp = P / NPX, px = P - sp * NPX, y = px / C::WI, x = px - y * C::WI;
          img[sp * C::IMG + y * C::RPITCH + x * C::PITCH + (lid & 63)] = keep.left;
        }
      }
#pragma unroll
      for (int d = 0; d < NR; ++d)
#pragma unroll
        for (int j = 0; j < C::WP; ++j) wv[d][j] = keep.w[d * C::WP + j];
      w_commit(0, wv[0]);
      w_commit(1, wv[1]);
    } else {
    w_issue(0, wv[0]);
#pragma unroll
    for (int j = 0; j < C::P0; ++j) r0[j] = row_issue(j / C::PO, C::first_row(j % C::PO));
#pragma unroll
    for (int d = 1; d < NR; ++d) w_issue(d, wv[d]);
    w_commit(0, wv[0]);
#pragma unroll
    for (int j = 0; j < C::P0; ++j) row_commit(j / C::PO, C::first_row(j % C::PO), r0[j]);
    w_commit(1, wv[1]);                                                  // (the matrix waves prefetch chunk 1's first fragments before barrier #1)
    }
#if !defined(SS_ABL) || SS_ABL != 9
    SS_STAMP_T(256, 6);
#endif
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
      } else w_commit(i + 3, wv[0]);
      if constexpr (C::PR > 0 && !(MODE & 2)) {
        if (i >= C::LAG && i - C::LAG < C::NI) {
#pragma unroll
          for (int j = 0; j < C::PR; ++j) { const int q = (i - C::LAG) * C::PR + j; if (q < C::NQ) row_commit(q % C::NS, C::rest_row(q / C::NS), rr[i % C::LAG][j]); }
        }
      }
#ifdef SDQN_TIMING
      if (c.dbg & 2) continue;
#endif
      if (i + 4 < C::NCH) w_issue(i + 4, wv[0]);
      if constexpr (C::PR > 0 && !(MODE & 2)) {
        if (i < C::NI) {
#pragma unroll
          for (int j = 0; j < C::PR; ++j) { const int q = i * C::PR + j; if (q < C::NQ) rr[i % C::LAG][j] = row_issue(q % C::NS, C::rest_row(q / C::NS)); }
        }
      }
      stg_barrier();                                                   // barrier #(i + 1)
    }
    // ---- final phase: finished tiles, two per round, leave as whole 256-byte rows ----
    float* const obase = c.out + ((int64_t)z * c.B + n0) * (C::NPOS * NO);
    const int nrows = nvalid * C::NPOS;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)obase, 0, nrows * NO * 4, 0x00020000);
    const int orow = lid >> 4, ocol = 4 * (lid & 15);
    // the collection buffers alternate: round p (tiles 2 p, 2 p + 1) in outl (even p) / in the ring slot of chunk KO - 1, free from
    // barrier #KO on (odd p).  ONE barrier per round: A (p) publishes round p, and — this wave reaches it only after it has read round
    // p - 1 — tells the matrix waves that round p - 1's buffer may be overwritten (by round p + 1).  Between A (p) and A (p + 1) lie the
    // MFMAs of a whole pair (1 500+ cycles): room for the ~30 vector instructions of a round, of which the staging wave gets about one
    // into every gap of its SIMD's MFMA stream
    const int ovo = (orow * NO + ocol) * 4;                              // byte offset of this lane's piece inside a tile's 16 rows (+ the tile: scalar)
    const int olane = orow * WPITCH + ocol;                              // (integer indexes: a run-time choice between two POINTERS makes hipcc read through a flat one)
    const int obuf1 = (int)(((C::KO + FC) % NR) * C::WCH) - NR * C::WCH;   // the second buffer relative to outl (it lies in front of it)
    constexpr int NP = (C::NT + 1) / 2;
    if constexpr (MODE & 1) {                                            // layer 1 of a chain: the next layer's first weight chunks, requested now
#pragma unroll
      for (int d = 0; d < NR; ++d) {
        const int pc = CN::rord(d / CN::S) * CN::S + d % CN::S;
#pragma unroll
        for (int j = 0; j < CN::WP; ++j) keep.w[d * CN::WP + j] = ldb(reinterpret_cast<const char*>(cn_w[z]), wvo, pc * (CN::CI * NO * 4) + 4096 * j);
      }
    }
    auto round = [&](int pi) {
      stg_barrier();                                                   // A of round pi: both tiles are collected
      const float* const ob = outl + olane + ((pi & 1) ? obuf1 : 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * pi + u;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ob + u * C::OUTT);
        const f32x2 z2 = {0.0f, 0.0f};
        const f32x2 lo = __builtin_elementwise_max(__builtin_shufflevector(v, v, 0, 1), z2), hi = __builtin_elementwise_max(__builtin_shufflevector(v, v, 2, 3), z2);   // Rectlin (deepqnetwork.py:85-87)
        if constexpr (MODE & 1) keep.v[2 * pi + u] = f32x4{lo[0], lo[1], hi[0], hi[1]};
        const u32x4 w = {__float_as_uint(lo[0]), __float_as_uint(lo[1]), __float_as_uint(hi[0]), __float_as_uint(hi[1])};
        const int soff = t < C::NT ? t * (16 * NO * 4) : nrows * NO * 4;   // (a tile past the last one, rows past the batch: dropped by the buffer's range check)
#ifdef SDQN_TIMING
        if (c.dbg & 4) continue;
#endif
        if (c.wt) __builtin_amdgcn_raw_buffer_store_b128(w, rs, ovo, soff, 16); else __builtin_amdgcn_raw_buffer_store_b128(w, rs, ovo, soff, 0);
      }
    };
    if constexpr (MODE & 1) {                                            // (unrolled: the kept pieces need compile-time register indexes)
#pragma unroll
      for (int pi = 0; pi < NP; ++pi) round(pi);
    } else {
#pragma unroll 1
      for (int pi = 0; pi < NP; ++pi) round(pi);
    }
    if constexpr (C::LV > 0) {
      // the left-over positions: the matrix waves' four partial sums (published with the last round, in chunk KO's ring slot — the other
      // collection buffer is still being read when they are written: [wave][LV][64]) added in wave order, Rectlin, one 256-byte row each
      if (lid < C::LV * NO) {
        const int l = lid >> 6, n = lid & 63, P = 16 * C::NT + l;
        const float* const q = wr + (C::KO % NR) * C::WCH + l * NO + n;
        const float v = ((q[0] + q[C::LV * NO]) + q[2 * C::LV * NO]) + q[3 * C::LV * NO];
        if constexpr (MODE & 1) keep.left = fmaxf(v, 0.0f);
        if (P < nrows) {
          float* const dst = obase + (size_t)P * NO + n;
          if (c.wt) __hip_atomic_store(dst, fmaxf(v, 0.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *dst = fmaxf(v, 0.0f);
        }
      }
    }
#if !defined(SS_ABL) || SS_ABL != 9
    SS_STAMP_T(256, 7);
#endif
    return;
  }

  // ================= matrix waves: wave w owns output maps 16 w .. 16 w + 15 of every position =================
  // MFMA operand roles: the WEIGHTS are the row operand (lane (m, kq): W[k-slot kq][map 16 w + m]), the patches the column operand (lane
  // (m, kq): position m of the tile, k-slot kq), so D[map][position]: a lane ends up with maps 16 w + 4 kq .. + 3 of position m — one
  // 16-byte LDS store per finished tile
  const int m = lane & 15, kq = lane >> 4;
  int ap[C::NT];                                                         // float index in img of this lane's patch origin per tile (+ its 4 channels), at the CURRENT kernel row
#pragma unroll
  for (int t = 0; t < C::NT; ++t) {
    int P = 16 * t + m; if (P > C::NS * C::NPOS - 1) P = C::NS * C::NPOS - 1;
    const int s = P / C::NPOS, pos = P - s * C::NPOS, p = pos / C::QO, q = pos - p * C::QO;
    ap[t] = s * C::IMG + (C::ST * p) * C::RPITCH + (C::ST * q) * C::PITCH + 4 * kq;
    // (one register per tile, unrelated as far as hipcc knows: it otherwise derives the origins of the second sample's tiles from the
    //  first's with a v_add of a 32-bit literal in front of every read whose offset no longer fits the 16-bit field — between the MFMAs)
    asm volatile("" : "+v"(ap[t]));
  }
  const float* const wl = wr + (4 * kq) * WPITCH + 16 * wave + m;        // B fragment: k-row 4 kq (+ 16 g + j), map 16 w + m
  // the left-over positions (Cfg::LV) on the 4 x 4 x 1 shape: this wave's channel quarter CQ w .. of every chunk; A operand = lane's
  // position (lane & 3, clamped), B operand = W[k][map lane]; D register i of lane l = position i, map l
  constexpr int CQ = C::CI / 4;
  int lap = 0;                                                           // float index in img of the lane's left-over position (+ this wave's channels), at the current kernel row
  if constexpr (C::LV > 0) {
    const int l = (lane & 3) < C::LV ? (lane & 3) : C::LV - 1;
    const int P = 16 * C::NT + l, sp = P / C::NPOS, pos = P - sp * C::NPOS, pp = pos / C::QO, qq = pos - pp * C::QO;
    lap = sp * C::IMG + (C::ST * pp) * C::RPITCH + (C::ST * qq) * C::PITCH + wave * CQ;
    asm volatile("" : "+v"(lap));
  }
  const float* const wll = wr + (wave * CQ) * WPITCH + lane;
  f32x4 lacc[4];                                                         // four chains (channel c of the quarter -> chain c & 3): back to back on ONE accumulator the 4 x 4 x 1 MFMAs
#pragma unroll                                                           // wait out their result latency (measured: 170 cycles per conv2 chunk for 8 of them)
  for (int q = 0; q < 4; ++q) lacc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 lav[CQ / 4];
  float lbv[CQ];
  auto left_load = [&](int s_, int slot) {
    if constexpr (C::LV > 0) {
#pragma unroll
      for (int c4 = 0; c4 < CQ / 4; ++c4) lav[c4] = *reinterpret_cast<const f32x4*>(img + lap + s_ * C::PITCH + 4 * c4);
#pragma unroll
      for (int c1 = 0; c1 < CQ; ++c1) lbv[c1] = wll[slot * C::WCH + c1 * WPITCH];
    }
  };
  auto left_mfma = [&]() {
    if constexpr (C::LV > 0) {
#pragma unroll
      for (int c1 = 0; c1 < CQ; ++c1) lacc[c1 & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(lav[c1 / 4][c1 % 4], lbv[c1], lacc[c1 & 3], 0, 0, 0);
    }
  };
  f32x4 acc[C::NT];
#pragma unroll
  for (int t = 0; t < C::NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  SS_STAMP_T(0, 0);
  __syncthreads();                                                       // barrier #0
  SS_STAMP_T(0, 1);
  // K-outer phase, software-pipelined by hand: the fragments of the NEXT group — (i, gq + 1), or (i + 1, 0) — are read from LDS while the
  // MFMAs of group (i, gq) run (left alone hipcc reads a whole chunk's fragments at the top of the iteration and waits for them).
  // Reading chunk i + 1's first group BEFORE barrier #(i + 1) is safe: weight chunk i + 1 and the image rows it needs are visible since
  // barrier #i at the latest (the staging waves' schedule above).  The loop runs over kernel ROWS with the S taps of a row unrolled:
  // every fragment address is then ap[t] + a compile-time offset (tap s, channel group, next row) — with a run-time chunk index the 11-22
  // v_add_u32 per chunk sat between the MFMAs and cost ~270 cycles per chunk (an issue slot between two MFMAs is not free)
  constexpr int RK = C::KO / C::S, TAIL = C::KO - RK * C::S;             // whole kernel rows of the K-outer phase + taps of a partial one
  f32x4 av[2][C::NT];
  float bv[2][4];
  // group (dr, s_, gq): kernel row (current + dr), tap s_, channels 16 gq ..; ring slot of its chunk: compile-time when S == NR
  auto load_group = [&](f32x4* a_, float* b_, int dr, int s_, int gq, int slot) {
#if defined(SS_ABL) && SS_ABL == 2
    return;
#endif
    const float* const wi = wl + slot * C::WCH + (16 * gq) * WPITCH;
#pragma unroll
    for (int j = 0; j < 4; ++j) b_[j] = wi[j * WPITCH];
#pragma unroll
    for (int t = 0; t < C::NT; ++t) a_[t] = *reinterpret_cast<const f32x4*>(img + ap[t] + dr * C::RPITCH + s_ * C::PITCH + 16 * gq);
  };
  auto mfma_group = [&](int buf, bool with_left) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < C::NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[buf][j], av[buf][t][j], acc[t], 0, 0, 0);
    if (with_left) left_mfma();
    // the next group's reads, ONE between every three MFMAs: issued as a burst at the top of the group (all four matrix waves at once,
    // right behind the barrier) they fill the LDS queue and the in-order wave cannot issue its next MFMA for 300-400 cycles per group
    // (best effort: the pattern lists more read slots than a group has reads; the left-over chunk's reads and MFMAs ride along)
#pragma unroll
    for (int q = 0; q < (4 * C::NT + CQ) / 3; ++q) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); }
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // one chunk = tap s_ of the current kernel row; `i` = its logical chunk index (ring slot i % NR); `bump` != 0: the row's last tap — the
  // tile origins move to the next kernel row of the order (by bump floats) before its first group is read; `last`: nothing to read behind it
  auto chunk = [&](int i, int s_, int bump, bool row_end, bool last) {
#pragma unroll
    for (int gq = 0; gq < C::GR; ++gq) {
      const int nb = (gq + 1) & 1;
      if (gq + 1 < C::GR) load_group(av[nb], bv[nb], 0, s_, gq + 1, i % NR);
      else if (!last) {
        if (row_end) {
#pragma unroll
          for (int t = 0; t < C::NT; ++t) ap[t] += bump;                 // (this group's fragments are in registers already)
          lap += bump;
          load_group(av[nb], bv[nb], 0, 0, 0, (i + 1) % NR);
        } else load_group(av[nb], bv[nb], 0, s_ + 1, 0, (i + 1) % NR);
      }
      if (gq == 0) left_load(s_, i % NR);                              // (this chunk's left-over operands: used by its LAST group)
      mfma_group(gq & 1, gq + 1 == C::GR);
    }
    // barrier #(i + 1), WITHOUT __syncthreads()'s fence: that would drain lgkmcnt and expose the reads just issued for chunk i + 1.  This
    // wave wrote nothing; the operand makes the wait for chunk i's last B fragments (and, LDS reads returning in order, for every read of
    // ring slot i % NR) precede the barrier, behind which the staging waves refill that slot; the reads still in flight touch other slots
#ifdef SDQN_TIMING
    if (!(c.dbg & 2))
#endif
    asm volatile("s_barrier" :: "v"(bv[(C::GR - 1) & 1][3]) : "memory");
  };
#pragma unroll
  for (int t = 0; t < C::NT; ++t) ap[t] += C::rord(0) * C::RPITCH;
  lap += C::rord(0) * C::RPITCH;
  load_group(av[0], bv[0], 0, 0, 0, 0);
#pragma unroll 1
  for (int ri = 0; ri < RK; ++ri) {
    int bump = 0;
#pragma unroll
    for (int q = 0; q < RK; ++q) if (ri == q) bump = (C::rord(q + 1) - C::rord(q)) * C::RPITCH;
#pragma unroll
    for (int s_ = 0; s_ < C::S; ++s_) {
      chunk(C::S == NR ? s_ : ri * C::S + s_, s_, bump, s_ + 1 == C::S, false);     // (the last row's final read: the tail's / the final phase's first group — harmless)
#ifdef SDQN_TIMING
#if !defined(SS_ABL) || SS_ABL != 9
      if (ri == 0 && s_ == 0) SS_STAMP_T(0, 2);
      if (ri == 1 && s_ == 0) SS_STAMP_T(0, 3);
#endif
#endif
    }
  }
#pragma unroll
  for (int s_ = 0; s_ < TAIL; ++s_) chunk(RK * C::S + s_, s_, 0, false, s_ + 1 == TAIL);
  SS_STAMP_T(0, 4);
  // ---- final phase: the last FC chunks (taps TAIL .. of kernel row RK, and on), two tiles at a time, B fragments in registers; a step =
  // one 16-channel group of one tap for both tiles of the pair (8 MFMAs), its two A fragments read two steps ahead through a ring of
  // three, across pair boundaries too — the collection of a finished pair (barrier B, Rectlin, LDS stores, barrier A) exposes no read ----
  float bf[FC][C::GR][4];
#pragma unroll
  for (int f = 0; f < FC; ++f)
#pragma unroll
    for (int gq = 0; gq < C::GR; ++gq)
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[f][gq][j] = wl[((C::KO + f) % NR) * C::WCH + (16 * gq + j) * WPITCH];
  constexpr int NP = (C::NT + 1) / 2, SPP = FC * C::GR, NSTEP = NP * SPP;     // pairs, steps per pair, steps
  f32x4 fa[3][2];
  auto fin_load_a = [&](int u, int w) {                                  // A fragment w (tile of the pair) of step u (compile-time u)
    const int pi = u / SPP, f = (u % SPP) / C::GR, gq = u % C::GR;
    const int t0 = 2 * pi, t1 = t0 + 1 < C::NT ? t0 + 1 : t0;
    const int ci = C::KO + f, dr = 0, s_ = ci % C::S;                    // (the final chunks are taps of kernel row rord(RK): Cfg's chunk-schedule assert)
    fa[u % 3][w] = *reinterpret_cast<const f32x4*>(img + ap[w ? t1 : t0] + dr * C::RPITCH + s_ * C::PITCH + 16 * gq);
  };
  auto fin_load = [&](int u) { fin_load_a(u, 0); fin_load_a(u, 1); };
  fin_load(0); fin_load(1);
  // the collection of a finished pair (two 16-byte LDS stores per lane into the round's buffer, barrier A) rides in the NEXT pair's steps
  // 1..3 — its accumulators are final and stay where they are, so nothing waits: done at the pair's end it idled the matrix pipe ~500
  // cycles per pair (result latency + stores + lgkmcnt(0) + barrier).  The buffers alternate (staging waves' comment above)
  auto collect = [&](int pi) {                                           // (raw sums: the staging waves apply the Rectlin on their way out)
    const int t0 = 2 * pi; const bool two = t0 + 1 < C::NT;
    float* const ob = outl + ((pi & 1) ? (int)(((C::KO + FC) % NR) * C::WCH) - NR * C::WCH : 0) + m * WPITCH + 16 * wave + 4 * kq;
    *reinterpret_cast<f32x4*>(ob) = acc[t0];
    if (two) *reinterpret_cast<f32x4*>(ob + C::OUTT) = acc[two ? t0 + 1 : t0];
  };
#pragma unroll
  for (int u = 0; u < NSTEP; ++u) {
    const int pi = u / SPP, k = u % SPP, f = k / C::GR, gq = k % C::GR;
    const int t0 = 2 * pi;
    const bool two = t0 + 1 < C::NT;
#if defined(SS_ABL) && SS_ABL == 9
    if (u == 0) SS_STAMP_T(0, 2);
    if (u == SPP) SS_STAMP_T(0, 3);
    if (u == 2 * SPP) SS_STAMP_T(0, 6);
    if (u == 3 * SPP) SS_STAMP_T(0, 7);
#endif
    // (the left-over positions' final chunks ride in the LAST pair's steps: operands read in step f, used in step f + 1)
    if constexpr (C::LV > 0) {
      if (pi == NP - 1 && k >= 1 && k <= FC) left_mfma();
      if (pi == NP - 1 && k < FC) left_load((C::KO + k) % C::S, (C::KO + k) % NR);
    }
    if (pi > 0 && k == 1) collect(pi - 1);
    if (pi > 0 && k == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                        // A (pi - 1) (the stores and every read issued so far are two steps old)
    // the order below IS the issue order (a sched_barrier behind every pair of MFMAs): left to itself the scheduler puts the two MFMAs
    // of ONE accumulator back to back — 40 cycles each instead of 32 (the 16 x 16 x 4 shape's dependent latency)
    if (u + 2 < NSTEP) fin_load_a(u + 2, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[t0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[f][gq][j], fa[u % 3][0][j], acc[t0], 0, 0, 0);
      if (two) acc[t0 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[f][gq][j], fa[u % 3][1][j], acc[two ? t0 + 1 : t0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (j == 0 && u + 2 < NSTEP) fin_load_a(u + 2, 1);
    }
  }
  collect(NP - 1);                                                       // the last pair — and this wave's partial sums of the left-over positions
  if constexpr (C::LV > 0) {
    float* const lb = wr + (C::KO % NR) * C::WCH + wave * (C::LV * NO) + lane;      // (chunk KO's ring slot: every read of it — bf, the left-over operands — is done)
#pragma unroll
    for (int l = 0; l < C::LV; ++l) lb[l * NO] = (lacc[0][l] + lacc[1][l]) + (lacc[2][l] + lacc[3][l]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  SS_STAMP_T(0, 5);
}

template <class C>
__global__ void __launch_bounds__(512) conv_ss_kernel(const Args c) {
  __shared__ __attribute__((aligned(16))) float smem[C::LDS];
  Keep<0, 0> none;
  conv_ss_body<C>(c, smem, none);
}

// conv2 -> conv3 for the SAME samples in one launch (round 6): both layers are sample-stationary with the same samples per workgroup, so
// a workgroup's conv3 needs nothing another workgroup computes — it follows its conv2 behind a workgroup barrier instead of a kernel
// boundary (~1.5 us + the second launch's ramp).  Layer 1's output still goes to memory (the backward pass reads it); layer 2's copy
// never leaves the CU: every staging thread keeps the pieces it stored in registers and writes them into layer 2's LDS image behind
// the barrier — no re-read, no wait for the stores — and layer 2's first weight chunks were requested when layer 1's final phase began.
struct ChainArgs { Args l1, l2; };
template <class CA, class CB>
__global__ void __launch_bounds__(512) conv_ss_chain_kernel(const ChainArgs c) {
  static_assert(CA::NS == CB::NS && CA::NPOS == CB::HI * CB::WI && CB::CI == NO, "layer 2 consumes layer 1's samples");
  __shared__ __attribute__((aligned(16))) float smem[CA::LDS > CB::LDS ? CA::LDS : CB::LDS];
  typedef Keep<2 * ((CA::NT + 1) / 2), NR * CB::WP> K;
  K keep;
  conv_ss_body<CA, 1, CB, K>(c.l1, smem, keep, c.l2.w);
  __syncthreads();                                                       // every LDS access of layer 1 is done before layer 2 overwrites the images
  conv_ss_body<CB, 2, NoNext, K>(c.l2, smem, keep);
}

template <class CA, class CB>
inline hipError_t launch_chain(const ChainArgs& c, int nz, hipStream_t s) {
  SDQN_LAUNCH((conv_ss_chain_kernel<CA, CB>), dim3(nz * c.l1.G), dim3(512), 0, s, c);
  return hipGetLastError();
}

template <class C>
inline hipError_t launch(const Args& c, int nz, hipStream_t s) {
  SDQN_LAUNCH((conv_ss_kernel<C>), dim3(nz * c.G), dim3(512), 0, s, c);
  return hipGetLastError();
}

}  // namespace ss
}  // namespace sdqn
