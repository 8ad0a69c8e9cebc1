// bt_map.h — index maps of the BLOCK-TILE engine (gemm_engine_bt.h), the routine of the throughput regime (B >= 128).
// Host + device: tests/emul/emul_bt.cpp executes the very same maps on the CPU (simulated LDS, simulated MFMA lanes), so the
// loader coverage, the two LDS panel layouts, the k-slot agreement of A and B and the tile -> (m, n) maps are validated
// without a GPU.
//
// One workgroup of NT = 256 threads (4 wave64s) owns a BM x BN block of C.  Per 32-deep K-chunk the workgroup stages ONE
// A panel (BM x 32) and ONE B panel (32 x BN) in LDS — every operand element is fetched from L2 once per workgroup and read
// by all waves that need it — and wave (wm, wn) of the WM x WN wave grid multiplies its SM x SN sub-tiles of 32 x 32 out of
// LDS.  Two panel layouts, chosen by the operand's memory-contiguous dimension (problems.h: A_K / B_K):
//   KM  (k-contiguous: im2col rows, activations, dgrad weights): [x][k], pitch 36 floats.  Filled by 16-byte loads along k
//       (8 lanes = one 128-byte row chunk), read back as the lane's own 4 x 16 bytes: ds_read_b128 at row (l & 31),
//       k = 8 j + 4 (l >> 5) — bank (36 x) mod 64 is distinct for the 16 lanes of every ds_read_b128 group.
//   MK  (x-contiguous: weights [k][n], dense deltas, the wgrad im2col operand): [k][x], pitch BX.  Filled by 16-byte loads
//       along x, read back with ds_read_b32 at [kslot][x0 + (l & 31)]: 32 consecutive lanes, conflict-free.
// k-slot <-> logical k is the engine-wide kslot(t, h) = 8 (t >> 2) + 4 h + (t & 3): step t of the chunk, half-wave h.
#pragma once
#include "problems.h"

namespace sdqn {
namespace bt {

constexpr int NT = 256;           // threads per workgroup
constexpr int BK = 32;            // K-chunk depth (fp32: 16 steps of v_mfma_f32_32x32x2_f32)
constexpr int KM_PITCH = 36;      // floats per row of a KM panel (32 + 4: 16-byte aligned rows, conflict-free b128 reads)

SDQN_HD constexpr int kslot(int t, int h) { return 8 * (t >> 2) + 4 * h + (t & 3); }

// ---- loader items: every thread moves `passes(BX)` float4 per operand and chunk -----------------------------------------
SDQN_HD constexpr int passes(int BX) { return BX / 32; }
// KM: item (row, k4): 8 threads cover the 32 k of one row
SDQN_HD constexpr int km_item_row(int tid, int p) { return (tid >> 3) + 32 * p; }
SDQN_HD constexpr int km_item_k(int tid) { return (tid & 7) * 4; }
SDQN_HD constexpr int km_off(int x, int k) { return x * KM_PITCH + k; }
SDQN_HD constexpr int km_floats(int BX) { return BX * KM_PITCH; }
// MK: item (k, x4): BX / 4 threads cover one k-row of the panel
SDQN_HD constexpr int mk_item_k(int BX, int tid, int p) { return tid / (BX / 4) + (NT / (BX / 4)) * p; }
SDQN_HD constexpr int mk_item_x(int BX, int tid) { return (tid % (BX / 4)) * 4; }
SDQN_HD constexpr int mk_off(int BX, int k, int x) { return k * BX + x; }
SDQN_HD constexpr int mk_floats(int BX) { return BK * BX; }

SDQN_HD constexpr int panel_floats(bool kcontig, int BX) { return kcontig ? km_floats(BX) : mk_floats(BX); }

// ---- fragment of lane (i = l & 31, h = l >> 5) for sub-tile row/column x0 + i, MFMA step t ------------------------------
SDQN_HD constexpr int frag_off(bool kcontig, int BX, int x, int t, int h) {
  return kcontig ? km_off(x, kslot(t, h)) : mk_off(BX, kslot(t, h), x);
}
// accumulator register r of lane l is C[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31] of its 32 x 32 sub-tile
SDQN_HD constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

}  // namespace bt
}  // namespace sdqn
