// problems.h — every GEMM-shaped stage of the DQN train step expressed as
//   C(m, n) = sum_k A(m, k) * B(k, n)      A(m,k) = srcA[a_row(m) + a_col(k)]
//                                            B(k,n) = srcB[b_row(k) + b_col(n)]
// i.e. im2col is never materialised: it is a separable gather performed by the tile engine's operand
// loaders (gemm_engine.h).  The functions are __host__ __device__ so tests/emul can execute the very
// same index math on the CPU.  Per problem: A_K/B_K say which dimension of the operand is contiguous in
// memory, *_REG mark plain row-major matrices (fast path), store() is the fused epilogue.
//
// Reference: the layer stack of src/deepqnetwork.py:83-91 and Neon's
// fprop/bprop/update semantics.
//
// Internal layouts (ours, chosen for coalesced loads; Neon layouts only at the C ABI):
//   activations / deltas : NHWC  [z][n][y][x][c]            (row-major GEMM outputs)
//   W1i [(c,r,s)=256][32]   == Neon (CRS, K)
//   W2i [(r,s,c)=512][64]   rows permuted from Neon's (c,r,s)
//   W3i [(r,s,c)=576][64]   rows permuted from Neon's (c,r,s)
//   W4i [(pix,f)=3136][512] == transpose of Neon (512, 3136[(f,pix)])
//   W5i [A][512]            == Neon
//   d3p [n][11][11][64]  conv3-output delta, zero-padded by 2 (full correlation for dgrad); d3 = dense copy
//   d2p [n][11][11][64]  conv2-output delta, zero-padded by 1 (stride-2 parity decomposition); d2 = dense copy
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define SDQN_HD __host__ __device__ inline
#else
#define SDQN_HD inline
#endif

namespace sdqn {

constexpr int H0 = 84, W0 = 84, C0 = 4;
constexpr int FRAME = H0 * W0;          // 7056 B
constexpr int STATE = C0 * FRAME;       // 28224 B
constexpr int ST1 = 4, P1 = 20, Q1 = 20, K1 = 32;   // conv 8x8 s4   deepqnetwork.py:83
constexpr int ST2 = 2, P2 = 9, Q2 = 9, K2 = 64;     // conv 4x4 s2   :85
constexpr int P3 = 7, Q3 = 7, K3 = 64;              // conv 3x3 s1   :87
constexpr int PIX1 = P1 * Q1, PIX2 = P2 * Q2, PIX3 = P3 * Q3;
constexpr int CRS1 = 256, CRS2 = 512, CRS3 = 576;
constexpr int NFC = 512, NIN4 = PIX3 * K3;          // Affine 512    :89
constexpr int PD3 = 11, PD2 = 11;                   // padded delta planes
constexpr int NW1 = CRS1 * K1, NW2 = CRS2 * K2, NW3 = CRS3 * K3, NW4 = NIN4 * NFC;
constexpr int OFF1 = 0, OFF2 = OFF1 + NW1, OFF3 = OFF2 + NW2, OFF4 = OFF3 + NW3, OFF5 = OFF4 + NW4;
constexpr int MAX_ACTIONS = 18;
// --batch_norm: BatchNorm after conv1..3 and fc4 (features per layer), parameter block appended to the flat buffers:
// [beta_l | gamma_l] per layer at BN_OFF(l), then the running statistics [gmean_l | gvar_l] at BN_PARAMS + BN_OFF(l)
constexpr int BN_LAYERS = 4;
constexpr int BN_PARAMS = 2 * (32 + 64 + 64 + 512);      // 1344
SDQN_HD constexpr int bn_features(int l) { return l == 0 ? 32 : (l == 3 ? 512 : 64); }
SDQN_HD constexpr int bn_off(int l) { return l == 0 ? 0 : (l == 1 ? 64 : (l == 2 ? 192 : 320)); }

#if defined(__HIPCC__)
typedef _Float16 half_t;                                             // IEEE binary16 storage type of the fp16 mode
typedef _Float16 half8 __attribute__((ext_vector_type(8)));          // one f16-MFMA operand fragment (16 B)
#else
typedef uint16_t half_t;                                             // opaque on the host (emulator never runs fp16 problems)
#endif

struct MetaRec {            // device mirror of (rewards, actions, terminals) of one ring slot
  int64_t reward;
  uint8_t action;
  uint8_t terminal;
  uint8_t pad[6];
};

struct StepArgs {
  const uint8_t* src;       // ring mirror (from_ring) or staging states [2][B][STATE]
  const int64_t* idx;       // sampled indexes [B] in DEVICE memory (copied from the pinned slot by prep_kernel)
  int from_ring;
  int B, A, nz;
  const float* theta[2];    // flat parameter buffers: [0] online, [1] target
  float* a1;                // [2][B*400][32]
  float* a2;                // [2][B*81][64]
  float* a3;                // [2][B*49][64]
  float* slab4;             // fc4 fwd split-K slabs [S4][2][B][512]
  float* a4;                // [2][B][512]
  float* d4;                // [B][512]
  float* d3p;               // [B][11][11][64]
  float* d2p;               // [B][11][11][64]
  float* d3;                // [B*49][64]  dense copy of the conv3-output delta (regular B operand of conv3 wgrad)
  float* d2;                // [B*81][64]  dense copy of the conv2-output delta (regular B operand of conv2 wgrad)
  float* d1;                // [B][20][20][32]
  float* g;                 // flat gradient sum (internal layout)
  float* slab1;             // conv wgrad split-K slabs [ns][NWx]
  float* slab2;
  float* slab3;
  int S4;                   // fc4 fwd K-splits
  int tps1, tps2, tps3;     // K-tiles (of 32) per wgrad split
  // single-GPU fast path: RMSProp of the fc4 weights (95 % of all parameters) fused into the fc4 wgrad
  // epilogue, so the 6.4 MB gradient is never written to / re-read from HBM
  int fuse_rms;
  int f4w_first, f4w_count; // fc4 wgrad tiles [first, first+count) handled by THIS launch (tiles are spread over
                            // the three backward launches so the 25.7 MB fused RMSProp RMW streams in the background)
  // ---- fp16 mode (--datatype float16): activations / deltas / MFMA weight operands in half, everything else fp32
  half_t *h_a1, *h_a2, *h_a3;      // [2][B*PIX][K] like a1..a3
  half_t *h_d4, *h_d3p, *h_d3, *h_d2p, *h_d2, *h_d1;   // deltas, pre-multiplied by loss_scale
  const half_t* wh[2];             // half copy of theta / theta- in the master (internal) layout: dgrad B operand
  const half_t* wht[2];            // half copy TRANSPOSED per layer ([n][k], k contiguous): forward B operand
  half_t *wh_w, *wht_w;            // writable aliases of wh[0] / wht[0] (refreshed by the optimizer epilogues)
  float loss_scale, inv_loss_scale;
  int h16;                         // 1: fp16 mode
  int xcd_map;              // XCD-contiguous workgroup->tile map (cuts fabric traffic to ~algorithmic): bit i = problem i of the launch
  int reserved_[12];        // (keeps the field offsets of the round-1 layout: the scalar-load schedule hipcc derives from them is part of
                            //  the tuned kernels — a re-packed struct measured 1 % slower; host-only tuning lives in LaunchTune, kernels.h)
  float* __restrict__ theta_w;   // online parameters, writable alias of theta[0]
  float* __restrict__ state;     // RMSProp state
  float bsz, rho, one_minus_rho, lr, eps;
  // --batch_norm (deepqnetwork.py:26,83-89): launch_kernel picks the *Raw forward problems (linear output without the
  // Rectlin: the BatchNorm + Rectlin pass of bn_kernels.hip follows) and the head variant that reads an activated a4
  int bn;
  const int64_t* idx_t;     // hoist option only (Conv1FwdTarget): the NEXT step's indexes, whose target conv1 rides in this step's K_BWD2
  // ---- round 3 (appended: every earlier field keeps its offset) ----
  // fc4_wgrad + fused RMSProp riding in the fc4_dgrad launch (sdqn_kernels_r3.hip): the dgrad tile of W4 row block j publishes
  // "my reads of W4 rows [32 j, 32 j + 32) are done" as f4d_flags[16 j] = f4d_epoch; an fc4_wgrad tile of the same row block
  // waits for it before its in-place RMSProp store (a write-after-read hand-off: no data crosses, so no cache visibility issue)
  unsigned* f4d_flags;      // [NIN4 / 32][16] one word per row block, 64 B apart; [NIN4 / 32 * 16] = sticky time-out word
  unsigned f4d_epoch;       // step number (monotonic, never 0): flags never need a reset
  int r3_pad_;
  // conv1 forward on packed-bf16 MFMA (sdqn_kernels_r3.hip): W1 of the online [0] / target [1] net as THREE bf16 planes whose sum is
  // the fp32 weight exactly, each [32 output maps][256 k] with k contiguous — written by whoever writes W1 (update kernel, set_weights,
  // target sync)
  const unsigned short* w1p[2];
  // ---- round 4 (appended): bf16 planes of the conv2 / conv3 / fc4 weights for the block-tile engine's plane mode (gemm_engine_bt.h:
  // bt_tile_xp).  Three planes whose sum is the fp32 weight exactly (split_bf16x3), plane stride XP_PLANE elements, written by whoever
  // writes the weights (update kernel, fc4_wgrad's fused RMSProp epilogue, set_weights, target sync, DP broadcast):
  //   wpm    : online net, MASTER layout (same element index as theta): the k-contiguous B operand of the dgrads
  //   wpt[z] : per net, every layer TRANSPOSED ([n][K], k contiguous: the fp16 mode's wht indexing): the B operand of conv2 / conv3 forward
  const unsigned short* wpm;
  const unsigned short* wpt[2];
  int xp;                   // 0: fp32 MFMA everywhere; 9 / 6: plane mode with 9 / 6 exact partial products per fp32 product
  int xp_pad_;
};
constexpr int XP_PLANE = OFF5;            // elements per weight plane (conv1 .. fc4; fc5 is not a GEMM stage of the engine)

// A9 + A10 in Neon's operation order (the library is built with -ffp-contract=off: one rounding per op)
// grad / be.bsz (A9), exactly: for a power-of-two divisor (B = 32, 256, ...; R*B under data parallel) the quotient
// equals the product with the exactly representable reciprocal, which saves one of the two IEEE divisions per weight
SDQN_HD float div_bsz(float x, float bsz) {
  union { float f; uint32_t u; } b, r; b.f = bsz;
  if ((b.u & 0x007FFFFFu) == 0u && b.u >= 0x00800000u && b.u < 0x7E800000u) { r.u = 0x7F000000u - b.u; return x * r.f; }
  return x / bsz;
}
SDQN_HD float rms_step(float w, float& st, float gsum, float bsz, float rho, float omr, float lr, float eps) {
  const float g = div_bsz(gsum, bsz);                 // grad / be.bsz
  st = rho * st + (g * g) * omr;                      // state = rho*state + g^2*(1-rho)
  return w - (g * lr) / (sqrtf(st + eps) + eps);
}

#if defined(__HIPCC__)
// two weights at once: the multiplies / adds become packed-fp32 instructions (v_pk_mul_f32, v_pk_add_f32), the square
// root and the division stay scalar IEEE sequences — operation for operation the same arithmetic as rms_step
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ inline void rms_step2(float& w0, float& w1, float& st0, float& st1, float gs0, float gs1,
                                 float bsz, float rho, float omr, float lr, float eps) {
  union { float f; uint32_t u; } b, r; b.f = bsz;
  f2v gsum = {gs0, gs1}, g;
  if ((b.u & 0x007FFFFFu) == 0u && b.u >= 0x00800000u && b.u < 0x7E800000u) { r.u = 0x7F000000u - b.u; g = gsum * r.f; }
  else { g.x = gs0 / bsz; g.y = gs1 / bsz; }
  f2v st = {st0, st1};
  const f2v a = rho * st, q = (g * g) * omr;
  st = a + q;
  const f2v num = g * lr, rad = st + eps;
  f2v den; den.x = sqrtf(rad.x); den.y = sqrtf(rad.y);
  den = den + eps;
  f2v quo; quo.x = num.x / den.x; quo.y = num.y / den.y;
  const f2v w = (f2v){w0, w1} - quo;
  w0 = w.x; w1 = w.y; st0 = st.x; st1 = st.y;
}
#endif

// fp32 -> three bf16 (bit patterns) with hi + mid + lo == w EXACTLY: bf16 keeps 8 significant bits, every residual of a
// round-to-nearest split has at most 16, then 8 — all representable, so the two subtractions are exact (weights are finite and far
// from the subnormal range)
SDQN_HD uint16_t bf16_rn(float f) {
  union { float f; uint32_t u; } x; x.f = f;
  return (uint16_t)((x.u + 0x7FFFu + ((x.u >> 16) & 1u)) >> 16);
}
SDQN_HD float bf16_up(uint16_t b) { union { float f; uint32_t u; } x; x.u = (uint32_t)b << 16; return x.f; }
SDQN_HD void split_bf16x3(float w, uint16_t& hi, uint16_t& mid, uint16_t& lo) {
  hi = bf16_rn(w); const float r1 = w - bf16_up(hi);
  mid = bf16_rn(r1); const float r2 = r1 - bf16_up(mid);
  lo = bf16_rn(r2);
}
constexpr int W1P_PLANE = K1 * CRS1;      // elements per plane: [32 maps][256 k]

SDQN_HD int64_t sbase(const StepArgs& a, int z, int n) {
  // replay_memory.py:71-72: prestate = screens[i-4:i], poststate = screens[i-3:i+1]
  return a.from_ring ? (a.idx[n] - C0 + z) * (int64_t)FRAME : ((int64_t)z * a.B + n) * (int64_t)STATE;
}
// deepqnetwork.py:100 be.divide(input, 255): correctly-rounded x/255 without a divide — q = x*r, one fma
// Newton correction (bit-identical to IEEE x/255.0f for all 256 byte values: tests/test_emul.py checks).
SDQN_HD float norm_u8(uint32_t v) {
  const float x = (float)v, r = 1.0f / 255.0f;
  const float q = x * r;
  const float e = fmaf(-q, 255.0f, x);
  return fmaf(e, r, q);
}
struct f4 { float x, y, z, w; };
SDQN_HD f4 ld4(const float* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float4 v = *reinterpret_cast<const float4*>(p); f4 o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w; return o;
#else
  f4 o; o.x = p[0]; o.y = p[1]; o.z = p[2]; o.w = p[3]; return o;
#endif
}
// bytes p[0], p[4], p[8], p[12] (four consecutive conv1 output positions of one patch element), normalised:
// ONE unaligned 16-byte load on the device (gfx9+ global loads take any byte alignment), v_cvt_f32_ubyte0 per dword.
// Reads up to 12 bytes past the last byte used: frame buffers are allocated with SRC_PAD bytes of slack.
constexpr int SRC_PAD = 16;
SDQN_HD f4 ld4_u8_stride4(const uint8_t* p) {
  f4 o;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(1)));
  const u32x4 w = *reinterpret_cast<const u32x4*>(p);
  o.x = norm_u8(w.x & 255u); o.y = norm_u8(w.y & 255u); o.z = norm_u8(w.z & 255u); o.w = norm_u8(w.w & 255u);
#else
  o.x = norm_u8(p[0]); o.y = norm_u8(p[4]); o.z = norm_u8(p[8]); o.w = norm_u8(p[12]);
#endif
  return o;
}
SDQN_HD f4 ld4_u8(const uint8_t* p) {       // 4 consecutive bytes (4-byte aligned by construction) -> 4 normalised floats
  uint32_t w;
#if defined(__HIP_DEVICE_COMPILE__)
  w = *reinterpret_cast<const uint32_t*>(p);
#else
  w = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
#endif
  f4 o; o.x = norm_u8(w & 255u); o.y = norm_u8((w >> 8) & 255u); o.z = norm_u8((w >> 16) & 255u); o.w = norm_u8(w >> 24); return o;
}

SDQN_HD int ceil_div(int a, int b) { return (a + b - 1) / b; }

// XCD-aware workgroup -> tile map (guide T1).  The dispatcher places workgroup b on XCD b % 8, each XCD with its
// own L2.  Tiles are numbered x-fastest (then y, then split/net z), so NEIGHBOURING tile ids share operands
// (adjacent im2col rows and halos; the same K-range of activations for all (crs, f) tiles of a wgrad split).
// Giving every XCD one CONTIGUOUS run of tile ids keeps those re-reads in one L2 instead of eight.  Bijective
// for any workgroup count; placement only changes speed, never results.  Measured (profiles/README.md): fabric
// traffic drops to ~1.0-1.7x algorithmic (bwd1 15.0 -> 3.8 MB) but the step gets 3-4 % SLOWER at B=32 and B=256 —
// these launches are latency-bound and eight L2s fetching a tile's neighbourhood in parallel beat one — so the
// map is an option (StepArgs::xcd_map, sdqn_net_set_option "xcd_map"), off by default.
SDQN_HD int xcd_tile_id(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, i = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}
// the same for a sub-range [s, s + n) of a grid (one problem of a multi-problem launch): workgroup b's XCD is
// still b % 8; the range's workgroups on XCD x get one contiguous run of the range's n tile ids
SDQN_HD int xcd_tile_id_range(int b, int s, int n) {
  const int x = b & 7;
  int before = 0, mine_first = 0;
  for (int y = 0; y < 8; ++y) {
    const int first = s + ((y - (s & 7) + 8) & 7);              // first workgroup of the range on XCD y
    const int cnt = first < s + n ? (s + n - first + 7) >> 3 : 0;
    if (y < x) before += cnt;
    if (y == x) mine_first = first;
  }
  return before + ((b - mine_first) >> 3);
}


// ---- im2col index helpers -------------------------------------------------------------
SDQN_HD int64_t row1(const StepArgs& a, int z, int m) {      // conv1 patch origin in the byte source
  int n = m / PIX1, pix = m - n * PIX1, p = pix / Q1, q = pix - p * Q1;
  return sbase(a, z, n) + (int64_t)(p * ST1) * W0 + q * ST1;
}
SDQN_HD int col1(int k) { int c = k >> 6, r = (k >> 3) & 7, s = k & 7; return c * FRAME + r * W0 + s; }
SDQN_HD int row2(const StepArgs& a, int z, int m) {          // conv2 patch origin in a1 (NHWC)
  int n = m / PIX2, pix = m - n * PIX2, p = pix / Q2, q = pix - p * Q2;
  return (((z * a.B + n) * P1 + p * ST2) * Q1 + q * ST2) * K1;
}
SDQN_HD int col2(int k) { int r = k >> 7, s = (k >> 5) & 3, c = k & 31; return (r * Q1 + s) * K1 + c; }
SDQN_HD int row3(const StepArgs& a, int z, int m) {          // conv3 patch origin in a2 (NHWC)
  int n = m / PIX3, pix = m - n * PIX3, p = pix / Q3, q = pix - p * Q3;
  return (((z * a.B + n) * P2 + p) * Q2 + q) * K2;
}
SDQN_HD int col3(int k) { int rs = k >> 6, c = k & 63, r = rs / 3, s = rs - r * 3; return (r * Q2 + s) * K2 + c; }
SDQN_HD int prow2(int m) {                                   // (n,p,q) of conv2 output -> d2p (pad 1)
  int n = m / PIX2, pix = m - n * PIX2, p = pix / Q2, q = pix - p * Q2;
  return ((n * PD2 + p + 1) * PD2 + q + 1) * K2;
}

// =========================== forward =====================================================
struct Conv1Fwd {   // fused gather + normalise + conv1 + ReLU: replay_memory.py:71-72 + deepqnetwork.py:94-100,83
  static constexpr bool A_K = true, B_K = false;     // operand contiguous along k (-> LDS transpose) or along m/n
  typedef int64_t aoff_t;
  // operand descriptors for the engine's fast paths: *_REG = plain row-major [k][x] matrix with row pitch *_LD
  static constexpr bool A_REG = false, A_U8 = true, B_REG = true; static constexpr int A_LD = 0, B_LD = K1;
  SDQN_HD static const float* a_ptr(const StepArgs& a, int z) { (void)z; return (const float*)nullptr; }
  SDQN_HD static const float* b_ptr(const StepArgs& a, int z) { (void)z; return a.theta[z] + OFF1; }
#if defined(__HIPCC__)
  struct Epi {};
  __device__ static void epi_begin(const StepArgs&, int, int, int, Epi&) {}
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi&);
#endif
  SDQN_HD static int M(const StepArgs& a) { return a.B * PIX1; }
  SDQN_HD static int N(const StepArgs&) { return K1; }
  SDQN_HD static int nbz(const StepArgs& a) { return a.nz; }
  SDQN_HD static void ksplit(const StepArgs&, int bz, int& z, int& ks, int& kb, int& ke) { z = bz; ks = 0; kb = 0; ke = CRS1; }
  SDQN_HD static aoff_t a_row(const StepArgs& a, int z, int m) { return row1(a, z, m); }
  SDQN_HD static aoff_t a_col(const StepArgs&, int, int k) { return col1(k); }
  SDQN_HD static float a_load(const StepArgs& a, int, aoff_t o) { return norm_u8(a.src[o]); }
  SDQN_HD static f4 a_load4(const StepArgs& a, int, aoff_t o) { return ld4_u8(a.src + o); }
  SDQN_HD static int b_row(const StepArgs&, int, int k) { return k * K1; }
  SDQN_HD static int b_col(const StepArgs&, int, int n) { return n; }
  SDQN_HD static float b_load(const StepArgs& a, int z, int o) { return a.theta[z][OFF1 + o]; }
  SDQN_HD static f4 b_load4(const StepArgs& a, int z, int o) { return ld4(a.theta[z] + OFF1 + o); }
  SDQN_HD static void store(const StepArgs& a, int z, int, int m, int n, float v) {
    a.a1[((int64_t)z * M(a) + m) * K1 + n] = fmaxf(v, 0.0f);
  }
};

struct Conv2Fwd {   // deepqnetwork.py:85
  static constexpr bool A_K = true, B_K = false;     // operand contiguous along k (-> LDS transpose) or along m/n
  typedef int aoff_t;
  // operand descriptors for the engine's fast paths: *_REG = plain row-major [k][x] matrix with row pitch *_LD
  static constexpr bool A_REG = false, A_U8 = false, B_REG = true; static constexpr int A_LD = 0, B_LD = K2;
  SDQN_HD static const float* a_ptr(const StepArgs& a, int z) { (void)z; return a.a1; }
  SDQN_HD static const float* b_ptr(const StepArgs& a, int z) { (void)z; return a.theta[z] + OFF2; }
#if defined(__HIPCC__)
  struct Epi {};
  __device__ static void epi_begin(const StepArgs&, int, int, int, Epi&) {}
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi&);
#endif
  SDQN_HD static int M(const StepArgs& a) { return a.B * PIX2; }
  SDQN_HD static int N(const StepArgs&) { return K2; }
  SDQN_HD static int nbz(const StepArgs& a) { return a.nz; }
  SDQN_HD static void ksplit(const StepArgs&, int bz, int& z, int& ks, int& kb, int& ke) { z = bz; ks = 0; kb = 0; ke = CRS2; }
  SDQN_HD static aoff_t a_row(const StepArgs& a, int z, int m) { return row2(a, z, m); }
  SDQN_HD static aoff_t a_col(const StepArgs&, int, int k) { return col2(k); }
  SDQN_HD static float a_load(const StepArgs& a, int, aoff_t o) { return a.a1[o]; }
  SDQN_HD static f4 a_load4(const StepArgs& a, int, aoff_t o) { return ld4(a.a1 + o); }
  SDQN_HD static int b_row(const StepArgs&, int, int k) { return k * K2; }
  SDQN_HD static int b_col(const StepArgs&, int, int n) { return n; }
  SDQN_HD static float b_load(const StepArgs& a, int z, int o) { return a.theta[z][OFF2 + o]; }
  SDQN_HD static f4 b_load4(const StepArgs& a, int z, int o) { return ld4(a.theta[z] + OFF2 + o); }
  SDQN_HD static void store(const StepArgs& a, int z, int, int m, int n, float v) {
    a.a2[((int64_t)z * M(a) + m) * K2 + n] = fmaxf(v, 0.0f);
  }
};

struct Conv3Fwd {   // deepqnetwork.py:87
  static constexpr bool A_K = true, B_K = false;     // operand contiguous along k (-> LDS transpose) or along m/n
  typedef int aoff_t;
  // operand descriptors for the engine's fast paths: *_REG = plain row-major [k][x] matrix with row pitch *_LD
  static constexpr bool A_REG = false, A_U8 = false, B_REG = true; static constexpr int A_LD = 0, B_LD = K3;
  SDQN_HD static const float* a_ptr(const StepArgs& a, int z) { (void)z; return a.a2; }
  SDQN_HD static const float* b_ptr(const StepArgs& a, int z) { (void)z; return a.theta[z] + OFF3; }
#if defined(__HIPCC__)
  struct Epi {};
  __device__ static void epi_begin(const StepArgs&, int, int, int, Epi&) {}
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi&);
#endif
  SDQN_HD static int M(const StepArgs& a) { return a.B * PIX3; }
  SDQN_HD static int N(const StepArgs&) { return K3; }
  SDQN_HD static int nbz(const StepArgs& a) { return a.nz; }
  SDQN_HD static void ksplit(const StepArgs&, int bz, int& z, int& ks, int& kb, int& ke) { z = bz; ks = 0; kb = 0; ke = CRS3; }
  SDQN_HD static aoff_t a_row(const StepArgs& a, int z, int m) { return row3(a, z, m); }
  SDQN_HD static aoff_t a_col(const StepArgs&, int, int k) { return col3(k); }
  SDQN_HD static float a_load(const StepArgs& a, int, aoff_t o) { return a.a2[o]; }
  SDQN_HD static f4 a_load4(const StepArgs& a, int, aoff_t o) { return ld4(a.a2 + o); }
  SDQN_HD static int b_row(const StepArgs&, int, int k) { return k * K3; }
  SDQN_HD static int b_col(const StepArgs&, int, int n) { return n; }
  SDQN_HD static float b_load(const StepArgs& a, int z, int o) { return a.theta[z][OFF3 + o]; }
  SDQN_HD static f4 b_load4(const StepArgs& a, int z, int o) { return ld4(a.theta[z] + OFF3 + o); }
  SDQN_HD static void store(const StepArgs& a, int z, int, int m, int n, float v) {
    a.a3[((int64_t)z * M(a) + m) * K3 + n] = fmaxf(v, 0.0f);
  }
};

struct Fc4Fwd {     // deepqnetwork.py:89, split-K over S4 slabs; bias-free, ReLU applied by the head kernel
  static constexpr bool A_K = true, B_K = false;     // operand contiguous along k (-> LDS transpose) or along m/n
  typedef int aoff_t;
  // operand descriptors for the engine's fast paths: *_REG = plain row-major [k][x] matrix with row pitch *_LD
  static constexpr bool A_REG = false, A_U8 = false, B_REG = true; static constexpr int A_LD = 0, B_LD = NFC;
  SDQN_HD static const float* a_ptr(const StepArgs& a, int z) { (void)z; return a.a3; }
  SDQN_HD static const float* b_ptr(const StepArgs& a, int z) { (void)z; return a.theta[z] + OFF4; }
#if defined(__HIPCC__)
  struct Epi {};
  __device__ static void epi_begin(const StepArgs&, int, int, int, Epi&) {}
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi&);
#endif
  SDQN_HD static int M(const StepArgs& a) { return a.B; }
  SDQN_HD static int N(const StepArgs&) { return NFC; }
  SDQN_HD static int nbz(const StepArgs& a) { return a.nz * a.S4; }
  SDQN_HD static void ksplit(const StepArgs& a, int bz, int& z, int& ks, int& kb, int& ke) {
    z = bz / a.S4; ks = bz - z * a.S4;
    int per = ceil_div(NIN4 / 32, a.S4) * 32;
    kb = ks * per; ke = kb + per; if (ke > NIN4) ke = NIN4; if (kb > NIN4) kb = NIN4;
  }
  SDQN_HD static aoff_t a_row(const StepArgs& a, int z, int m) { return (z * a.B + m) * NIN4; }
  SDQN_HD static aoff_t a_col(const StepArgs&, int, int k) { return k; }
  SDQN_HD static float a_load(const StepArgs& a, int, aoff_t o) { return a.a3[o]; }
  SDQN_HD static f4 a_load4(const StepArgs& a, int, aoff_t o) { return ld4(a.a3 + o); }
  SDQN_HD static int b_row(const StepArgs&, int, int k) { return k * NFC; }
  SDQN_HD static int b_col(const StepArgs&, int, int n) { return n; }
  SDQN_HD static float b_load(const StepArgs& a, int z, int o) { return a.theta[z][OFF4 + o]; }
  SDQN_HD static f4 b_load4(const StepArgs& a, int z, int o) { return ld4(a.theta[z] + OFF4 + o); }
  SDQN_HD static void store(const StepArgs& a, int z, int ks, int m, int n, float v) {
    a.slab4[(((int64_t)ks * 2 + z) * a.B + m) * NFC + n] = v;
  }
};

// The same forward stages for the TARGET net only (z = 1 whatever StepArgs::nz says): problems of the multi-problem
// launches that carry the next step's target forward (deepqnetwork.py:119-125 depends on theta- and the sampled
// poststates only — not on anything the current step computes).  Same tiles, same K split: bit-identical values.
template <class P> struct TargetOnly : P {
  SDQN_HD static int nbz(const StepArgs&) { return 1; }
  SDQN_HD static void ksplit(const StepArgs& a, int, int& z, int& ks, int& kb, int& ke) { P::ksplit(a, 1, z, ks, kb, ke); }
};
struct Conv1FwdTarget : Conv1Fwd {       // target conv1 of the NEXT step: gathers with StepArgs::idx_t (poststates = screens[i-3 : i+1])
  SDQN_HD static int nbz(const StepArgs&) { return 1; }
  SDQN_HD static void ksplit(const StepArgs& a, int, int& z, int& ks, int& kb, int& ke) { Conv1Fwd::ksplit(a, 1, z, ks, kb, ke); }
  SDQN_HD static aoff_t a_row(const StepArgs& a, int, int m) {
    int n = m / PIX1, pix = m - n * PIX1, p = pix / Q1, q = pix - p * Q1;
    return (a.idx_t[n] - C0 + 1) * (int64_t)FRAME + (int64_t)(p * ST1) * W0 + q * ST1;
  }
};
struct Fc4FwdTarget : Fc4Fwd {
  SDQN_HD static int nbz(const StepArgs& a) { return a.S4; }
  SDQN_HD static void ksplit(const StepArgs& a, int bz, int& z, int& ks, int& kb, int& ke) { Fc4Fwd::ksplit(a, bz + a.S4, z, ks, kb, ke); }
};

// =========================== backward (online net, z = 0) ===================================
struct Fc4Dgrad {   // delta3 = (W4^T delta4) * 1[a3 > 0]  (A5, A8), written straight into the padded d3p
  static constexpr bool A_K = true, B_K = true;     // operand contiguous along k (-> LDS transpose) or along m/n
  typedef int aoff_t;
  // operand descriptors for the engine's fast paths: *_REG = plain row-major [k][x] matrix with row pitch *_LD
  static constexpr bool A_REG = false, A_U8 = false, B_REG = false; static constexpr int A_LD = 0, B_LD = 0;
  SDQN_HD static const float* a_ptr(const StepArgs& a, int z) { (void)z; return a.d4; }
  SDQN_HD static const float* b_ptr(const StepArgs& a, int z) { (void)z; return a.theta[0] + OFF4; }
#if defined(__HIPCC__)
  struct Epi {};
  __device__ static void epi_begin(const StepArgs&, int, int, int, Epi&) {}
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi&);
#endif
  SDQN_HD static int M(const StepArgs& a) { return a.B; }
  SDQN_HD static int N(const StepArgs&) { return NIN4; }
  SDQN_HD static int nbz(const StepArgs&) { return 1; }
  SDQN_HD static void ksplit(const StepArgs&, int, int& z, int& ks, int& kb, int& ke) { z = 0; ks = 0; kb = 0; ke = NFC; }
  SDQN_HD static aoff_t a_row(const StepArgs&, int, int m) { return m * NFC; }
  SDQN_HD static aoff_t a_col(const StepArgs&, int, int k) { return k; }
  SDQN_HD static float a_load(const StepArgs& a, int, aoff_t o) { return a.d4[o]; }
  SDQN_HD static f4 a_load4(const StepArgs& a, int, aoff_t o) { return ld4(a.d4 + o); }
  SDQN_HD static int b_row(const StepArgs&, int, int k) { return k; }
  SDQN_HD static int b_col(const StepArgs&, int, int n) { return n * NFC; }
  SDQN_HD static float b_load(const StepArgs& a, int, int o) { return a.theta[0][OFF4 + o]; }
  SDQN_HD static f4 b_load4(const StepArgs& a, int, int o) { return ld4(a.theta[0] + OFF4 + o); }
  SDQN_HD static void store(const StepArgs& a, int, int, int m, int n, float v) {
    int pix = n >> 6, f = n & 63, p = pix / Q3, q = pix - p * Q3;
    bool on = a.a3[(int64_t)m * NIN4 + n] > 0.0f;
    const float dv = on ? v : 0.0f;
    a.d3p[((m * PD3 + p + 2) * PD3 + q + 2) * K3 + f] = dv;      // padded plane: operand of conv3 dgrad
    a.d3[(int64_t)m * NIN4 + n] = dv;                             // dense [(n,pix)][f]: operand of conv3 wgrad
  }
};

struct Fc4Wgrad {   // gW4 = delta4 . a3^T (sum over batch, A8) in the W4i layout; no split (K = B)
  static constexpr bool A_K = false, B_K = false;     // operand contiguous along k (-> LDS transpose) or along m/n
  typedef int aoff_t;
  // operand descriptors for the engine's fast paths: *_REG = plain row-major [k][x] matrix with row pitch *_LD
  static constexpr bool A_REG = true, A_U8 = false, B_REG = true; static constexpr int A_LD = NIN4, B_LD = NFC;
  SDQN_HD static const float* a_ptr(const StepArgs& a, int z) { (void)z; return a.a3; }
  SDQN_HD static const float* b_ptr(const StepArgs& a, int z) { (void)z; return a.d4; }
  SDQN_HD static int M(const StepArgs&) { return NIN4; }
  SDQN_HD static int N(const StepArgs&) { return NFC; }
  SDQN_HD static int nbz(const StepArgs&) { return 1; }
  SDQN_HD static void ksplit(const StepArgs& a, int, int& z, int& ks, int& kb, int& ke) { z = 0; ks = 0; kb = 0; ke = a.B; }
  SDQN_HD static aoff_t a_row(const StepArgs&, int, int m) { return m; }
  SDQN_HD static aoff_t a_col(const StepArgs&, int, int k) { return k * NIN4; }
  SDQN_HD static float a_load(const StepArgs& a, int, aoff_t o) { return a.a3[o]; }
  SDQN_HD static f4 a_load4(const StepArgs& a, int, aoff_t o) { return ld4(a.a3 + o); }
  SDQN_HD static int b_row(const StepArgs&, int, int k) { return k * NFC; }
  SDQN_HD static int b_col(const StepArgs&, int, int n) { return n; }
  SDQN_HD static float b_load(const StepArgs& a, int, int o) { return a.d4[o]; }
  SDQN_HD static f4 b_load4(const StepArgs& a, int, int o) { return ld4(a.d4 + o); }
  SDQN_HD static void store(const StepArgs& a, int, int, int m, int n, float v) {
    const int64_t e = OFF4 + (int64_t)m * NFC + n;
    if (a.fuse_rms) { float st = a.state[e]; a.theta_w[e] = rms_step(a.theta_w[e], st, v, a.bsz, a.rho, a.one_minus_rho, a.lr, a.eps); a.state[e] = st; }
    else a.g[e] = v;
  }
#if defined(__HIPCC__)
  // single-wave epilogue (B <= 32): the read-modify-write's 32 loads (theta, state) are issued at kernel
  // entry (epi_begin) so they fly under the operand loads and the MFMAs
  struct Epi { float w[16], st[16]; };
  // accumulator register r of lane l is element (m0 + (r&3) + 8(r>>2) + 4(l>>5), n0 + (l&31)): one 32-bit base offset
  // per lane + compile-time row offsets (uniform base pointer + 32-bit offset addressing, no 64-bit math per element)
  // BYTE offsets in 32 bits added to the uniform base pointer: the form hipcc turns into `global_load v, v_off32, s[base]`
  __device__ static uint32_t epi_base(int m0, int n0, int lane) { return 4u * (uint32_t)(OFF4 + (m0 + 4 * (lane >> 5)) * NFC + n0 + (lane & 31)); }
  __device__ static constexpr uint32_t epi_row(int r) { return 4u * (uint32_t)(((r & 3) + 8 * (r >> 2)) * NFC); }
  __device__ static float ldb(const float* p, uint32_t byte_off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + byte_off); }
  __device__ static void stb(float* p, uint32_t byte_off, float v) { *reinterpret_cast<float*>(reinterpret_cast<char*>(p) + byte_off) = v; }
  __device__ static void epi_begin(const StepArgs& a, int m0, int n0, int lane, Epi& e) {
    if (!a.fuse_rms) return;
    const float* __restrict__ tw = a.theta_w; const float* __restrict__ sp = a.state;
    const uint32_t base = epi_base(m0, n0, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) { e.w[r] = ldb(tw, base + epi_row(r)); e.st[r] = ldb(sp, base + epi_row(r)); }
  }
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi& e) {
    const uint32_t base = epi_base(m0, n0, lane);
    if (!a.fuse_rms) {
      float* __restrict__ gp = a.g;
#pragma unroll
      for (int r = 0; r < 16; ++r) stb(gp, base + epi_row(r), v[r]);
      return;
    }
#pragma unroll
    for (int r = 0; r < 16; r += 2) rms_step2(e.w[r], e.w[r + 1], e.st[r], e.st[r + 1], v[r], v[r + 1], a.bsz, a.rho, a.one_minus_rho, a.lr, a.eps);
    float* __restrict__ tw = a.theta_w; float* __restrict__ sp = a.state;
#pragma unroll
    for (int r = 0; r < 16; ++r) { stb(tw, base + epi_row(r), e.w[r]); stb(sp, base + epi_row(r), e.st[r]); }
  }
#endif
};

#if defined(__HIPCC__)
// fc4_dgrad + fc4_wgrad in ONE launch (round 3).  At B <= 32 the dgrad has a single M tile, so W4i rows [32 j, 32 j + 32)
// are read by exactly ONE workgroup (dgrad tile j) — and rewritten in place by the 16 fc4_wgrad tiles (j, 0..15) when RMSProp
// is fused into their epilogue.  Tile j publishes the end of its reads, the writers wait for it: the 25.7 MB read-modify-write
// stream of W4 + its optimizer state then runs on the 158 CUs the 98 dgrad workgroups leave idle instead of lengthening bwd3.
struct Fc4DgradSig : Fc4Dgrad {
  static constexpr bool SIGNALS = true;
  // called by thread 0 after the workgroup barrier that follows every wave's main loop (all operand loads have returned)
  __device__ static void signal(const StepArgs& a, int bx, int by, int bz) {
    (void)bx; (void)bz;
    __hip_atomic_store(a.f4d_flags + by * 16, a.f4d_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};
struct Fc4WgradWait : Fc4Wgrad {
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi& e) {
    if (a.fuse_rms) {                                  // (materialised gradient: written to g, nothing to wait for)
      const unsigned* f = a.f4d_flags + (m0 >> 5) * 16;
      int spins = 0;                                   // wave-uniform loop: every lane reads the same word
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.f4d_epoch) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > 4000000) {                       // ~ seconds: the producer tile never ran -> report, never hang
          __hip_atomic_store(a.f4d_flags + (NIN4 / 32) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    Fc4Wgrad::store16(a, z, ks, m0, n0, lane, M, N, v, e);
  }
};
#endif

struct Conv3Dgrad { // delta2 = full-correlation(d3p, W3) * 1[a2 > 0], written into the padded d2p
  static constexpr bool A_K = true, B_K = true;     // operand contiguous along k (-> LDS transpose) or along m/n
  typedef int aoff_t;
  // operand descriptors for the engine's fast paths: *_REG = plain row-major [k][x] matrix with row pitch *_LD
  static constexpr bool A_REG = false, A_U8 = false, B_REG = false; static constexpr int A_LD = 0, B_LD = 0;
  SDQN_HD static const float* a_ptr(const StepArgs& a, int z) { (void)z; return a.d3p; }
  SDQN_HD static const float* b_ptr(const StepArgs& a, int z) { (void)z; return a.theta[0] + OFF3; }
#if defined(__HIPCC__)
  struct Epi {};
  __device__ static void epi_begin(const StepArgs&, int, int, int, Epi&) {}
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi&);
#endif
  SDQN_HD static int M(const StepArgs& a) { return a.B * PIX2; }
  SDQN_HD static int N(const StepArgs&) { return K2; }
  SDQN_HD static int nbz(const StepArgs&) { return 1; }
  SDQN_HD static void ksplit(const StepArgs&, int, int& z, int& ks, int& kb, int& ke) { z = 0; ks = 0; kb = 0; ke = CRS3; }
  SDQN_HD static aoff_t a_row(const StepArgs&, int, int m) {
    int n = m / PIX2, pix = m - n * PIX2, y = pix / Q2, x = pix - y * Q2;
    return ((n * PD3 + y + 2) * PD3 + x + 2) * K3;
  }
  SDQN_HD static aoff_t a_col(const StepArgs&, int, int k) {
    int rs = k >> 6, f = k & 63, r = rs / 3, s = rs - r * 3;
    return -(r * PD3 + s) * K3 + f;
  }
  SDQN_HD static float a_load(const StepArgs& a, int, aoff_t o) { return a.d3p[o]; }
  SDQN_HD static f4 a_load4(const StepArgs& a, int, aoff_t o) { return ld4(a.d3p + o); }
  SDQN_HD static int b_row(const StepArgs&, int, int k) { return (k >> 6) * (K2 * K3) + (k & 63); }
  SDQN_HD static int b_col(const StepArgs&, int, int c) { return c * K3; }
  SDQN_HD static float b_load(const StepArgs& a, int, int o) { return a.theta[0][OFF3 + o]; }
  SDQN_HD static f4 b_load4(const StepArgs& a, int, int o) { return ld4(a.theta[0] + OFF3 + o); }
  SDQN_HD static void store(const StepArgs& a, int, int, int m, int c, float v) {
    bool on = a.a2[(int64_t)m * K2 + c] > 0.0f;
    const float dv = on ? v : 0.0f;
    a.d2p[prow2(m) + c] = dv;                                     // padded plane: operand of conv2 dgrad
    a.d2[(int64_t)m * K2 + c] = dv;                               // dense: operand of conv2 wgrad
  }
};

struct Conv3Wgrad { // gW3[(r,s,c)][f] = sum_(n,p,q) a2 patch * delta3   (Neon update_conv), split-K slabs
  static constexpr bool A_K = false, B_K = false;     // operand contiguous along k (-> LDS transpose) or along m/n
  typedef int aoff_t;
  // operand descriptors for the engine's fast paths: *_REG = plain row-major [k][x] matrix with row pitch *_LD
  static constexpr bool A_REG = false, A_U8 = false, B_REG = true; static constexpr int A_LD = 0, B_LD = K3;
  SDQN_HD static const float* a_ptr(const StepArgs& a, int z) { (void)z; return a.a2; }
  SDQN_HD static const float* b_ptr(const StepArgs& a, int z) { (void)z; return a.d3; }
#if defined(__HIPCC__)
  struct Epi {};
  __device__ static void epi_begin(const StepArgs&, int, int, int, Epi&) {}
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi&);
#endif
  SDQN_HD static int Kt(const StepArgs& a) { return a.B * PIX3; }
  SDQN_HD static int M(const StepArgs&) { return CRS3; }
  SDQN_HD static int N(const StepArgs&) { return K3; }
  SDQN_HD static int nbz(const StepArgs& a) { return ceil_div(ceil_div(Kt(a), 32), a.tps3); }
  SDQN_HD static void ksplit(const StepArgs& a, int bz, int& z, int& ks, int& kb, int& ke) {
    z = 0; ks = bz; kb = bz * a.tps3 * 32; ke = kb + a.tps3 * 32; if (ke > Kt(a)) ke = Kt(a);
  }
  SDQN_HD static aoff_t a_row(const StepArgs&, int, int m) { return col3(m); }
  SDQN_HD static aoff_t a_col(const StepArgs& a, int, int k) { return row3(a, 0, k); }
  SDQN_HD static float a_load(const StepArgs& a, int, aoff_t o) { return a.a2[o]; }
  SDQN_HD static f4 a_load4(const StepArgs& a, int, aoff_t o) { return ld4(a.a2 + o); }
  SDQN_HD static int b_row(const StepArgs&, int, int k) { return k * K3; }
  SDQN_HD static int b_col(const StepArgs&, int, int n) { return n; }
  SDQN_HD static float b_load(const StepArgs& a, int, int o) { return a.d3[o]; }
  SDQN_HD static f4 b_load4(const StepArgs& a, int, int o) { return ld4(a.d3 + o); }
  SDQN_HD static void store(const StepArgs& a, int, int ks, int m, int n, float v) { a.slab3[(int64_t)ks * NW3 + m * K3 + n] = v; }
};

struct Conv2Dgrad { // stride-2 dgrad as 4 parity classes (z = py*2+px), each a dense 2x2 correlation over d2p
  static constexpr bool A_K = true, B_K = true;     // operand contiguous along k (-> LDS transpose) or along m/n
  typedef int aoff_t;
  // operand descriptors for the engine's fast paths: *_REG = plain row-major [k][x] matrix with row pitch *_LD
  static constexpr bool A_REG = false, A_U8 = false, B_REG = false; static constexpr int A_LD = 0, B_LD = 0;
  SDQN_HD static const float* a_ptr(const StepArgs& a, int z) { (void)z; return a.d2p; }
  SDQN_HD static const float* b_ptr(const StepArgs& a, int z) { (void)z; return a.theta[0] + OFF2; }
#if defined(__HIPCC__)
  struct Epi {};
  __device__ static void epi_begin(const StepArgs&, int, int, int, Epi&) {}
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi&);
#endif
  SDQN_HD static int M(const StepArgs& a) { return a.B * 100; }
  SDQN_HD static int N(const StepArgs&) { return K1; }
  SDQN_HD static int nbz(const StepArgs&) { return 4; }
  SDQN_HD static void ksplit(const StepArgs&, int bz, int& z, int& ks, int& kb, int& ke) { z = bz; ks = 0; kb = 0; ke = 256; }
  SDQN_HD static aoff_t a_row(const StepArgs&, int, int m) {
    int n = m / 100, pix = m - n * 100, i = pix / 10, j = pix - i * 10;
    return ((n * PD2 + i + 1) * PD2 + j + 1) * K2;
  }
  SDQN_HD static aoff_t a_col(const StepArgs&, int, int k) {
    int ab = k >> 6, f = k & 63, aa = ab >> 1, bb = ab & 1;
    return -(aa * PD2 + bb) * K2 + f;
  }
  SDQN_HD static float a_load(const StepArgs& a, int, aoff_t o) { return a.d2p[o]; }
  SDQN_HD static f4 a_load4(const StepArgs& a, int, aoff_t o) { return ld4(a.d2p + o); }
  SDQN_HD static int b_row(const StepArgs&, int z, int k) {
    int py = z >> 1, px = z & 1, ab = k >> 6, f = k & 63, aa = ab >> 1, bb = ab & 1;
    return ((py + 2 * aa) * 4 + (px + 2 * bb)) * (K1 * K2) + f;
  }
  SDQN_HD static int b_col(const StepArgs&, int, int c) { return c * K2; }
  SDQN_HD static float b_load(const StepArgs& a, int, int o) { return a.theta[0][OFF2 + o]; }
  SDQN_HD static f4 b_load4(const StepArgs& a, int, int o) { return ld4(a.theta[0] + OFF2 + o); }
  SDQN_HD static void store(const StepArgs& a, int z, int, int m, int c, float v) {
    int py = z >> 1, px = z & 1;
    int n = m / 100, pix = m - n * 100, i = pix / 10, j = pix - i * 10;
    int o = ((n * P1 + 2 * i + py) * Q1 + 2 * j + px) * K1 + c;
    a.d1[o] = a.a1[o] > 0.0f ? v : 0.0f;
  }
};

struct Conv2Wgrad {
  static constexpr bool A_K = false, B_K = false;     // operand contiguous along k (-> LDS transpose) or along m/n
  typedef int aoff_t;
  // operand descriptors for the engine's fast paths: *_REG = plain row-major [k][x] matrix with row pitch *_LD
  static constexpr bool A_REG = false, A_U8 = false, B_REG = true; static constexpr int A_LD = 0, B_LD = K2;
  SDQN_HD static const float* a_ptr(const StepArgs& a, int z) { (void)z; return a.a1; }
  SDQN_HD static const float* b_ptr(const StepArgs& a, int z) { (void)z; return a.d2; }
#if defined(__HIPCC__)
  struct Epi {};
  __device__ static void epi_begin(const StepArgs&, int, int, int, Epi&) {}
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi&);
#endif
  SDQN_HD static int Kt(const StepArgs& a) { return a.B * PIX2; }
  SDQN_HD static int M(const StepArgs&) { return CRS2; }
  SDQN_HD static int N(const StepArgs&) { return K2; }
  SDQN_HD static int nbz(const StepArgs& a) { return ceil_div(ceil_div(Kt(a), 32), a.tps2); }
  SDQN_HD static void ksplit(const StepArgs& a, int bz, int& z, int& ks, int& kb, int& ke) {
    z = 0; ks = bz; kb = bz * a.tps2 * 32; ke = kb + a.tps2 * 32; if (ke > Kt(a)) ke = Kt(a);
  }
  SDQN_HD static aoff_t a_row(const StepArgs&, int, int m) { return col2(m); }
  SDQN_HD static aoff_t a_col(const StepArgs& a, int, int k) { return row2(a, 0, k); }
  SDQN_HD static float a_load(const StepArgs& a, int, aoff_t o) { return a.a1[o]; }
  SDQN_HD static f4 a_load4(const StepArgs& a, int, aoff_t o) { return ld4(a.a1 + o); }
  SDQN_HD static int b_row(const StepArgs&, int, int k) { return k * K2; }
  SDQN_HD static int b_col(const StepArgs&, int, int n) { return n; }
  SDQN_HD static float b_load(const StepArgs& a, int, int o) { return a.d2[o]; }
  SDQN_HD static f4 b_load4(const StepArgs& a, int, int o) { return ld4(a.d2 + o); }
  SDQN_HD static void store(const StepArgs& a, int, int ks, int m, int n, float v) { a.slab2[(int64_t)ks * NW2 + m * K2 + n] = v; }
};

struct Conv1Wgrad { // re-gathers the normalised u8 patches from the ring (no fp32 input copy is ever stored)
  static constexpr bool A_K = false, B_K = false;     // operand contiguous along k (-> LDS transpose) or along m/n
  typedef int64_t aoff_t;
  // operand descriptors for the engine's fast paths: *_REG = plain row-major [k][x] matrix with row pitch *_LD
  static constexpr bool A_REG = false, A_U8 = true, B_REG = true; static constexpr int A_LD = 0, B_LD = K1;
  SDQN_HD static const float* a_ptr(const StepArgs& a, int z) { (void)z; return (const float*)nullptr; }
  SDQN_HD static const float* b_ptr(const StepArgs& a, int z) { (void)z; return a.d1; }
#if defined(__HIPCC__)
  struct Epi {};
  __device__ static void epi_begin(const StepArgs&, int, int, int, Epi&) {}
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi&);
#endif
  SDQN_HD static int Kt(const StepArgs& a) { return a.B * PIX1; }
  SDQN_HD static int M(const StepArgs&) { return CRS1; }
  SDQN_HD static int N(const StepArgs&) { return K1; }
  SDQN_HD static int nbz(const StepArgs& a) { return ceil_div(ceil_div(Kt(a), 32), a.tps1); }
  SDQN_HD static void ksplit(const StepArgs& a, int bz, int& z, int& ks, int& kb, int& ke) {
    z = 0; ks = bz; kb = bz * a.tps1 * 32; ke = kb + a.tps1 * 32; if (ke > Kt(a)) ke = Kt(a);
  }
  SDQN_HD static aoff_t a_row(const StepArgs&, int, int m) { return col1(m); }
  SDQN_HD static aoff_t a_col(const StepArgs& a, int, int k) { return row1(a, 0, k); }
  SDQN_HD static float a_load(const StepArgs& a, int, aoff_t o) { return norm_u8(a.src[o]); }
  SDQN_HD static f4 a_load4(const StepArgs& a, int, aoff_t o) { return ld4_u8(a.src + o); }
  static constexpr bool A_GROUP4 = true;              // k = (n, y, x), x fastest: +4 bytes per k inside aligned groups of 4
  SDQN_HD static f4 a_load_group4(const StepArgs& a, int, aoff_t o) { return ld4_u8_stride4(a.src + o); }
  SDQN_HD static int b_row(const StepArgs&, int, int k) { return k * K1; }
  SDQN_HD static int b_col(const StepArgs&, int, int n) { return n; }
  SDQN_HD static float b_load(const StepArgs& a, int, int o) { return a.d1[o]; }
  SDQN_HD static f4 b_load4(const StepArgs& a, int, int o) { return ld4(a.d1 + o); }
  SDQN_HD static void store(const StepArgs& a, int, int ks, int m, int n, float v) { a.slab1[(int64_t)ks * NW1 + m * K1 + n] = v; }
};

#if defined(__HIPCC__)
#define SDQN_STORE16_DEFAULT(P) __device__ inline void P::store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, P::Epi&) { \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) { const int ml = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), nl = lane & 31; \
    if (m0 + ml < M && n0 + nl < N) P::store(a, z, ks, m0 + ml, n0 + nl, v[r]); } }
SDQN_STORE16_DEFAULT(Conv1Fwd)
SDQN_STORE16_DEFAULT(Conv2Fwd)
SDQN_STORE16_DEFAULT(Conv3Fwd)
SDQN_STORE16_DEFAULT(Fc4Fwd)
SDQN_STORE16_DEFAULT(Fc4Dgrad)
SDQN_STORE16_DEFAULT(Conv3Dgrad)
SDQN_STORE16_DEFAULT(Conv3Wgrad)
SDQN_STORE16_DEFAULT(Conv2Dgrad)
SDQN_STORE16_DEFAULT(Conv2Wgrad)
SDQN_STORE16_DEFAULT(Conv1Wgrad)
#endif

// --batch_norm forward: the same three conv problems storing the raw linear output x_l (no Rectlin) into a.a1/a2/a3,
// which the orchestration points at the x buffers for that launch
struct Conv1FwdRaw : Conv1Fwd {
  SDQN_HD static void store(const StepArgs& a, int z, int, int m, int n, float v) { a.a1[((int64_t)z * M(a) + m) * K1 + n] = v; }
#if defined(__HIPCC__)
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi&);
#endif
};
struct Conv2FwdRaw : Conv2Fwd {
  SDQN_HD static void store(const StepArgs& a, int z, int, int m, int n, float v) { a.a2[((int64_t)z * M(a) + m) * K2 + n] = v; }
#if defined(__HIPCC__)
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi&);
#endif
};
struct Conv3FwdRaw : Conv3Fwd {
  SDQN_HD static void store(const StepArgs& a, int z, int, int m, int n, float v) { a.a3[((int64_t)z * M(a) + m) * K3 + n] = v; }
#if defined(__HIPCC__)
  __device__ static void store16(const StepArgs& a, int z, int ks, int m0, int n0, int lane, int M, int N, const float* v, Epi&);
#endif
};
#if defined(__HIPCC__)
SDQN_STORE16_DEFAULT(Conv1FwdRaw)
SDQN_STORE16_DEFAULT(Conv2FwdRaw)
SDQN_STORE16_DEFAULT(Conv3FwdRaw)
#endif

// ---- Neon <-> internal parameter layouts (host side; C ABI boundary) ---------------------------
// returns the internal flat index (relative to the layer's OFFx) of Neon element (row, col)
inline int64_t neon_to_internal(int layer, int64_t row, int64_t col) {
  switch (layer) {
    case 0: return row * K1 + col;                                               // identical
    case 1: { int c = (int)(row / 16), rs = (int)(row % 16); return ((int64_t)rs * 32 + c) * K2 + col; }
    case 2: { int c = (int)(row / 9), rs = (int)(row % 9); return ((int64_t)rs * 64 + c) * K3 + col; }
    case 3: { int f = (int)(col / PIX3), pix = (int)(col % PIX3); return ((int64_t)pix * K3 + f) * NFC + row; }
    default: return row * NFC + col;                                             // fc5 identical
  }
}
inline void layer_dims(int layer, int A, int64_t& rows, int64_t& cols, int64_t& off) {
  switch (layer) {
    case 0: rows = CRS1; cols = K1; off = OFF1; break;
    case 1: rows = CRS2; cols = K2; off = OFF2; break;
    case 2: rows = CRS3; cols = K3; off = OFF3; break;
    case 3: rows = NFC; cols = NIN4; off = OFF4; break;
    default: rows = A; cols = NFC; off = OFF5; break;
  }
}

}  // namespace sdqn
