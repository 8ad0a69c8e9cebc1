// kernels.h — launch interface between the host orchestration (sdqn_api_*.hip) and the device
// code (sdqn_kernels.hip).  Kernel ids double as the profiler's slots.
#pragma once
#include <hip/hip_runtime.h>
#include "problems.h"
#include "launch.h"

namespace sdqn {

enum KernelId {
  K_CONV1_FWD = 0, K_CONV2_FWD, K_CONV3_FWD, K_FC4_FWD, K_HEAD,
  K_FC4_DGRAD, K_FC4_WGRAD, K_CONV3_DGRAD, K_CONV3_WGRAD, K_CONV2_DGRAD, K_CONV2_WGRAD,
  K_CONV1_WGRAD, K_UPDATE, K_ALLREDUCE, K_GATHER, K_PREP,
  K_BWD3,      // one launch: conv3_dgrad + conv3_wgrad + fc4_wgrad (all depend on fc4_dgrad only)
  K_BWD2,      // one launch: conv2_dgrad + conv2_wgrad (both depend on conv3_dgrad only) + a share of fc4_wgrad
  K_BWD1,      // one launch: conv1_wgrad + the last share of fc4_wgrad
  K_BN,        // --batch_norm: one BatchNorm layer, forward ([partial +] apply) or backward (partial + apply)
  K_F4D_F4W,   // round 3, one launch: fc4_dgrad + fc4_wgrad (+ fused RMSProp of W4) — the wgrad only needs delta4 and a3
  K_BWD3_CONV, // round 3: bwd3 without the fc4 share (conv3_dgrad + conv3_wgrad)
  K_UPD_CONV1, // round 3: update(i) + conv1_fwd(i + 1) in one launch
  K_HEAD_F4D,  // round 3: head + fc4_dgrad in one launch (the dgrad tiles fetch their W4 panels while the head runs)
  K_WGRADS,    // round 4 (float16, B >= 128): fc4_wgrad (+ fused RMSProp) || conv3_wgrad || conv2_wgrad in one launch, after the block-tile dgrad chain
  K_ACT,       // round 4: the acting forward (batch of one) as ONE launch (sdqn_act.hip)
  K_COUNT
};
const char* kernel_name(int id);


struct HeadArgs {
  const uint8_t* st_actions;    // minibatch metadata (staged from the host, or gathered by prep_kernel)
  const int64_t* st_rewards;
  const uint8_t* st_terminals;
  float* q;                     // [2][B][A] q-values of both nets
  float* maxq;                  // [B]
  float* dq;                    // [B][A] clipped deltas
  float* cost_terms;            // [B]  0.5 * delta^2 (pre-clip)
  double discount, min_reward, max_reward;
  float clip_error;
  int train;                    // 0: predict only (z = 0)
  // hoist: one extra workgroup copies the NEXT step's sampled indexes from their pinned slot into device memory, so that
  // the target conv1 riding in this step's K_BWD2 launch reads them from HBM (next_B = 0: nothing to copy)
  const int64_t* next_idx_pinned; int64_t* next_idx_dev; int next_B;
};

struct PrepArgs {                // pinned index slot + ring metadata -> device-resident (idx, a, r, t) of this step
  const int64_t* idx_pinned;    // [B] zero-copy view of the pinned slot
  const MetaRec* meta;          // ring metadata mirror
  int64_t* idx;                 // [B] device
  uint8_t* actions;             // [B] device staging shared with the host-minibatch path
  int64_t* rewards;
  uint8_t* terminals;
  int B;
  // B <= 32: the indexes themselves (host data at launch time) ride in the kernel arguments, so the prep block of the update launch
  // starts with a load from the argument segment instead of a zero-copy read of pinned host memory over PCIe (~2 us: it was the longest
  // dependency chain of the whole update launch)
  int idx_in_valid;
  int64_t idx_in[32];
};

struct UpdateArgs {
  float* theta;                 // online parameters (flat, internal layout)
  float* state;                 // RMSProp state
  float* g;                     // flat gradient sum
  const float* slab[3];
  int ns[3];
  const float* dq;              // [B][A]
  const float* a4;              // online a4 [B][512]
  const float* cost_terms;
  float* cost_out;              // [1]
  double* cost_accum;           // running sum over steps
  int B, A;
  int mode;                     // 0 fused reduce+apply, 1 reduce only (-> g), 2 apply only (g already reduced)
  int skip_fc4;                 // fc4 already updated inside fc4_wgrad's epilogue (StepArgs::fuse_rms), or by the only_fc4 launch
  int only_fc4;                 // overlapped data parallel: this launch (on the comm stream) applies the fc4 range only
  PrepArgs next;                // next.B > 0: also do the NEXT step's prep (train_many samples one step ahead)
  float bsz;                    // divisor of A9 (B, or R*B under data parallel)
  float rho, one_minus_rho, lr, eps;
  int opt;                      // 0 RMSProp, 1 Adam, 2 Adadelta (deepqnetwork.py:50-59)
  float* state2;                // Adam v / Adadelta E[dx^2]
  float beta1, one_minus_beta1, beta2, one_minus_beta2, lr_t;   // Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t), t = epoch+1
  half_t* wh;                   // fp16 mode: half copies of theta refreshed by the update (master layout / transposed)
  half_t* wht;
  unsigned short* w1p;          // conv1's three bf16 planes of the ONLINE net, rewritten with W1 (nullptr: not maintained)
  unsigned short* wpm;          // round 4: bf16 planes of conv2 / conv3 / fc4 weights, master layout (nullptr: not maintained — B < 128)
  unsigned short* wpt;          //          ... conv2 / conv3 transposed ([n][K])
  int wt;                       // 1: the new parameters / optimizer state leave with write-through (sc1) stores
  unsigned* w1_ctr;             // fused update + conv1 launch only: counts the W1 blocks whose write-through stores are out (monotonic across launches)
  int64_t bn_first;             // --batch_norm: element offset of the [beta|gamma] block (bn_update_kernel); BN_PARAMS elements
  const int* ovf_flag;          // fp16 data parallel (update_kernel<true>): != 0 -> the all-reduced half gradient overflowed, leave theta / state untouched
  int64_t* ovf_count;           //   ... and count the skipped step
  int ovf_dynamic;              //   1: block 0 also moves the payload scale (state[1]) — halve on overflow, double after 200 clean steps
};

struct BnArgs {                  // one BatchNorm layer (bn_kernels.hip); activations NHWC: rows x C, C contiguous
  int layer;                    // 0..3 = after conv1, conv2, conv3, fc4
  int C, rows;                  // features; rows per net (B*P*Q, or B for fc4)
  int nz;                       // nets in this forward pass (2 = online + target, 1 = predict)
  int train;                    // 1: z = 0 normalises with batch statistics and updates the running ones
  const float* x;               // raw linear output [nz][rows][C]; fc4: the split-K slabs [S4][2][B][512]
  int S4, B;                    // fc4 only (S4 > 0)
  float* a;                     // activated output [nz][rows][C]
  float* theta[2];              // flat parameter buffers (BatchNorm block at off_bn)
  int64_t off_bn;
  double* partial;              // [row blocks][C][2]
  float* mean; float* rstd;     // [C] batch statistics of z = 0 (kept for the backward pass)
  float* d;                     // backward: dense delta [rows][C], transformed in place
  float* dpad;                  // optional zero-padded copy [n][PD][PD][C] (operand of the next dgrad)
  int PQ, Qw, PD, pad;
  float* g;                     // flat gradient buffer
};

struct GatherArgs {
  const uint8_t* ring;          // [size][84][84]
  const MetaRec* meta;
  const int64_t* idx;           // [B]
  uint8_t* pre;                 // [B][4][84][84]
  uint8_t* post;
  uint8_t* actions;             // [B]
  int64_t* rewards;
  uint8_t* terminals;
  int B;
};

#if defined(__HIPCC__)
// ------------------------------------------------------------------------------------------------
// one parameter of Neon's optimizers [neon-recalled, SURVEY.md A9/A10 + §8a-bis "non-default branches"];
// every one starts with grad = grad / be.bsz
__device__ inline float opt_apply(float w, float& s1, float& s2, float gsum, const UpdateArgs& u) {
  if (u.opt == 0) return rms_step(w, s1, gsum, u.bsz, u.rho, u.one_minus_rho, u.lr, u.eps);
  const float g = div_bsz(gsum, u.bsz);
  if (u.opt == 1) {                                   // Adam: m, v; bias correction folded into lr_t (t = epoch + 1)
    s1 = s1 * u.beta1 + u.one_minus_beta1 * g;
    s2 = s2 * u.beta2 + (u.one_minus_beta2 * g) * g;
    return w - (u.lr_t * s1) / (sqrtf(s2) + u.eps);
  }
  s1 = s1 * u.rho + (u.one_minus_rho * g) * g;        // Adadelta: E[g^2], E[dx^2]
  const float upd = sqrtf((s2 + u.eps) / (s1 + u.eps)) * g;
  s2 = s2 * u.rho + (u.one_minus_rho * upd) * upd;
  return w - upd;
}
#endif

// host-side launch choices that never reach a kernel (kept out of StepArgs: kernel-argument bytes are not free)
struct LaunchTune {
  int nw_override[12];      // tuning hook: waves per tile for kernel id i (0 = built-in choice)
  const int64_t* host_idx;  // ring paths, B <= 32: this step's sampled indexes in HOST memory (they ride in the kernel arguments of conv1_bf16_kernel)
  int r3_xcd;               // round-3 kernels' XCD-contiguous tile maps: bit 0 conv1_fwd (bf16), bit 1 conv1_wgrad (bf16)
  int wt;                   // write-through (sc1) epilogue stores per launch: 1 conv2_fwd, 2 conv3_fwd, 4 fc4_fwd, 8 fc4_dgrad, 16 bwd3, 32 bwd2, 64 conv1_wgrad, 128 conv1_fwd
  int bt[K_COUNT];          // B >= 128, float32: block-tile engine (sdqn_kernels_bt.hip) per kernel id; 0 = built-in block shape, n > 0 = menu entry, < 0 = latency engine
  int r3;                   // round-3 launch variants (sdqn_kernels_r3.hip); bit 0: this K_FC4_DGRAD launch also carries the fc4_wgrad tiles; bit 1: conv3_fwd on 36-deep K-chunks; bit 2: conv1_fwd on packed-bf16 MFMA
};
hipError_t launch_kernel(int id, const StepArgs& a, const LaunchTune& t, hipStream_t s);     // the GEMM-shaped stages (single or multi-problem launches)
hipError_t launch_head(const StepArgs& a, const HeadArgs& h, hipStream_t s, bool q_system_scope = false);   // q_system_scope: h.q is mapped host memory (acting path)
hipError_t launch_update(const UpdateArgs& u, hipStream_t s);
hipError_t launch_gather(const GatherArgs& g, hipStream_t s, const int64_t* host_idx = nullptr);   // host_idx (B <= 256): indexes inside the kernel arguments
hipError_t launch_bn_forward(const BnArgs& b, hipStream_t s);      // [partial +] apply
hipError_t launch_bn_backward(const BnArgs& b, hipStream_t s);     // partial + apply
hipError_t launch_bn_update(const UpdateArgs& u, hipStream_t s);   // optimizer step of the [beta | gamma] block (g already holds the sums)
hipError_t launch_prep(const PrepArgs& p, hipStream_t s, double* zero8 = nullptr);   // zero8: an 8-byte accumulator the launch also clears
hipError_t launch_grad_to_half(const float* g, half_t* gh, int64_t n, int* state, hipStream_t s);       // fp16 DP payload; state = {flag, log2 scale, good steps}
hipError_t launch_grad_from_half(const half_t* gh, float* g, int64_t n, int* state, hipStream_t s);
// ---- the acting forward as ONE launch (sdqn_act.hip): float32, standard geometry, no batch-norm ---------------------------------------
constexpr int ACT_GRID = 256;                     // workgroups of 256 threads (one per CU when the chip is idle; any placement is correct)
constexpr int ACT_XCC_FLOATS = 21248 + 8 * 32 * 64;  // one XCC's scratch: a1 [400][32] | a2 [81][64] | a3 [49][64] (+ pad) | fc4 partials [8 stripes][32 chunks][64]
constexpr int ACT_KCH = 32;                       // fc4 K-chunks per stripe of 64 hidden units
constexpr int ACT_Q_STRIDE = 32;                  // the launch delivers 8 stripe partials of the Q-vector: q[stripe * 32 + action]; the host adds them in stripe order
constexpr int ACT_CTL_WORDS = 832;                // control block of one launch; 4 rotate (a launch clears the one after next)
constexpr int ACT_STAMPS = 40;                    // (timing build of the kernel: {kind, clock64} pairs per workgroup)
struct ActArgs {
  const uint8_t* state;       // [4][84*84] bytes, oldest frame first (the device state buffer's window)
  const float* theta;         // online parameters
  float* scratch;             // [8][ACT_XCC_FLOATS]
  unsigned* ctl;              // [4][ACT_CTL_WORDS], zero before the first launch
  float* q;                   // [8][ACT_Q_STRIDE] destination: stripe partials
  unsigned long long* stamps; // nullptr, or [ACT_GRID][2 * ACT_STAMPS] (tools/exp/act_stamps.py)
  int A;
  unsigned seq;               // launch number (selects the control block / partial slot)
};
hipError_t launch_act(const ActArgs& a, bool q_system_scope, hipStream_t s);
hipError_t launch_w1_planes(const float* theta, unsigned short* w1p, hipStream_t s);   // conv1's three bf16 weight planes of one net (problems.h: split_bf16x3)
hipError_t launch_refresh16(const float* theta, half_t* wh, half_t* wht, hipStream_t s);   // fp16 mode: rebuild both half copies
hipError_t launch_refresh_planes(const float* theta, unsigned short* wpm, unsigned short* wpt, hipStream_t s);   // plane mode: rebuild the bf16 planes of conv2 / conv3 (both layouts) and fc4 (master; wpm may be nullptr: target net)

}  // namespace sdqn
