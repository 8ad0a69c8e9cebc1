// sdqn_api_step.hip — the train / predict step: launch orchestration (deepqnetwork.py:107-186) and the fused replay loops (agent.py:108-114)
#include "api_internal.h"

hipError_t dp_allreduce(sdqn_net_s* h, void* buf, size_t count, int dtype, void* comm, hipStream_t s) {
  h->nccl_rc = g_rccl.AllReduce(buf, buf, count, dtype, /*ncclSum*/ 0, comm, s);
  return h->nccl_rc == 0 ? hipSuccess : hipErrorUnknown;
}

// ---- the step ---------------------------------------------------------------------------------------------
StepArgs step_args(sdqn_net_s* h) {
  StepArgs a; memset(&a, 0, sizeof a);
  a.B = h->B; a.A = h->A; a.nz = 2; a.theta[0] = h->theta; a.theta[1] = h->theta_t;
  a.a1 = h->a1; a.a2 = h->a2; a.a3 = h->a3; a.slab4 = h->slab4; a.a4 = h->a4; a.d4 = h->d4; a.d3p = h->d3p; a.d2p = h->d2p; a.d3 = h->d3; a.d2 = h->d2;
  a.d1 = h->d1; a.g = h->g; a.slab1 = h->slab1; a.slab2 = h->slab2; a.slab3 = h->slab3;
  a.S4 = h->S4; a.tps1 = h->tps1; a.tps2 = h->tps2; a.tps3 = h->tps3;

  a.xcd_map = h->xcd_map ? 7 : 0;
  if (h->cfg.datatype == 1) {
    a.h16 = h->h16_wgrad_mfma ? 2 : 1; a.h_a1 = h->h_a1; a.h_a2 = h->h_a2; a.h_a3 = h->h_a3; a.h_d4 = h->h_d4; a.h_d3p = h->h_d3p; a.h_d3 = h->h_d3;
    a.h_d2p = h->h_d2p; a.h_d2 = h->h_d2; a.h_d1 = h->h_d1; a.wh[0] = h->wh[0]; a.wh[1] = h->wh[1]; a.wht[0] = h->wht[0]; a.wht[1] = h->wht[1];
    a.wh_w = h->wh[0]; a.wht_w = h->wht[0];
    a.loss_scale = (float)h->cfg.loss_scale; a.inv_loss_scale = (float)(1.0 / h->cfg.loss_scale);
  }
  a.f4w_first = 0; a.f4w_count = (NIN4 / 32) * (NFC / 32);
  a.w1p[0] = h->w1p[0]; a.w1p[1] = h->w1p[1];
  a.fuse_rms = (!h->comm && !h->keep_grads && !h->grad_only && h->cfg.optimizer == 0) ? 1 : 0;
  a.theta_w = h->theta; a.state = h->state; a.bsz = (float)h->B;
  a.rho = (float)h->cfg.decay_rate; a.one_minus_rho = (float)(1.0 - h->cfg.decay_rate);
  a.lr = (float)h->cfg.learning_rate; a.eps = (float)h->cfg.epsilon;
  return a;
}
HeadArgs head_args(sdqn_net_s* h, int train) {
  HeadArgs hd; memset(&hd, 0, sizeof hd);
  hd.st_actions = h->st_act; hd.st_rewards = h->st_rew; hd.st_terminals = h->st_term;
  hd.q = h->q; hd.maxq = h->maxq; hd.dq = h->dq; hd.cost_terms = h->cost_terms;
  hd.discount = h->cfg.discount_rate; hd.min_reward = h->cfg.min_reward; hd.max_reward = h->cfg.max_reward;
  hd.clip_error = (float)h->cfg.clip_error; hd.train = train;
  return hd;
}
// Overlapped data parallel: the previous step's fc4 all-reduce + update may still be running on g_comm.  Everything
// on the library stream that touches W4, its optimizer state or the fc4 gradient must come after it.
int join_comm(sdqn_net_s* h) {
  if (h->w4_pending) { HIPCHK(hipStreamWaitEvent(g_stream, h->ev_w4, 0)); h->w4_pending = false; }
  return SDQN_OK;
}
// --batch_norm: one BatchNorm layer's arguments (bn_kernels.hip)
BnArgs bn_args(sdqn_net_s* h, const StepArgs& a, int layer, int train) {
  BnArgs b; memset(&b, 0, sizeof b);
  const int pix[4] = {PIX1, PIX2, PIX3, 1};
  b.layer = layer; b.C = bn_features(layer); b.rows = a.B * pix[layer]; b.nz = a.nz; b.train = train; b.B = a.B;
  b.theta[0] = h->theta; b.theta[1] = h->theta_t; b.off_bn = h->NPW; b.partial = h->bn_partial;
  b.mean = h->bn_mean + bn_off(layer) / 2; b.rstd = h->bn_rstd + bn_off(layer) / 2; b.g = h->g;
  switch (layer) {
    case 0: b.x = h->x1; b.a = h->a1; b.d = h->d1; break;
    case 1: b.x = h->x2; b.a = h->a2; b.d = h->d2; b.dpad = h->d2p; b.PQ = PIX2; b.Qw = Q2; b.PD = PD2; b.pad = 1; break;
    case 2: b.x = h->x3; b.a = h->a3; b.d = h->d3; b.dpad = h->d3p; b.PQ = PIX3; b.Qw = Q3; b.PD = PD3; b.pad = 2; break;
    default: b.x = h->slab4; b.S4 = a.S4; b.a = h->a4; b.d = h->d4; break;
  }
  return b;
}
// tuning hook: per-launch XCD map mask override (sdqn_net_set_option "xcd:<id>", value = mask + 1; 0 = built-in)
#define XCD_TUNE(ARGS, KID) do { if (h->xcd_mask[KID] > 0) (ARGS).xcd_map = h->xcd_mask[KID] - 1; } while (0)
hipError_t launch_tuned(sdqn_net_s* h, int id, StepArgs a, hipStream_t s, int r3) {
  XCD_TUNE(a, id);
  LaunchTune t;
  for (int i = 0; i < 12; ++i) t.nw_override[i] = h->nw_override[i];
  for (int i = 0; i < K_COUNT; ++i) t.bt[i] = h->bt_on ? h->bt[i] : -1;
  t.r3 = r3; t.host_idx = h->host_idx_cur; t.r3_xcd = h->r3_xcd; t.wt = h->wt;
  return launch_kernel(id, a, t, s);
}
int run_forward(sdqn_net_s* h, const StepArgs& a, const HeadArgs& hd) {
  if (h->bn) {
    // deepqnetwork.py:83-89 with batch_norm: [Convolution|Linear] -> BatchNorm -> Rectlin.  The GEMM stage writes the raw
    // linear output (x_l), the BatchNorm pass turns it into the activation the next stage reads; training-mode
    // statistics for the online net of a train step (:129), running statistics for the target net (:120) and predict (:180)
    StepArgs f = a; f.bn = 1;
    f.a1 = h->x1; LAUNCH(K_CONV1_FWD, launch_tuned(h, K_CONV1_FWD, f, g_stream)); f.a1 = h->a1;
    LAUNCH(K_BN, launch_bn_forward(bn_args(h, a, 0, hd.train), g_stream));
    f.a2 = h->x2; LAUNCH(K_CONV2_FWD, launch_tuned(h, K_CONV2_FWD, f, g_stream)); f.a2 = h->a2;
    LAUNCH(K_BN, launch_bn_forward(bn_args(h, a, 1, hd.train), g_stream));
    f.a3 = h->x3; LAUNCH(K_CONV3_FWD, launch_tuned(h, K_CONV3_FWD, f, g_stream)); f.a3 = h->a3;
    LAUNCH(K_BN, launch_bn_forward(bn_args(h, a, 2, hd.train), g_stream));
    { int rc = join_comm(h); if (rc) return rc; }
    LAUNCH(K_FC4_FWD, launch_tuned(h, K_FC4_FWD, f, g_stream));
    LAUNCH(K_BN, launch_bn_forward(bn_args(h, a, 3, hd.train), g_stream));
    LAUNCH(K_HEAD, launch_head(f, hd, g_stream));
    return SDQN_OK;
  }
  // XCD-contiguous tile map where it wins time (tools/sweep_xcd.py, tools/ab_options.py): conv1_fwd +0.5 %, conv2_fwd
  // +0.2 %, fc4_fwd +0.6 % of the step rate; slower for conv3_fwd, fc4_dgrad and every backward launch
  StepArgs fm = a; fm.xcd_map = 1; fm.idx_t = nullptr;
  LAUNCH(K_CONV1_FWD, launch_tuned(h, K_CONV1_FWD, fm, g_stream, (h->conv1_bf16 && h->nw_override[K_CONV1_FWD] == 0) ? 4 : 0));
  { StepArgs f2 = fm; if (h->B >= 128) f2.xcd_map = a.xcd_map;          // block-tile routines (B >= 128): conv2_fwd 28.8 / 8.8 us round-robin, 28.9 / 9.0 on the map (fp32 / float16)
    LAUNCH(K_CONV2_FWD, launch_tuned(h, K_CONV2_FWD, f2, g_stream)); }
  { StepArgs f3 = fm; f3.xcd_map = a.xcd_map;
    const int c36 = (h->conv3_c36 && h->nw_override[K_CONV3_FWD] == 0) ? 2 : 0;
    LAUNCH(K_CONV3_FWD, launch_tuned(h, K_CONV3_FWD, f3, g_stream, c36)); }
  { int rc = join_comm(h); if (rc) return rc; }                // conv1..3 of this step overlap the previous step's fc4 all-reduce
  LAUNCH(K_FC4_FWD, launch_tuned(h, K_FC4_FWD, fm, g_stream));
  LAUNCH(K_HEAD, launch_head(a, hd, g_stream, h->head_q_system));
  return SDQN_OK;
}
UpdateArgs make_update_args(sdqn_net_s* h, const StepArgs& a) {
  UpdateArgs u; memset(&u, 0, sizeof u);
  u.theta = h->theta; u.state = h->state; u.g = h->g;
  u.slab[0] = h->slab1; u.slab[1] = h->slab2; u.slab[2] = h->slab3; u.ns[0] = h->ns1; u.ns[1] = h->ns2; u.ns[2] = h->ns3;
  u.dq = h->dq; u.a4 = h->a4; u.cost_terms = h->cost_terms; u.cost_out = h->cost_out; u.cost_accum = h->cost_accum;
  u.B = h->B; u.A = h->A;
  u.rho = (float)h->cfg.decay_rate; u.one_minus_rho = (float)(1.0 - h->cfg.decay_rate);
  u.lr = (float)h->cfg.learning_rate; u.eps = (float)h->cfg.epsilon;
  u.skip_fc4 = a.fuse_rms;
  u.opt = h->cfg.optimizer; u.state2 = h->state2;
  if (h->cfg.datatype == 1) { u.wh = h->wh[0]; u.wht = h->wht[0]; }
  u.w1p = h->w1p[0];
  u.wt = (h->wt >> 8) & 1;
  u.bn_first = h->bn ? h->NPW : 0;
  if (u.opt == 1) {            // Neon Adam [neon-recalled]: t = epoch + 1, l = lr*sqrt(1-b2^t)/(1-b1^t), math in Python floats
    const double b1 = h->cfg.beta_1, b2 = h->cfg.beta_2, t = (double)h->epoch + 1.0;
    u.beta1 = (float)b1; u.one_minus_beta1 = (float)(1.0 - b1); u.beta2 = (float)b2; u.one_minus_beta2 = (float)(1.0 - b2);
    u.lr_t = (float)(h->cfg.learning_rate * sqrt(1.0 - pow(b2, t)) / (1.0 - pow(b1, t)));
  }
  return u;
}
// One train step after the minibatch is in place (deepqnetwork.py:119-172): forward of both nets, head, backward, optimizer.
// Which launches run is a function of (batch regime, datatype, batch_norm, data-parallel form) only — `step_structure` below names it,
// DESIGN.md 12 tabulates it, tests/test_step_structure.py enumerates it:
//   fused      fc4_dgrad | bwd3 = conv3_dgrad || conv3_wgrad || fc4_wgrad (+ RMSProp of W4) | bwd2 = conv2_dgrad || conv2_wgrad | bwd1 = conv1_wgrad
//   h16_bt     float16, B >= 128: fc4_dgrad | conv3_dgrad | conv2_dgrad | wgrads = fc4_wgrad (+ RMSProp) || conv3_wgrad || conv2_wgrad | bwd1
//   dp_overlap fused, with ALL of fc4_wgrad in bwd3 and its all-reduce + update on the second communicator's stream
//   unfused    one launch per problem (option fused_launches = 0; batch_norm inserts its own passes between them)
StepStructure step_structure(const sdqn_net_s* h) {
  if (h->comm && h->comm2 && h->dp_overlap && h->fused_launches) return STEP_DP_OVERLAP;
  if (h->cfg.datatype == 1 && h->B >= 128 && h->bt_on && !h->bn && h->fused_launches && h->bt[K_CONV3_DGRAD] >= 0 && h->bt[K_CONV2_DGRAD] >= 0 &&
      h->nw_override[K_CONV3_DGRAD] == 0 && h->nw_override[K_CONV2_DGRAD] == 0) return STEP_H16_BT;
  return h->fused_launches ? STEP_FUSED : STEP_UNFUSED;
}
// how the optimizer pass runs: 0 one update launch (fc4's RMSProp rode in its weight-gradient tiles) | 1 serial data parallel: local sums,
// one all-reduce of the flat gradient, apply | 2 overlapped data parallel (fc4's part already on its way) | 3 grad_only (local sums, nothing applied)
UpdateForm update_form(const sdqn_net_s* h) {
  if (step_structure(h) == STEP_DP_OVERLAP) return UPD_DP_OVERLAP;
  if (h->comm) return UPD_DP_SERIAL;
  return h->grad_only ? UPD_GRAD_ONLY : UPD_SINGLE;
}
extern "C" int sdqn_net_step_structure(sdqn_net_t h, int* structure, int* update) {
  ARGCHK(h && structure && update, "NULL argument");
  if (h->gen) { *structure = 4; *update = 0; return SDQN_OK; }
  *structure = (int)step_structure(h); *update = (int)update_form(h);
  return SDQN_OK;
}
int run_train(sdqn_net_s* h, const StepArgs& a, const HeadArgs& hd, const PrepArgs* next) {
  int rc = run_forward(h, a, hd);
  if (rc) return rc;
  // Backward.  Critical path: fc4_dgrad -> conv3_dgrad -> conv2_dgrad -> conv1_wgrad; the other weight gradients only need the delta of
  // their layer and share a launch with the dgrad that is computed from the same delta.
  // --batch_norm: the delta arriving at layer l (masked by its Rectlin) first goes back through BatchNorm l, in place
#define BN_BWD(L) do { if (h->bn) LAUNCH(K_BN, launch_bn_backward(bn_args(h, a, (L), 1), g_stream)); } while (0)
  BN_BWD(3);
  const StepStructure st = step_structure(h);
  // conv1's weight gradient on packed-bf16 MFMA (sdqn_kernels_r3.hip) in every launch structure (same bits); B < 128 only: in the
  // throughput regime the on-the-fly split of delta1 makes it VALU-bound (3 580 vs 4 063 steps/s at B = 256; option value 2 forces it)
  const bool c1w = (h->conv1w_bf16 == 2 || (h->conv1w_bf16 == 1 && h->B < 128)) && h->cfg.datatype == 0 && !h->bn && h->nw_override[K_CONV1_WGRAD] == 0;
  // round 4, B >= 128 float32: the backward launches on the XCD-contiguous block / tile maps (the blocks that share a weight panel share an L2):
  // fc4_dgrad 15.1 -> 14.4 us, bwd2 38.9 -> 37.5, bwd3 39.7 -> 39.2; 4 802 -> 4 867 steps/s at B = 256 (placement only: same bits)
  const bool bt_xcd = h->B >= 128 && h->cfg.datatype == 0 && h->bt_on && !h->bn && h->bt_xcd;
  { StepArgs fd = a; if (bt_xcd) fd.xcd_map |= 1;
    LAUNCH(K_FC4_DGRAD, launch_tuned(h, K_FC4_DGRAD, fd, g_stream)); }
  BN_BWD(2);
  const int f4_tiles = (NIN4 / 32) * (NFC / 32);
  if (st == STEP_DP_OVERLAP) {
    // data parallel, overlapped: ALL of fc4_wgrad rides the first backward launch, so the 6.4 MB fc4 gradient is
    // complete two launches before the step ends; its all-reduce and its optimizer update run on g_comm
    // (second communicator) under K_BWD2, K_BWD1, the conv/fc5 all-reduce + update and the next step's conv1..3.
    StepArgs b3 = a, b2 = a, b1 = a;
    b3.f4w_first = 0; b3.f4w_count = f4_tiles; b2.f4w_count = b1.f4w_count = 0;
    LAUNCH(K_BWD3, launch_tuned(h, K_BWD3, b3, g_stream));
    HIPCHK(hipEventRecord(h->ev_g4, g_stream));
    HIPCHK(hipStreamWaitEvent(g_comm, h->ev_g4, 0));
    LAUNCH_ON(g_comm, K_ALLREDUCE, dp_allreduce(h, h->g + OFF4, (size_t)NW4, /*ncclFloat32*/ 7, h->comm2, g_comm));
    UpdateArgs u4 = make_update_args(h, a);
    u4.mode = 2; u4.only_fc4 = 1; u4.skip_fc4 = 0; u4.bsz = (float)h->B * (float)h->nranks;
    LAUNCH_ON(g_comm, K_UPDATE, launch_update(u4, g_comm));
    HIPCHK(hipEventRecord(h->ev_w4, g_comm));
    h->w4_pending = true;
    BN_BWD(1);
    LAUNCH(K_BWD2, launch_tuned(h, K_BWD2, b2, g_stream));
    BN_BWD(0);
    LAUNCH(K_BWD1, launch_tuned(h, K_BWD1, b1, g_stream, c1w ? 8 : 0));
  } else if (st == STEP_H16_BT) {
    // round 4, float16 at B >= 128: the two dgrads run on the half block-tile routine as launches of their own (15.6 / 18.1 -> ~7 / 8 us:
    // operands leave L2 once per workgroup), and every weight gradient that does not need delta1 shares ONE launch behind them (it packs
    // better than the two fused backward launches did) — five launches where there were four, 19 us less (tools/exp/README.md)
    StepArgs w = a; w.f4w_first = 0; w.f4w_count = f4_tiles;
    if (h->bt_xcd) w.xcd_map |= 7;          // the weight-gradient launch on XCD-contiguous block maps (the blocks of a K slab share their operand rows): 20.1 -> 17.0 us at B = 256
    StepArgs b1 = a; b1.f4w_count = 0; b1.xcd_map |= 2;
    LAUNCH(K_CONV3_DGRAD, launch_tuned(h, K_CONV3_DGRAD, a, g_stream));
    LAUNCH(K_CONV2_DGRAD, launch_tuned(h, K_CONV2_DGRAD, a, g_stream));
    // round 6: conv1's weight gradient (c1w_h_kernel's K-slab workgroups) as a fourth block range of the weight-gradient launch — delta1 is
    // complete by then — where that kernel applies (packed-fp16 weight gradients, slabs of whole 80-position chunks) and nothing asks for
    // another form (bt:18 / bt:22 = 0; option c1w_in_wgrads: 0 = off, 1 = its workgroups last (built-in), 2 = first)
    const int merge = (h->c1w_in_wgrads && h->h16_wgrad_mfma && (h->tps1 * 32) % 80 == 0 && h->bt[K_BWD1] == 0 && h->bt[K_WGRADS] == 0 &&
                       h->nw_override[K_CONV1_WGRAD] == 0) ? (h->c1w_in_wgrads == 2 ? 48 : 16) : 0;
    LAUNCH(K_WGRADS, launch_tuned(h, K_WGRADS, w, g_stream, merge));
    LAUNCH(K_BWD1, launch_tuned(h, K_BWD1, b1, g_stream, merge));
  } else if (st == STEP_FUSED) {
    // fc4 wgrad (1568 tiles at B <= 32) may be spread over the three backward launches as background traffic (options f4_share3 / f4_share2;
    // built-in: all of it in bwd3); for B > 32 (K-split workgroups) it all rides in the first one
    StepArgs b3 = a, b2 = a, b1 = a;
    if (h->B <= 32) {
      const int s3 = h->f4_share[0] * f4_tiles / 100, s2 = h->f4_share[1] * f4_tiles / 100;
      b3.f4w_first = 0; b3.f4w_count = s3;
      b2.f4w_first = s3; b2.f4w_count = s2;
      b1.f4w_first = s3 + s2; b1.f4w_count = f4_tiles - s3 - s2;
    } else { b3.f4w_first = 0; b3.f4w_count = f4_tiles; b2.f4w_count = b1.f4w_count = 0; }
    if (bt_xcd) { b3.xcd_map |= 7; b2.xcd_map |= 7; }
    // B <= 32: the fc4_wgrad tiles of bwd3 (third problem of the launch) on the XCD-contiguous map — a tile row's 16 tiles share a3's columns
    // (10.29 -> 10.10 us, 15 315 -> 15 345 steps/s in alternating rate loops; the conv3 problems are slower on it: round-robin as before)
    if (h->B <= 32 && h->cfg.datatype == 0 && !h->bn && h->bt_xcd) b3.xcd_map |= 4;
    LAUNCH(K_BWD3, launch_tuned(h, K_BWD3, b3, g_stream));
    BN_BWD(1);
    LAUNCH(K_BWD2, launch_tuned(h, K_BWD2, b2, g_stream));
    BN_BWD(0);
    // conv1_wgrad on the XCD-contiguous tile map: the 8 m-tiles of a K-slab read the same frames, so a slab's tiles belong on ONE XCD's L2
    // (L2 <-> fabric traffic of the launch 15.8 -> 4.0 MB = 1.4x algorithmic, rocprofv3 PMC; step rate -0.1 %: the re-reads were MALL hits)
    b1.xcd_map |= 2;
    LAUNCH(K_BWD1, launch_tuned(h, K_BWD1, b1, g_stream, (c1w && b1.f4w_count == 0) ? 8 : 0));
  } else {
    // fc4_wgrad may update W4 in place (fused RMSProp): it must not start before fc4_dgrad has read W4 — same stream, after it
    LAUNCH(K_FC4_WGRAD, launch_tuned(h, K_FC4_WGRAD, a, g_stream));            // needs d4, a3
    LAUNCH(K_CONV3_WGRAD, launch_tuned(h, K_CONV3_WGRAD, a, g_stream));        // needs d3p, a2
    LAUNCH(K_CONV3_DGRAD, launch_tuned(h, K_CONV3_DGRAD, a, g_stream));
    BN_BWD(1);
    LAUNCH(K_CONV2_WGRAD, launch_tuned(h, K_CONV2_WGRAD, a, g_stream));        // needs d2p, a1
    LAUNCH(K_CONV2_DGRAD, launch_tuned(h, K_CONV2_DGRAD, a, g_stream));
    BN_BWD(0);
    LAUNCH(K_CONV1_WGRAD, launch_tuned(h, K_CONV1_WGRAD, a, g_stream, c1w ? 8 : 0));
  }
  UpdateArgs u = make_update_args(h, a);
  if (next) u.next = *next;                 // (memset above left next.B = 0 otherwise)
  switch (update_form(h)) {
  case UPD_DP_OVERLAP:
    // conv + fc5 gradients (0.3 MB): reduce the slabs, all-reduce the two ranges as one RCCL group on the library
    // stream (first communicator), apply; the fc4 part is already on its way on g_comm
    u.mode = 1; u.bsz = (float)h->B; u.next.B = 0; u.skip_fc4 = 1;
    LAUNCH(K_UPDATE, launch_update(u, g_stream));
    if (g_rccl.GroupStart && g_rccl.GroupEnd) NCCLCHK(g_rccl.GroupStart());
    LAUNCH(K_ALLREDUCE, dp_allreduce(h, h->g, (size_t)OFF4, 7, h->comm, g_stream));
    LAUNCH(K_ALLREDUCE, dp_allreduce(h, h->g + OFF5, (size_t)(h->NP - OFF5), 7, h->comm, g_stream));
    if (g_rccl.GroupStart && g_rccl.GroupEnd) NCCLCHK(g_rccl.GroupEnd());
    u.mode = 2; u.bsz = (float)h->B * (float)h->nranks; u.skip_fc4 = 1;
    if (next) u.next = *next;
    LAUNCH(K_UPDATE, launch_update(u, g_stream));
    if (h->bn) LAUNCH(K_BN, launch_bn_update(u, g_stream));
    break;
  case UPD_DP_SERIAL:
    // synchronous data parallel: local gradient sums -> one RCCL all-reduce of the flat buffer -> identical RMSProp
    u.mode = 1; u.bsz = (float)h->B; u.next.B = 0;
    LAUNCH(K_UPDATE, launch_update(u, g_stream));
    if (h->cfg.datatype == 1 && h->dp_half && h->gh) {
      // fp16 mode: half payload, fp32 accumulation in the optimizer, overflow -> the step is skipped on all ranks
      LAUNCH(K_UPDATE, launch_grad_to_half(h->g, h->gh, h->NP, h->ovf_flag, g_stream));
      LAUNCH(K_ALLREDUCE, dp_allreduce(h, h->gh, (size_t)h->NP, /*ncclFloat16*/ 6, h->comm, g_stream));
      LAUNCH(K_UPDATE, launch_grad_from_half(h->gh, h->g, h->NP, h->ovf_flag, g_stream));
      u.ovf_flag = h->ovf_flag; u.ovf_count = h->ovf_count; u.ovf_dynamic = h->dp_half_scale_log2 < 0 ? 1 : 0;
    } else
    LAUNCH(K_ALLREDUCE, dp_allreduce(h, h->g, (size_t)h->NP, /*ncclFloat32*/ 7, h->comm, g_stream));
    u.mode = 2; u.bsz = (float)h->B * (float)h->nranks;
    if (next) u.next = *next;
    LAUNCH(K_UPDATE, launch_update(u, g_stream));
    if (h->bn) LAUNCH(K_BN, launch_bn_update(u, g_stream));
    break;
  case UPD_GRAD_ONLY:
    u.mode = 1; u.bsz = (float)h->B;                                            // local sums -> g, nothing applied
    LAUNCH(K_UPDATE, launch_update(u, g_stream));
    break;
  default:
    u.mode = 0; u.bsz = (float)h->B;
    LAUNCH(K_UPDATE, launch_update(u, g_stream));
    if (h->bn) LAUNCH(K_BN, launch_bn_update(u, g_stream));
  }
  h->train_iterations += 1;                                                   // deepqnetwork.py:168
  h->spec_pending = false;                 // the online parameters move: a forward enqueued before this step no longer is "predict now"
  return SDQN_OK;
}
int read_cost(sdqn_net_s* h, float* cost_out) {
  HIPCHK(hipMemcpyAsync(h->h_f, h->cost_out, 4, hipMemcpyDeviceToHost, g_stream));
  HIPCHK(hipStreamSynchronize(g_stream));
  *cost_out = h->h_f[0];
  return SDQN_OK;
}

extern "C" int sdqn_net_predict_f64(sdqn_net_t h, const uint8_t* states, double* q_out) {
  ARGCHK(h && states && q_out, "NULL argument");
  if (h->gen) { GENCHK(h->gen->predict_host(states, h->B, q_out, true)); return SDQN_OK; }
  std::vector<float> tmp((size_t)h->B * h->A);
  int rc = sdqn_net_predict(h, states, tmp.data()); if (rc) return rc;
  for (size_t i = 0; i < tmp.size(); ++i) q_out[i] = (double)tmp[i];
  return SDQN_OK;
}
extern "C" int sdqn_net_predict(sdqn_net_t h, const uint8_t* states, float* q_out) {
  ARGCHK(h && states && q_out, "NULL argument");
  if (h->gen) { GENCHK(h->gen->predict_host(states, h->B, q_out, false)); return SDQN_OK; }
  HIPCHK(hipMemcpyAsync(h->st_states, states, (size_t)h->B * STATE, hipMemcpyHostToDevice, g_stream));
  StepArgs a = step_args(h); a.nz = 1; a.from_ring = 0; a.src = h->st_states;
  HeadArgs hd = head_args(h, 0);
  int rc = run_forward(h, a, hd); if (rc) return rc;
  HIPCHK(hipMemcpyAsync(h->h_f, h->q, (size_t)h->B * h->A * 4, hipMemcpyDeviceToHost, g_stream));
  HIPCHK(hipStreamSynchronize(g_stream));
  memcpy(q_out, h->h_f, (size_t)h->B * h->A * 4);                             // (B, A): deepqnetwork.py:186 qvalues.T
  return SDQN_OK;
}

// the one-launch forward's 8 stripe partials [8][ACT_Q_STRIDE] -> Q-values, added in stripe order; false if a stripe never arrived
bool act_sum_partials(const float* part, int A, float* q_out) {
  for (int k = 0; k < A; ++k) {
    float qv = 0.0f;
    for (int sp = 0; sp < 8; ++sp) {
      uint32_t w; memcpy(&w, part + sp * ACT_Q_STRIDE + k, 4);
      if (w == 0xFFFFFFFFu) return false;
      qv = sp ? qv + part[sp * ACT_Q_STRIDE + k] : part[sp * ACT_Q_STRIDE + k];
    }
    q_out[k] = qv;
  }
  return true;
}
extern "C" int sdqn_net_predict_one(sdqn_net_t h, const uint8_t* state, float* q_out) {
  ARGCHK(h && state && q_out, "NULL argument");
  if (h->gen) { GENCHK(h->gen->predict_host(state, 1, q_out, false)); return SDQN_OK; }
  HIPCHK(hipMemcpyAsync(h->st_states, state, (size_t)STATE, hipMemcpyHostToDevice, g_stream));
  if (h->act_on && !h->prof_on) {            // the one-launch forward (sdqn_act.hip): the same kernel predict_state runs, the same numbers
    { int rcj = join_comm(h); if (rcj) return rcj; }
    ActArgs aa; memset(&aa, 0, sizeof aa);
    aa.state = h->st_states; aa.theta = h->theta; aa.scratch = h->act_scratch; aa.ctl = h->act_ctl;
    aa.q = h->act_q; aa.A = h->A; aa.seq = h->act_seq++;
    HIPCHK(hipMemsetAsync(h->act_q, 0xFF, (size_t)Q_SLOT_FLOATS * 4, g_stream));          // (an abandoned launch leaves NaNs, checked below)
    LAUNCH(K_ACT, launch_act(aa, false, g_stream));
    HIPCHK(hipMemcpyAsync(h->h_f, h->act_q, (size_t)Q_SLOT_FLOATS * 4, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    if (act_sum_partials(h->h_f, h->A, q_out)) return SDQN_OK;
    h->act_on = false; h->act_fallbacks += 1;
    fprintf(stderr, "simple_dqn_amd: the one-launch acting forward did not complete; using the five-launch forward from now on\n");
  }
  StepArgs a = step_args(h); a.B = 1; a.nz = 1; a.from_ring = 0; a.src = h->st_states;   // same buffers, batch of one
  HeadArgs hd = head_args(h, 0);
  int rc = run_forward(h, a, hd); if (rc) return rc;
  HIPCHK(hipMemcpyAsync(h->h_f, h->q, (size_t)h->A * 4, hipMemcpyDeviceToHost, g_stream));
  HIPCHK(hipStreamSynchronize(g_stream));
  memcpy(q_out, h->h_f, (size_t)h->A * 4);
  return SDQN_OK;
}

extern "C" int sdqn_net_train_host(sdqn_net_t h, const uint8_t* pre, const uint8_t* actions, const int64_t* rewards,
                                   const uint8_t* post, const uint8_t* terminals, float* cost_out) {
  // the one-shot "minibatch buffers are clean" declarations are consumed FIRST: an argument error below must not leave one armed for a
  // later call whose buffers were edited in place
  sdqn_replay_s* owner = nullptr;                               // (handles are created / destroyed / used from ONE host thread: sdqn.h)
  bool reuse = false;                                           // train on the device copy the last gather left (no state upload)
  for (sdqn_replay_s* r : g_replays) {
    if (h && pre && post && pre == r->h_pre && post == r->h_post && r->B == h->B) { owner = r; reuse = r->mb_clean_declared && (r->mb_host_gen == r->mb_dev_gen || r->mb_clean_on_device); }
    r->mb_clean_declared = false; r->mb_clean_on_device = false;          // one-shot, whoever it was meant for
  }
  ARGCHK(h && pre && actions && rewards && post && terminals, "NULL argument");
  for (int i = 0; i < h->B; ++i) ARGCHK(actions[i] < h->A, "action %d out of range at %d", (int)actions[i], i);
  const bool ours = owner != nullptr;
  if (h->gen) {
    ARGCHK(!ours || (size_t)owner->state == h->gen->state_bytes(), "replay geometry differs from the network's");
    if (reuse) GENCHK(h->gen->train_dev_host_meta(owner->d_pre, owner->d_post, actions, rewards, terminals, h->epoch));
    else GENCHK(h->gen->train_host(pre, actions, rewards, post, terminals, h->epoch));
    h->train_iterations += 1;
    if (cost_out) { double c; GENCHK(h->gen->read_cost(&c)); *cost_out = (float)c; }
    return SDQN_OK;
  }
  const size_t sb = (size_t)h->B * STATE, small = (size_t)h->B * 10;
  // No stream synchronisation (round 1 paid a full PCIe + sync bubble per step here): the caller's arrays are free to
  // change after return because they are either copied into a pinned double buffer of the library first (pageable
  // arrays), or ARE the pinned minibatch buffers of one of this library's ReplayMemory handles — what getMinibatch() returns
  // for prestates / poststates —, whose next overwrite by the library (gather + D2H) is ordered behind this H2D on the library
  // stream; a HOST write to them could still race the DMA, so in that case the call returns only after the upload has completed
  // (event wait AFTER every launch of the step is enqueued: the GPU never idles for it, the host waits ~30 us it would otherwise
  // spend ahead of the stream).  Either way: once train() has returned the caller's five arrays are free, like the reference's.
  // Round 3: when those pinned buffers still hold exactly what the last gather put on the device and the caller says it has not
  // written into them (`reuse`), nothing is uploaded at all — the step reads the device copy in place.
  // Round 5: actions / rewards / terminals too.  The gather that filled the device minibatch also left ring[idx] of the three small
  // arrays next to it (d_rew | d_act | d_term) and the library kept what those values were (mb_snap, taken from the host master at
  // enqueue time); when the caller passes back exactly those values — what getMinibatch() returned, untouched — the step reads the
  // device copy and the call uploads nothing at all (no 10 B x batch H2D, no staging slot, no event: ~6 us of a ~75 us iteration).
  // Any difference (a caller that clips rewards, edits an action, ...) takes the upload below, as before.
  ARGCHK(!ours || owner->tuned_geom, "replay geometry differs from the network's");
  const size_t nb = (size_t)h->B;
  const bool small_dev = reuse && owner->mb_snap && owner->mb_snap_gen == owner->mb_dev_gen &&
                         !memcmp(owner->mb_snap, rewards, nb * 8) && !memcmp(owner->mb_snap + nb * 8, actions, nb) &&
                         !memcmp(owner->mb_snap + nb * 9, terminals, nb);
  if (!small_dev) {
    const int sl = h->stage_next; h->stage_next ^= 1;
    if (!h->h_stage[sl]) {
      HIPCHK(hipHostMalloc((void**)&h->h_stage[sl], 2 * sb + small, hipHostMallocDefault));
      HIPCHK(hipEventCreateWithFlags(&h->stage_ev[sl], hipEventDisableTiming));
    }
    if (h->stage_busy[sl]) { HIPCHK(hipEventSynchronize(h->stage_ev[sl])); h->stage_busy[sl] = false; }
    uint8_t* st = h->h_stage[sl];
    if (!ours) { memcpy(st, pre, sb); memcpy(st + sb, post, sb); }
    uint8_t* sm = st + 2 * sb;                                                  // [rewards 8 B | actions B | terminals B], as on the device
    memcpy(sm, rewards, nb * 8); memcpy(sm + nb * 8, actions, nb); memcpy(sm + nb * 9, terminals, nb);
    // every copy is a packet of its own in the stream: 2 instead of 5 (1 with `reuse`, 0 with `small_dev`)
    if (!reuse) {
      HIPCHK(hipMemcpyAsync(h->st_states, ours ? pre : st, 2 * sb, hipMemcpyHostToDevice, g_stream));     // (a ReplayMemory's pre | post are one block too)
      if (ours) { HIPCHK(hipEventRecord(owner->mb_upload_ev, g_stream)); }   // waited for before this call returns
    }
    HIPCHK(hipMemcpyAsync(h->st_rew, sm, small, hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipEventRecord(h->stage_ev[sl], g_stream)); h->stage_busy[sl] = true;
  }
  h->tuple_calls += 1; h->tuple_states_skipped += reuse ? 1 : 0; h->tuple_small_skipped += small_dev ? 1 : 0;
  StepArgs a = step_args(h); a.from_ring = 0; a.src = reuse ? owner->d_pre : h->st_states;
  HeadArgs hd = head_args(h, 1);
  if (small_dev) { hd.st_actions = owner->d_act; hd.st_rewards = owner->d_rew; hd.st_terminals = owner->d_term; }
  int rc = run_train(h, a, hd); if (rc) return rc;
  if (ours && !reuse) { HIPCHK(hipEventSynchronize(owner->mb_upload_ev)); }
  if (cost_out) return read_cost(h, cost_out);
  return SDQN_OK;
}

extern "C" int sdqn_net_tuple_counters(sdqn_net_t h, int64_t* calls, int64_t* states_in_place, int64_t* nothing_uploaded) {
  ARGCHK(h, "NULL handle");
  if (calls) *calls = h->tuple_calls; if (states_in_place) *states_in_place = h->tuple_states_skipped;
  if (nothing_uploaded) *nothing_uploaded = h->tuple_small_skipped;
  return SDQN_OK;
}

PrepArgs prep_args(sdqn_net_s* h, sdqn_replay_s* r, const int64_t* pinned_idx) {
  PrepArgs p; memset(&p, 0, sizeof p);
  p.idx_pinned = pinned_idx; p.meta = r->d_meta; p.idx = h->d_idx; p.actions = h->st_act;
  p.rewards = h->st_rew; p.terminals = h->st_term; p.B = h->B;
  if (h->B <= 32 && h->prep_inline) {            // the slot's host copy (pinned_idx is its device alias)
    memcpy(p.idx_in, r->h_idx + (pinned_idx - r->d_idx_view), (size_t)h->B * sizeof(int64_t));
    p.idx_in_valid = 1;
  }
  return p;
}
// ring paths take (a, r, t) from the ring: an action the network has no output for would index past the Q row in the
// head kernel (which clamps) and train garbage silently; the tuple API checks the same thing (sdqn_net_train_host)
int check_ring_actions(sdqn_net_s* h, sdqn_replay_s* r, const int64_t* idx) {
  for (int i = 0; i < r->B; ++i)
    ARGCHK(idx[i] >= 0 && idx[i] < r->size && r->actions[idx[i]] < h->A,
           "ring slot %lld holds action %d but the network has %d actions", (long long)idx[i], (int)r->actions[idx[i]], h->A);
  return SDQN_OK;
}
// do_prep: launch the standalone prep for THIS step; next_pinned: fold the NEXT step's prep into the update
int train_replay_slot(sdqn_net_s* h, sdqn_replay_s* r, const int64_t* pinned_idx, bool do_prep,
                             const int64_t* next_pinned, double* zero8) {
  if (do_prep) { PrepArgs p = prep_args(h, r, pinned_idx); LAUNCH(K_PREP, launch_prep(p, g_stream, zero8)); }
  StepArgs a = step_args(h); a.from_ring = 1; a.src = r->d_ring; a.idx = h->d_idx;
  HeadArgs hd = head_args(h, 1);
  // the slot's HOST address (pinned_idx is its device alias): conv1's tiles take their indexes from the kernel arguments
  h->host_idx_cur = r->h_idx + (pinned_idx - r->d_idx_view);
  int rc;
  if (next_pinned) { PrepArgs np = prep_args(h, r, next_pinned); rc = run_train(h, a, hd, &np); }
  else rc = run_train(h, a, hd, nullptr);
  h->host_idx_cur = nullptr;
  return rc;
}
// float64 / other geometries: sample on the host, gather on the device into the replay handle's minibatch buffers, train from there
int gen_train_replay(sdqn_net_s* h, sdqn_replay_s* r, const int64_t* idx_host) {
  ARGCHK((size_t)r->state == h->gen->state_bytes(), "replay geometry (%dx%d, history %d) differs from the network's", r->H, r->W, r->hist);
  int slot; const int64_t* didx; int rc = check_ring_actions(h, r, idx_host); if (rc) return rc;
  rc = replay_push_idx(r, idx_host, &slot, &didx); if (rc) return rc;
  rc = replay_gather_generic(r, didx); if (rc) return rc;
  rc = replay_release_idx_batched(r, slot, false); if (rc) return rc;
  GENCHK(h->gen->train_dev(r->d_pre, r->d_post, r->d_act, r->d_rew, r->d_term, h->epoch));
  h->train_iterations += 1;
  return SDQN_OK;
}
extern "C" int sdqn_net_train_replay(sdqn_net_t h, sdqn_replay_t r, const int64_t* idx_host, float* cost_out) {
  ARGCHK(h && r && idx_host, "NULL argument");
  ARGCHK(r->B == h->B, "replay batch_size %d != network batch_size %d", r->B, h->B);
  if (h->gen) {
    int rc = gen_train_replay(h, r, idx_host); if (rc) return rc;
    if (cost_out) { double c; GENCHK(h->gen->read_cost(&c)); *cost_out = (float)c; }
    return SDQN_OK;
  }
  ARGCHK(r->tuned_geom, "replay geometry (%dx%d, history %d) differs from the network's (84x84, 4)", r->H, r->W, r->hist);
  int slot; const int64_t* didx; int rc = check_ring_actions(h, r, idx_host); if (rc) return rc;
  rc = replay_push_idx(r, idx_host, &slot, &didx); if (rc) return rc;
  rc = train_replay_slot(h, r, didx); if (rc) return rc;
  rc = replay_release_idx_batched(r, slot, false); if (rc) return rc;
  if (cost_out) return read_cost(h, cost_out);
  return SDQN_OK;
}
extern "C" int sdqn_net_train_many(sdqn_net_t h, sdqn_replay_t r, uint32_t* mt, int n_steps, float* mean_cost) {
  ARGCHK(h && r && mt && n_steps >= 0, "bad arguments");
  ARGCHK(r->B == h->B, "replay batch_size %d != network batch_size %d", r->B, h->B);
  if (h->gen) {
    std::vector<int64_t> gi((size_t)r->B);
    GENCHK(h->gen->reset_cost_sum());
    for (int i = 0; i < n_steps; ++i) {
      int rc = sample_checked(mt, r->terminals, r->count, r->current, r->hist, r->B, gi.data(), nullptr); if (rc) return rc;
      rc = gen_train_replay(h, r, gi.data()); if (rc) return rc;
    }
    int rc = replay_flush_pending(r); if (rc) return rc;
    if (mean_cost) { double sum; GENCHK(h->gen->read_cost_sum(&sum)); *mean_cost = n_steps ? (float)(sum / n_steps) : 0.0f; }
    return SDQN_OK;
  }
  ARGCHK(r->tuned_geom, "replay geometry (%dx%d, history %d) differs from the network's (84x84, 4)", r->H, r->W, r->hist);
  std::vector<int64_t> idx((size_t)r->B);
  if (n_steps == 0) HIPCHK(hipMemsetAsync(h->cost_accum, 0, 8, g_stream));       // (otherwise the first step's prep launch clears it)
  // sample one step ahead: step i's update launch also performs step i+1's prep (index copy + metadata gather)
  int slot = -1, next_slot = -1; const int64_t *pinned = nullptr, *next_pinned = nullptr;
  if (n_steps > 0) {
    int rc = sample_checked(mt, r->terminals, r->count, r->current, r->hist, r->B, idx.data(), nullptr); if (rc) return rc;
    rc = check_ring_actions(h, r, idx.data()); if (rc) return rc;
    rc = replay_push_idx(r, idx.data(), &slot, &pinned); if (rc) return rc;
  }
  for (int i = 0; i < n_steps; ++i) {
    next_pinned = nullptr;
    if (i + 1 < n_steps) {
      int rc = sample_checked(mt, r->terminals, r->count, r->current, r->hist, r->B, idx.data(), nullptr); if (rc) return rc;
      rc = check_ring_actions(h, r, idx.data()); if (rc) return rc;
      rc = replay_push_idx(r, idx.data(), &next_slot, &next_pinned); if (rc) return rc;
    }
    int rc = train_replay_slot(h, r, pinned, /*do_prep=*/i == 0, next_pinned, i == 0 ? h->cost_accum : nullptr); if (rc) return rc;
    rc = replay_release_idx_batched(r, slot, false);
    if (rc) return rc;
    slot = next_slot; pinned = next_pinned;
  }
  if (mean_cost) {
    HIPCHK(hipMemcpyAsync(h->h_f, h->cost_accum, 8, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    *mean_cost = n_steps ? (float)(*(double*)h->h_f / n_steps) : 0.0f;
  }
  return SDQN_OK;
}
// ---- train_many without waiting for the cost (agent.py:108-114 + deepqnetwork.py:168-172 when the callback can take the cost later) ----------
// The mean cost of the call's steps is copied into a pinned ring slot by the stream itself; sdqn_net_cost_collect polls the slot (bounded).
// A ticket is valid until COST_RING further deferred calls have been made.
extern "C" int sdqn_net_train_many_deferred(sdqn_net_t h, sdqn_replay_t r, uint32_t* mt, int n_steps, int64_t* ticket) {
  ARGCHK(h && ticket && n_steps >= 1, "bad arguments");
  if (!h->cost_ring) HIPCHK(hipHostMalloc((void**)&h->cost_ring, COST_RING * sizeof(double), hipHostMallocDefault));
  const int slot = (int)(h->cost_ticket % COST_RING);
  if (h->gen) {
    float c = 0.0f; int rc = sdqn_net_train_many(h, r, mt, n_steps, &c); if (rc) return rc;
    h->cost_ring[slot] = (double)c * n_steps;
  } else {
    int rc = sdqn_net_train_many(h, r, mt, n_steps, nullptr); if (rc) return rc;
    uint64_t s1 = ~0ull; memcpy(&h->cost_ring[slot], &s1, 8);                          // sentinel: a NaN no cost sum produces
    HIPCHK(hipMemcpyAsync(&h->cost_ring[slot], h->cost_accum, 8, hipMemcpyDeviceToHost, g_stream));
  }
  h->cost_steps[slot] = n_steps;
  *ticket = h->cost_ticket++;
  return SDQN_OK;
}
extern "C" int sdqn_net_cost_collect(sdqn_net_t h, int64_t ticket, float* mean_cost) {
  ARGCHK(h && mean_cost, "NULL argument");
  ARGCHK(h->cost_ring && ticket >= 0 && ticket < h->cost_ticket && ticket + COST_RING > h->cost_ticket, "stale or unknown cost ticket");
  const int slot = (int)(ticket % COST_RING);
  volatile uint64_t* w = reinterpret_cast<volatile uint64_t*>(&h->cost_ring[slot]);
  const auto t0 = std::chrono::steady_clock::now();
  while (*w == ~0ull) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) { HIPCHK(hipStreamSynchronize(g_stream)); break; }
  }
  ARGCHK(*w != ~0ull, "the cost of ticket %lld was never delivered", (long long)ticket);
  uint64_t bits = *w; double sum; memcpy(&sum, &bits, 8);
  *mean_cost = (float)(sum / h->cost_steps[slot]);
  return SDQN_OK;
}
extern "C" int sdqn_mt_words(uint64_t* words) { ARGCHK(words, "NULL argument"); *words = mt_words_drawn(); return SDQN_OK; }
