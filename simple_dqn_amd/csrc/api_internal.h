// api_internal.h — what the translation units of the host side share (sdqn_api_*.hip): handles, library-wide state, error / launch
// macros, and the prototypes of the functions that cross a file boundary.  Nothing here is part of the C ABI (include/sdqn.h).
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <string>
#include <vector>
#include <chrono>
#include <thread>

#include "../../include/sdqn.h"
#include "kernels.h"
#include "generic_net.h"
#include "sampler.h"

using namespace sdqn;


static constexpr int Q_SLOT_FLOATS = 8 * ACT_Q_STRIDE;   // one slot: [A] Q-values (head kernel) or 8 stripe partials [8][ACT_Q_STRIDE] (one-launch forward)
static constexpr int COST_RING = 64;
static constexpr int Q_SLOTS = 8;       // host-mapped Q-value slots of the acting path (sdqn_net_predict_state)

extern thread_local std::string g_err;
void set_error(const char* fmt, ...);
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return SDQN_ERR_HIP; } } while (0)
#define ARGCHK(c, ...) do { if (!(c)) { set_error(__VA_ARGS__); return SDQN_ERR_ARG; } } while (0)

extern hipStream_t g_stream;      // the library stream: everything is ordered on it
extern hipStream_t g_side;        // second stream (the dp probe's cross-stream round trip)
extern hipStream_t g_comm;        // data parallel: the fc4 gradient all-reduce + fc4 update run here, beside the compute stream
extern hipEvent_t g_ev[5];        // fork/join events (timing disabled)
extern int g_dev;                 // device the library streams live on (bound by the first device call)
#define STREAMCHK() do { int r_ = ensure_stream(); if (r_) return r_; } while (0)

const int NSLOT = 64;     // pinned index slots: kernels read the sampled indexes zero-copy
struct sdqn_replay_s {
  int64_t size = 0; int H = 0, W = 0, hist = 0, B = 0, flags = 0;
  int64_t frame = 0, state = 0;                    // bytes per screen / per state (hist screens); the tuned kernels need 84 x 84 x 4
  bool tuned_geom = true;
  int64_t count = 0, current = 0;
  uint8_t* screens = nullptr; uint8_t* actions = nullptr; int64_t* rewards = nullptr; uint8_t* terminals = nullptr;  // pinned master
  MetaRec* h_meta = nullptr;                       // pinned packed metadata (source of the per-add H2D)
  uint8_t* d_ring = nullptr; MetaRec* d_meta = nullptr;
  uint8_t *d_pre = nullptr, *d_post = nullptr, *d_act = nullptr, *d_term = nullptr; int64_t* d_rew = nullptr;
  uint8_t *h_pre = nullptr, *h_post = nullptr, *h_act = nullptr, *h_term = nullptr; int64_t* h_rew = nullptr;
  int64_t* h_idx = nullptr; int64_t* d_idx_view = nullptr;      // [NSLOT][B] pinned + its device alias
  hipEvent_t slot_ev[NSLOT]; bool slot_busy[NSLOT]; int next_slot = 0;
  int slot_cover[NSLOT]; int pending[NSLOT]; int npending = 0;   // batched release (train_many): slot s is free once slot_ev[slot_cover[s]] has completed
  // tuple API: the device copy of the gathered minibatch (d_pre | d_post) stays valid after getMinibatch() has brought it down; a
  // train(tuple) call on the very same pinned arrays may read it in place instead of uploading 2 x B x state bytes again — when the
  // host copy is as new as the device copy (generations) AND the caller has declared that it did not write into the host arrays
  // (sdqn_replay_declare_minibatch_clean: one-shot, consumed by the next sdqn_net_train_host)
  uint64_t mb_dev_gen = 1, mb_host_gen = 0; bool mb_clean_declared = false, mb_clean_on_device = false;
  hipEvent_t mb_upload_ev = nullptr;      // tuple API: the H2D of h_pre | h_post issued by sdqn_net_train_host (waited for before that call returns)
  // what the last sdqn_replay_gather left in d_rew | d_act | d_term, as [rewards 8 B x B | actions B | terminals B] taken from the host
  // master at enqueue time (== the device metadata in stream order), and the device generation it belongs to: a tuple whose small
  // arrays equal it trains on the device copy (sdqn_net_train_host)
  uint8_t* mb_snap = nullptr; uint64_t mb_snap_gen = 0;
  uint64_t mb_gather_gen = 0;   // device-minibatch generation the last sdqn_replay_gather left (sdqn_replay_declare_minibatch_on_device(h, UINT64_MAX) names it)
};
extern std::vector<sdqn_replay_s*> g_replays;      // live handles: sdqn_net_train_host recognises their pinned minibatch buffers

struct Id128 { char b[128]; };   // ncclUniqueId is passed BY VALUE to ncclCommInitRank
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Id128, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;   // optional (replica sync at dp_init)
  int (*CommDestroy)(void*) = nullptr;
  int (*CommSplit)(void*, int, int, void**, void*) = nullptr;      // optional (second communicator for the overlapped all-reduce)
  int (*CommAbort)(void*) = nullptr;                              // optional (tears down a communicator whose collective never completed)
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;
  int (*CommUserRank)(void*, int*) = nullptr;
  int (*CommCuDevice)(void*, int*) = nullptr;
};
extern Rccl g_rccl;
#define NCCLCHK(x) do { int e_ = (x); if (e_ != 0) { set_error("%s -> %s", #x, g_rccl.GetErrorString ? g_rccl.GetErrorString(e_) : "rccl error"); return SDQN_ERR_RCCL; } } while (0)

struct ProfPair { int id; hipEvent_t a, b; };
struct sdqn_net_s {
  GenericNet* gen = nullptr;               // float64 / non-84x84x4 configurations: the whole network lives there (generic_net.hip)
  sdqn_net_cfg cfg; int B = 0, A = 0; int64_t NP = 0;   // NP: floats per flat buffer (weights [+ BatchNorm params + running stats])
  int64_t NPW = 0;                         // weights only = offset of the BatchNorm block
  bool bn = false;                         // --batch_norm
  float *x1 = nullptr, *x2 = nullptr, *x3 = nullptr;          // raw linear outputs [2][B*PIX][K] (BatchNorm input; kept for the backward pass)
  float *bn_mean = nullptr, *bn_rstd = nullptr; double* bn_partial = nullptr;
  float *theta = nullptr, *theta_t = nullptr, *state = nullptr, *state2 = nullptr, *g = nullptr;
  int epoch = 0;
  float *a1 = nullptr, *a2 = nullptr, *a3 = nullptr, *slab4 = nullptr, *a4 = nullptr, *d4 = nullptr;
  float *d3 = nullptr, *d2 = nullptr;
  // fp16 mode (cfg.datatype == 1)
  half_t *h_a1 = nullptr, *h_a2 = nullptr, *h_a3 = nullptr, *h_d4 = nullptr, *h_d3p = nullptr, *h_d3 = nullptr,
         *h_d2p = nullptr, *h_d2 = nullptr, *h_d1 = nullptr, *wh[2] = {nullptr, nullptr}, *wht[2] = {nullptr, nullptr};
  float *d3p = nullptr, *d2p = nullptr, *d1 = nullptr, *slab1 = nullptr, *slab2 = nullptr, *slab3 = nullptr;
  float *q = nullptr, *maxq = nullptr, *dq = nullptr, *cost_terms = nullptr, *cost_out = nullptr; double* cost_accum = nullptr;
  uint8_t *st_states = nullptr, *st_act = nullptr, *st_term = nullptr; int64_t* st_rew = nullptr; int64_t* d_idx = nullptr;
  int64_t* d_idx_t = nullptr;              // hoist: the NEXT step's indexes (copied from their pinned slot by an extra workgroup of the head launch)
                                           // Built, bit-identical, measured 1.8 % SLOWER (tools/exp/README.md) -> off; set_option "hoist"
  float* h_f = nullptr;                    // pinned scratch for small read-backs
  // acting path (round 4): the head kernel of a predict_state forward writes its Q-values straight into mapped host memory (q_host; q_host_dev
  // = its device alias) and the host polls for them.  spec_*: a forward enqueued AHEAD of its use by sdqn_net_act_step (speculation) — valid
  // while the state buffer generation and the parameters are what they were when it was enqueued
  float *q_host = nullptr, *q_host_dev = nullptr;      // Q_SLOTS slots of 32 floats: every enqueued acting forward gets its OWN slot, so a
  int q_slot = 0;                                       // speculation that was dropped (still in flight) cannot write into the slot being polled
  bool head_q_system = false;              // (run_forward: this forward's head writes system-scope)
  // the acting forward as ONE launch (sdqn_act.hip; float32, no batch-norm): per-XCC scratch copies, fc4 partial slots, control blocks
  float *act_scratch = nullptr, *act_q = nullptr; unsigned* act_ctl = nullptr; unsigned act_seq = 0;
  bool act_on = false, act_last = false, act_inject = false; int act_fallbacks = 0;
  int64_t tuple_calls = 0, tuple_states_skipped = 0, tuple_small_skipped = 0;   // sdqn_net_train_host: calls / calls without the state upload / without any upload
      // act_inject (tests): the next one-launch forward finds its work already claimed and delivers nothing     // act_last: the forward being collected came from that launch
  // deferred cost read-back (sdqn_net_train_many_deferred): pinned ring of cost sums the stream copies into
  double* cost_ring = nullptr; int cost_steps[64] = {0}; int64_t cost_ticket = 0;
  bool spec_pending = false; const void* spec_sb = nullptr; uint64_t spec_gen = 0;
  uint8_t* h_stage[2] = {nullptr, nullptr}; hipEvent_t stage_ev[2] = {nullptr, nullptr}; bool stage_busy[2] = {false, false}; int stage_next = 0;
                                           // tuple API (sdqn_net_train_host): pinned double buffer for the caller's pageable minibatch
  int S4 = 7, S4_cap = 7, tps1 = 1, tps2 = 1, tps3 = 1, ns1 = 1, ns2 = 1, ns3 = 1;
  int64_t train_iterations = 0;
  bool keep_grads = false;                 // true: fc4 gradient materialised in g (readable with which=3), no fused RMSProp
  half_t* gh = nullptr; int* ovf_flag = nullptr; int64_t* ovf_count = nullptr;    // fp16 data parallel: half gradient payload, overflow flag / skipped steps
  int dp_half = 1, dp_half_scale_log2 = -1;  // fp16 mode: all-reduce the gradient as half; -1 = dynamic payload scale (starts at 2^10, device-side), >= 0 = fixed 2^n
  bool h16_wgrad_mfma = true;              // fp16 mode: weight gradients on packed-fp16 MFMA (LDS transposes); false = fp32 MFMA with half operands
  int c1w_in_wgrads = 1;                   // fp16 mode, B >= 128: conv1's weight gradient as a block range of the weight-gradient launch (0 off, 1 last, 2 first)
  bool half_payload_pending = false;       // sdqn_net_grad_from_half ran: the next sdqn_net_apply_update honours the overflow flag / moves the scale
  bool grad_only = false;                  // true: a train step stops after the local gradient sums (update mode 1): what a
                                           // data-parallel rank has before the all-reduce; sdqn_net_apply_update finishes it
  int nw_override[12] = {0};               // tuning hook
  int f4_share[2] = {100, 0};               // % of the fc4-wgrad tiles in bwd3 / bwd2 (rest in bwd1)
  int ns_cap[3] = {1, 1, 1};               // slabs the split-K buffers were allocated for (tuning hook "tps:<layer>")
  bool bt_xcd = true;                      // round 4, B >= 128: XCD-contiguous block maps for fc4_dgrad / bwd3 / bwd2 (float32) and the weight-gradient launch (float16); option "bt_xcd"
  bool bt_on = true; int bt[K_COUNT] = {0};  // round 4, B >= 128 float32: block-tile engine (sdqn_kernels_bt.hip); per kernel id 0 = built-in block shape, n = menu entry, -1 = latency engine
  int xcd_mask[K_COUNT] = {0};             // tuning hook "xcd:<kernel id>": per-launch problem mask (-1 = built-in)
  bool xcd_map = false;                    // XCD-contiguous tile map for EVERY launch: traffic ~ algorithmic, step ~1 % slower (bwd3);
                                           // built-in: only where it also wins time (fc4_fwd: the 7 K-slabs of a tile's W4 panel share an L2)
  bool conv1_bf16 = true;                  // round 3: conv1_fwd on packed-bf16 MFMA (bytes x 3-way bf16 split of W1; sdqn_kernels_r3.hip)
  unsigned short* w1p[2] = {nullptr, nullptr};   // the three bf16 planes of W1, online / target net ([3][32][256] each)
  const int64_t* host_idx_cur = nullptr;   // ring paths: host copy of the indexes of the step being enqueued (valid during run_train only)
  int r3_xcd = 2;                          // XCD-contiguous tile maps of the round-3 kernels: bit 0 conv1_fwd (bf16), bit 1 conv1_wgrad (bf16)
  bool prep_inline = true;                 // B <= 32: the next step's indexes ride in the update launch's kernel arguments (no PCIe read in its prep block)
  int wt = 511;                            // write-through epilogue stores, bit per launch (kernels.h: LaunchTune::wt; bit 8 = the update kernel)
  int conv1w_bf16 = 1;                 // round 3: conv1_wgrad on packed-bf16 MFMA (bytes x on-the-fly bf16 split of delta1)
  bool conv3_c36 = true;                   // round 3: conv3_fwd on 36-deep K-chunks (one chunk per wave; sdqn_kernels_r3.hip)
  bool fused_launches = true;              // independent backward stages share one launch (K_BWD3, K_BWD2)
  // profiler
  bool prof_on = false; int prof_filter = -1;
  int prof_mode = 1;                                        // 1: kernel-packet timestamps (hipExtLaunchKernel), 0: hipEventRecord markers around the launch
  int prof_every = 1; int64_t prof_seen[K_COUNT] = {0};     // bracket only every prof_every-th launch of a kernel (an event pair costs ~2-3 us of queue time)
  std::vector<ProfPair> prof_pending; std::vector<hipEvent_t> prof_free;
  double prof_ms[K_COUNT]; int64_t prof_n[K_COUNT];
  // data parallel
  void* comm = nullptr; int rank = 0, nranks = 1; int nccl_rc = 0;
  // overlapped data parallel (run_train): comm2 carries the fc4 gradient (95 % of the bytes) on g_comm while the
  // compute stream finishes the backward pass and starts the next forward; ev_w4 = "W4 of the last step is updated"
  void* comm2 = nullptr; hipEvent_t ev_g4 = nullptr, ev_w4 = nullptr; bool w4_pending = false;
  bool dp_sync_replicas = true;   // dp_init broadcasts rank 0's theta / theta_t / optimizer state (set_option "dp_sync_replicas" 0: keep own)
  // Overlapped form (fc4's 95 % of the payload all-reduced + applied on a second communicator / stream under the rest of the step).
  // dp_overlap_req: -1 AUTO (default, round 4): sdqn_dp_init creates the second communicator whenever nranks >= 2, but the form only
  //   becomes ACTIVE (dp_overlap) after sdqn_dp_probe succeeded on EVERY rank (the caller votes over its control plane) and
  //   sdqn_dp_set_overlap(1) was called; any rank timing out -> sdqn_dp_set_overlap(0) on all ranks: second communicator torn down,
  //   the serial form (one all-reduce on the library stream) runs.  1: forced on at dp_init (single-rank tests), 0: never.
  int dp_overlap_req = -1;
  bool dp_overlap = false;        // the overlapped form is active
  int dp_probe_result = -1;       // -1 not probed, 0 timed out / failed, 1 ok (sdqn_dp_probe)
  int wt_kid = -1; unsigned long long* wt_words = nullptr;     // timing build: per-wave stamps around launches of this id (pinned: {buffer, null, blocks})
  std::vector<void*> allocs;
};
#define GENCHK(x) do { hipError_t ge_ = (x); if (ge_ != hipSuccess) { set_error("%s -> %s", #x, hipGetErrorString(ge_)); return ge_ == hipErrorInvalidValue ? SDQN_ERR_ARG : SDQN_ERR_HIP; } } while (0)

constexpr int SB_SLOTS = 64;
uint64_t next_statebuf_gen();
struct sdqn_statebuf_s {
  uint8_t* d = nullptr;          // [SB_SLOTS][FRAME]
  uint8_t* host = nullptr;       // [hist][FRAME] mirror in state_buffer.py order (oldest first)
  uint8_t* stage = nullptr;      // pinned [SB_SLOTS][FRAME]: staging slot i feeds ring slot i
  int hist = 0;
  int64_t frame = 0;             // bytes per screen
  int pos = 0;                   // slot of the newest frame; window = slots [pos-hist+1, pos]
  uint64_t gen = next_statebuf_gen();   // bumped by every add / reset: identifies the state a speculative forward was enqueued for
                                 // (own 2^40 range per buffer: a buffer allocated where a destroyed one lay never matches its generations)
};

// which launches a train step is made of / how its optimizer pass runs (sdqn_api_step.hip: step_structure, update_form)
enum StepStructure { STEP_FUSED = 0, STEP_H16_BT = 1, STEP_DP_OVERLAP = 2, STEP_UNFUSED = 3 };
enum UpdateForm { UPD_SINGLE = 0, UPD_DP_SERIAL = 1, UPD_DP_OVERLAP = 2, UPD_GRAD_ONLY = 3 };
// ---- functions that cross a file boundary -------------------------------------------------------------------------------
int ensure_stream();
int sample_checked(uint32_t* mt, const uint8_t* terminals, int64_t count, int64_t current, int hist,
                          int batch, int64_t* idx_out, int64_t* draws_out);
int replay_free(sdqn_replay_s* r);
int replay_flush_pending(sdqn_replay_s* r);
int replay_push_idx(sdqn_replay_s* r, const int64_t* idx, int* slot_out, const int64_t** dev);
int replay_release_idx(sdqn_replay_s* r, int slot);
int replay_release_idx_batched(sdqn_replay_s* r, int slot, bool flush);
GatherArgs gather_args(sdqn_replay_s* r, const int64_t* didx);
int replay_gather_generic(sdqn_replay_s* r, const int64_t* didx);
int rccl_load(const char* path);
int dalloc(sdqn_net_s* h, void** p, size_t bytes, bool zero = true);
int net_free(sdqn_net_s* h);
float* which_buf(sdqn_net_s* h, int which);
bool bn_layer_span(sdqn_net_s* h, int which, int layer, float** base, int64_t* n);
int gen_set(sdqn_net_s* h, int which, int layer, const void* w, int64_t n, bool f64);
int gen_get(sdqn_net_s* h, int which, int layer, void* w, int64_t n, bool f64);
int prof_collect(sdqn_net_s* h);
int prof_event(sdqn_net_s* h, hipEvent_t* e);
hipError_t dp_allreduce(sdqn_net_s* h, void* buf, size_t count, int dtype, void* comm, hipStream_t s);
StepArgs step_args(sdqn_net_s* h);
HeadArgs head_args(sdqn_net_s* h, int train);
int join_comm(sdqn_net_s* h);
BnArgs bn_args(sdqn_net_s* h, const StepArgs& a, int layer, int train);
hipError_t launch_tuned(sdqn_net_s* h, int id, StepArgs a, hipStream_t s, int r3 = 0);
int run_forward(sdqn_net_s* h, const StepArgs& a, const HeadArgs& hd);
UpdateArgs make_update_args(sdqn_net_s* h, const StepArgs& a);
StepStructure step_structure(const sdqn_net_s* h);
UpdateForm update_form(const sdqn_net_s* h);
int run_train(sdqn_net_s* h, const StepArgs& a, const HeadArgs& hd, const PrepArgs* next = nullptr);
int read_cost(sdqn_net_s* h, float* cost_out);
bool act_sum_partials(const float* part, int A, float* q_out);
const uint8_t* statebuf_window(sdqn_statebuf_s* s);
int predict_state_enqueue(sdqn_net_s* h, sdqn_statebuf_s* sb);
bool act_trace();
int predict_state_collect(sdqn_net_s* h, float* q_out);
PrepArgs prep_args(sdqn_net_s* h, sdqn_replay_s* r, const int64_t* pinned_idx);
int check_ring_actions(sdqn_net_s* h, sdqn_replay_s* r, const int64_t* idx);
int train_replay_slot(sdqn_net_s* h, sdqn_replay_s* r, const int64_t* pinned_idx, bool do_prep = true,
                             const int64_t* next_pinned = nullptr, double* zero8 = nullptr);
int gen_train_replay(sdqn_net_s* h, sdqn_replay_s* r, const int64_t* idx_host);

inline bool prof_single_kernel(int kid) { return kid != K_ALLREDUCE && kid != K_BN; }
#ifdef SDQN_TIMING
namespace sdqn { hipError_t set_wave_timing_buffer(unsigned long long* const* p, const unsigned* nb, hipStream_t s); }
// timing build: sdqn_debug_time_step_waves arms per-wave stamps around ONE launch id of a real train step (h->wt_kid), stream-ordered
#define SDQN_WAVE_TIMING_ARM(STRM, KID) do { if (h->wt_kid == (KID) && h->wt_words) HIPCHK(sdqn::set_wave_timing_buffer((unsigned long long* const*)h->wt_words, (const unsigned*)(h->wt_words + 2), (STRM))); } while (0)
#define SDQN_WAVE_TIMING_DISARM(STRM, KID) do { if (h->wt_kid == (KID) && h->wt_words) HIPCHK(sdqn::set_wave_timing_buffer((unsigned long long* const*)(h->wt_words + 1), (const unsigned*)(h->wt_words + 2), (STRM))); } while (0)
#else
#define SDQN_WAVE_TIMING_ARM(STRM, KID) do {} while (0)
#define SDQN_WAVE_TIMING_DISARM(STRM, KID) do {} while (0)
#endif
#define LAUNCH_ON(STRM, KID, expr) do { \
  const bool pf_ = h->prof_on && (h->prof_filter < 0 || h->prof_filter == (KID)) && (h->prof_seen[KID]++ % h->prof_every) == 0; ProfPair pp_; \
  const bool px_ = pf_ && h->prof_mode == 1 && prof_single_kernel(KID); \
  if (pf_) { pp_.id = (KID); int r1_ = prof_event(h, &pp_.a); if (r1_) return r1_; r1_ = prof_event(h, &pp_.b); if (r1_) return r1_; \
             if (px_) { sdqn::LaunchEvents& le = sdqn::launch_events(); le.start = pp_.a; le.stop = pp_.b; le.used = false; } \
             else HIPCHK(hipEventRecord(pp_.a, (STRM))); } \
  SDQN_WAVE_TIMING_ARM(STRM, KID); \
  hipError_t le_ = (expr); \
  SDQN_WAVE_TIMING_DISARM(STRM, KID); \
  bool pu_ = true; \
  if (px_) { sdqn::LaunchEvents& le = sdqn::launch_events(); pu_ = le.used; le.start = le.stop = nullptr; le.used = false; } \
  if (le_ != hipSuccess) { \
    if ((KID) == K_ALLREDUCE && h->nccl_rc != 0) { \
      set_error("ncclAllReduce (rank %d of %d) -> %s", h->rank, h->nranks, g_rccl.GetErrorString ? g_rccl.GetErrorString(h->nccl_rc) : "rccl error"); \
      return SDQN_ERR_RCCL; } \
    set_error("launch %s -> %s", kernel_name(KID), hipGetErrorString(le_)); return SDQN_ERR_HIP; } \
  if (pf_ && !pu_) { h->prof_free.push_back(pp_.a); h->prof_free.push_back(pp_.b); }     /* (a launch path that did not take the events: no sample) */ \
  else if (pf_) { if (!px_) HIPCHK(hipEventRecord(pp_.b, (STRM))); h->prof_pending.push_back(pp_); \
             if (h->prof_pending.size() > 16384) { int r2_ = prof_collect(h); if (r2_) return r2_; } } \
} while (0)
#define LAUNCH(KID, expr) LAUNCH_ON(g_stream, KID, expr)

