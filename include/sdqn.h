/* sdqn.h — C ABI of libsdqn_hip.so: the MI355X (gfx950) DQN training-step hot path.
 *
 * The reference (tambetm/simple_dqn) has no FFI of its own: its boundary for this
 * path is the duck-typed Python API of two classes.  Each entry point below names
 * the reference interface (file:line under /root/reference) that it replaces; the
 * ctypes binding that turns them back into those classes is
 * simple_dqn_amd/_lib.py.
 *
 * Conventions
 *   - every function returns 0 on success, a negative sdqn_status on failure;
 *     sdqn_last_error() returns a thread-local message for the last failure.
 *     No exception crosses the boundary.
 *   - handles are opaque, owned by the library, single-owner (not thread-safe),
 *     matching the reference's single-threaded caller (src/agent.py:96-116).
 *   - every device operation is enqueued on ONE library stream per process and is
 *     asynchronous unless the call returns host data (then it synchronises).
 *   - host pointers passed in are borrowed for the duration of the call only;
 *     pointers handed out (sdqn_replay_host_ptrs) live as long as the handle.
 *   - there is NO CPU fallback: without a gfx950 device every device entry point
 *     fails with SDQN_ERR_HIP.
 */
#ifndef SDQN_H
#define SDQN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  SDQN_OK = 0,
  SDQN_ERR_ARG = -1,     /* precondition / shape violation: the reference raises AssertionError */
  SDQN_ERR_HIP = -2,     /* HIP runtime failure (incl. "no device") */
  SDQN_ERR_RCCL = -3,    /* RCCL failure */
  SDQN_ERR_STATE = -4    /* call not valid in the handle's current state */
} sdqn_status;

typedef struct sdqn_replay_s* sdqn_replay_t;
typedef struct sdqn_net_s* sdqn_net_t;

#define SDQN_MT_WORDS 625          /* CPython random.getstate()[1]: 624 state words + position */

/* ---- misc ---------------------------------------------------------------- */
const char* sdqn_last_error(void);
int sdqn_version(void);
int sdqn_device_count(int* n);
/* replaces --device_id (src/main.py:52 -> gen_backend(device_id=...), deepqnetwork.py:29-34).  One device per process:
 * the first device call binds the library; asking for the bound device again is a no-op, any other device afterwards
 * is SDQN_ERR_STATE (never a silent run on the wrong GPU). */
int sdqn_set_device(int dev);
int sdqn_get_device(int* dev);                /* the device the library is bound to (binds the current one if none yet) */
int sdqn_device_sync(void);

/* ---- index sampler: pure host, no device needed ---------------------------
 * replaces the rejection sampler in ReplayMemory.getMinibatch, src/replay_memory.py:54-68,
 * bit-exact against CPython 3.10's random.randint stream (Lib/random.py randint ->
 * _randbelow_with_getrandbits; Modules/_randommodule.c genrand_uint32).               */
int sdqn_mt_seed(uint32_t mt[SDQN_MT_WORDS], uint64_t seed);            /* random.seed(int), src/main.py:89-90 */
int sdqn_mt_randint(uint32_t mt[SDQN_MT_WORDS], int64_t a, int64_t b, int64_t* out);
int sdqn_sample_indices(uint32_t mt[SDQN_MT_WORDS], const uint8_t* terminals, int64_t count,
                        int64_t current, int history_length, int batch,
                        int64_t* idx_out /*[batch]*/, int64_t* draws_out /*nullable*/);

/* ---- replay memory: src/replay_memory.py ---------------------------------- */
#define SDQN_REPLAY_HBM_MIRROR 1   /* ring master in pinned host DRAM + mirror in HBM (default) */
#define SDQN_REPLAY_ZERO_COPY  2   /* no mirror: kernels read the pinned ring over PCIe */

/* ReplayMemory.__init__, replay_memory.py:7-24 */
int sdqn_replay_create(sdqn_replay_t* h, int64_t size, int screen_height, int screen_width,
                       int history_length, int batch_size, int flags);
int sdqn_replay_destroy(sdqn_replay_t h);
/* the numpy attributes screens/actions/rewards/terminals (replay_memory.py:10-13) are views of these */
int sdqn_replay_host_ptrs(sdqn_replay_t h, uint8_t** screens, uint8_t** actions, int64_t** rewards,
                          uint8_t** terminals);
/* preallocated minibatch outputs prestates/poststates (replay_memory.py:21-22) + a/r/t, pinned host */
int sdqn_replay_minibatch_ptrs(sdqn_replay_t h, uint8_t** pre, uint8_t** post, uint8_t** actions,
                               int64_t** rewards, uint8_t** terminals);
/* ReplayMemory.add, replay_memory.py:26-34 (also enqueues the 7 KB H2D into the mirror) */
int sdqn_replay_add(sdqn_replay_t h, int action, int64_t reward, const uint8_t* screen, int terminal);
int sdqn_replay_get_state(sdqn_replay_t h, int64_t* count, int64_t* current);
int sdqn_replay_set_state(sdqn_replay_t h, int64_t count, int64_t current);
/* after writing the host views directly (bulk fill): copy slots [first, first+n) into the mirror */
int sdqn_replay_upload(sdqn_replay_t h, int64_t first, int64_t n);
/* the same for the packed metadata only (actions / rewards / terminals edited in place: 16 B per slot instead of 7 KB) */
int sdqn_replay_upload_meta(sdqn_replay_t h, int64_t first, int64_t n);
/* replay_memory.py:54-68 on this ring (terminals/count/current of the handle) */
int sdqn_replay_sample(sdqn_replay_t h, uint32_t mt[SDQN_MT_WORDS], int64_t* idx_out, int64_t* draws_out);
/* replay_memory.py:71-78: HIP gather of (s, a, r, s', terminal) by index into device HBM (async) */
int sdqn_replay_gather(sdqn_replay_t h, const int64_t* idx_host /*[batch]*/);
/* replay_memory.py:79: the device minibatch copied into the buffers of sdqn_replay_minibatch_ptrs (sync) */
int sdqn_replay_minibatch_to_host(sdqn_replay_t h);
/* One-shot declaration by the caller of sdqn_net_train_host: "the pinned prestates / poststates buffers of this handle
 * (sdqn_replay_minibatch_ptrs) still hold what the last sdqn_replay_minibatch_to_host wrote — I have not modified them".  If that
 * call is then made on exactly those buffers and no other gather has run since, the step reads the device copy of the minibatch in
 * place and uploads only the 10 x batch_size bytes of (rewards, actions, terminals).  Never required for correctness: without it
 * (C callers that write into the buffers, arrays from elsewhere) the states are uploaded as before.  simple_dqn_amd's
 * ReplayMemory exposes the two buffers as write-tracking numpy views and declares this only while they are untouched. */
int sdqn_replay_declare_minibatch_clean(sdqn_replay_t h);
/* replay_memory.py:79 without the wait (round 5): the reference's getMinibatch() returns arrays; most callers hand them straight back
 * to DeepQNetwork.train (agent.py:112-114) and never read the 1.8 MB of states on the host.  sdqn_replay_minibatch_gen reports the
 * generation of the device minibatch (bumped by every gather launch) and of the host copy (set by sdqn_replay_minibatch_to_host), so a
 * host binding can hand out LAZY state arrays and fetch them only when somebody looks.  sdqn_replay_declare_minibatch_on_device is the
 * one-shot declaration for that case: "the two state arguments of the next sdqn_net_train_host stand for device minibatch `gen` — the
 * host buffers may not hold it yet, and I have not written into them".  Honoured only while `gen` still IS the device minibatch's
 * generation (gen = 0: whatever it holds now — the reference's aliased buffers always show the latest gather); otherwise the call uploads
 * the host buffers as always.  gen = UINT64_MAX names the minibatch of the LAST sdqn_replay_gather and is for a caller that has not
 * fetched it: SDQN_ERR_STATE if another launch has overwritten the device minibatch since (round 6: those states exist nowhere).  rewards / actions / terminals always come from the arguments — compared by value with what that gather
 * left on the device (ring[idx] of the three arrays, snapshotted by sdqn_replay_gather): when they are equal the step reads the device
 * copy and the call uploads nothing; any edit (a clipped reward, another action) is uploaded and used, as before. */
int sdqn_replay_minibatch_gen(sdqn_replay_t h, uint64_t* device_gen, uint64_t* host_gen);
int sdqn_replay_declare_minibatch_on_device(sdqn_replay_t h, uint64_t gen);
/* device-side timing of the last n gather launches, for bench.py (HIP events on the library stream) */
int sdqn_replay_bench_gather(sdqn_replay_t h, const int64_t* idx_host, int iters, float* ms_per_launch);
/* the same with a different index set per launch (idx_host = nsets x batch_size, cycled): a repeated set is served from L2 / MALL
 * after its first launch, getMinibatch() (replay_memory.py:54-79) never gathers the same states twice in a row */
int sdqn_replay_bench_gather_sets(sdqn_replay_t h, const int64_t* idx_host, int nsets, int iters, float* ms_per_launch);

/* ---- Q-network: src/deepqnetwork.py ---------------------------------------- */
typedef struct {
  int batch_size;          /* args.batch_size      deepqnetwork.py:19 */
  int history_length;      /* args.history_length  :21 (main.py:34).  84 x 84 x 4 runs on the tuned kernels; any other    */
  int screen_height;       /* :22 (main.py:28)      geometry the layer stack of :83-91 accepts runs on the generic    */
  int screen_width;        /* :22 (main.py:27)      im2col + GEMM path (csrc/generic_net.hip): float32 / float64 only */
  int num_actions;         /* :18 */
  int target_enabled;      /* bool(args.target_steps) :64 */
  int optimizer;           /* args.optimizer :50-59: 0 rmsprop, 1 adam, 2 adadelta */
  int datatype;            /* args.datatype :33 (main.py:53): 0 float32; 1 float16 = half activations/deltas/MFMA weight operands,
                              fp32 accumulation, master weights and optimizer state (BASELINE.json configs[4]);
                              2 float64 = the whole step in double (generic path) */
  /* Python floats (doubles) exactly as argparse hands them over; the library rounds to fp32
   * where Neon's fp32 backend would (lr, decay, epsilon, clip) and keeps doubles where the
   * reference does host-side float math (discount, reward clip: deepqnetwork.py:136-143).  */
  double discount_rate;    /* :20 */
  double clip_error;       /* :23 (0 disables clipping, :158) */
  double min_reward;       /* :24 */
  double max_reward;       /* :25 */
  double learning_rate;    /* :51 */
  double decay_rate;       /* :52 */
  double epsilon;          /* Neon defaults: RMSProp 1e-6, Adam 1e-8, Adadelta 1e-6 */
  double beta_1;           /* Adam (Neon default 0.9) */
  double beta_2;           /* Adam (Neon default 0.999) */
  double loss_scale;       /* float16 mode: static power-of-two scale of the stored deltas (0 -> 1024) */
  double batch_norm;       /* args.batch_norm :26 (0 / 1): Neon BatchNorm after conv1..3 and fc4 (float32 only) */
} sdqn_net_cfg;

/* DeepQNetwork.__init__, deepqnetwork.py:16-75 (weights start at zero: inject with set_weights) */
int sdqn_net_create(sdqn_net_t* h, const sdqn_net_cfg* cfg);
int sdqn_net_destroy(sdqn_net_t h);
/* which: 0 online theta, 1 target theta-, 2 optimizer state (RMSProp s / Adam m / Adadelta E[g^2]),
 * 3 last gradient sum (get only), 4 second optimizer state (Adam v / Adadelta E[dx^2]).
 * layer: 0..4 = conv1, conv2, conv3, fc4, fc5.  With batch_norm also 5..8 = the BatchNorm layers after conv1, conv2,
 * conv3, fc4: 2*C floats [beta | gamma] (C = 32, 64, 64, 512); for those layers which = 5 / 6 reads or writes the
 * running statistics [gmean | gvar] of the online / target net.  Data in Neon layout:
 * conv (C*R*S, K) rows c*R*S+r*S+s; fc4 (512, 3136) nin in (K,P,Q) order; fc5 (A, 512).  */
int sdqn_net_layer_size(sdqn_net_t h, int layer, int64_t* n);
int sdqn_net_set_weights(sdqn_net_t h, int which, int layer, const float* w, int64_t n);
int sdqn_net_get_weights(sdqn_net_t h, int which, int layer, float* w, int64_t n);   /* sync */
/* the same in double: what a float64 network exchanges without a round trip through float (other networks convert) */
int sdqn_net_set_weights_f64(sdqn_net_t h, int which, int layer, const double* w, int64_t n);
int sdqn_net_get_weights_f64(sdqn_net_t h, int which, int layer, double* w, int64_t n);   /* sync */
/* DeepQNetwork.predict, deepqnetwork.py:174-186: states u8[B,4,84,84] -> q float[B,A] (sync) */
int sdqn_net_predict(sdqn_net_t h, const uint8_t* states, float* q_out);
int sdqn_net_predict_f64(sdqn_net_t h, const uint8_t* states, double* q_out);      /* float64 networks: Q-values in double */
/* acting path: Q-values of ONE state u8[4,84,84] -> float[A].  The reference pads the
 * current state to a full minibatch of zero rows only because Neon cannot change its batch size
 * (src/state_buffer.py:13-24, src/agent.py:55-61) and then uses row 0; this computes exactly that row. */
int sdqn_net_predict_one(sdqn_net_t h, const uint8_t* state, float* q_out);
/* Device-resident StateBuffer (src/state_buffer.py:3-27): the last `hist` screens of the
 * acting agent live in HBM, so each environment step uploads one 7 KB frame instead of a padded minibatch
 * (src/agent.py:55-61 builds u8[B,4,84,84] per step).  A host mirror backs getState()/getStateMinibatch(). */
typedef struct sdqn_statebuf_s* sdqn_statebuf_t;
int sdqn_statebuf_create(sdqn_statebuf_t* out, int screen_height, int screen_width, int history_length);
int sdqn_statebuf_destroy(sdqn_statebuf_t s);
int sdqn_statebuf_add(sdqn_statebuf_t s, const uint8_t* screen);   /* state_buffer.py:15-18: shift left, append */
int sdqn_statebuf_reset(sdqn_statebuf_t s);                         /* state_buffer.py:26-27 */
int sdqn_statebuf_get(sdqn_statebuf_t s, uint8_t* state_out);       /* host mirror copy u8[hist,H,W] (no sync) */
int sdqn_statebuf_read_device(sdqn_statebuf_t s, uint8_t* state_out);   /* test hook: D2H of the device window (sync) */
/* Q-values of the buffered state -> float[A] (sync): sdqn_net_predict_one without the state upload */
int sdqn_net_predict_state(sdqn_net_t h, sdqn_statebuf_t s, float* q_out);
/* agent.py:55-59: greedy action of the buffered state = first index of the maximal Q-value (np.argmax); q_out float[A] nullable */
int sdqn_net_act_greedy(sdqn_net_t h, sdqn_statebuf_t s, int* action, float* q_out);
/* One environment transition in one call (agent.py:48-85: `self.buf.add(screen)` [+ `self.mem.add(action, reward, screen, terminal)`,
 * replay_memory.py:26-34, when `r` is not NULL]); speculate != 0 also enqueues the acting forward of the NEW state, whose Q-values the next
 * sdqn_net_predict_state on the same buffer collects without launching anything (dropped if the buffer or the parameters change first). */
int sdqn_net_act_step(sdqn_net_t h, sdqn_statebuf_t sb, sdqn_replay_t r, const uint8_t* screen, int action, int64_t reward,
                      int terminal, int speculate);
/* Test / measurement hook of the one-launch acting forward (float32, no batch_norm, standard geometry; option "act_kernel"): one blocking
 * forward of the buffered state with per-workgroup phase stamps.  q_out float[A], stamps_out uint64[256][80] ({kind, clock} pairs; the last
 * word of a workgroup's row = its XCC id); either may be NULL. */
int sdqn_net_debug_act(sdqn_net_t h, sdqn_statebuf_t sb, float* q_out, unsigned long long* stamps_out);
/* DeepQNetwork.train, deepqnetwork.py:107-172, minibatch given as host arrays.
 * cost_out nullable: NULL -> the step itself is not waited for.  Buffer contract: when the call returns, all five arrays
 * are free to be overwritten — pageable arrays were copied into a pinned double buffer; pre / post that ARE a
 * ReplayMemory's pinned minibatch buffers (sdqn_replay_minibatch_ptrs) are uploaded in place and the call returns only
 * after that upload has completed (the step's kernels are already enqueued behind it).
 * Threading: handles are created, used and destroyed from ONE host thread (no internal locking). */
int sdqn_net_train_host(sdqn_net_t h, const uint8_t* pre, const uint8_t* actions, const int64_t* rewards,
                        const uint8_t* post, const uint8_t* terminals, float* cost_out);
/* How sdqn_net_train_host has been served on this handle: calls / calls that read the states from the device minibatch in place (no
 * 2 x batch x state H2D) / calls that uploaded nothing at all (small arrays equal to what the gather left on the device).  Any NULL. */
int sdqn_net_tuple_counters(sdqn_net_t h, int64_t* calls, int64_t* states_in_place, int64_t* nothing_uploaded);
/* same step with the minibatch gathered on the device straight from the replay ring:
 * fuses replay_memory.py:71-78 with deepqnetwork.py:94-100 (no u8 minibatch is materialised) */
int sdqn_net_train_replay(sdqn_net_t h, sdqn_replay_t r, const int64_t* idx_host, float* cost_out);
/* n_steps x { sample (replay_memory.py:54-68) ; train } without returning to Python:
 * the loop body of Agent.train, src/agent.py:108-114 */
int sdqn_net_train_many(sdqn_net_t h, sdqn_replay_t r, uint32_t mt[SDQN_MT_WORDS], int n_steps,
                        float* mean_cost /*nullable*/);
/* The same without waiting for the cost (deepqnetwork.py:168-172 when the stats callback can take the cost later): the stream copies the
 * call's cost sum into a pinned ring slot; sdqn_net_cost_collect(ticket) polls it (bounded) and returns the mean cost of the call's steps.
 * A ticket stays valid for 64 further deferred calls. */
int sdqn_net_train_many_deferred(sdqn_net_t h, sdqn_replay_t r, uint32_t mt[SDQN_MT_WORDS], int n_steps, int64_t* ticket);
int sdqn_net_cost_collect(sdqn_net_t h, int64_t ticket, float* mean_cost);
/* 32-bit MT19937 outputs the library has drawn since it was loaded: a caller that handed over a COPY of random.getstate()[1] advances its
 * own generator by the difference (random.getrandbits(32 * words)) instead of importing the 625 words back */
int sdqn_mt_words(uint64_t* words);
/* DeepQNetwork.update_target_network, deepqnetwork.py:102-105 */
int sdqn_net_update_target(sdqn_net_t h);
int sdqn_net_sync(sdqn_net_t h);
/* The two halves of a data-parallel step without a communicator (the arithmetic every rank performs
 * around the all-reduce).  With option "grad_only" = 1 a train step ends after the local gradient SUMS are in the flat
 * gradient buffer (readable / writable per layer with which = 3); sdqn_net_apply_update then runs the optimizer on
 * whatever that buffer holds with divisor bsz (A9: grad / be.bsz, deepqnetwork.py:165 — nranks * batch_size under DP). */
int sdqn_net_apply_update(sdqn_net_t h, double bsz);
/* float16 mode, data parallel (no reference counterpart: deepqnetwork.py:46-48 disables Neon's DP; SURVEY.md §8e "fp16
 * config: all-reduce fp16 grads with fp32 accumulation in RMSProp"): the two passes that bracket the half all-reduce, callable
 * on their own so the exchange itself can be done by the caller.  to_half: the flat fp32 gradient sums * 2^k as IEEE half
 * (uint16 bit patterns; k = device-side payload scale).  from_half: a summed half payload back into the fp32 gradient buffer
 * (/ 2^k); a non-finite value raises the overflow flag and the next sdqn_net_apply_update skips the step (parameters and
 * optimizer state untouched), counts it (sdqn_net_overflow_steps) and, in dynamic mode, halves the scale; 200 clean steps
 * double it.  n = number of values of the flat buffer (sum of sdqn_net_layer_size over all layers). */
int sdqn_net_grad_to_half(sdqn_net_t h, uint16_t* half_out, int64_t n);
int sdqn_net_grad_from_half(sdqn_net_t h, const uint16_t* half_in, int64_t n);
/* {overflow flag of the last from-half pass, log2 of the payload scale, clean steps since the scale last moved}; sync */
int sdqn_net_half_payload_state(sdqn_net_t h, int* flag, int* scale_log2, int* clean_steps);
/* q-values of the last train step: preq float[B,A] (online, prestates), maxpostq float[B] (sync) */
int sdqn_net_last_q(sdqn_net_t h, float* preq, float* maxpostq);
int sdqn_net_train_iterations(sdqn_net_t h, int64_t* n);     /* deepqnetwork.py:168 */
/* float16 mode under data parallel: the gradient is all-reduced as IEEE half with a dynamic power-of-two payload scale
 * (device-side: halved on overflow, doubled after 200 clean steps; fp32 accumulation in the optimizer; options "dp_half",
 * "dp_half_scale_log2" = fixed scale).  A step whose summed half gradient is not finite is skipped on every rank;
 * this returns how many were (sync).  Always 0 in float32 mode. */
int sdqn_net_overflow_steps(sdqn_net_t h, int64_t* n);
/* the `epoch` argument of DeepQNetwork.train (deepqnetwork.py:107,165): Neon's Adam bias-corrects with t = epoch + 1 */
int sdqn_net_set_epoch(sdqn_net_t h, int epoch);

/* Which launches a train step of this handle is made of, and how its optimizer pass runs — a function of (batch regime, datatype,
 * batch_norm, data-parallel form, option fused_launches) only (tests/test_step_structure.py enumerates the combinations):
 *   structure: 0 fused (fc4_dgrad | bwd3 | bwd2 | bwd1)   1 float16 block-tile (fc4_dgrad | conv3_dgrad | conv2_dgrad | wgrads | bwd1)
 *              2 fused with fc4's all-reduce + update overlapped on the second communicator   3 one launch per problem   4 generic path
 *   update:    0 one update launch   1 serial data parallel (local sums, all-reduce, apply)   2 overlapped data parallel   3 grad_only */
int sdqn_net_step_structure(sdqn_net_t h, int* structure, int* update);
/* options: "grad_only" (see sdqn_net_apply_update), "keep_gradients" (1: the fc4 gradient is materialised and readable with which=3; 0 (default): on one
 * GPU RMSProp of fc4 is fused into the wgrad epilogue), "two_streams" (0 default; 1: wgrad kernels overlap the dgrad chain on a side stream), "fused_launches" (1 default:
 * independent backward stages share one grid), "xcd_map" (0 default = only where it wins time: conv1/conv2/fc4 forward; 1: the
 * XCD-contiguous workgroup->tile map for every launch), "dp_overlap" (BEFORE sdqn_dp_init: -1 default = auto — second communicator for
 * nranks >= 2, activated by sdqn_dp_probe + vote + sdqn_dp_set_overlap; 1 forced on; 0 one all-reduce on the library stream), "dp_sync_replicas" (1 default: sdqn_dp_init broadcasts rank 0's online net,
 * target net and optimizer state so every learner starts from — and keeps — the same network; 0 BEFORE sdqn_dp_init: keep own), "profile_every" (N: sdqn_net_profile times every N-th
 * launch), "profile_mode" (1 default: the launch records its own dispatch-packet begin / end timestamps into the event pair through hipExtLaunchKernel —
 * what rocprofv3 --kernel-trace reports, nothing added to the queue; 0: hipEventRecord markers around the launch, ~2.6 us more per launch), "nw:<kernel id>" / "xcd:<kernel id>" / "f4_share3" / "f4_share2" (tuning hooks).
 * Round-3 switches of the default fp32 / float16 step, every one bit-identical to its alternative unless noted (tools/exp/README.md has the
 * measurements): "wt" (bit mask, default 511: write-through epilogue stores per launch), "conv1_bf16" / "conv1w_bf16" (1 default: conv1
 * forward / weight gradient on packed-bf16 MFMA with an exact 3-way split; 0: the fp32-MFMA engine — last bits differ), "conv3_c36" (1 default:
 * conv3 forward on 36-deep K-chunks; last bits differ), "r3_xcd" (tile maps of the two conv1 kernels), and three in-launch hand-offs that
 * measured slower than kernel boundaries and default to 0: "f4w_early" (fc4_wgrad inside the fc4_dgrad launch), "fuse_upd" (update(i) +
 * conv1(i+1)), "head_f4d" (head + fc4_dgrad).  A float64 / non-84x84x4 network (generic path) accepts and ignores the tuning options.
 * Round 4: "bt" (1 default: batch_size >= 128 runs on the block-tile engine, gemm_engine_bt.h; 0: the latency engine's launch forms),
 * "bt:<kernel id>" (menu entry of that launch: 0 built-in block shape, -1 latency engine / previous kernel, n > 0 other shapes — tuning
 * surface of tools/sweep_bt.py), "tps:<layer>" (K chunks per weight-gradient slab), "act_kernel" (1 default where available — float32, no
 * batch_norm, 84x84x4: the acting forward of sdqn_net_predict_state / _predict_one is ONE launch, sdqn_act.hip; 0: the five batched
 * forward kernels at batch 1; test hook "act_inject_failure": the next such launch delivers nothing, which exercises the host's fall-back to the
 * five launches).  The step structures that were built, tested and measured SLOWER — "hoist", "f4w_early", "fuse_upd",
 * "head_f4d", "two_streams", "fwd_rb", "bwd_order", "rb:<id>", "bt_x", "btx:<id>", "bt_planes" — left the library in round 5
 * (tools/exp/experiments_r04.patch re-creates the last tree that contained them, results in tools/exp/README.md); a non-zero value for one of
 * these names is refused with SDQN_ERR_ARG, zero is accepted and does nothing. */
int sdqn_net_set_option(sdqn_net_t h, const char* name, int value);

/* test hook: raw read-back of an internal device buffer ("a1","a2","a3","a4","d4","d3p","d2p","d1","q",
 * "dq","g","theta","cost_terms"; internal layouts documented in simple_dqn_amd/csrc/problems.h) */
int sdqn_net_debug_read(sdqn_net_t h, const char* name, float* out, int64_t n);

/* per-kernel device timing (HIP events on the library stream; see option "profile_mode"), for bench.py's roofline leg.
 * kernel < 0 brackets every kernel of the step, otherwise only that kernel id. */
int sdqn_net_profile(sdqn_net_t h, int enable, int kernel);
int sdqn_net_profile_count(int* n_kernels);
int sdqn_net_profile_read(sdqn_net_t h, int kernel, const char** name, double* total_ms, int64_t* launches);
int sdqn_net_profile_reset(sdqn_net_t h);

/* ---- data parallel: one learner per GPU, one RCCL all-reduce of the flat gradient per step ---
 * (no reference counterpart: deepqnetwork.py:46-48 disables Neon's DP) */
int sdqn_dp_unique_id(const char* rccl_path, char id[128]);
int sdqn_dp_init(sdqn_net_t h, const char* rccl_path, const char id[128], int rank, int nranks);
int sdqn_dp_shutdown(sdqn_net_t h);
/* what RCCL reports about the live communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice; -1 without one)
 * and the device the library is bound to: lets a multi-GPU record prove that RCCL spanned N ranks on N devices */
int sdqn_dp_info(sdqn_net_t h, int* comm_ranks, int* comm_rank, int* comm_device, int* bound_device);
/* Overlapped form of the step (fc4's 95 % of the gradient all-reduced + applied on a second communicator / stream under the rest of the
 * step; no reference counterpart, deepqnetwork.py:46-48).  With option "dp_overlap" = -1 (auto, default) sdqn_dp_init creates the second
 * communicator for nranks >= 2 but leaves the form INACTIVE; the caller then
 *   1. runs sdqn_dp_probe on every rank (two rounds of the overlapped collectives with bounded waits; *ok = 1: drained in time here),
 *   2. agrees over its control plane (all ranks' ok AND-ed),
 *   3. calls sdqn_dp_set_overlap(agreed) on EVERY rank: 1 activates the form, 0 tears the second communicator down (ncclCommAbort on a
 *      rank whose probe never drained) and the serial form — one all-reduce on the library stream — runs.
 * sdqn_dp_form reports what runs: 0 no communicator, 1 serial, 2 overlapped; *probe = -1 not probed / 0 timed out / 1 ok;
 * *second_comm = 1 while the second communicator exists. */
int sdqn_dp_probe(sdqn_net_t h, int timeout_ms, int inject_timeout, int* ok);
int sdqn_dp_set_overlap(sdqn_net_t h, int on);
int sdqn_dp_form(sdqn_net_t h, int* form, int* probe, int* second_comm);

#ifdef __cplusplus
}
#endif
#endif /* SDQN_H */
