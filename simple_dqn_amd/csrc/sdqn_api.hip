// sdqn_api.hip — host side of libsdqn_hip.so: handles, memory, step orchestration, C ABI (include/sdqn.h).
// No CPU fallback lives here: every device entry point needs a HIP device and fails loudly without one.
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

// ------------------------------------------------------------------------------------------------
// Step structures that were built, tested bit-identical and measured SLOWER than the default (tools/exp/README.md) live behind the
// compile-time switch SDQN_EXPERIMENTS (`make experiments` -> libsdqn_hip_exp.so, never loaded by the package unless
// SDQN_LIB_VARIANT=experiments): hoist, f4w_early, fuse_upd, head_f4d, two_streams, fwd_rb, bwd_order, rb:<id>.  In the product build the
// constant below is false, every branch they guard folds away and their kernels are not compiled: one step structure per
// (B regime, datatype).
#ifdef SDQN_EXPERIMENTS
static constexpr bool EXPERIMENTS = true;
#else
static constexpr bool EXPERIMENTS = false;
#endif
#define EXP_OPTION_REFUSED(NAME) do { set_error("option %s is an experiment (measured slower than the default step; tools/exp/README.md): " \
                                                "build the library with `make -C simple_dqn_amd/csrc experiments` and load it with SDQN_LIB_VARIANT=experiments", NAME); \
                                      return SDQN_ERR_ARG; } while (0)
static constexpr int Q_SLOT_FLOATS = 8 * ACT_Q_STRIDE;   // one slot: [A] Q-values (head kernel) or 8 stripe partials [8][ACT_Q_STRIDE] (one-launch forward)
static constexpr int COST_RING = 64;
static constexpr int Q_SLOTS = 8;       // host-mapped Q-value slots of the acting path (sdqn_net_predict_state)
static thread_local std::string g_err;
static void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_err = buf;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return SDQN_ERR_HIP; } } while (0)
#define ARGCHK(c, ...) do { if (!(c)) { set_error(__VA_ARGS__); return SDQN_ERR_ARG; } } while (0)

static hipStream_t g_stream = nullptr;      // the library stream: everything is ordered on it
static hipStream_t g_side = nullptr;        // side stream: weight-gradient kernels run beside the dgrad chain
static hipStream_t g_comm = nullptr;        // data parallel: the fc4 gradient all-reduce + fc4 update run here, beside the compute stream
static hipEvent_t g_ev[5];                  // fork/join events between the two (timing disabled)
static int g_dev = -1;                      // device the library streams live on (bound by the first device call)
static int ensure_stream() {
  if (g_stream) return SDQN_OK;
  int n = 0;
  HIPCHK(hipGetDeviceCount(&n));
  if (n <= 0) { set_error("no HIP device visible (libsdqn_hip has no CPU path)"); return SDQN_ERR_HIP; }
  HIPCHK(hipGetDevice(&g_dev));
  HIPCHK(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&g_comm, hipStreamNonBlocking));
  for (int i = 0; i < 5; ++i) HIPCHK(hipEventCreateWithFlags(&g_ev[i], hipEventDisableTiming));
  return SDQN_OK;
}
#define STREAMCHK() do { int r_ = ensure_stream(); if (r_) return r_; } while (0)

extern "C" const char* sdqn_last_error(void) { return g_err.c_str(); }
extern "C" int sdqn_version(void) { return 100; }
extern "C" int sdqn_device_count(int* n) { ARGCHK(n, "n is NULL"); HIPCHK(hipGetDeviceCount(n)); return SDQN_OK; }
extern "C" int sdqn_set_device(int dev) {
  // one device per process (one process per GPU): the first device call binds the library streams; asking for the
  // bound device again is a no-op, asking for another one is an error instead of a silent run on the wrong GPU
  if (g_stream) {
    if (dev == g_dev) return SDQN_OK;
    set_error("libsdqn_hip is already bound to device %d (asked for %d): one device per process", g_dev, dev);
    return SDQN_ERR_STATE;
  }
  int n = 0;
  HIPCHK(hipGetDeviceCount(&n));
  ARGCHK(dev >= 0 && dev < n, "device_id %d out of range (%d visible)", dev, n);
  HIPCHK(hipSetDevice(dev));
  return SDQN_OK;
}
extern "C" int sdqn_get_device(int* dev) {
  ARGCHK(dev, "dev is NULL"); STREAMCHK(); *dev = g_dev; return SDQN_OK;
}
extern "C" int sdqn_device_sync(void) { STREAMCHK(); HIPCHK(hipStreamSynchronize(g_stream)); return SDQN_OK; }

// ---- sampler (pure host) -------------------------------------------------------------------------
extern "C" int sdqn_mt_seed(uint32_t* mt, uint64_t seed) { ARGCHK(mt, "mt is NULL"); MT::seed(mt, seed); return SDQN_OK; }
extern "C" int sdqn_mt_randint(uint32_t* mt, int64_t a, int64_t b, int64_t* out) {
  ARGCHK(mt && out && b >= a, "bad randint arguments");
  ARGCHK(mt[624] <= 624, "corrupt MT state (position %u)", mt[624]);
  MT m(mt); *out = m.randint(a, b); return SDQN_OK;
}
static int sample_checked(uint32_t* mt, const uint8_t* terminals, int64_t count, int64_t current, int hist,
                          int batch, int64_t* idx_out, int64_t* draws_out) {
  ARGCHK(mt && terminals && idx_out, "NULL argument");
  ARGCHK(mt[624] <= 624, "corrupt MT state (position %u)", mt[624]);
  ARGCHK(count > hist, "replay memory must hold more than history_length frames (count=%lld)", (long long)count);  // :52
  ARGCHK(batch > 0 && hist > 0 && current >= 0, "bad sampler arguments");
  // guard against a ring with no admissible index (the reference would spin forever)
  bool any_ok = false;
  for (int64_t i = hist; i < count && !any_ok; ++i) {
    if (i >= current && i - hist < current) continue;
    bool t = false;
    for (int64_t k = i - hist; k < i; ++k) t |= terminals[k] != 0;
    any_ok = !t;
  }
  ARGCHK(any_ok, "no admissible index in the ring (every window straddles the write pointer or a terminal)");
  int64_t d = sample_indices(mt, terminals, count, current, hist, batch, idx_out);
  if (draws_out) *draws_out = d;
  return SDQN_OK;
}
extern "C" int sdqn_sample_indices(uint32_t* mt, const uint8_t* terminals, int64_t count, int64_t current,
                                   int hist, int batch, int64_t* idx_out, int64_t* draws_out) {
  return sample_checked(mt, terminals, count, current, hist, batch, idx_out, draws_out);
}

// ---- replay -----------------------------------------------------------------------------------------
static const int NSLOT = 64;     // pinned index slots: kernels read the sampled indexes zero-copy
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
};

static std::vector<sdqn_replay_s*> g_replays;      // live handles: sdqn_net_train_host recognises their pinned minibatch buffers
static int replay_free(sdqn_replay_s* r) {
  if (!r) return SDQN_OK;
  for (size_t i = 0; i < g_replays.size(); ++i) if (g_replays[i] == r) { g_replays.erase(g_replays.begin() + i); break; }
  if (g_stream) hipStreamSynchronize(g_stream);
  if (!(r->flags & SDQN_REPLAY_ZERO_COPY)) { hipFree(r->d_ring); hipFree(r->d_meta); }
  hipFree(r->d_pre); hipFree(r->d_rew);                   // (d_post / d_act / d_term live inside these two blocks, likewise on the host)
  hipHostFree(r->screens); hipHostFree(r->actions); hipHostFree(r->rewards); hipHostFree(r->terminals);
  hipHostFree(r->h_meta); hipHostFree(r->h_pre); hipHostFree(r->h_rew); hipHostFree(r->h_idx);
  for (int i = 0; i < NSLOT; ++i) if (r->slot_ev[i]) hipEventDestroy(r->slot_ev[i]);
  if (r->mb_upload_ev) hipEventDestroy(r->mb_upload_ev);
  delete r;
  return SDQN_OK;
}

extern "C" int sdqn_replay_create(sdqn_replay_t* out, int64_t size, int H, int W, int hist, int batch, int flags) {
  ARGCHK(out, "handle pointer is NULL");
  ARGCHK(size > hist && batch > 0, "bad replay geometry (size=%lld, batch=%d)", (long long)size, batch);
  ARGCHK(H > 0 && W > 0 && hist > 0 && H <= 4096 && W <= 4096 && hist <= 64, "bad screen geometry %dx%d, history_length %d", H, W, hist);
  if (!flags) flags = SDQN_REPLAY_HBM_MIRROR;
  STREAMCHK();
  sdqn_replay_s* r = new sdqn_replay_s();
  memset(r->slot_ev, 0, sizeof r->slot_ev); memset(r->slot_busy, 0, sizeof r->slot_busy); memset(r->slot_cover, 0, sizeof r->slot_cover); r->npending = 0;
  r->size = size; r->H = H; r->W = W; r->hist = hist; r->B = batch; r->flags = flags;
  r->frame = (int64_t)H * W; r->state = r->frame * hist; r->tuned_geom = (H == H0 && W == W0 && hist == C0);
  const int64_t FRAME = r->frame, STATE = r->state;     // (shadow the 84 x 84 x 4 constants of problems.h in this function)
  const unsigned hf = hipHostMallocMapped | hipHostMallocPortable;
#define RCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s -> %s", #x, hipGetErrorString(e_)); replay_free(r); return SDQN_ERR_HIP; } } while (0)
  RCHK(hipHostMalloc((void**)&r->screens, (size_t)size * FRAME + SRC_PAD, hf));     // + slack: conv1_wgrad's 16-byte patch loads (problems.h)
  RCHK(hipHostMalloc((void**)&r->actions, (size_t)size, hf));
  RCHK(hipHostMalloc((void**)&r->rewards, (size_t)size * 8, hf));
  RCHK(hipHostMalloc((void**)&r->terminals, (size_t)size, hf));
  RCHK(hipHostMalloc((void**)&r->h_meta, (size_t)size * sizeof(MetaRec), hf));
  memset(r->h_meta, 0, (size_t)size * sizeof(MetaRec));
  if (flags & SDQN_REPLAY_ZERO_COPY) {
    RCHK(hipHostGetDevicePointer((void**)&r->d_ring, r->screens, 0));
    RCHK(hipHostGetDevicePointer((void**)&r->d_meta, r->h_meta, 0));
  } else {
    RCHK(hipMalloc((void**)&r->d_ring, (size_t)size * FRAME + SRC_PAD));
    RCHK(hipMalloc((void**)&r->d_meta, (size_t)size * sizeof(MetaRec)));
    RCHK(hipMemsetAsync(r->d_meta, 0, (size_t)size * sizeof(MetaRec), g_stream));
  }
  const size_t sb = (size_t)batch * STATE;
  // the gathered minibatch as two blocks — [pre | post] and [rewards 8 B | actions B | terminals B] — on the device and in pinned host
  // memory alike: getMinibatch() brings it down with two copies and the tuple API sends it back up with two (every copy is a stream packet)
  RCHK(hipMalloc((void**)&r->d_pre, 2 * sb + SRC_PAD)); r->d_post = r->d_pre + sb;     // (+ slack: the step may read it in place like a staging buffer)
  RCHK(hipMalloc((void**)&r->d_rew, (size_t)batch * 10));
  r->d_act = reinterpret_cast<uint8_t*>(r->d_rew) + (size_t)batch * 8; r->d_term = r->d_act + batch;
  RCHK(hipHostMalloc((void**)&r->h_pre, 2 * sb, hf)); r->h_post = r->h_pre + sb;
  RCHK(hipHostMalloc((void**)&r->h_rew, (size_t)batch * 10, hf));
  r->h_act = reinterpret_cast<uint8_t*>(r->h_rew) + (size_t)batch * 8; r->h_term = r->h_act + batch;
  RCHK(hipHostMalloc((void**)&r->h_idx, (size_t)NSLOT * batch * 8, hf));
  RCHK(hipHostGetDevicePointer((void**)&r->d_idx_view, r->h_idx, 0));
  for (int i = 0; i < NSLOT; ++i) RCHK(hipEventCreateWithFlags(&r->slot_ev[i], hipEventDisableTiming));
  RCHK(hipEventCreateWithFlags(&r->mb_upload_ev, hipEventDisableTiming));
#undef RCHK
  g_replays.push_back(r);
  *out = r;
  return SDQN_OK;
}
extern "C" int sdqn_replay_destroy(sdqn_replay_t r) { return replay_free(r); }

extern "C" int sdqn_replay_host_ptrs(sdqn_replay_t r, uint8_t** screens, uint8_t** actions, int64_t** rewards, uint8_t** terminals) {
  ARGCHK(r, "NULL handle");
  if (screens) *screens = r->screens; if (actions) *actions = r->actions;
  if (rewards) *rewards = r->rewards; if (terminals) *terminals = r->terminals;
  return SDQN_OK;
}
extern "C" int sdqn_replay_minibatch_ptrs(sdqn_replay_t r, uint8_t** pre, uint8_t** post, uint8_t** actions, int64_t** rewards, uint8_t** terminals) {
  ARGCHK(r, "NULL handle");
  if (pre) *pre = r->h_pre; if (post) *post = r->h_post; if (actions) *actions = r->h_act;
  if (rewards) *rewards = r->h_rew; if (terminals) *terminals = r->h_term;
  return SDQN_OK;
}

extern "C" int sdqn_replay_add(sdqn_replay_t r, int action, int64_t reward, const uint8_t* screen, int terminal) {
  ARGCHK(r && screen, "NULL argument");
  const int64_t FRAME = r->frame;
  const int64_t c = r->current;                                   // replay_memory.py:29-32
  r->actions[c] = (uint8_t)action; r->rewards[c] = reward; r->terminals[c] = terminal ? 1 : 0;
  memcpy(r->screens + c * FRAME, screen, FRAME);
  MetaRec& m = r->h_meta[c];
  m.reward = reward; m.action = (uint8_t)action; m.terminal = terminal ? 1 : 0;
  if (!(r->flags & SDQN_REPLAY_ZERO_COPY)) {
    HIPCHK(hipMemcpyAsync(r->d_ring + c * FRAME, r->screens + c * FRAME, FRAME, hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipMemcpyAsync(r->d_meta + c, &m, sizeof(MetaRec), hipMemcpyHostToDevice, g_stream));
  }
  if (c + 1 > r->count) r->count = c + 1;                         // :33
  r->current = (c + 1) % r->size;                                 // :34
  return SDQN_OK;
}
extern "C" int sdqn_replay_get_state(sdqn_replay_t r, int64_t* count, int64_t* current) {
  ARGCHK(r, "NULL handle"); if (count) *count = r->count; if (current) *current = r->current; return SDQN_OK;
}
extern "C" int sdqn_replay_set_state(sdqn_replay_t r, int64_t count, int64_t current) {
  ARGCHK(r && count >= 0 && count <= r->size && current >= 0 && current < r->size, "bad count/current");
  r->count = count; r->current = current; return SDQN_OK;
}
extern "C" int sdqn_replay_upload(sdqn_replay_t r, int64_t first, int64_t n) {
  ARGCHK(r && first >= 0 && n >= 0 && first + n <= r->size, "bad upload range");
  const int64_t FRAME = r->frame;
  for (int64_t i = first; i < first + n; ++i) {
    MetaRec& m = r->h_meta[i];
    m.reward = r->rewards[i]; m.action = r->actions[i]; m.terminal = r->terminals[i] ? 1 : 0;
  }
  if (!(r->flags & SDQN_REPLAY_ZERO_COPY) && n > 0) {
    HIPCHK(hipMemcpyAsync(r->d_ring + first * FRAME, r->screens + first * FRAME, (size_t)n * FRAME, hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipMemcpyAsync(r->d_meta + first, r->h_meta + first, (size_t)n * sizeof(MetaRec), hipMemcpyHostToDevice, g_stream));
  }
  HIPCHK(hipStreamSynchronize(g_stream));
  return SDQN_OK;
}
// metadata only (actions / rewards / terminals of slots [first, first + n) re-packed and sent): 16 B per slot instead of 7 KB
extern "C" int sdqn_replay_upload_meta(sdqn_replay_t r, int64_t first, int64_t n) {
  ARGCHK(r && first >= 0 && n >= 0 && first + n <= r->size, "bad upload range");
  for (int64_t i = first; i < first + n; ++i) {
    MetaRec& m = r->h_meta[i];
    m.reward = r->rewards[i]; m.action = r->actions[i]; m.terminal = r->terminals[i] ? 1 : 0;
  }
  if (!(r->flags & SDQN_REPLAY_ZERO_COPY) && n > 0)
    HIPCHK(hipMemcpyAsync(r->d_meta + first, r->h_meta + first, (size_t)n * sizeof(MetaRec), hipMemcpyHostToDevice, g_stream));
  HIPCHK(hipStreamSynchronize(g_stream));
  return SDQN_OK;
}
extern "C" int sdqn_replay_sample(sdqn_replay_t r, uint32_t* mt, int64_t* idx_out, int64_t* draws_out) {
  ARGCHK(r, "NULL handle");
  return sample_checked(mt, r->terminals, r->count, r->current, r->hist, r->B, idx_out, draws_out);
}

// take the next pinned index slot (waiting for its previous consumer), fill it, return its device alias
static int replay_flush_pending(sdqn_replay_s* r) {          // one event for every slot released since the last one
  if (r->npending == 0) return SDQN_OK;
  const int last = r->pending[r->npending - 1];
  HIPCHK(hipEventRecord(r->slot_ev[last], g_stream));
  for (int i = 0; i < r->npending; ++i) { r->slot_cover[r->pending[i]] = last; r->slot_busy[r->pending[i]] = true; }
  r->npending = 0;
  return SDQN_OK;
}
static int replay_push_idx(sdqn_replay_s* r, const int64_t* idx, int* slot_out, const int64_t** dev) {
  const int s = r->next_slot; r->next_slot = (s + 1) % NSLOT;
  for (int i = 0; i < r->npending; ++i)            // (cannot happen inside train_many: a slot comes round after NSLOT pushes, a batch is 16)
    if (r->pending[i] == s) { int rc_ = replay_flush_pending(r); if (rc_) return rc_; break; }
  if (r->slot_busy[s]) { HIPCHK(hipEventSynchronize(r->slot_ev[r->slot_cover[s]])); r->slot_busy[s] = false; }
  int64_t* dst = r->h_idx + (size_t)s * r->B;
  for (int i = 0; i < r->B; ++i) {
    ARGCHK(idx[i] >= r->hist && idx[i] < r->count, "index %lld out of range (count %lld)", (long long)idx[i], (long long)r->count);
    dst[i] = idx[i];
  }
  *slot_out = s; *dev = r->d_idx_view + (size_t)s * r->B;
  return SDQN_OK;
}
static int replay_release_idx(sdqn_replay_s* r, int slot) {
  HIPCHK(hipEventRecord(r->slot_ev[slot], g_stream)); r->slot_busy[slot] = true; r->slot_cover[slot] = slot; return SDQN_OK;
}
// The train paths release their index slots in batches: an event record is a packet of its own in the dependent launch chain and
// costs ~2.8 us of GPU time — one per step took 3.9 % off the step rate (12 998 -> 13 500 steps/s, tools/exp/README.md) and made every
// call's first ~20 steps slow.  One record per SLOT_BATCH releases covers all the slots used
// since the previous one; a slot is reused NSLOT = 64 pushes after its use, so its covering event is recorded long before.
static const int SLOT_BATCH = 16;
static int replay_release_idx_batched(sdqn_replay_s* r, int slot, bool flush) {
  r->pending[r->npending++] = slot;
  if (flush || r->npending >= SLOT_BATCH) return replay_flush_pending(r);
  return SDQN_OK;
}

static GatherArgs gather_args(sdqn_replay_s* r, const int64_t* didx) {
  r->mb_dev_gen++;                                  // (every launch built from these arguments overwrites the device minibatch)
  GatherArgs g; g.ring = r->d_ring; g.meta = r->d_meta; g.idx = didx; g.pre = r->d_pre; g.post = r->d_post;
  g.actions = r->d_act; g.rewards = r->d_rew; g.terminals = r->d_term; g.B = r->B; return g;
}
static int replay_gather_generic(sdqn_replay_s* r, const int64_t* didx) {      // any geometry (generic_net.hip)
  r->mb_dev_gen++;
  GatherGenericArgs g; g.ring = r->d_ring; g.meta = r->d_meta; g.idx = didx; g.pre = r->d_pre; g.post = r->d_post;
  g.actions = r->d_act; g.rewards = r->d_rew; g.terminals = r->d_term; g.B = r->B; g.hist = r->hist; g.frame = r->frame;
  HIPCHK(launch_gather_generic(g, g_stream));
  return SDQN_OK;
}
extern "C" int sdqn_replay_gather(sdqn_replay_t r, const int64_t* idx_host) {
  ARGCHK(r && idx_host, "NULL argument");
  if (!r->tuned_geom) {
    int slot; const int64_t* didx; int rc = replay_push_idx(r, idx_host, &slot, &didx); if (rc) return rc;
    rc = replay_gather_generic(r, didx); if (rc) return rc;
    return replay_release_idx(r, slot);
  }
  if (r->B <= 256) {               // the indexes ride in the kernel arguments: no pinned slot, no release event (sdqn_kernels.hip)
    for (int i = 0; i < r->B; ++i)
      ARGCHK(idx_host[i] >= r->hist && idx_host[i] < r->count, "index %lld out of range (count %lld)", (long long)idx_host[i], (long long)r->count);
    HIPCHK(launch_gather(gather_args(r, nullptr), g_stream, idx_host));
    return SDQN_OK;
  }
  int slot; const int64_t* didx; int rc = replay_push_idx(r, idx_host, &slot, &didx); if (rc) return rc;
  HIPCHK(launch_gather(gather_args(r, didx), g_stream));
  return replay_release_idx(r, slot);
}
extern "C" int sdqn_replay_minibatch_to_host(sdqn_replay_t r) {
  ARGCHK(r, "NULL handle");
  const size_t sb = (size_t)r->B * r->state;
  HIPCHK(hipMemcpyAsync(r->h_pre, r->d_pre, 2 * sb, hipMemcpyDeviceToHost, g_stream));                 // [pre | post]
  HIPCHK(hipMemcpyAsync(r->h_rew, r->d_rew, (size_t)r->B * 10, hipMemcpyDeviceToHost, g_stream));      // [rewards | actions | terminals]
  HIPCHK(hipStreamSynchronize(g_stream));
  r->mb_host_gen = r->mb_dev_gen;
  return SDQN_OK;
}
extern "C" int sdqn_replay_declare_minibatch_clean(sdqn_replay_t r) {
  ARGCHK(r, "NULL handle"); r->mb_clean_declared = true; r->mb_clean_on_device = false; return SDQN_OK;
}
extern "C" int sdqn_replay_minibatch_gen(sdqn_replay_t r, uint64_t* device_gen, uint64_t* host_gen) {
  ARGCHK(r, "NULL handle");
  if (device_gen) *device_gen = r->mb_dev_gen;
  if (host_gen) *host_gen = r->mb_host_gen;
  return SDQN_OK;
}
extern "C" int sdqn_replay_declare_minibatch_on_device(sdqn_replay_t r, uint64_t gen) {
  ARGCHK(r, "NULL handle");
  r->mb_clean_declared = true; r->mb_clean_on_device = gen == r->mb_dev_gen;       // (a stale generation: the host buffers are uploaded as always)
  return SDQN_OK;
}
extern "C" int sdqn_replay_bench_gather(sdqn_replay_t r, const int64_t* idx_host, int iters, float* ms_per_launch) {
  ARGCHK(r && idx_host && iters > 0 && ms_per_launch, "bad arguments");
  ARGCHK(r->tuned_geom, "bench_gather times the 84x84x4 kernel");
  int slot; const int64_t* didx; int rc = replay_push_idx(r, idx_host, &slot, &didx); if (rc) return rc;
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  GatherArgs g = gather_args(r, didx);
  const int64_t* inl = r->B <= 256 ? idx_host : nullptr;                     // the launch form sdqn_replay_gather uses
  HIPCHK(launch_gather(g, g_stream, inl));                                   // warm
  HIPCHK(hipEventRecord(e0, g_stream));
  for (int i = 0; i < iters; ++i) HIPCHK(launch_gather(g, g_stream, inl));
  HIPCHK(hipEventRecord(e1, g_stream));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  *ms_per_launch = ms / iters;
  hipEventDestroy(e0); hipEventDestroy(e1);
  return replay_release_idx(r, slot);
}
// As above with a DIFFERENT index set per launch (idx_host = nsets x batch_size indexes, cycled): getMinibatch() never gathers the
// same states twice in a row, and a repeated set is served from L2 / MALL after its first launch instead of HBM.
extern "C" int sdqn_replay_bench_gather_sets(sdqn_replay_t r, const int64_t* idx_host, int nsets, int iters, float* ms_per_launch) {
  ARGCHK(r && idx_host && nsets > 0 && iters > 0 && ms_per_launch, "bad arguments");
  ARGCHK(r->tuned_geom, "bench_gather_sets times the 84x84x4 kernel");
  const int B = r->B;
  for (int64_t i = 0; i < (int64_t)nsets * B; ++i)
    ARGCHK(idx_host[i] >= r->hist && idx_host[i] < r->count, "index %lld out of range (count %lld)", (long long)idx_host[i], (long long)r->count);
  int64_t* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, (size_t)nsets * B * sizeof(int64_t)));
  HIPCHK(hipMemcpy(d, idx_host, (size_t)nsets * B * sizeof(int64_t), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  GatherArgs g = gather_args(r, d);
  HIPCHK(launch_gather(g, g_stream));                                        // warm (code object, not data: set 0 comes round last)
  HIPCHK(hipEventRecord(e0, g_stream));
  for (int i = 0; i < iters; ++i) {
    const size_t set = (size_t)((i + 1) % nsets) * B;
    g.idx = d + set; HIPCHK(launch_gather(g, g_stream, B <= 256 ? idx_host + set : nullptr));
  }
  HIPCHK(hipEventRecord(e1, g_stream));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  *ms_per_launch = ms / iters;
  hipEventDestroy(e0); hipEventDestroy(e1);
  HIPCHK(hipFree(d));
  return SDQN_OK;
}

// ---- RCCL (resolved at run time from the library the process already uses) ------------------------------
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
static Rccl g_rccl;
static int rccl_load(const char* path) {
  if (g_rccl.lib) return SDQN_OK;
  const char* p = (path && *path) ? path : "librccl.so.1";
  void* lib = dlopen(p, RTLD_NOW | RTLD_LOCAL);
  if (!lib) { set_error("dlopen(%s) failed: %s", p, dlerror()); return SDQN_ERR_RCCL; }
  g_rccl.GetUniqueId = (int (*)(void*))dlsym(lib, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(void**, int, Id128, int))dlsym(lib, "ncclCommInitRank");
  g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclAllReduce");
  g_rccl.Broadcast = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclBroadcast");
  g_rccl.CommDestroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
  g_rccl.CommSplit = (int (*)(void*, int, int, void**, void*))dlsym(lib, "ncclCommSplit");
  g_rccl.CommAbort = (int (*)(void*))dlsym(lib, "ncclCommAbort");
  g_rccl.GroupStart = (int (*)())dlsym(lib, "ncclGroupStart");
  g_rccl.GroupEnd = (int (*)())dlsym(lib, "ncclGroupEnd");
  g_rccl.CommCount = (int (*)(void*, int*))dlsym(lib, "ncclCommCount");
  g_rccl.CommUserRank = (int (*)(void*, int*))dlsym(lib, "ncclCommUserRank");
  g_rccl.CommCuDevice = (int (*)(void*, int*))dlsym(lib, "ncclCommCuDevice");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) {
    set_error("%s lacks the nccl* entry points", p); dlclose(lib); return SDQN_ERR_RCCL;
  }
  g_rccl.lib = lib;
  return SDQN_OK;
}
#define NCCLCHK(x) do { int e_ = (x); if (e_ != 0) { set_error("%s -> %s", #x, g_rccl.GetErrorString ? g_rccl.GetErrorString(e_) : "rccl error"); return SDQN_ERR_RCCL; } } while (0)

// ---- network -------------------------------------------------------------------------------------------
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
  bool hoist = false;                      // train_many: the next step's target-net forward rides in this step's launches (B <= 32, fp32).
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
  bool act_on = false, act_last = false, act_inject = false; int act_fallbacks = 0;     // act_inject (tests): the next one-launch forward finds its work already claimed and delivers nothing     // act_last: the forward being collected came from that launch
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
  bool half_payload_pending = false;       // sdqn_net_grad_from_half ran: the next sdqn_net_apply_update honours the overflow flag / moves the scale
  bool grad_only = false;                  // true: a train step stops after the local gradient sums (update mode 1): what a
                                           // data-parallel rank has before the all-reduce; sdqn_net_apply_update finishes it
  int nw_override[12] = {0};               // tuning hook
  int bwd_order = 0;                       // experiment: problem order inside the fused backward launches (sdqn_kernels_ext.hip)
  int f4_share[2] = {100, 0};               // % of the fc4-wgrad tiles in bwd3 / bwd2 (rest in bwd1)
  int ns_cap[3] = {1, 1, 1};               // slabs the split-K buffers were allocated for (tuning hook "tps:<layer>")
  int rb[12] = {0};                        // B >= 128: register-blocked routine, menu entry per kernel id (0 = unblocked)
  // plane mode of the block-tile engine (B >= 128, float32): bf16 planes of the conv2 / conv3 / fc4 weights (problems.h: StepArgs::wpm / wpt),
  // kept current wherever the weights are written; xp = partial products per fp32 product (9 exact / 6; 0 = fp32 MFMA, no planes used)
  unsigned short *wpm = nullptr, *wpt[2] = {nullptr, nullptr}; int xp = 0;
  int btx[K_COUNT] = {0};                    // block-tile engine arithmetic per kernel id: 0 fp32 MFMA, 9 / 6 exact bf16x3 operand splits on packed-bf16 MFMA
  bool bt_xcd = true;                      // round 4, B >= 128: XCD-contiguous block maps for fc4_dgrad / bwd3 / bwd2 (float32) and the weight-gradient launch (float16); option "bt_xcd"
  bool bt_on = true; int bt[K_COUNT] = {0};  // round 4, B >= 128 float32: block-tile engine (sdqn_kernels_bt.hip); per kernel id 0 = built-in block shape, n = menu entry, -1 = latency engine
  int xcd_mask[K_COUNT] = {0};             // tuning hook "xcd:<kernel id>": per-launch problem mask (-1 = built-in)
  bool xcd_map = false;                    // XCD-contiguous tile map for EVERY launch: traffic ~ algorithmic, step ~1 % slower (bwd3);
                                           // built-in: only where it also wins time (fc4_fwd: the 7 K-slabs of a tile's W4 panel share an L2)
  bool f4w_early = false;                  // round 3 (B <= 32, fp32): fc4_wgrad + fused RMSProp ride in the fc4_dgrad launch (K_F4D_F4W) instead of bwd3
  bool conv1_bf16 = true;                  // round 3: conv1_fwd on packed-bf16 MFMA (bytes x 3-way bf16 split of W1; sdqn_kernels_r3.hip)
  unsigned short* w1p[2] = {nullptr, nullptr};   // the three bf16 planes of W1, online / target net ([3][32][256] each)
  const int64_t* host_idx_cur = nullptr;   // ring paths: host copy of the indexes of the step being enqueued (valid during run_train only)
  int r3_xcd = 2;                          // XCD-contiguous tile maps of the round-3 kernels: bit 0 conv1_fwd (bf16), bit 1 conv1_wgrad (bf16)
  bool prep_inline = true;                 // B <= 32: the next step's indexes ride in the update launch's kernel arguments (no PCIe read in its prep block)
  int wt = 511;                            // write-through epilogue stores, bit per launch (kernels.h: LaunchTune::wt; bit 8 = the update kernel)
  int fwd_rb = 0;                          // experiment: bit 0 conv2_fwd, bit 1 conv3_fwd on the 1 x 2 register-blocked routine (one workgroup per 32 x 64 block)
  bool handoff_launched = false;           // a launch with an in-launch hand-off (f4w_early / fuse_upd) was enqueued since the last sync
  int fuse_dbg = 0;                        // experiment only: 1 = the online conv1 blocks do not wait (WRONG results, timing of the wait)
  bool fuse_upd = false;                   // round 3: inside train_many, update(i) and conv1_fwd(i + 1) are ONE launch (sdqn_kernels_r3.hip: upd_conv1_kernel)
  bool has_pending_upd = false; UpdateArgs pending_upd;      // the deferred optimizer pass of the previous step (never outlives a train_many call)
  bool head_f4d = false; bool skip_head = false; unsigned hf_epochs = 0;   // head + fc4_dgrad as one launch (w1_ctr[4] counts head-block arrivals, [5] = time-out word)
  unsigned* w1_ctr = nullptr; unsigned w1_epochs = 0;        // [0] W1 blocks counted in (monotonic: 64 per fused launch), [1] sticky time-out word
  int conv1w_bf16 = 1;                 // round 3: conv1_wgrad on packed-bf16 MFMA (bytes x on-the-fly bf16 split of delta1)
  bool conv3_c36 = true;                   // round 3: conv3_fwd on 36-deep K-chunks (one chunk per wave; sdqn_kernels_r3.hip)
  unsigned* f4d_flags = nullptr;           // [NIN4 / 32][16] write-after-read flags of that launch + one sticky time-out word
  bool fused_launches = true;              // independent backward stages share one launch (K_BWD3, K_BWD2)
  bool two_streams = false;                // weight-gradient kernels on the side stream (measured slower eagerly: event waits)
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
  std::vector<void*> allocs;
};

static int dalloc(sdqn_net_s* h, void** p, size_t bytes, bool zero = true) {
  HIPCHK(hipMalloc(p, bytes)); h->allocs.push_back(*p);
  if (zero) HIPCHK(hipMemsetAsync(*p, 0, bytes, g_stream));
  return SDQN_OK;
}
static int join_comm(sdqn_net_s* h);
static int net_free(sdqn_net_s* h) {
  if (!h) return SDQN_OK;
  if (g_stream) hipStreamSynchronize(g_stream);
  delete h->gen; h->gen = nullptr;
  if (g_comm) hipStreamSynchronize(g_comm);
  if (h->comm2 && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm2);
  if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
  if (h->ev_g4) hipEventDestroy(h->ev_g4);
  if (h->ev_w4) hipEventDestroy(h->ev_w4);
  for (void* p : h->allocs) hipFree(p);
  hipHostFree(h->h_f);
  if (h->q_host) hipHostFree(h->q_host);
  if (h->cost_ring) hipHostFree(h->cost_ring);
  for (int i = 0; i < 2; ++i) { if (h->h_stage[i]) hipHostFree(h->h_stage[i]); if (h->stage_ev[i]) hipEventDestroy(h->stage_ev[i]); }
  for (auto& pp : h->prof_pending) { hipEventDestroy(pp.a); hipEventDestroy(pp.b); }
  for (auto e : h->prof_free) hipEventDestroy(e);
  delete h;
  return SDQN_OK;
}

extern "C" int sdqn_net_create(sdqn_net_t* out, const sdqn_net_cfg* c) {
  ARGCHK(out && c, "NULL argument");
  ARGCHK(c->batch_size > 0 && c->batch_size <= 4096, "bad batch_size %d", c->batch_size);
  ARGCHK(c->num_actions > 0 && c->num_actions <= MAX_ACTIONS, "num_actions must be in 1..%d (got %d)", MAX_ACTIONS, c->num_actions);
  ARGCHK(c->optimizer >= 0 && c->optimizer <= 2, "unknown optimizer %d", c->optimizer);
  ARGCHK(c->datatype >= 0 && c->datatype <= 2, "datatype must be 0 (float32), 1 (float16) or 2 (float64)");
  ARGCHK(!(c->batch_norm != 0.0 && c->datatype != 0), "batch_norm is float32 only");
  ARGCHK(c->screen_height > 0 && c->screen_width > 0 && c->history_length > 0 && c->screen_height <= 4096 && c->screen_width <= 4096 &&
         c->history_length <= 64, "bad screen geometry %dx%d, history_length %d", c->screen_height, c->screen_width, c->history_length);
  const bool tuned_geom = c->screen_height == H0 && c->screen_width == W0 && c->history_length == C0;
  ARGCHK(tuned_geom || (c->datatype != 1 && c->batch_norm == 0.0),
         "float16 and batch_norm are implemented for 84x84 screens with history_length 4 (got %dx%d, %d)", c->screen_height, c->screen_width, c->history_length);
  STREAMCHK();
  sdqn_net_s* h = new sdqn_net_s();
  h->cfg = *c; h->B = c->batch_size; h->A = c->num_actions; h->NPW = OFF5 + (int64_t)h->A * NFC;
  if (c->datatype == 2 || !tuned_geom) {            // main.py:27-28,34,53: same layer stack, other sizes / float64 arithmetic
    std::string err;
    h->gen = make_generic_net(*c, g_stream, &err);
    if (!h->gen) { set_error("%s", err.c_str()); delete h; return SDQN_ERR_HIP; }
    h->NP = h->NPW = h->gen->param_count();
    memset(h->prof_ms, 0, sizeof h->prof_ms); memset(h->prof_n, 0, sizeof h->prof_n);
    *out = h;
    return SDQN_OK;
  }
  h->bn = c->batch_norm != 0.0;
  h->NP = h->NPW + (h->bn ? 2 * BN_PARAMS : 0);
  memset(h->prof_ms, 0, sizeof h->prof_ms); memset(h->prof_n, 0, sizeof h->prof_n);
  const int B = h->B;
  auto pick = [](int T, int target) { int t = (T + target - 1) / target; return t < 1 ? 1 : t; };
  // wgrad split-K: 16 waves per workgroup take one 32-deep chunk each at B = 32 (more per wave for larger B)
  const int T1 = ceil_div(B * PIX1, 32), T2 = ceil_div(B * PIX2, 32), T3 = ceil_div(B * PIX3, 32);
  // B = 32: 16 / 8 / 13 chunks per slab.  conv1 / conv2 wgrad: one chunk per wave of the 16- / 8-wave workgroup (tools/sweep_tps.py: conv2
  // with 8 instead of 14 chunks per slab 12 570 -> 12 840 steps/s); conv3's slabs ride under the fc4 RMSProp stream: fewer, longer ones
  h->tps1 = pick(T1, 25); h->tps2 = T2 < 8 ? T2 : 8; h->tps3 = pick(T3, 4);
  if (B >= 128) {
    // throughput regime: the chunks per slab stay about what they are at B = 32 and the NUMBER of slabs grows with B
    // (tools/sweep_tps.py at B = 256, steps/s: 64/54/49 chunks per slab 3 430 -> 50/18/20 3 620; float16 5 380 -> 100/18/20 5 790)
    h->tps1 = c->datatype == 1 ? 100 : 50; h->tps2 = 18; h->tps3 = 20;
    // round 4, float32 on the block-tile engine (one 64 x 64 block of a slab per workgroup, its chunks in sequence): shorter slabs
    // (tools/sweep_bt.py at B = 256: conv2_wgrad 18 -> 9 chunks per slab, bwd2 49.6 -> 43.8 us; conv3_wgrad 20 -> 14: 41.5 -> 40.7 us)
    // conv1's weight gradient (c1w_bt_kernel: one workgroup per slab of whole 80-position chunks, all 256 x 32 outputs): 10 x 32 = 320
    // positions per slab = 4 chunks, 320 workgroups at B = 256
    if (c->datatype == 0 && !h->bn) { h->tps2 = 9; h->tps3 = 14; h->tps1 = 10; }
    // float16: conv1's weight gradient is one workgroup per slab of whole 80-position chunks too (c1w_h_kernel): 10 x 32 = 320 positions
    if (c->datatype == 1 && !h->bn) h->tps1 = 10;
    if (h->tps1 > T1) h->tps1 = T1; if (h->tps2 > T2) h->tps2 = T2; if (h->tps3 > T3) h->tps3 = T3;
  }
  // (the register-blocked routine, gemm_engine_rb.h, is available per kernel id through set_option "rb:<id>" / "tps:<l>":
  //  measured slower than these choices at B = 256 in every fused launch — tools/exp/README.md — so it is off by default)
  h->ns1 = ceil_div(T1, h->tps1); h->ns2 = ceil_div(T2, h->tps2); h->ns3 = ceil_div(T3, h->tps3);
  // fc4 forward K-splits: parallelism at B = 32; at B >= 128 the M x N tiles fill the chip in fp32 (3 620 -> 3 650 steps/s at B = 256),
  // not in float16 where a wave owns a 64 x 64 block (S4 = 1: 4 850 steps/s, 7: 5 770)
  h->S4 = (B >= 128 && c->datatype == 0) ? 1 : 7;
#define NCHK(x) do { int r_ = (x); if (r_) { net_free(h); return r_; } } while (0)
  NCHK(dalloc(h, (void**)&h->theta, h->NP * 4));
  if (c->target_enabled) NCHK(dalloc(h, (void**)&h->theta_t, h->NP * 4)); else h->theta_t = h->theta;   // deepqnetwork.py:64-73
  NCHK(dalloc(h, (void**)&h->state, h->NP * 4));
  if (c->optimizer != 0) NCHK(dalloc(h, (void**)&h->state2, h->NP * 4));
  NCHK(dalloc(h, (void**)&h->g, h->NP * 4));
  NCHK(dalloc(h, (void**)&h->a1, (size_t)2 * B * PIX1 * K1 * 4));
  NCHK(dalloc(h, (void**)&h->a2, (size_t)2 * B * PIX2 * K2 * 4));
  NCHK(dalloc(h, (void**)&h->a3, (size_t)2 * B * PIX3 * K3 * 4));
  h->S4_cap = 7;
  NCHK(dalloc(h, (void**)&h->slab4, (size_t)h->S4_cap * 2 * B * NFC * 4));
  NCHK(dalloc(h, (void**)&h->a4, (size_t)2 * B * NFC * 4));
  NCHK(dalloc(h, (void**)&h->d4, (size_t)B * NFC * 4));
  NCHK(dalloc(h, (void**)&h->d3p, (size_t)B * PD3 * PD3 * K3 * 4));    // borders stay zero for ever
  NCHK(dalloc(h, (void**)&h->d2p, (size_t)B * PD2 * PD2 * K2 * 4));
  NCHK(dalloc(h, (void**)&h->d1, (size_t)B * PIX1 * K1 * 4));
  NCHK(dalloc(h, (void**)&h->d3, (size_t)B * PIX3 * K3 * 4));
  NCHK(dalloc(h, (void**)&h->d2, (size_t)B * PIX2 * K2 * 4));
  // room for the "tps:<layer>" tuning hook down to 8 chunks per slab (at least 64 slabs)
  { const int Ts[3] = {T1, T2, T3}; const int ns[3] = {h->ns1, h->ns2, h->ns3};
    for (int l = 0; l < 3; ++l) { int c = ceil_div(Ts[l], l == 0 ? 5 : 8); if (c < 64) c = 64; if (c < ns[l]) c = ns[l]; h->ns_cap[l] = c; } }
  NCHK(dalloc(h, (void**)&h->slab1, (size_t)h->ns_cap[0] * NW1 * 4));
  NCHK(dalloc(h, (void**)&h->slab2, (size_t)h->ns_cap[1] * NW2 * 4));
  NCHK(dalloc(h, (void**)&h->slab3, (size_t)h->ns_cap[2] * NW3 * 4));
  if (h->bn) {
    NCHK(dalloc(h, (void**)&h->x1, (size_t)2 * B * PIX1 * K1 * 4));
    NCHK(dalloc(h, (void**)&h->x2, (size_t)2 * B * PIX2 * K2 * 4));
    NCHK(dalloc(h, (void**)&h->x3, (size_t)2 * B * PIX3 * K3 * 4));
    NCHK(dalloc(h, (void**)&h->bn_mean, (size_t)BN_PARAMS / 2 * 4));
    NCHK(dalloc(h, (void**)&h->bn_rstd, (size_t)BN_PARAMS / 2 * 4));
    const int max_rb = (B * PIX1 + 255) / 256;
    NCHK(dalloc(h, (void**)&h->bn_partial, (size_t)max_rb * 512 * 2 * 8));
    // BatchNorm init [neon-recalled]: beta = 0, gamma = 1, running mean / variance = 0
    std::vector<float> blk((size_t)BN_PARAMS, 0.0f);
    for (int l = 0; l < BN_LAYERS; ++l) for (int cc = 0; cc < bn_features(l); ++cc) blk[(size_t)bn_off(l) + bn_features(l) + cc] = 1.0f;
    HIPCHK(hipStreamSynchronize(g_stream));
    { hipError_t e_ = hipMemcpy(h->theta + h->NPW, blk.data(), (size_t)BN_PARAMS * 4, hipMemcpyHostToDevice);
      if (e_ == hipSuccess && h->theta_t != h->theta) e_ = hipMemcpy(h->theta_t + h->NPW, blk.data(), (size_t)BN_PARAMS * 4, hipMemcpyHostToDevice);
      if (e_ != hipSuccess) { set_error("hipMemcpy -> %s", hipGetErrorString(e_)); net_free(h); return SDQN_ERR_HIP; } }
  }
  if (c->datatype == 1) {
    if (h->cfg.loss_scale == 0) h->cfg.loss_scale = 1024.0;
    NCHK(dalloc(h, (void**)&h->h_a1, (size_t)2 * B * PIX1 * K1 * 2));
    NCHK(dalloc(h, (void**)&h->h_a2, (size_t)2 * B * PIX2 * K2 * 2));
    NCHK(dalloc(h, (void**)&h->h_a3, (size_t)2 * B * PIX3 * K3 * 2));
    NCHK(dalloc(h, (void**)&h->h_d4, (size_t)B * NFC * 2));
    NCHK(dalloc(h, (void**)&h->h_d3p, (size_t)B * PD3 * PD3 * K3 * 2));      // borders stay zero
    NCHK(dalloc(h, (void**)&h->h_d2p, (size_t)B * PD2 * PD2 * K2 * 2));
    NCHK(dalloc(h, (void**)&h->h_d3, (size_t)B * PIX3 * K3 * 2));
    NCHK(dalloc(h, (void**)&h->h_d2, (size_t)B * PIX2 * K2 * 2));
    NCHK(dalloc(h, (void**)&h->h_d1, (size_t)B * PIX1 * K1 * 2));
    for (int zz = 0; zz < (c->target_enabled ? 2 : 1); ++zz) {
      NCHK(dalloc(h, (void**)&h->wh[zz], (size_t)OFF5 * 2));
      NCHK(dalloc(h, (void**)&h->wht[zz], (size_t)OFF5 * 2));
    }
    if (!c->target_enabled) { h->wh[1] = h->wh[0]; h->wht[1] = h->wht[0]; }
    NCHK(dalloc(h, (void**)&h->gh, (size_t)h->NP * 2));
    NCHK(dalloc(h, (void**)&h->ovf_flag, 16));
    { const int st0[4] = {0, 10, 0, 0}; HIPCHK(hipMemcpyAsync(h->ovf_flag, st0, 16, hipMemcpyHostToDevice, g_stream)); HIPCHK(hipStreamSynchronize(g_stream)); }
    NCHK(dalloc(h, (void**)&h->ovf_count, 16));
  }
  NCHK(dalloc(h, (void**)&h->q, (size_t)2 * B * h->A * 4));
  NCHK(dalloc(h, (void**)&h->maxq, (size_t)B * 4));
  NCHK(dalloc(h, (void**)&h->dq, (size_t)B * h->A * 4));
  NCHK(dalloc(h, (void**)&h->cost_terms, (size_t)B * 4));
  NCHK(dalloc(h, (void**)&h->cost_out, 16));
  NCHK(dalloc(h, (void**)&h->cost_accum, 16));
  NCHK(dalloc(h, (void**)&h->st_states, (size_t)2 * B * STATE + SRC_PAD));
  // the minibatch's small arrays in ONE block [rewards 8 B | actions B | terminals B]: the tuple API uploads them with one copy
  NCHK(dalloc(h, (void**)&h->st_rew, (size_t)B * 10));
  h->st_act = reinterpret_cast<uint8_t*>(h->st_rew) + (size_t)B * 8; h->st_term = h->st_act + B;
  NCHK(dalloc(h, (void**)&h->f4d_flags, (size_t)(NIN4 / 32 + 1) * 16 * 4));
  NCHK(dalloc(h, (void**)&h->w1_ctr, 64));
  if (c->datatype == 0) {                  // (all-zero planes == all-zero W1, which is what the zeroed theta holds until set_weights)
    NCHK(dalloc(h, (void**)&h->w1p[0], (size_t)3 * W1P_PLANE * 2));
    if (c->target_enabled) NCHK(dalloc(h, (void**)&h->w1p[1], (size_t)3 * W1P_PLANE * 2)); else h->w1p[1] = h->w1p[0];
  }
  // plane mode (experiments build only: measured 9 % / 17 % SLOWER than fp32 MFMA with 6 / 9 partial products, tools/exp/README.md)
  if (EXPERIMENTS && c->datatype == 0 && B >= 128 && !h->bn) {   // its weight planes (3 x bf16 of every conv1..fc4 weight slot: 10 MB each)
    NCHK(dalloc(h, (void**)&h->wpm, (size_t)3 * XP_PLANE * 2));
    NCHK(dalloc(h, (void**)&h->wpt[0], (size_t)3 * XP_PLANE * 2));
    if (c->target_enabled) NCHK(dalloc(h, (void**)&h->wpt[1], (size_t)3 * XP_PLANE * 2)); else h->wpt[1] = h->wpt[0];
    h->xp = 0;                                                    // (option bt_planes = 6 / 9 turns it on)
  }
  if (c->datatype == 0 && !h->bn) {
    NCHK(dalloc(h, (void**)&h->act_scratch, (size_t)8 * ACT_XCC_FLOATS * 4));
    NCHK(dalloc(h, (void**)&h->act_q, (size_t)Q_SLOT_FLOATS * 4));
    NCHK(dalloc(h, (void**)&h->act_ctl, (size_t)4 * ACT_CTL_WORDS * 4));
    h->act_on = true;
  }
  NCHK(dalloc(h, (void**)&h->d_idx, (size_t)B * 8));
  NCHK(dalloc(h, (void**)&h->d_idx_t, (size_t)B * 8));
  { hipError_t e = hipHostMalloc((void**)&h->h_f, (size_t)(2 * B * MAX_ACTIONS + B + 64 + Q_SLOT_FLOATS) * 8, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostMalloc((void**)&h->q_host, Q_SLOTS * Q_SLOT_FLOATS * sizeof(float), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&h->q_host_dev, h->q_host, 0);
    if (e != hipSuccess) { set_error("hipHostMalloc -> %s", hipGetErrorString(e)); net_free(h); return SDQN_ERR_HIP; } }
#undef NCHK
  HIPCHK(hipStreamSynchronize(g_stream));
  *out = h;
  return SDQN_OK;
}
extern "C" int sdqn_net_destroy(sdqn_net_t h) { return net_free(h); }

static float* which_buf(sdqn_net_s* h, int which) {
  switch (which) { case 0: return h->theta; case 1: return h->theta_t; case 2: return h->state; case 3: return h->g;
                   case 4: return h->state2; default: return nullptr; }
}
// BatchNorm pseudo-layers 5..8 (batch_norm only): [beta | gamma] at NPW + bn_off(l); which 5 / 6 = running statistics
static bool bn_layer_span(sdqn_net_s* h, int which, int layer, float** base, int64_t* n) {
  if (!h->bn || layer < 5 || layer > 8) return false;
  const int l = layer - 5;
  *n = 2 * bn_features(l);
  float* buf = nullptr; int64_t extra = 0;
  switch (which) {
    case 0: buf = h->theta; break;
    case 1: buf = h->theta_t; break;
    case 2: buf = h->state; break;
    case 3: buf = h->g; break;
    case 4: buf = h->state2; break;
    case 5: buf = h->theta; extra = BN_PARAMS; break;
    case 6: buf = h->theta_t; extra = BN_PARAMS; break;
    default: break;
  }
  if (!buf) return false;
  *base = buf + h->NPW + extra + bn_off(l);
  return true;
}
extern "C" int sdqn_net_layer_size(sdqn_net_t h, int layer, int64_t* n) {
  ARGCHK(h && n && layer >= 0 && layer < (h->bn ? 9 : 5), "bad arguments");
  if (h->gen) { *n = h->gen->layer_size(layer); return SDQN_OK; }
  if (layer >= 5) { *n = 2 * bn_features(layer - 5); return SDQN_OK; }
  int64_t rows, cols, off; layer_dims(layer, h->A, rows, cols, off);
  *n = rows * cols;
  return SDQN_OK;
}
#define GENCHK(x) do { hipError_t ge_ = (x); if (ge_ != hipSuccess) { set_error("%s -> %s", #x, hipGetErrorString(ge_)); return ge_ == hipErrorInvalidValue ? SDQN_ERR_ARG : SDQN_ERR_HIP; } } while (0)
static int gen_set(sdqn_net_s* h, int which, int layer, const void* w, int64_t n, bool f64) {
  ARGCHK(layer >= 0 && layer < 5 && which >= 0 && which <= 4 && which != 3, "bad arguments (which %d, layer %d)", which, layer);
  ARGCHK(which != 4 || h->cfg.optimizer != 0, "this optimizer has no second state");
  ARGCHK(n == h->gen->layer_size(layer), "layer %d holds %lld values, got %lld", layer, (long long)h->gen->layer_size(layer), (long long)n);
  GENCHK(h->gen->set_param(which, layer, w, f64));
  return SDQN_OK;
}
static int gen_get(sdqn_net_s* h, int which, int layer, void* w, int64_t n, bool f64) {
  ARGCHK(layer >= 0 && layer < 5 && which >= 0 && which <= 4, "bad arguments (which %d, layer %d)", which, layer);
  ARGCHK(which != 4 || h->cfg.optimizer != 0, "this optimizer has no second state");
  ARGCHK(n == h->gen->layer_size(layer), "layer %d holds %lld values, got %lld", layer, (long long)h->gen->layer_size(layer), (long long)n);
  GENCHK(h->gen->get_param(which, layer, w, f64));
  return SDQN_OK;
}
// double-precision forms of set_weights / get_weights / predict / last_q: what a `--datatype float64` network (main.py:53) exchanges
// without a round trip through float.  On float32 / float16 networks they convert.
extern "C" int sdqn_net_set_weights_f64(sdqn_net_t h, int which, int layer, const double* w, int64_t n) {
  ARGCHK(h && w && n >= 0, "NULL argument");
  if (h->gen) return gen_set(h, which, layer, w, n, true);
  std::vector<float> tmp((size_t)n); for (int64_t i = 0; i < n; ++i) tmp[(size_t)i] = (float)w[i];
  return sdqn_net_set_weights(h, which, layer, tmp.data(), n);
}
extern "C" int sdqn_net_get_weights_f64(sdqn_net_t h, int which, int layer, double* w, int64_t n) {
  ARGCHK(h && w && n >= 0, "NULL argument");
  if (h->gen) return gen_get(h, which, layer, w, n, true);
  std::vector<float> tmp((size_t)n);
  int rc = sdqn_net_get_weights(h, which, layer, tmp.data(), n); if (rc) return rc;
  for (int64_t i = 0; i < n; ++i) w[i] = (double)tmp[(size_t)i];
  return SDQN_OK;
}
extern "C" int sdqn_net_set_weights(sdqn_net_t h, int which, int layer, const float* w, int64_t n) {
  ARGCHK(h && w, "NULL argument");
  if (h->gen) return gen_set(h, which, layer, w, n, false);
  h->spec_pending = false;                 // (parameters may change under a speculative acting forward)
  if (layer >= 5) {
    float* base; int64_t cnt;
    ARGCHK(which != 3 && bn_layer_span(h, which, layer, &base, &cnt), "no such BatchNorm buffer (which %d, layer %d)", which, layer);
    ARGCHK(n == cnt, "layer %d holds %lld values, got %lld", layer, (long long)cnt, (long long)n);
    { int rc_ = join_comm(h); if (rc_) return rc_; } HIPCHK(hipStreamSynchronize(g_stream));
    HIPCHK(hipMemcpy(base, w, (size_t)n * 4, hipMemcpyHostToDevice));
    return SDQN_OK;
  }
  ARGCHK(layer >= 0 && layer < 5 && which >= 0 && which <= 4, "bad arguments");
  ARGCHK(which_buf(h, which), "this optimizer has no second state");
  int64_t rows, cols, off; layer_dims(layer, h->A, rows, cols, off);
  ARGCHK(n == rows * cols, "layer %d holds %lld values, got %lld", layer, (long long)(rows * cols), (long long)n);
  std::vector<float> tmp((size_t)n);
  for (int64_t r = 0; r < rows; ++r) for (int64_t c = 0; c < cols; ++c) tmp[(size_t)neon_to_internal(layer, r, c)] = w[r * cols + c];
  { int rc_ = join_comm(h); if (rc_) return rc_; } HIPCHK(hipStreamSynchronize(g_stream));
  HIPCHK(hipMemcpy(which_buf(h, which) + off, tmp.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  if (h->cfg.datatype == 1 && which <= 1) {      // fp16 mode: the half copies follow the master weights
    const int zz = (which == 1 && h->theta_t != h->theta) ? 1 : 0;
    HIPCHK(launch_refresh16(zz ? h->theta_t : h->theta, h->wh[zz], h->wht[zz], g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
  }
  if (h->wpm && which <= 1 && layer >= 1 && layer <= 3) {   // plane mode: the bf16 planes of conv2 / conv3 / fc4 follow the weights
    const int zz = (which == 1 && h->theta_t != h->theta) ? 1 : 0;
    HIPCHK(launch_refresh_planes(zz ? h->theta_t : h->theta, zz ? nullptr : h->wpm, h->wpt[zz], g_stream));
    if (!zz && h->theta_t == h->theta) {}        // (no target net: wpt[1] aliases wpt[0])
    HIPCHK(hipStreamSynchronize(g_stream));
  }
  if (h->w1p[0] && which <= 1 && layer == 0) {   // conv1's bf16 planes follow W1
    const int zz = (which == 1 && h->theta_t != h->theta) ? 1 : 0;
    HIPCHK(launch_w1_planes(zz ? h->theta_t : h->theta, h->w1p[zz], g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
  }
  return SDQN_OK;
}
extern "C" int sdqn_net_get_weights(sdqn_net_t h, int which, int layer, float* w, int64_t n) {
  ARGCHK(h && w, "NULL argument");
  if (h->gen) return gen_get(h, which, layer, w, n, false);
  if (layer >= 5) {
    float* base; int64_t cnt;
    ARGCHK(bn_layer_span(h, which, layer, &base, &cnt), "no such BatchNorm buffer (which %d, layer %d)", which, layer);
    ARGCHK(n == cnt, "layer %d holds %lld values, got %lld", layer, (long long)cnt, (long long)n);
    { int rc_ = join_comm(h); if (rc_) return rc_; } HIPCHK(hipStreamSynchronize(g_stream));
    HIPCHK(hipMemcpy(w, base, (size_t)n * 4, hipMemcpyDeviceToHost));
    return SDQN_OK;
  }
  ARGCHK(layer >= 0 && layer < 5 && which >= 0 && which <= 4, "bad arguments");
  ARGCHK(which_buf(h, which), "this optimizer has no second state");
  int64_t rows, cols, off; layer_dims(layer, h->A, rows, cols, off);
  ARGCHK(n == rows * cols, "layer %d holds %lld values, got %lld", layer, (long long)(rows * cols), (long long)n);
  std::vector<float> tmp((size_t)n);
  { int rc_ = join_comm(h); if (rc_) return rc_; } HIPCHK(hipStreamSynchronize(g_stream));
  HIPCHK(hipMemcpy(tmp.data(), which_buf(h, which) + off, (size_t)n * 4, hipMemcpyDeviceToHost));
  for (int64_t r = 0; r < rows; ++r) for (int64_t c = 0; c < cols; ++c) w[r * cols + c] = tmp[(size_t)neon_to_internal(layer, r, c)];
  return SDQN_OK;
}

// ---- profiler -------------------------------------------------------------------------------------------
static int prof_collect(sdqn_net_s* h) {
  if (h->prof_pending.empty()) return SDQN_OK;
  HIPCHK(hipStreamSynchronize(g_stream));
  HIPCHK(hipStreamSynchronize(g_side));
  HIPCHK(hipStreamSynchronize(g_comm));
  for (auto& p : h->prof_pending) {
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, p.a, p.b));
    h->prof_ms[p.id] += ms; h->prof_n[p.id] += 1;
    h->prof_free.push_back(p.a); h->prof_free.push_back(p.b);
  }
  h->prof_pending.clear();
  return SDQN_OK;
}
static int prof_event(sdqn_net_s* h, hipEvent_t* e) {
  if (!h->prof_free.empty()) { *e = h->prof_free.back(); h->prof_free.pop_back(); return SDQN_OK; }
  HIPCHK(hipEventCreate(e)); return SDQN_OK;
}
// profile_mode 1 (default): the launch itself records its dispatch packet's begin / end timestamps into the pair (launch.h:
// what rocprofv3 --kernel-trace reports, nothing added to the queue); 0, and always for launches that are not ONE kernel
// (RCCL, BatchNorm's two passes): hipEventRecord markers around the launch (adds ~2.6 us of packet processing to the figure)
static inline bool prof_single_kernel(int kid) { return kid != K_ALLREDUCE && kid != K_BN; }
#define LAUNCH_ON(STRM, KID, expr) do { \
  const bool pf_ = h->prof_on && (h->prof_filter < 0 || h->prof_filter == (KID)) && (h->prof_seen[KID]++ % h->prof_every) == 0; ProfPair pp_; \
  const bool px_ = pf_ && h->prof_mode == 1 && prof_single_kernel(KID); \
  if (pf_) { pp_.id = (KID); int r1_ = prof_event(h, &pp_.a); if (r1_) return r1_; r1_ = prof_event(h, &pp_.b); if (r1_) return r1_; \
             if (px_) { sdqn::LaunchEvents& le = sdqn::launch_events(); le.start = pp_.a; le.stop = pp_.b; le.used = false; } \
             else HIPCHK(hipEventRecord(pp_.a, (STRM))); } \
  hipError_t le_ = (expr); \
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

extern "C" int sdqn_net_profile(sdqn_net_t h, int enable, int kernel) {
  ARGCHK(h && kernel < K_COUNT, "bad arguments");
  if (h->gen) return SDQN_OK; h->prof_on = enable != 0; h->prof_filter = kernel; return SDQN_OK;
}
extern "C" int sdqn_net_profile_count(int* n) { ARGCHK(n, "NULL"); *n = K_COUNT; return SDQN_OK; }
extern "C" int sdqn_net_profile_read(sdqn_net_t h, int kernel, const char** name, double* total_ms, int64_t* launches) {
  ARGCHK(h && kernel >= 0 && kernel < K_COUNT, "bad arguments");
  int rc = prof_collect(h); if (rc) return rc;
  if (name) *name = kernel_name(kernel); if (total_ms) *total_ms = h->prof_ms[kernel]; if (launches) *launches = h->prof_n[kernel];
  return SDQN_OK;
}
extern "C" int sdqn_net_profile_reset(sdqn_net_t h) {
  ARGCHK(h, "NULL handle"); int rc = prof_collect(h); if (rc) return rc;
  memset(h->prof_ms, 0, sizeof h->prof_ms); memset(h->prof_n, 0, sizeof h->prof_n);
  memset(h->prof_seen, 0, sizeof h->prof_seen);      // launch 0 after a reset is bracketed again (profile_every counts from the reset)
  return SDQN_OK;
}

// RCCL all-reduce on behalf of LAUNCH_ON: a failure keeps RCCL's own message (h->nccl_rc / sdqn_last_error) and is
// reported as SDQN_ERR_RCCL by the macro instead of an anonymous hipErrorUnknown
static hipError_t dp_allreduce(sdqn_net_s* h, void* buf, size_t count, int dtype, void* comm, hipStream_t s) {
  h->nccl_rc = g_rccl.AllReduce(buf, buf, count, dtype, /*ncclSum*/ 0, comm, s);
  return h->nccl_rc == 0 ? hipSuccess : hipErrorUnknown;
}

// ---- the step ---------------------------------------------------------------------------------------------
static StepArgs step_args(sdqn_net_s* h) {
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
  a.wpm = h->wpm; a.wpt[0] = h->wpt[0]; a.wpt[1] = h->wpt[1]; a.xp = (h->wpm && h->bt_on) ? h->xp : 0;
  a.f4d_flags = h->f4d_flags; a.f4d_epoch = (unsigned)(h->train_iterations + 1);      // (never 0; one train step per value)
  a.fuse_rms = (!h->comm && !h->keep_grads && !h->grad_only && h->cfg.optimizer == 0) ? 1 : 0;
  a.theta_w = h->theta; a.state = h->state; a.bsz = (float)h->B;
  a.rho = (float)h->cfg.decay_rate; a.one_minus_rho = (float)(1.0 - h->cfg.decay_rate);
  a.lr = (float)h->cfg.learning_rate; a.eps = (float)h->cfg.epsilon;
  return a;
}
static HeadArgs head_args(sdqn_net_s* h, int train) {
  HeadArgs hd; memset(&hd, 0, sizeof hd);
  hd.st_actions = h->st_act; hd.st_rewards = h->st_rew; hd.st_terminals = h->st_term;
  hd.q = h->q; hd.maxq = h->maxq; hd.dq = h->dq; hd.cost_terms = h->cost_terms;
  hd.discount = h->cfg.discount_rate; hd.min_reward = h->cfg.min_reward; hd.max_reward = h->cfg.max_reward;
  hd.clip_error = (float)h->cfg.clip_error; hd.train = train;
  return hd;
}
// Overlapped data parallel: the previous step's fc4 all-reduce + update may still be running on g_comm.  Everything
// on the library stream that touches W4, its optimizer state or the fc4 gradient must come after it.
static int join_comm(sdqn_net_s* h) {
  if (h->w4_pending) { HIPCHK(hipStreamWaitEvent(g_stream, h->ev_w4, 0)); h->w4_pending = false; }
  return SDQN_OK;
}
// --batch_norm: one BatchNorm layer's arguments (bn_kernels.hip)
static BnArgs bn_args(sdqn_net_s* h, const StepArgs& a, int layer, int train) {
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
static hipError_t launch_tuned(sdqn_net_s* h, int id, StepArgs a, hipStream_t s, int hoist = 0, int r3 = 0) {
  XCD_TUNE(a, id);
  LaunchTune t;
  for (int i = 0; i < 12; ++i) { t.nw_override[i] = h->nw_override[i]; t.rb[i] = EXPERIMENTS ? h->rb[i] : 0; }
  for (int i = 0; i < K_COUNT; ++i) { t.bt[i] = h->bt_on ? h->bt[i] : -1; t.btx[i] = h->btx[i]; }
  t.hoist = EXPERIMENTS ? hoist : 0; t.order = EXPERIMENTS ? h->bwd_order : 0; t.r3 = r3; t.host_idx = h->host_idx_cur; t.r3_xcd = h->r3_xcd; t.wt = h->wt;
  return launch_kernel(id, a, t, s);
}
static int run_forward(sdqn_net_s* h, const StepArgs& a, const HeadArgs& hd, int hoist = 0) {
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
  if (EXPERIMENTS && (hoist & 2)) {
    // hoist, second half: target conv1 / conv2 of THIS step ran inside the previous step's K_BWD2 / K_BWD1; its conv3 and fc4
    // ride in this step's conv1 / conv2 launches, whose own tiles are the online net's only (nz = 1).  The head then finds
    // the split-K slabs of both nets as usual.
    StepArgs f1 = a; f1.nz = 1; f1.idx_t = nullptr;
    StepArgs c1 = f1; c1.xcd_map = 1;                           // problem 0 (online conv1 / conv2) on the XCD-contiguous map, as below
    StepArgs c2 = f1; c2.xcd_map = 3;                           // + the riding target fc4 (the K-slabs of a W4 panel share an L2)
    LAUNCH(K_CONV1_FWD, launch_tuned(h, K_CONV1_FWD, c1, g_stream, 2));
    LAUNCH(K_CONV2_FWD, launch_tuned(h, K_CONV2_FWD, c2, g_stream, 2));
    LAUNCH(K_CONV3_FWD, launch_tuned(h, K_CONV3_FWD, f1, g_stream));
    { int rc = join_comm(h); if (rc) return rc; }
    StepArgs f4 = f1; f4.xcd_map = 1;
    LAUNCH(K_FC4_FWD, launch_tuned(h, K_FC4_FWD, f4, g_stream));
    LAUNCH(K_HEAD, launch_head(a, hd, g_stream));
    return SDQN_OK;
  }
  // XCD-contiguous tile map where it wins time (tools/sweep_xcd.py, tools/ab_options.py): conv1_fwd +0.5 %, conv2_fwd
  // +0.2 %, fc4_fwd +0.6 % of the step rate; slower for conv3_fwd, fc4_dgrad and every backward launch
  StepArgs fm = a; fm.xcd_map = 1; fm.idx_t = nullptr;
#ifdef SDQN_EXPERIMENTS
  if (h->has_pending_upd) {
    // the previous step's optimizer pass rides in front of this step's conv1 (train_many only): one launch, the online conv1
    // workgroups wait for the 64 W1 blocks of the same launch (upd_conv1_kernel)
    h->has_pending_upd = false; h->handoff_launched = true;
    h->w1_epochs += 1;
    UpdateArgs pu = h->pending_upd; pu.w1_ctr = h->w1_ctr;
    LAUNCH(K_UPD_CONV1, launch_upd_conv1(pu, fm, h->host_idx_cur, h->w1_ctr, h->fuse_dbg == 1 ? 0u : 64u * h->w1_epochs, h->w1_ctr + 1, h->r3_xcd & 1, g_stream));
  } else
#endif
  LAUNCH(K_CONV1_FWD, launch_tuned(h, K_CONV1_FWD, fm, g_stream, 0, (h->conv1_bf16 && !(EXPERIMENTS && h->hoist) && h->nw_override[K_CONV1_FWD] == 0) ? 4 : 0));
  { StepArgs f2 = fm; if (h->B >= 128) f2.xcd_map = a.xcd_map;          // block-tile routines (B >= 128): conv2_fwd 28.8 / 8.8 us round-robin, 28.9 / 9.0 on the map (fp32 / float16)
    LAUNCH(K_CONV2_FWD, launch_tuned(h, K_CONV2_FWD, f2, g_stream, 0, (EXPERIMENTS && (h->fwd_rb & 1)) ? 16 : 0)); }
  { StepArgs f3 = fm; f3.xcd_map = a.xcd_map;
    const int c36 = (EXPERIMENTS && (h->fwd_rb & 2)) ? 32 : ((h->conv3_c36 && !(EXPERIMENTS && h->hoist) && h->nw_override[K_CONV3_FWD] == 0) ? 2 : 0);      // (hoist: the riding target conv3 uses the 32-deep routine)
    LAUNCH(K_CONV3_FWD, launch_tuned(h, K_CONV3_FWD, f3, g_stream, 0, c36)); }
  { int rc = join_comm(h); if (rc) return rc; }                // conv1..3 of this step overlap the previous step's fc4 all-reduce
  LAUNCH(K_FC4_FWD, launch_tuned(h, K_FC4_FWD, fm, g_stream));
  if (!(EXPERIMENTS && h->skip_head)) LAUNCH(K_HEAD, launch_head(a, hd, g_stream, h->head_q_system));       // (skip_head: run_train launches it together with fc4_dgrad)
  return SDQN_OK;
}
static UpdateArgs make_update_args(sdqn_net_s* h, const StepArgs& a) {
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
  u.wpm = h->wpm; u.wpt = h->wpt[0];
  u.wt = (h->wt >> 8) & 1;
  u.bn_first = h->bn ? h->NPW : 0;
  if (u.opt == 1) {            // Neon Adam [neon-recalled]: t = epoch + 1, l = lr*sqrt(1-b2^t)/(1-b1^t), math in Python floats
    const double b1 = h->cfg.beta_1, b2 = h->cfg.beta_2, t = (double)h->epoch + 1.0;
    u.beta1 = (float)b1; u.one_minus_beta1 = (float)(1.0 - b1); u.beta2 = (float)b2; u.one_minus_beta2 = (float)(1.0 - b2);
    u.lr_t = (float)(h->cfg.learning_rate * sqrt(1.0 - pow(b2, t)) / (1.0 - pow(b1, t)));
  }
  return u;
}
static int run_train(sdqn_net_s* h, const StepArgs& a, const HeadArgs& hd, const PrepArgs* next = nullptr, int hoist = 0, bool defer_update = false) {
  // round 3: head + fc4_dgrad in ONE launch — the 98 dgrad tiles fetch their W4 panels while the B head workgroups run, then pick up
  // delta4 through an in-launch hand-off (sdqn_kernels_r3.hip: head_f4d_kernel).  Same arithmetic and summation order: bit-identical.
  const bool hf = EXPERIMENTS && h->head_f4d && hd.train && h->B <= 32 && h->A <= 8 && a.nz == 2 && h->cfg.datatype == 0 && !h->bn && !hoist && !h->f4w_early &&
                  h->S4 == 7 && h->nw_override[K_FC4_DGRAD] == 0 && h->nw_override[K_HEAD] == 0 && hd.next_B == 0 && h->w1_ctr;
  h->skip_head = hf;
  int rc = run_forward(h, a, hd, hoist);
  h->skip_head = false;
  if (rc) return rc;
  // Backward.  Critical path on the library stream: fc4_dgrad -> conv3_dgrad -> conv2_dgrad -> conv1_wgrad.
  // The three other weight-gradient kernels only need the delta of their layer, so they run beside it
  // on the side stream (fork after the producer of their delta, join before the update).
  const bool two_streams = EXPERIMENTS && h->two_streams;
  hipStream_t ss = two_streams ? g_side : g_stream;
  // --batch_norm: the delta arriving at layer l (masked by its Rectlin) first goes back through BatchNorm l, in place
#define BN_BWD(L) do { if (h->bn) LAUNCH(K_BN, launch_bn_backward(bn_args(h, a, (L), 1), g_stream)); } while (0)
  BN_BWD(3);
  const bool dp_ov = h->comm && h->comm2 && h->dp_overlap && h->fused_launches && !two_streams;
  // round 3: fc4_wgrad (needs delta4 and a3 only) rides in the fc4_dgrad launch, whose 98 workgroups leave 158 CUs idle; with the
  // fused RMSProp its in-place update of W4 is ordered behind the dgrad's reads by per-row-block flags (sdqn_kernels_r3.hip).
  // Same tiles, same K split as the bwd3 form: bit-identical.  Not for the overlapped-DP / two-stream / hoist / fp16 / bn variants.
  // conv1's weight gradient on packed-bf16 MFMA (sdqn_kernels_r3.hip) in EVERY launch structure (fused / unfused / two streams: same bits);
  // B < 128 only: in the throughput regime the on-the-fly split of delta1 makes it VALU-bound (measured 3 580 vs 4 063 steps/s at B = 256;
  // option value 2 forces it for experiments)
  const bool c1w = (h->conv1w_bf16 == 2 || (h->conv1w_bf16 == 1 && h->B < 128)) && h->cfg.datatype == 0 && !h->bn && !hoist && h->nw_override[K_CONV1_WGRAD] == 0;
  const bool bt_xcd = h->B >= 128 && h->cfg.datatype == 0 && h->bt_on && !h->bn && h->bt_xcd;
  const bool f4_early = EXPERIMENTS && h->f4w_early && h->B <= 32 && h->cfg.datatype == 0 && !h->bn && h->fused_launches && !two_streams &&
                        !dp_ov && !hoist && h->bwd_order == 0 && h->f4_share[0] == 100 && h->f4_share[1] == 0 && h->nw_override[K_FC4_DGRAD] == 0;
#ifdef SDQN_EXPERIMENTS
  if (hf) {
    h->handoff_launched = true; h->hf_epochs += 1;
    LAUNCH(K_HEAD_F4D, launch_head_f4d(a, hd, h->w1_ctr + 4, (unsigned)h->B * h->hf_epochs, h->w1_ctr + 5, g_stream));
  }
  else
#endif
  if (f4_early) { h->handoff_launched = true; LAUNCH(K_F4D_F4W, launch_tuned(h, K_FC4_DGRAD, a, g_stream, 0, 1)); }
  else {
    // round 4, B >= 128 float32: the backward launches on the XCD-contiguous block / tile maps (the blocks that share a weight panel share an L2):
    // fc4_dgrad 15.1 -> 14.4 us, bwd2 38.9 -> 37.5, bwd3 39.7 -> 39.2; 4 802 -> 4 867 steps/s at B = 256 (placement only: same bits)
    StepArgs fd = a; if (bt_xcd) fd.xcd_map |= 1;
    LAUNCH(K_FC4_DGRAD, launch_tuned(h, K_FC4_DGRAD, fd, g_stream));
  }
  BN_BWD(2);
  // round 4, float16 at B >= 128: the two dgrads run on the half block-tile routine as launches of their own (15.6 / 18.1 -> ~7 / 8 us:
  // operands leave L2 once per workgroup), and every weight gradient that does not need delta1 shares ONE launch behind them (it packs
  // better than the two fused backward launches did): fc4_dgrad, conv3_dgrad, conv2_dgrad, {fc4_wgrad + RMSProp || conv3_wgrad ||
  // conv2_wgrad}, conv1_wgrad — five launches where there were four, 19 us less (tools/exp/README.md)
  const bool h16_bt = h->cfg.datatype == 1 && h->B >= 128 && h->bt_on && !h->bn && h->fused_launches && !two_streams && !dp_ov &&
                      h->bt[K_CONV3_DGRAD] >= 0 && h->bt[K_CONV2_DGRAD] >= 0 && h->nw_override[K_CONV3_DGRAD] == 0 && h->nw_override[K_CONV2_DGRAD] == 0;
  if (dp_ov) {
    // data parallel, overlapped: ALL of fc4_wgrad rides the first backward launch, so the 6.4 MB fc4 gradient is
    // complete two launches before the step ends; its all-reduce and its optimizer update run on g_comm
    // (second communicator) under K_BWD2, K_BWD1, the conv/fc5 all-reduce + update and the next step's conv1..3.
    StepArgs b3 = a, b2 = a, b1 = a;
    b3.f4w_first = 0; b3.f4w_count = (NIN4 / 32) * (NFC / 32); b2.f4w_count = b1.f4w_count = 0;
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
    LAUNCH(K_BWD1, launch_tuned(h, K_BWD1, b1, g_stream, 0, c1w ? 8 : 0));
  } else
  if (h16_bt) {
    StepArgs w = a; w.f4w_first = 0; w.f4w_count = (NIN4 / 32) * (NFC / 32);
    if (h->bt_xcd) w.xcd_map |= 7;          // the weight-gradient launch on XCD-contiguous block maps (the blocks of a K slab share their operand rows): 20.1 -> 17.0 us at B = 256
    StepArgs b1 = a; b1.f4w_count = 0; b1.xcd_map |= 2;
    LAUNCH(K_CONV3_DGRAD, launch_tuned(h, K_CONV3_DGRAD, a, g_stream));
    LAUNCH(K_CONV2_DGRAD, launch_tuned(h, K_CONV2_DGRAD, a, g_stream));
    LAUNCH(K_WGRADS, launch_tuned(h, K_WGRADS, w, g_stream));
    LAUNCH(K_BWD1, launch_tuned(h, K_BWD1, b1, g_stream));
  } else
  if (h->fused_launches && !two_streams) {
    // fc4 wgrad (1568 tiles at B <= 32) is spread over the three backward launches as background traffic;
    // for B > 32 (K-split workgroups) it all rides in the first one
    const int f4_tiles = (NIN4 / 32) * (NFC / 32);
    StepArgs b3 = a, b2 = a, b1 = a;
    if (f4_early) { b3.f4w_count = b2.f4w_count = b1.f4w_count = 0; }
    else if (h->B <= 32) {
      const int s3 = h->f4_share[0] * f4_tiles / 100, s2 = h->f4_share[1] * f4_tiles / 100;
      b3.f4w_first = 0; b3.f4w_count = s3;
      b2.f4w_first = s3; b2.f4w_count = s2;
      b1.f4w_first = s3 + s2; b1.f4w_count = f4_tiles - s3 - s2;
    } else { b3.f4w_first = 0; b3.f4w_count = f4_tiles; b2.f4w_count = b1.f4w_count = 0; }
    if (bt_xcd) { b3.xcd_map |= 7; b2.xcd_map |= 7; }
    // B <= 32: the fc4_wgrad tiles of bwd3 (third problem of the launch) on the XCD-contiguous map — a tile row's 16 tiles share a3's columns
    // (10.29 -> 10.10 us, 15 315 -> 15 345 steps/s in alternating rate loops; the conv3 problems are slower on it: round-robin as before)
    if (h->B <= 32 && h->cfg.datatype == 0 && !h->bn && h->bt_xcd) b3.xcd_map |= 4;
    if (f4_early) LAUNCH(K_BWD3_CONV, launch_tuned(h, K_BWD3, b3, g_stream)); else LAUNCH(K_BWD3, launch_tuned(h, K_BWD3, b3, g_stream));
    BN_BWD(1);
    LAUNCH(K_BWD2, launch_tuned(h, K_BWD2, b2, g_stream, hoist & 1));
    BN_BWD(0);
    // conv1_wgrad on the XCD-contiguous tile map: the 8 m-tiles of a K-slab read the same frames, so a slab's tiles belong on ONE XCD's L2
    // (L2 <-> fabric traffic of the launch 15.8 -> 4.0 MB = 1.4x algorithmic, rocprofv3 PMC; step rate -0.1 %: the re-reads were MALL hits)
    b1.xcd_map |= 2;
    LAUNCH(K_BWD1, launch_tuned(h, K_BWD1, b1, g_stream, hoist & 1, (c1w && b1.f4w_count == 0) ? 8 : 0));
  } else {
  if (two_streams) { HIPCHK(hipEventRecord(g_ev[1], g_stream)); HIPCHK(hipStreamWaitEvent(ss, g_ev[1], 0)); }
  // fc4_wgrad may update W4 in place (fused RMSProp): it must not start before fc4_dgrad has read W4
  LAUNCH_ON(ss, K_FC4_WGRAD, launch_tuned(h, K_FC4_WGRAD, a, ss));            // needs d4, a3
  LAUNCH_ON(ss, K_CONV3_WGRAD, launch_tuned(h, K_CONV3_WGRAD, a, ss));        // needs d3p, a2
  LAUNCH(K_CONV3_DGRAD, launch_tuned(h, K_CONV3_DGRAD, a, g_stream));
  BN_BWD(1);
  if (two_streams) { HIPCHK(hipEventRecord(g_ev[2], g_stream)); HIPCHK(hipStreamWaitEvent(ss, g_ev[2], 0)); }
  LAUNCH_ON(ss, K_CONV2_WGRAD, launch_tuned(h, K_CONV2_WGRAD, a, ss));        // needs d2p, a1
  LAUNCH(K_CONV2_DGRAD, launch_tuned(h, K_CONV2_DGRAD, a, g_stream));
  BN_BWD(0);
  LAUNCH(K_CONV1_WGRAD, launch_tuned(h, K_CONV1_WGRAD, a, g_stream, 0, c1w ? 8 : 0));
  if (two_streams) { HIPCHK(hipEventRecord(g_ev[3], ss)); HIPCHK(hipStreamWaitEvent(g_stream, g_ev[3], 0)); }
  }
  UpdateArgs u = make_update_args(h, a);
  if (next) u.next = *next;                 // (memset above left next.B = 0 otherwise)
  if (dp_ov) {
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
  } else   if (h->comm) {
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
  } else if (h->grad_only) {
    u.mode = 1; u.bsz = (float)h->B;                                            // local sums -> g, nothing applied
    LAUNCH(K_UPDATE, launch_update(u, g_stream));
  } else if (EXPERIMENTS && defer_update && a.fuse_rms && !h->bn) {
    u.mode = 0; u.bsz = (float)h->B;
    h->pending_upd = u; h->has_pending_upd = true;            // launched together with the next step's conv1 (run_forward)
  } else {
    u.mode = 0; u.bsz = (float)h->B;
    LAUNCH(K_UPDATE, launch_update(u, g_stream));
    if (h->bn) LAUNCH(K_BN, launch_bn_update(u, g_stream));
  }
  h->train_iterations += 1;                                                   // deepqnetwork.py:168
  h->spec_pending = false;                 // the online parameters move: a forward enqueued before this step no longer is "predict now"
  return SDQN_OK;
}
static int read_cost(sdqn_net_s* h, float* cost_out) {
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
static bool act_sum_partials(const float* part, int A, float* q_out) {
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

// ---- device-resident StateBuffer --------------------------------------------------------------------------
// A ring of SB_SLOTS frame slots in HBM; frame t goes to slot `pos`, the current state is the contiguous window
// of the `hist` slots ending there.  When the ring is full the last hist-1 frames are copied back to the start
// (one 21 KB D2D every SB_SLOTS-hist+1 adds), so an add is ONE 7 KB H2D from a pinned staging slot.
static constexpr int SB_SLOTS = 64;
static uint64_t next_statebuf_gen() { static uint64_t n = 0; return (++n << 40) | 1; }
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
extern "C" int sdqn_statebuf_create(sdqn_statebuf_t* out, int H, int W, int hist) {
  ARGCHK(out, "NULL argument");
  ARGCHK(H > 0 && W > 0 && hist > 0 && hist < SB_SLOTS / 2 && H <= 4096 && W <= 4096, "bad screen geometry %dx%d, history_length %d", H, W, hist);
  STREAMCHK();
  sdqn_statebuf_s* s = new sdqn_statebuf_s(); s->hist = hist; s->pos = hist - 1; s->frame = (int64_t)H * W;
  const int64_t FRAME = s->frame;
  hipError_t e = hipMalloc((void**)&s->d, (size_t)SB_SLOTS * FRAME);
  if (e == hipSuccess) e = hipMemsetAsync(s->d, 0, (size_t)SB_SLOTS * FRAME, g_stream);
  if (e == hipSuccess) e = hipHostMalloc((void**)&s->stage, (size_t)SB_SLOTS * FRAME, hipHostMallocDefault);
  if (e != hipSuccess) { set_error("statebuf_create -> %s", hipGetErrorString(e)); sdqn_statebuf_destroy(s); return SDQN_ERR_HIP; }
  s->host = (uint8_t*)calloc((size_t)hist, FRAME);
  if (!s->host) { set_error("out of host memory"); sdqn_statebuf_destroy(s); return SDQN_ERR_HIP; }
  *out = s; return SDQN_OK;
}
extern "C" int sdqn_statebuf_destroy(sdqn_statebuf_t s) {
  if (!s) return SDQN_OK;
  if (g_stream) hipStreamSynchronize(g_stream);
  if (s->d) hipFree(s->d);
  if (s->stage) hipHostFree(s->stage);
  free(s->host);
  delete s; return SDQN_OK;
}
extern "C" int sdqn_statebuf_add(sdqn_statebuf_t s, const uint8_t* screen) {
  ARGCHK(s && screen, "NULL argument");
  const int64_t FRAME = s->frame;
  s->gen += 1;
  memmove(s->host, s->host + FRAME, (size_t)(s->hist - 1) * FRAME);           // state_buffer.py:17
  memcpy(s->host + (size_t)(s->hist - 1) * FRAME, screen, FRAME);             // :18
  if (s->pos + 1 == SB_SLOTS) {
    // wrap: the newest hist-1 frames move to the front; the sync also retires every staging slot of this lap
    HIPCHK(hipMemcpyAsync(s->d, s->d + (size_t)(SB_SLOTS - (s->hist - 1)) * FRAME, (size_t)(s->hist - 1) * FRAME,
                          hipMemcpyDeviceToDevice, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    s->pos = s->hist - 2;
  }
  s->pos += 1;
  uint8_t* src = s->stage + (size_t)s->pos * FRAME;
  memcpy(src, screen, FRAME);
  HIPCHK(hipMemcpyAsync(s->d + (size_t)s->pos * FRAME, src, FRAME, hipMemcpyHostToDevice, g_stream));
  return SDQN_OK;
}
extern "C" int sdqn_statebuf_reset(sdqn_statebuf_t s) {
  ARGCHK(s, "NULL handle");
  const int64_t FRAME = s->frame;
  s->gen += 1;
  memset(s->host, 0, (size_t)s->hist * FRAME);                                // state_buffer.py:27
  HIPCHK(hipMemsetAsync(s->d + (size_t)(s->pos - s->hist + 1) * FRAME, 0, (size_t)s->hist * FRAME, g_stream));
  return SDQN_OK;
}
extern "C" int sdqn_statebuf_get(sdqn_statebuf_t s, uint8_t* out) {
  ARGCHK(s && out, "NULL argument"); memcpy(out, s->host, (size_t)s->hist * s->frame); return SDQN_OK;
}
static const uint8_t* statebuf_window(sdqn_statebuf_s* s) { return s->d + (size_t)(s->pos - s->hist + 1) * s->frame; }
extern "C" int sdqn_statebuf_read_device(sdqn_statebuf_t s, uint8_t* out) {
  ARGCHK(s && out, "NULL argument");
  HIPCHK(hipMemcpyAsync(out, statebuf_window(s), (size_t)s->hist * s->frame, hipMemcpyDeviceToHost, g_stream));
  HIPCHK(hipStreamSynchronize(g_stream));
  return SDQN_OK;
}
// Acting forward of the buffered state, batch of one, read in place from HBM.  Its head kernel writes the Q-values with system-scope stores
// into mapped host memory that the host pre-filled with a sentinel (an all-ones NaN no sum produces): no D2H copy packet, no stream
// synchronisation — the host polls the A words (bounded; falls back to a blocking wait).  37 -> ~29 us per call on MI355X.
static const uint32_t Q_SENTINEL = 0xFFFFFFFFu;
static int predict_state_enqueue(sdqn_net_s* h, sdqn_statebuf_s* sb) {
  // a fresh slot per forward: the stream runs forwards in order, so by the time a slot comes round again (Q_SLOTS forwards later) any
  // dropped speculation that wrote into it has long finished
  h->q_slot = (h->q_slot + 1) % Q_SLOTS;
  volatile uint32_t* qh = reinterpret_cast<volatile uint32_t*>(h->q_host + h->q_slot * Q_SLOT_FLOATS);
  for (int k = 0; k < h->A; ++k) qh[k] = Q_SENTINEL;
  h->act_last = false;
  if (h->act_on && !h->prof_on) {                                // one launch: conv1 .. fc5 (sdqn_act.hip); 8 stripe partials come back
    { int rcj = join_comm(h); if (rcj) return rcj; }            // (data parallel, overlapped form: W4's update runs on the second stream)
    for (int sp = 1; sp < 8; ++sp) for (int k = 0; k < h->A; ++k) qh[sp * ACT_Q_STRIDE + k] = Q_SENTINEL;
    ActArgs aa; memset(&aa, 0, sizeof aa);
    aa.state = statebuf_window(sb); aa.theta = h->theta; aa.scratch = h->act_scratch; aa.ctl = h->act_ctl;
    aa.q = h->q_host_dev + h->q_slot * Q_SLOT_FLOATS; aa.A = h->A; aa.seq = h->act_seq++;
    if (h->act_inject) {            // every ticket counter of this launch's control block far beyond the item count: all workgroups leave at once
      h->act_inject = false;
      HIPCHK(hipMemsetAsync(h->act_ctl + (size_t)(aa.seq & 3u) * ACT_CTL_WORDS, 0x7F, (size_t)ACT_CTL_WORDS * 4, g_stream));
    }
    LAUNCH(K_ACT, launch_act(aa, true, g_stream));
    h->act_last = true;
    h->spec_pending = true; h->spec_sb = sb; h->spec_gen = sb->gen;
    return SDQN_OK;
  }
  StepArgs a = step_args(h); a.B = 1; a.nz = 1; a.from_ring = 0; a.src = statebuf_window(sb);   // batch of one, read in place
  HeadArgs hd = head_args(h, 0);
  const bool direct = !h->bn;                                    // (--batch_norm: the plain head + a copy, as before)
  if (direct) hd.q = h->q_host_dev + h->q_slot * Q_SLOT_FLOATS;
  h->head_q_system = direct;
  const int rc = run_forward(h, a, hd);
  h->head_q_system = false;
  if (rc) return rc;
  if (!direct) HIPCHK(hipMemcpyAsync(h->q_host + h->q_slot * Q_SLOT_FLOATS, h->q, (size_t)h->A * 4, hipMemcpyDeviceToHost, g_stream));
  h->spec_pending = true; h->spec_sb = sb; h->spec_gen = sb->gen;
  return SDQN_OK;
}
// SDQN_ACT_TRACE=1: host-side segments of the acting path on stderr every 1000 calls (enqueue = sentinel fill + launch; wait = poll)
static bool act_trace() { static const bool on = getenv("SDQN_ACT_TRACE") != nullptr; return on; }
static double g_act_enq_ns = 0, g_act_wait_ns = 0; static long g_act_calls = 0;
static int predict_state_collect(sdqn_net_s* h, float* q_out) {
  volatile uint32_t* qh = reinterpret_cast<volatile uint32_t*>(h->q_host + h->q_slot * Q_SLOT_FLOATS);
  const int nparts = h->act_last ? 8 : 1;                       // (one-launch forward: 8 stripe partials, added here in stripe order)
  auto landed = [&]() { for (int sp = 0; sp < nparts; ++sp) for (int k = 0; k < h->A; ++k) if (qh[sp * ACT_Q_STRIDE + k] == Q_SENTINEL) return false; return true; };
  const auto t0 = std::chrono::steady_clock::now();
  while (!landed()) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {        // never an unbounded spin
      HIPCHK(hipStreamSynchronize(g_stream));
      if (!landed() && h->act_last && h->spec_sb) {
        // the one-launch forward gave up on a hand-off (its polls are bounded): never again in this process — the five-launch forward
        // of the same state instead, said once on stderr
        h->act_on = false; h->act_fallbacks += 1;
        fprintf(stderr, "simple_dqn_amd: the one-launch acting forward did not complete; using the five-launch forward from now on\n");
        sdqn_statebuf_s* sb = (sdqn_statebuf_s*)h->spec_sb;
        int rc = predict_state_enqueue(h, sb); if (rc) return rc;
        return predict_state_collect(h, q_out);
      }
      if (!landed()) { set_error("predict_state: the head kernel finished without delivering its Q-values"); return SDQN_ERR_STATE; }
      break;
    }
  }
  for (int k = 0; k < h->A; ++k) {
    float qv = 0.0f;
    for (int sp = 0; sp < nparts; ++sp) { uint32_t w = qh[sp * ACT_Q_STRIDE + k]; float f; memcpy(&f, &w, 4); qv = sp ? qv + f : f; }
    q_out[k] = qv;
  }
  h->spec_pending = false;
  if (act_trace()) {
    g_act_wait_ns += (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    if (++g_act_calls % 1000 == 0) { fprintf(stderr, "act trace: enqueue %.2f us, wait %.2f us per call\n", g_act_enq_ns / 1e6, g_act_wait_ns / 1e6); g_act_enq_ns = g_act_wait_ns = 0; }
  }
  return SDQN_OK;
}
extern "C" int sdqn_net_predict_state(sdqn_net_t h, sdqn_statebuf_t sb, float* q_out) {
  ARGCHK(h && sb && q_out, "NULL argument");
  ARGCHK((size_t)sb->hist * sb->frame == (h->gen ? h->gen->state_bytes() : (size_t)STATE), "state buffer geometry differs from the network's");
  if (h->gen) { GENCHK(h->gen->predict_dev(statebuf_window(sb), 1, q_out, false)); return SDQN_OK; }
  // a forward enqueued ahead by sdqn_net_act_step for exactly this state and these parameters: only collect it
  if (!(h->spec_pending && h->spec_sb == sb && h->spec_gen == sb->gen)) {
    const auto t0 = std::chrono::steady_clock::now();
    int rc = predict_state_enqueue(h, sb); if (rc) return rc;
    if (act_trace()) g_act_enq_ns += (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  }
  return predict_state_collect(h, q_out);
}
// agent.py:55-59: the greedy action of the buffered state (first index of the maximum, as np.argmax) — predict_state + argmax in one call
extern "C" int sdqn_net_act_greedy(sdqn_net_t h, sdqn_statebuf_t sb, int* action, float* q_out) {
  ARGCHK(h && sb && action, "NULL argument");
  float q[MAX_ACTIONS];
  int rc = sdqn_net_predict_state(h, sb, q); if (rc) return rc;
  const int A = h->A;
  int best = 0;
  for (int k = 1; k < A; ++k) if (q[k] > q[best] || (q[k] != q[k] && q[best] == q[best])) best = k;   // np.argmax: the first maximum, a NaN counts as one
  *action = best;
  if (q_out) memcpy(q_out, q, (size_t)A * 4);
  return SDQN_OK;
}
// One environment transition in ONE call (agent.py:48-85 + :62: `buf.add(screen)`, optionally `mem.add(action, reward, screen, terminal)`):
// the frame goes to the device-resident state buffer and, with a replay handle, into the ring; with `speculate` the acting forward of the
// NEW state is enqueued right away — the next step's sdqn_net_predict_state then finds its Q-values already on the host (or on their way)
// instead of starting five dependent launches.  Values are identical to a forward started later: the speculation is dropped whenever the
// state buffer or the online parameters change before it is used.
extern "C" int sdqn_net_act_step(sdqn_net_t h, sdqn_statebuf_t sb, sdqn_replay_t r, const uint8_t* screen, int action, int64_t reward,
                                 int terminal, int speculate) {
  ARGCHK(h && sb && screen, "NULL argument");
  ARGCHK(!r || r->frame == (int64_t)sb->frame, "the replay memory's screens (%lld bytes) and the state buffer's (%lld) differ: one screen pointer feeds both",
         (long long)(r ? r->frame : 0), (long long)sb->frame);             // (replay_memory.py:28's assert: sdqn_replay_add copies r->frame bytes)
  int rc = sdqn_statebuf_add(sb, screen); if (rc) return rc;
  if (r) { rc = sdqn_replay_add(r, action, reward, screen, terminal); if (rc) return rc; }
  if (speculate && !h->gen && (size_t)sb->hist * sb->frame == (size_t)STATE) return predict_state_enqueue(h, sb);
  return SDQN_OK;
}

// Test / measurement hook: ONE one-launch acting forward of the buffered state with per-workgroup phase stamps ({kind, clock64} pairs,
// sdqn_act.hip) — blocking; q_out [A], stamps_out [ACT_GRID][2 * ACT_STAMPS] (either may be NULL).  tools/exp/act_stamps.py reads them.
extern "C" int sdqn_net_debug_act(sdqn_net_t h, sdqn_statebuf_t sb, float* q_out, unsigned long long* stamps_out) {
  ARGCHK(h && sb, "NULL argument");
  ARGCHK(h->act_scratch && (size_t)sb->hist * sb->frame == (size_t)STATE, "the one-launch acting forward needs a float32 network without batch_norm and the standard geometry");
  const size_t nst = (size_t)ACT_GRID * 2 * ACT_STAMPS;
  unsigned long long* d_st = nullptr;
  { int rcj = join_comm(h); if (rcj) return rcj; }
  if (stamps_out) { HIPCHK(hipMalloc((void**)&d_st, nst * 8)); HIPCHK(hipMemsetAsync(d_st, 0, nst * 8, g_stream)); }
  ActArgs aa; memset(&aa, 0, sizeof aa);
  aa.state = statebuf_window(sb); aa.theta = h->theta; aa.scratch = h->act_scratch; aa.ctl = h->act_ctl;
  aa.q = h->act_q; aa.A = h->A; aa.seq = h->act_seq++; aa.stamps = d_st;
  HIPCHK(hipMemsetAsync(h->act_q, 0xFF, (size_t)Q_SLOT_FLOATS * 4, g_stream));
  hipError_t e = launch_act(aa, false, g_stream);
  if (e == hipSuccess) e = hipMemcpyAsync(h->h_f, h->act_q, (size_t)Q_SLOT_FLOATS * 4, hipMemcpyDeviceToHost, g_stream);
  if (e == hipSuccess && stamps_out) e = hipMemcpyAsync(stamps_out, d_st, nst * 8, hipMemcpyDeviceToHost, g_stream);
  if (e == hipSuccess) e = hipStreamSynchronize(g_stream);
  if (d_st) hipFree(d_st);
  HIPCHK(e);
  float qv[MAX_ACTIONS];
  if (!act_sum_partials(h->h_f, h->A, qv)) { set_error("the one-launch acting forward did not deliver every stripe"); return SDQN_ERR_STATE; }
  if (q_out) memcpy(q_out, qv, (size_t)h->A * 4);
  return SDQN_OK;
}

// Experiments build: the training forward conv chain (conv1 -> conv2 -> conv3 of ns (state, net) pairs per XCC) as ONE XCC-local launch in the
// acting kernel's arithmetic — a timing probe, outputs unused (sdqn_act.hip: chain_probe_kernel; tools/exp/chain_probe.py).
// us_per_launch: mean over `reps` launches of the dispatch packet's own begin -> end time (hipExtLaunchKernel events: the duration rocprofv3's
// kernel trace reports; the control block is re-zeroed by a memset before each launch, outside it).
extern "C" int sdqn_exp_chain_probe(sdqn_net_t h, int ns, int grid, int reps, float* us_per_launch, unsigned long long* stamps_out) {
#ifdef SDQN_EXPERIMENTS
  ARGCHK(h && h->act_scratch && ns != 0 && ns >= -8 && ns <= 8 && grid >= 8 && grid <= 4096 && reps >= 1 && us_per_launch, "bad arguments");
  uint8_t* st = nullptr; float* scr = nullptr; unsigned* ctl = nullptr; unsigned long long* d_st = nullptr;
  const int nsa = ns < 0 ? -ns : ns;
  const size_t nstate = (size_t)8 * nsa * STATE, nscr = (size_t)8 * nsa * ACT_XCC_FLOATS * 4, nctl = (size_t)(8 * 16 + 8 * 8 * 3 * 16 + 64) * 4;
  const size_t nst = (size_t)grid * 2 * ACT_STAMPS;
  HIPCHK(hipMalloc((void**)&st, nstate)); HIPCHK(hipMalloc((void**)&scr, nscr)); HIPCHK(hipMalloc((void**)&ctl, nctl));
  if (stamps_out) HIPCHK(hipMalloc((void**)&d_st, nst * 8));
  { std::vector<uint8_t> hs(nstate); uint32_t v = 12345u; for (auto& b : hs) { v = v * 1664525u + 1013904223u; b = (uint8_t)(v >> 24); }
    HIPCHK(hipMemcpy(st, hs.data(), nstate, hipMemcpyHostToDevice)); }
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  ActArgs aa; memset(&aa, 0, sizeof aa);
  aa.state = st; aa.theta = h->theta; aa.scratch = scr; aa.ctl = ctl; aa.q = h->act_q; aa.A = ns < 0 ? 1 : 0; aa.seq = 0;      // (ns < 0: static tickets)
  if (ns < 0) ns = -ns;
  double total = 0; hipError_t e = hipSuccess;
  for (int r = 0; r < reps + 3 && e == hipSuccess; ++r) {
    const bool last = r == reps + 2;
    aa.stamps = last ? d_st : nullptr;
    if (last && d_st) HIPCHK(hipMemsetAsync(d_st, 0, nst * 8, g_stream));
    HIPCHK(hipMemsetAsync(ctl, 0, nctl, g_stream));
    e = launch_chain_probe(aa, ns, grid, e0, e1, g_stream);
    HIPCHK(hipStreamSynchronize(g_stream));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 3 && !last) total += ms; else if (r >= 3 && last && reps == 1) total += ms;
  }
  *us_per_launch = (float)(total / (reps > 1 ? reps - 1 : 1) * 1e3);
  if (e == hipSuccess && stamps_out) e = hipMemcpy(stamps_out, d_st, nst * 8, hipMemcpyDeviceToHost);
  hipEventDestroy(e0); hipEventDestroy(e1); hipFree(st); hipFree(scr); hipFree(ctl); if (d_st) hipFree(d_st);
  HIPCHK(e);
  return SDQN_OK;
#else
  (void)h; (void)ns; (void)grid; (void)reps; (void)us_per_launch; (void)stamps_out;
  set_error("sdqn_exp_chain_probe is part of the experiments build (make -C simple_dqn_amd/csrc experiments)"); return SDQN_ERR_ARG;
#endif
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
  const int sl = h->stage_next; h->stage_next ^= 1;
  if (!h->h_stage[sl]) {
    HIPCHK(hipHostMalloc((void**)&h->h_stage[sl], 2 * sb + small, hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&h->stage_ev[sl], hipEventDisableTiming));
  }
  if (h->stage_busy[sl]) { HIPCHK(hipEventSynchronize(h->stage_ev[sl])); h->stage_busy[sl] = false; }
  ARGCHK(!ours || owner->tuned_geom, "replay geometry differs from the network's");
  uint8_t* st = h->h_stage[sl];
  if (!ours) { memcpy(st, pre, sb); memcpy(st + sb, post, sb); }
  uint8_t* sm = st + 2 * sb;                                                  // [rewards 8 B | actions B | terminals B], as on the device
  memcpy(sm, rewards, (size_t)h->B * 8); memcpy(sm + (size_t)h->B * 8, actions, h->B); memcpy(sm + (size_t)h->B * 9, terminals, h->B);
  // every copy is a packet of its own in the stream: 2 instead of 5 (1 with `reuse`)
  if (!reuse) {
    HIPCHK(hipMemcpyAsync(h->st_states, ours ? pre : st, 2 * sb, hipMemcpyHostToDevice, g_stream));     // (a ReplayMemory's pre | post are one block too)
    if (ours) { HIPCHK(hipEventRecord(owner->mb_upload_ev, g_stream)); }   // waited for before this call returns
  }
  HIPCHK(hipMemcpyAsync(h->st_rew, sm, small, hipMemcpyHostToDevice, g_stream));
  HIPCHK(hipEventRecord(h->stage_ev[sl], g_stream)); h->stage_busy[sl] = true;
  StepArgs a = step_args(h); a.from_ring = 0; a.src = reuse ? owner->d_pre : h->st_states;
  HeadArgs hd = head_args(h, 1);
  int rc = run_train(h, a, hd); if (rc) return rc;
  if (ours && !reuse) { HIPCHK(hipEventSynchronize(owner->mb_upload_ev)); }
  if (cost_out) return read_cost(h, cost_out);
  return SDQN_OK;
}

static PrepArgs prep_args(sdqn_net_s* h, sdqn_replay_s* r, const int64_t* pinned_idx) {
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
static int check_ring_actions(sdqn_net_s* h, sdqn_replay_s* r, const int64_t* idx) {
  for (int i = 0; i < r->B; ++i)
    ARGCHK(idx[i] >= 0 && idx[i] < r->size && r->actions[idx[i]] < h->A,
           "ring slot %lld holds action %d but the network has %d actions", (long long)idx[i], (int)r->actions[idx[i]], h->A);
  return SDQN_OK;
}
// do_prep: launch the standalone prep for THIS step; next_pinned: fold the NEXT step's prep into the update
// hoist_in: this step's target conv1 / conv2 already ran inside the previous step; hoist_out: this step carries the next one's
static bool hoist_possible(sdqn_net_s* h) {
  return EXPERIMENTS && h->hoist && h->B <= 32 && !h->bn && h->cfg.datatype == 0 && h->fused_launches && !h->two_streams &&
         !(h->comm && h->comm2 && h->dp_overlap) && h->f4_share[0] == 100 && h->f4_share[1] == 0 && h->theta_t != h->theta;
}
static int train_replay_slot(sdqn_net_s* h, sdqn_replay_s* r, const int64_t* pinned_idx, bool do_prep = true,
                             const int64_t* next_pinned = nullptr, bool hoist_in = false, bool hoist_out = false, double* zero8 = nullptr,
                             bool defer_update = false) {
  if (do_prep) { PrepArgs p = prep_args(h, r, pinned_idx); LAUNCH(K_PREP, launch_prep(p, g_stream, zero8)); }
  StepArgs a = step_args(h); a.from_ring = 1; a.src = r->d_ring; a.idx = h->d_idx;
  HeadArgs hd = head_args(h, 1);
  const int hoist = (hoist_out ? 1 : 0) | (hoist_in ? 2 : 0);
  if (hoist_out) { a.idx_t = h->d_idx_t; hd.next_idx_pinned = next_pinned; hd.next_idx_dev = h->d_idx_t; hd.next_B = h->B; }
  // the slot's HOST address (pinned_idx is its device alias): conv1's tiles take their indexes from the kernel arguments
  h->host_idx_cur = r->h_idx + (pinned_idx - r->d_idx_view);
  int rc;
  if (next_pinned) { PrepArgs np = prep_args(h, r, next_pinned); rc = run_train(h, a, hd, &np, hoist, defer_update); }
  else rc = run_train(h, a, hd, nullptr, hoist);
  h->host_idx_cur = nullptr;
  return rc;
}
// float64 / other geometries: sample on the host, gather on the device into the replay handle's minibatch buffers, train from there
static int gen_train_replay(sdqn_net_s* h, sdqn_replay_s* r, const int64_t* idx_host) {
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
  bool hoisted = false;                      // step i's target conv1 / conv2 were computed during step i - 1
  for (int i = 0; i < n_steps; ++i) {
    next_pinned = nullptr;
    if (i + 1 < n_steps) {
      int rc = sample_checked(mt, r->terminals, r->count, r->current, r->hist, r->B, idx.data(), nullptr); if (rc) return rc;
      rc = check_ring_actions(h, r, idx.data()); if (rc) return rc;
      rc = replay_push_idx(r, idx.data(), &next_slot, &next_pinned); if (rc) return rc;
    }
    // the target-net forward of step i+1 (theta- and the next indexes only) rides in step i's launches: run_forward / launch_kernel
    const bool hoist_out = next_pinned != nullptr && hoist_possible(h);
    // round 3: this step's optimizer pass is launched together with the NEXT step's conv1 (default fp32 single-learner path, B <= 32:
    // the fused kernel takes the next indexes from its arguments); the last step of a call keeps its own update launch
    const bool defer = EXPERIMENTS && next_pinned != nullptr && h->fuse_upd && h->B <= 32 && h->cfg.datatype == 0 && !h->bn && !h->comm && !h->grad_only &&
                       !h->keep_grads && !h->hoist && h->conv1_bf16 && h->fused_launches && !h->two_streams && h->nw_override[K_CONV1_FWD] == 0 &&
                       h->theta_t != h->theta && !(r->flags & SDQN_REPLAY_ZERO_COPY);
    int rc = train_replay_slot(h, r, pinned, /*do_prep=*/i == 0, next_pinned, hoisted, hoist_out, i == 0 ? h->cost_accum : nullptr, defer); if (rc) return rc;
    hoisted = hoist_out;
    rc = replay_release_idx_batched(r, slot, false);
    if (rc) return rc;
    slot = next_slot; pinned = next_pinned;
  }
  if (h->has_pending_upd) { set_error("internal: a deferred update outlived train_many"); return SDQN_ERR_STATE; }
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

extern "C" int sdqn_net_update_target(sdqn_net_t h) {
  ARGCHK(h, "NULL handle");
  if (h->gen) { GENCHK(h->gen->update_target()); return SDQN_OK; }
  { int rc = join_comm(h); if (rc) return rc; }
  if (h->theta_t != h->theta) {
    HIPCHK(hipMemcpyAsync(h->theta_t, h->theta, (size_t)h->NP * 4, hipMemcpyDeviceToDevice, g_stream));   // deepqnetwork.py:102-105
    if (h->w1p[0] && h->w1p[1] != h->w1p[0])
      HIPCHK(hipMemcpyAsync(h->w1p[1], h->w1p[0], (size_t)3 * W1P_PLANE * 2, hipMemcpyDeviceToDevice, g_stream));
    if (h->wpt[0] && h->wpt[1] != h->wpt[0])     // (the transposed conv planes of the three planes only: [OFF2, OFF4) of each)
      for (int q = 0; q < 3; ++q)
        HIPCHK(hipMemcpyAsync(h->wpt[1] + (size_t)q * XP_PLANE + OFF2, h->wpt[0] + (size_t)q * XP_PLANE + OFF2, (size_t)(OFF4 - OFF2) * 2, hipMemcpyDeviceToDevice, g_stream));
    if (h->cfg.datatype == 1) {
      HIPCHK(hipMemcpyAsync(h->wh[1], h->wh[0], (size_t)OFF5 * 2, hipMemcpyDeviceToDevice, g_stream));
      HIPCHK(hipMemcpyAsync(h->wht[1], h->wht[0], (size_t)OFF5 * 2, hipMemcpyDeviceToDevice, g_stream));
    }
  }
  return SDQN_OK;
}
// The second half of a data-parallel step without a communicator: the gradient sums currently in the flat buffer g
// (written by a grad_only step and/or sdqn_net_set_weights(which = 3)) are applied with divisor bsz — exactly what every
// rank does after the all-reduce with bsz = nranks * batch_size (A9: grad / be.bsz, deepqnetwork.py:165).
extern "C" int sdqn_net_apply_update(sdqn_net_t h, double bsz) {
  ARGCHK(h && bsz > 0, "bad arguments");
  if (h->gen) { set_error("data parallel (grad_only / apply_update) is implemented for the 84x84x4 float32 / float16 configurations"); return SDQN_ERR_STATE; }
  { int rc = join_comm(h); if (rc) return rc; }
  h->spec_pending = false;
  StepArgs a = step_args(h);
  UpdateArgs u = make_update_args(h, a);
  u.mode = 2; u.bsz = (float)bsz; u.skip_fc4 = 0;
  if (h->half_payload_pending) {           // the gradient came back through the half payload: same overflow rule as the RCCL path
    u.ovf_flag = h->ovf_flag; u.ovf_count = h->ovf_count; u.ovf_dynamic = h->dp_half_scale_log2 < 0 ? 1 : 0;
    h->half_payload_pending = false;
  }
  LAUNCH(K_UPDATE, launch_update(u, g_stream));
  if (h->bn) LAUNCH(K_BN, launch_bn_update(u, g_stream));
  return SDQN_OK;
}
// float16 data parallel without a communicator: the two passes that bracket ncclAllReduce(ncclFloat16) in run_train, callable
// on their own so that the exchange can be done by the caller (tests: gloo across two processes sharing one GPU).
//   to_half  : g * 2^k -> IEEE half (k = the device-side payload scale), copied to the caller's buffer
//   from_half: the caller's summed half payload -> g / 2^k in fp32; a non-finite value raises the step's overflow flag, which the
//              next sdqn_net_apply_update honours (parameters untouched, skipped-step counter + 1, dynamic scale halved)
extern "C" int sdqn_net_grad_to_half(sdqn_net_t h, uint16_t* out, int64_t n) {
  ARGCHK(h && out, "NULL argument");
  ARGCHK(h->gh && h->ovf_flag, "not a float16 network");
  ARGCHK(n == h->NP, "the flat gradient holds %lld values, got %lld", (long long)h->NP, (long long)n);
  { int rc = join_comm(h); if (rc) return rc; }
  LAUNCH(K_UPDATE, launch_grad_to_half(h->g, h->gh, h->NP, h->ovf_flag, g_stream));
  HIPCHK(hipStreamSynchronize(g_stream));
  HIPCHK(hipMemcpy(out, h->gh, (size_t)n * 2, hipMemcpyDeviceToHost));
  return SDQN_OK;
}
extern "C" int sdqn_net_grad_from_half(sdqn_net_t h, const uint16_t* in, int64_t n) {
  ARGCHK(h && in, "NULL argument");
  ARGCHK(h->gh && h->ovf_flag, "not a float16 network");
  ARGCHK(n == h->NP, "the flat gradient holds %lld values, got %lld", (long long)h->NP, (long long)n);
  { int rc = join_comm(h); if (rc) return rc; }
  HIPCHK(hipStreamSynchronize(g_stream));
  HIPCHK(hipMemcpy(h->gh, in, (size_t)n * 2, hipMemcpyHostToDevice));
  LAUNCH(K_UPDATE, launch_grad_from_half(h->gh, h->g, h->NP, h->ovf_flag, g_stream));
  h->half_payload_pending = true;
  return SDQN_OK;
}
// {overflow flag of the last from-half pass, log2 of the payload scale, clean steps since the scale last moved} (sync)
extern "C" int sdqn_net_half_payload_state(sdqn_net_t h, int* flag, int* scale_log2, int* clean_steps) {
  ARGCHK(h, "NULL handle");
  ARGCHK(h->ovf_flag, "not a float16 network");
  { int rc = join_comm(h); if (rc) return rc; } HIPCHK(hipStreamSynchronize(g_stream));
  int st[4]; HIPCHK(hipMemcpy(st, h->ovf_flag, 16, hipMemcpyDeviceToHost));
  if (flag) *flag = st[0]; if (scale_log2) *scale_log2 = st[1]; if (clean_steps) *clean_steps = st[2];
  return SDQN_OK;
}
extern "C" int sdqn_net_sync(sdqn_net_t h) {
  ARGCHK(h, "NULL handle");
  if (h->gen) { HIPCHK(hipStreamSynchronize(g_stream)); return SDQN_OK; }
  int rc = join_comm(h); if (rc) return rc;
  // short waits are polled (a blocking hipStreamSynchronize costs 10-20 us of wake-up latency: 1 % of a 20-step call); anything
  // longer than ~2 ms falls through to the blocking wait
  { const auto t0 = std::chrono::steady_clock::now();
    while (hipStreamQuery(g_stream) == hipErrorNotReady)
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    (void)hipGetLastError(); }
  HIPCHK(hipStreamSynchronize(g_stream));
  // in-launch hand-offs are bounded spins: a producer that never ran would show up here, never as a hung GPU.  Only launches of
  // the two opt-in hand-off variants can raise the words, so only calls that enqueued one pay the two small read-backs
  // (~40 us: measured as 6 % of the driver's 20-step timed region when they ran on every sync)
  if (!EXPERIMENTS || !h->handoff_launched) return SDQN_OK;
  h->handoff_launched = false;
  unsigned timed_out = 0;
  HIPCHK(hipMemcpy(&timed_out, h->f4d_flags + (NIN4 / 32) * 16, 4, hipMemcpyDeviceToHost));
  if (timed_out) { set_error("fc4_wgrad waited for a fc4_dgrad tile that never signalled (in-launch hand-off timed out): results are invalid"); return SDQN_ERR_STATE; }
  HIPCHK(hipMemcpy(&timed_out, h->w1_ctr + 5, 4, hipMemcpyDeviceToHost));
  if (timed_out) { set_error("fc4_dgrad waited for head workgroups of its own launch that never signalled (in-launch hand-off timed out): results are invalid"); return SDQN_ERR_STATE; }
  HIPCHK(hipMemcpy(&timed_out, h->w1_ctr + 1, 4, hipMemcpyDeviceToHost));
  if (timed_out) { set_error("conv1 waited for W1 blocks of the fused update that never signalled (in-launch hand-off timed out): results are invalid"); return SDQN_ERR_STATE; }
  return SDQN_OK;
}
extern "C" int sdqn_net_last_q(sdqn_net_t h, float* preq, float* maxpostq) {
  ARGCHK(h, "NULL handle");
  if (h->gen) { GENCHK(h->gen->last_q(preq, maxpostq, false)); return SDQN_OK; }
  const size_t nq = (size_t)h->B * h->A;
  HIPCHK(hipMemcpyAsync(h->h_f, h->q, nq * 4, hipMemcpyDeviceToHost, g_stream));
  HIPCHK(hipMemcpyAsync(h->h_f + nq, h->maxq, (size_t)h->B * 4, hipMemcpyDeviceToHost, g_stream));
  HIPCHK(hipStreamSynchronize(g_stream));
  if (preq) memcpy(preq, h->h_f, nq * 4);
  if (maxpostq) memcpy(maxpostq, h->h_f + nq, (size_t)h->B * 4);
  return SDQN_OK;
}
// fp16 data parallel: train steps whose all-reduced half gradient overflowed and were therefore skipped (sync)
extern "C" int sdqn_net_overflow_steps(sdqn_net_t h, int64_t* n) {
  ARGCHK(h && n, "NULL argument");
  *n = 0;
  if (!h->ovf_count) return SDQN_OK;
  { int rc = join_comm(h); if (rc) return rc; } HIPCHK(hipStreamSynchronize(g_stream));
  HIPCHK(hipMemcpy(n, h->ovf_count, 8, hipMemcpyDeviceToHost));
  return SDQN_OK;
}
extern "C" int sdqn_net_train_iterations(sdqn_net_t h, int64_t* n) { ARGCHK(h && n, "NULL"); *n = h->train_iterations; return SDQN_OK; }

extern "C" int sdqn_net_set_epoch(sdqn_net_t h, int epoch) { ARGCHK(h && epoch >= 0, "bad epoch"); h->epoch = epoch; return SDQN_OK; }

extern "C" int sdqn_net_set_option(sdqn_net_t h, const char* name, int value) {
  ARGCHK(h && name, "NULL argument");
  if (h->gen) {                                   // the generic path has no tuning knobs; the ones that change semantics are refused
    if (!strcmp(name, "dp_overlap") && value < 0) return SDQN_OK;      // (auto: nothing to overlap without a communicator)
    if (!strcmp(name, "grad_only") || !strcmp(name, "dp_overlap") || !strcmp(name, "keep_gradients")) {
      ARGCHK(value == 0 || !strcmp(name, "keep_gradients"), "option %s is implemented for the 84x84x4 float32 / float16 configurations", name);
    }
    return SDQN_OK;
  }
  if (!strcmp(name, "keep_gradients")) h->keep_grads = value != 0;
  else if (!strcmp(name, "grad_only")) h->grad_only = value != 0;
  else if (!strcmp(name, "h16_wgrad_mfma")) h->h16_wgrad_mfma = value != 0;
  else if (!strcmp(name, "hoist")) { if (!EXPERIMENTS && value) EXP_OPTION_REFUSED(name); h->hoist = value != 0; }
  else if (!strcmp(name, "dp_half")) h->dp_half = value != 0;              // fp16 mode: half (1, default) or fp32 (0) all-reduce payload
  else if (!strcmp(name, "dp_half_scale_log2")) {          // -1: dynamic (default); n >= 0: fixed payload scale 2^n
    ARGCHK(value >= -1 && value <= 40 && h->ovf_flag, "bad scale (or not a float16 network)");
    { int rc_ = join_comm(h); if (rc_) return rc_; } HIPCHK(hipStreamSynchronize(g_stream));
    h->dp_half_scale_log2 = value;
    const int st0[4] = {0, value < 0 ? 10 : value, 0, 0};
    HIPCHK(hipMemcpy(h->ovf_flag, st0, 16, hipMemcpyHostToDevice));
  }
  else if (!strcmp(name, "dp_half_scale_seed")) {          // dynamic mode kept, scale STARTS at 2^value (tests of the scale state machine)
    ARGCHK(value >= 0 && value <= 15 && h->ovf_flag, "bad scale seed (0..15; float16 networks only)");
    { int rc_ = join_comm(h); if (rc_) return rc_; } HIPCHK(hipStreamSynchronize(g_stream));
    h->dp_half_scale_log2 = -1;
    const int st0[4] = {0, value, 0, 0};
    HIPCHK(hipMemcpy(h->ovf_flag, st0, 16, hipMemcpyHostToDevice));
  }
  else if (!strcmp(name, "two_streams")) { if (!EXPERIMENTS && value) EXP_OPTION_REFUSED(name); ARGCHK(!(value && h->bn), "two_streams is not available with batch_norm"); h->two_streams = value != 0; }
  else if (!strcmp(name, "fused_launches")) h->fused_launches = value != 0;
  else if (!strcmp(name, "conv1_bf16")) h->conv1_bf16 = value != 0;     // 0: conv1_fwd on the fp32-MFMA engine (round-2 kernel)
  else if (!strcmp(name, "fuse_dbg")) { if (!EXPERIMENTS && value) EXP_OPTION_REFUSED(name); h->fuse_dbg = value; }
  else if (!strcmp(name, "fwd_rb")) { if (!EXPERIMENTS && value) EXP_OPTION_REFUSED(name); h->fwd_rb = value; }
  else if (!strcmp(name, "wt")) h->wt = value;
  else if (!strcmp(name, "prep_inline")) h->prep_inline = value != 0;
  else if (!strcmp(name, "r3_xcd")) h->r3_xcd = value;
  else if (!strcmp(name, "head_f4d")) { if (!EXPERIMENTS && value) EXP_OPTION_REFUSED(name); h->head_f4d = value != 0; }         // 1: head + fc4_dgrad in one launch (in-launch hand-off of delta4)
  else if (!strcmp(name, "fuse_upd")) { if (!EXPERIMENTS && value) EXP_OPTION_REFUSED(name); h->fuse_upd = value != 0; }         // 0: the optimizer pass is always its own launch
  else if (!strcmp(name, "conv1w_bf16")) h->conv1w_bf16 = value;   // 0: conv1_wgrad on the fp32-MFMA engine (round-2 kernel)
  else if (!strcmp(name, "conv3_c36")) h->conv3_c36 = value != 0;       // 0: conv3_fwd on the engine's 32-deep chunks (round-2 kernel)
  else if (!strcmp(name, "f4w_early")) { if (!EXPERIMENTS && value) EXP_OPTION_REFUSED(name); h->f4w_early = value != 0; }       // 0: fc4_wgrad inside bwd3 (round-2 launch structure)
  else if (!strcmp(name, "xcd_map")) h->xcd_map = value != 0;
  else if (!strcmp(name, "dp_sync_replicas")) h->dp_sync_replicas = value != 0;   // before dp_init
  else if (!strcmp(name, "dp_overlap")) {                  // before dp_init: -1 auto (probe + vote, default), 1 forced on, 0 single all-reduce on the library stream
    ARGCHK(value >= -2 && value <= 1, "dp_overlap must be -1 (auto), 0 or 1 (-2: auto also for a 1-rank communicator, tests)");
    ARGCHK(!h->comm, "dp_overlap is chosen before sdqn_dp_init (afterwards: sdqn_dp_set_overlap)");
    h->dp_overlap_req = value; h->dp_overlap = value == 1;
  }
  else if (!strcmp(name, "f4_share3")) h->f4_share[0] = value;
  else if (!strcmp(name, "f4_share2")) h->f4_share[1] = value;
  else if (!strcmp(name, "profile_mode")) { ARGCHK(value == 0 || value == 1, "profile_mode must be 0 (event markers) or 1 (kernel-packet timestamps)"); h->prof_mode = value; }
  else if (!strcmp(name, "profile_every")) { ARGCHK(value >= 1, "profile_every must be >= 1"); h->prof_every = value; }
  else if (!strncmp(name, "xcd:", 4)) {                    // tuning: XCD-map problem mask of kernel id (value = mask + 1, 0 = built-in)
    int id = atoi(name + 4);
    if (id < 0 || id >= K_COUNT || value < 0 || value > 8) { set_error("bad xcd override"); return SDQN_ERR_ARG; }
    h->xcd_mask[id] = value;
  }
  else if (!strncmp(name, "rb:", 3)) {                     // register-blocked routine (B >= 128): menu entry of kernel id, 0 = unblocked
    int id = atoi(name + 3);
    if (id < 0 || id >= 12 || value < 0 || value > 8) { set_error("bad rb override"); return SDQN_ERR_ARG; }
    if (!EXPERIMENTS && value) EXP_OPTION_REFUSED(name);
    h->rb[id] = value;
  }
  else if (!strcmp(name, "act_inject_failure")) {         // tests: the next one-launch acting forward delivers nothing (exercises the host's fallback)
    ARGCHK(h->act_scratch, "act_inject_failure needs a network with the one-launch acting forward");
    h->act_inject = value != 0;
  }
  else if (!strcmp(name, "act_kernel")) {                  // 1: acting forward as one launch (default where available), 0: the five forward launches
    ARGCHK(value == 0 || h->act_scratch, "act_kernel needs a float32 network without batch_norm");
    h->act_on = value != 0; h->spec_pending = false;
  }
  else if (!strcmp(name, "bt_xcd")) h->bt_xcd = value != 0;             // 0: round-robin block placement in the B >= 128 backward launches
  else if (!strcmp(name, "bt")) h->bt_on = value != 0;                  // 0: B >= 128 on the latency engine's launch forms (round 3)
  else if (!strcmp(name, "bt_planes")) {                   // plane mode: 9 / 6 partial products, 0 = fp32 MFMA (B >= 128 float32 networks only)
    ARGCHK(value == 0 || value == 6 || value == 9, "bt_planes must be 0, 6 or 9");
    if (!EXPERIMENTS && value) EXP_OPTION_REFUSED(name);
    ARGCHK(value == 0 || h->wpm, "bt_planes needs a float32 network with batch_size >= 128 (no batch_norm)");
    h->xp = value;
  }
  else if (!strcmp(name, "bt_x")) {                        // arithmetic of every block-tile launch: 0 fp32 MFMA, 9 / 6 exact bf16x3 splits
    if (!EXPERIMENTS && value) EXP_OPTION_REFUSED(name);
    ARGCHK(value == 0 || value == 6 || value == 9 || value == 19 || value == 16, "bt_x must be 0, 6 or 9");
    for (int i = 0; i < K_COUNT; ++i) h->btx[i] = value;
  }
  else if (!strncmp(name, "btx:", 4)) {
    int id = atoi(name + 4);
    if (!EXPERIMENTS && value) EXP_OPTION_REFUSED(name);
    ARGCHK(id >= 0 && id < K_COUNT && (value == 0 || value == 6 || value == 9), "bad btx override");
    h->btx[id] = value;
  }
  else if (!strncmp(name, "bt:", 3)) {                     // block-tile engine: menu entry of kernel id (0 built-in, -1 latency engine)
    int id = atoi(name + 3);
    if (id < 0 || id >= K_COUNT || value < -1 || value > (EXPERIMENTS ? 13 : 8)) { set_error("bad bt override"); return SDQN_ERR_ARG; }
    if (!EXPERIMENTS && id == K_WGRADS && value >= 6) EXP_OPTION_REFUSED(name);      // (one of the three weight gradients only: timing experiments)
    h->bt[id] = value;
  }
  else if (!strcmp(name, "bwd_order")) { if (!EXPERIMENTS && value) EXP_OPTION_REFUSED(name); h->bwd_order = value; }
  else if (!strcmp(name, "s4")) {                          // tuning: split-K slabs of the fc4 forward (1..7; 7 allocated)
    if (value < 1 || value > h->S4_cap) { set_error("bad s4 (1..%d)", h->S4_cap); return SDQN_ERR_ARG; }
    { int rc_ = join_comm(h); if (rc_) return rc_; } HIPCHK(hipStreamSynchronize(g_stream));
    h->S4 = value;
  }
  else if (!strncmp(name, "tps:", 4)) {                    // tuning: 32-deep K-chunks per split-K slab of conv layer 1..3 wgrad
    int l = atoi(name + 4);
    if (l < 1 || l > 3 || value < 1) { set_error("bad tps override"); return SDQN_ERR_ARG; }
    const int pix[3] = {PIX1, PIX2, PIX3};
    const int T = ceil_div(h->B * pix[l - 1], 32), ns = ceil_div(T, value);
    if (ns > h->ns_cap[l - 1]) { set_error("tps:%d = %d needs %d slabs (%d allocated)", l, value, ns, h->ns_cap[l - 1]); return SDQN_ERR_ARG; }
    { int rc_ = join_comm(h); if (rc_) return rc_; } HIPCHK(hipStreamSynchronize(g_stream));
    if (l == 1) { h->tps1 = value; h->ns1 = ns; } else if (l == 2) { h->tps2 = value; h->ns2 = ns; } else { h->tps3 = value; h->ns3 = ns; }
  }
  else if (!strncmp(name, "nw:", 3)) {                     // tuning: waves per tile of kernel id
    int id = atoi(name + 3);
    if (id < 0 || id >= 12 || !(value == 0 || value == 1 || value == 2 || value == 4 || value == 8 || value == 16 || (value == 9 && id == 2))) { set_error("bad nw override"); return SDQN_ERR_ARG; }
    h->nw_override[id] = value;
  }
  else { set_error("unknown option %s", name); return SDQN_ERR_ARG; }
  return SDQN_OK;
}

// test hook: raw read of an internal device buffer (internal layouts, see problems.h)
extern "C" int sdqn_net_debug_read(sdqn_net_t h, const char* name, float* out, int64_t n) {
  ARGCHK(h && name && out, "NULL argument");
  ARGCHK(!h->gen, "debug_read exposes the tuned path's internal buffers (84x84x4 float32 / float16 only)");
  const int B = h->B;
  struct { const char* n; float* p; int64_t len; } tab[] = {
    {"a1", h->a1, (int64_t)2 * B * PIX1 * K1}, {"a2", h->a2, (int64_t)2 * B * PIX2 * K2}, {"a3", h->a3, (int64_t)2 * B * PIX3 * K3},
    {"a4", h->a4, (int64_t)2 * B * NFC}, {"d4", h->d4, (int64_t)B * NFC}, {"d3p", h->d3p, (int64_t)B * PD3 * PD3 * K3},
    {"d2p", h->d2p, (int64_t)B * PD2 * PD2 * K2}, {"d1", h->d1, (int64_t)B * PIX1 * K1}, {"q", h->q, (int64_t)2 * B * h->A},
    {"dq", h->dq, (int64_t)B * h->A}, {"g", h->g, h->NP}, {"theta", h->theta, h->NP}, {"cost_terms", h->cost_terms, B}};
  for (auto& e : tab) if (!strcmp(e.n, name)) {
    ARGCHK(n <= e.len, "buffer %s holds %lld floats", name, (long long)e.len);
    { int rc_ = join_comm(h); if (rc_) return rc_; } HIPCHK(hipStreamSynchronize(g_stream));
    HIPCHK(hipMemcpy(out, e.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return SDQN_OK;
  }
  set_error("unknown buffer %s", name); return SDQN_ERR_ARG;
}

// ---- data parallel ---------------------------------------------------------------------------------------
extern "C" int sdqn_dp_unique_id(const char* rccl_path, char id[128]) {
  ARGCHK(id, "NULL id"); int rc = rccl_load(rccl_path); if (rc) return rc;
  NCCLCHK(g_rccl.GetUniqueId(id));
  return SDQN_OK;
}
extern "C" int sdqn_dp_init(sdqn_net_t h, const char* rccl_path, const char id[128], int rank, int nranks) {
  ARGCHK(h && id && nranks >= 1 && rank >= 0 && rank < nranks, "bad arguments");
  if (h->gen) { set_error("data parallel is implemented for the 84x84x4 float32 / float16 configurations"); return SDQN_ERR_STATE; }
  if (h->comm) { set_error("data parallel already initialised"); return SDQN_ERR_STATE; }
  int rc = rccl_load(rccl_path); if (rc) return rc;
  Id128 u; memcpy(u.b, id, 128);
  NCCLCHK(g_rccl.CommInitRank(&h->comm, nranks, u, rank));
  h->rank = rank; h->nranks = nranks;
  // second communicator (same ranks) for the overlapped fc4 all-reduce: two collectives may only be in flight at
  // once on different communicators.  Without ncclCommSplit the step falls back to one all-reduce on the library stream.
  h->comm2 = nullptr;
  const bool want2 = h->dp_overlap_req == 1 || h->dp_overlap_req == -2 || (h->dp_overlap_req == -1 && nranks >= 2);
  h->dp_overlap = h->dp_overlap_req == 1;                  // auto: inactive until probed + voted (sdqn_dp_probe / sdqn_dp_set_overlap)
  h->dp_probe_result = -1;
  if (g_rccl.CommSplit && want2) {
    if (g_rccl.CommSplit(h->comm, 0, rank, &h->comm2, nullptr) != 0) h->comm2 = nullptr;
  }
  if (!h->comm2) h->dp_overlap = false;
  if (h->comm2) {
    if (!h->ev_g4) HIPCHK(hipEventCreateWithFlags(&h->ev_g4, hipEventDisableTiming));
    if (!h->ev_w4) HIPCHK(hipEventCreateWithFlags(&h->ev_w4, hipEventDisableTiming));
  }
  // Replicas start identical BY CONSTRUCTION: rank 0's online net, target net and optimizer state are broadcast, so learners
  // created with different seeds (random_seed unset: main.py:89) still share one network and one target-net sync
  // (BASELINE configs[3]: "shared target-net sync" — afterwards every rank applies the same all-reduced gradient).
  if (g_rccl.Broadcast && h->dp_sync_replicas) {
    HIPCHK(hipStreamSynchronize(g_stream));
    NCCLCHK(g_rccl.Broadcast(h->theta, h->theta, (size_t)h->NP, 7 /* ncclFloat32 */, 0, h->comm, g_stream));
    if (h->theta_t != h->theta) NCCLCHK(g_rccl.Broadcast(h->theta_t, h->theta_t, (size_t)h->NP, 7, 0, h->comm, g_stream));
    NCCLCHK(g_rccl.Broadcast(h->state, h->state, (size_t)h->NP, 7, 0, h->comm, g_stream));
    if (h->state2) NCCLCHK(g_rccl.Broadcast(h->state2, h->state2, (size_t)h->NP, 7, 0, h->comm, g_stream));
    if (h->w1p[0]) {
      HIPCHK(launch_w1_planes(h->theta, h->w1p[0], g_stream));
      if (h->w1p[1] != h->w1p[0]) HIPCHK(launch_w1_planes(h->theta_t, h->w1p[1], g_stream));
    }
    if (h->cfg.datatype == 1) {
      HIPCHK(launch_refresh16(h->theta, h->wh[0], h->wht[0], g_stream));
      if (h->theta_t != h->theta) HIPCHK(launch_refresh16(h->theta_t, h->wh[1], h->wht[1], g_stream));
    }
    if (h->wpm) {
      HIPCHK(launch_refresh_planes(h->theta, h->wpm, h->wpt[0], g_stream));
      if (h->theta_t != h->theta) HIPCHK(launch_refresh_planes(h->theta_t, nullptr, h->wpt[1], g_stream));
    }
    HIPCHK(hipStreamSynchronize(g_stream));
    h->spec_pending = false;               // rank 0's parameters replaced ours: a forward enqueued before the sync is not "predict now"
  }
  return SDQN_OK;
}
// Start-up probe of the overlapped form: two rounds of exactly the collectives an overlapped step issues — the fc4
// range on the second communicator / communication stream, the conv + fc5 ranges as one group on the first communicator / library stream,
// concurrently — with BOUNDED waits (hipStreamQuery polling against a deadline: never a blocking wait on a stream that may never drain).
// *ok = 1: both streams drained in time on THIS rank.  The caller must agree over its control plane (every rank's ok AND-ed) and then call
// sdqn_dp_set_overlap on every rank with the same answer; ranks that disagree would deadlock in the first real step.  The gradient buffer
// used as payload is scratch between steps (every step rewrites it).  inject_timeout != 0 (tests): report a time-out after the real
// completion, so that the fallback path can be exercised on a healthy stack.
extern "C" int sdqn_dp_probe(sdqn_net_t h, int timeout_ms, int inject_timeout, int* ok) {
  ARGCHK(h && ok && timeout_ms > 0, "bad arguments");
  *ok = 0;
  if (h->gen || !h->comm) { set_error("sdqn_dp_probe needs a data-parallel network (sdqn_dp_init first)"); return SDQN_ERR_STATE; }
  if (!h->comm2) { h->dp_probe_result = 0; return SDQN_OK; }           // no second communicator: the serial form is the only one
  HIPCHK(hipStreamSynchronize(g_stream)); HIPCHK(hipStreamSynchronize(g_comm));
  for (int rep = 0; rep < 2; ++rep) {
    HIPCHK(hipEventRecord(h->ev_g4, g_stream));
    HIPCHK(hipStreamWaitEvent(g_comm, h->ev_g4, 0));
    NCCLCHK(g_rccl.AllReduce(h->g + OFF4, h->g + OFF4, (size_t)NW4, /*ncclFloat32*/ 7, 0, h->comm2, g_comm));
    HIPCHK(hipEventRecord(h->ev_w4, g_comm));
    if (g_rccl.GroupStart && g_rccl.GroupEnd) NCCLCHK(g_rccl.GroupStart());
    NCCLCHK(g_rccl.AllReduce(h->g, h->g, (size_t)OFF4, 7, 0, h->comm, g_stream));
    NCCLCHK(g_rccl.AllReduce(h->g + OFF5, h->g + OFF5, (size_t)(h->NP - OFF5), 7, 0, h->comm, g_stream));
    if (g_rccl.GroupStart && g_rccl.GroupEnd) NCCLCHK(g_rccl.GroupEnd());
    HIPCHK(hipStreamWaitEvent(g_stream, h->ev_w4, 0));
  }
  const auto t0 = std::chrono::steady_clock::now();
  bool done = false;
  while (!done) {
    const hipError_t a = hipStreamQuery(g_stream), b = hipStreamQuery(g_comm);
    if (a == hipSuccess && b == hipSuccess) { done = true; break; }
    if ((a != hipSuccess && a != hipErrorNotReady) || (b != hipSuccess && b != hipErrorNotReady)) break;
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms)) break;
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  (void)hipGetLastError();
  if (done) HIPCHK(hipMemsetAsync(h->g, 0, (size_t)h->NP * 4, g_stream));       // (the probe's sums are not a gradient)
  h->dp_probe_result = (done && !inject_timeout) ? 1 : 0;
  *ok = h->dp_probe_result;
  return SDQN_OK;
}
// The agreed answer of the vote: on = 1 activates the overlapped form (needs the second communicator), on = 0 tears the second
// communicator down — ncclCommAbort when this rank's probe never completed (a destroy would wait for the stuck collective), ncclCommDestroy
// otherwise — and every later step runs the serial form.
extern "C" int sdqn_dp_set_overlap(sdqn_net_t h, int on) {
  ARGCHK(h, "NULL handle");
  if (h->gen || !h->comm) { set_error("sdqn_dp_set_overlap needs a data-parallel network (sdqn_dp_init first)"); return SDQN_ERR_STATE; }
  if (on) {
    if (!h->comm2) { set_error("the overlapped form needs the second communicator (ncclCommSplit missing, or dp_overlap was 0 at sdqn_dp_init)"); return SDQN_ERR_STATE; }
    h->dp_overlap = true;
    return SDQN_OK;
  }
  h->dp_overlap = false;
  if (h->comm2) {
    const bool stuck = hipStreamQuery(g_comm) == hipErrorNotReady && h->dp_probe_result == 0;
    (void)hipGetLastError();
    if (stuck && g_rccl.CommAbort) { NCCLCHK(g_rccl.CommAbort(h->comm2)); }
    else { { int rc = join_comm(h); if (rc) return rc; } HIPCHK(hipStreamSynchronize(g_stream)); HIPCHK(hipStreamSynchronize(g_comm)); NCCLCHK(g_rccl.CommDestroy(h->comm2)); }
    h->comm2 = nullptr; h->w4_pending = false;
  }
  return SDQN_OK;
}
// which form runs: *form = 0 no communicator, 1 serial (one all-reduce on the library stream), 2 overlapped; *probe = -1 / 0 / 1;
// *second_comm = 1 while the second communicator exists (auto mode between sdqn_dp_init and the vote: present but inactive)
extern "C" int sdqn_dp_form(sdqn_net_t h, int* form, int* probe, int* second_comm) {
  ARGCHK(h, "NULL handle");
  if (form) *form = !h->comm ? 0 : ((h->comm2 && h->dp_overlap) ? 2 : 1);
  if (probe) *probe = h->dp_probe_result;
  if (second_comm) *second_comm = h->comm2 ? 1 : 0;
  return SDQN_OK;
}
// What RCCL itself reports about the communicator (not what the caller passed in): ranks it spans, this rank, its device.
// All -1 without a communicator.  bench.py gathers these so a multi-GPU record can show "RCCL saw N ranks".
extern "C" int sdqn_dp_info(sdqn_net_t h, int* comm_ranks, int* comm_rank, int* comm_device, int* bound_device) {
  ARGCHK(h, "NULL handle");
  int n = -1, r = -1, d = -1;
  if (h->comm) {
    if (g_rccl.CommCount) NCCLCHK(g_rccl.CommCount(h->comm, &n));
    if (g_rccl.CommUserRank) NCCLCHK(g_rccl.CommUserRank(h->comm, &r));
    if (g_rccl.CommCuDevice) NCCLCHK(g_rccl.CommCuDevice(h->comm, &d));
  }
  if (comm_ranks) *comm_ranks = n; if (comm_rank) *comm_rank = r; if (comm_device) *comm_device = d;
  if (bound_device) *bound_device = g_dev;
  return SDQN_OK;
}
extern "C" int sdqn_dp_shutdown(sdqn_net_t h) {
  ARGCHK(h, "NULL handle");
  { int rc = join_comm(h); if (rc) return rc; }
  if (h->comm) {
    HIPCHK(hipStreamSynchronize(g_stream)); HIPCHK(hipStreamSynchronize(g_comm));
    if (h->comm2) { NCCLCHK(g_rccl.CommDestroy(h->comm2)); h->comm2 = nullptr; }
    NCCLCHK(g_rccl.CommDestroy(h->comm)); h->comm = nullptr;
  }
  h->rank = 0; h->nranks = 1;
  return SDQN_OK;
}

#ifdef SDQN_TIMING
namespace sdqn { hipError_t set_timing_buffer(unsigned long long* p); }
// experiment-only build (make timing): run ONE kernel id of the step with phase stamps; returns [blocks][8] cycles
extern "C" int sdqn_debug_time_kernel(sdqn_net_t h, sdqn_replay_t r, const int64_t* idx_host, int kernel, unsigned long long* out, int max_blocks) {
  ARGCHK(h && r && idx_host && out, "NULL");
  unsigned long long* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, (size_t)max_blocks * 64));
  HIPCHK(hipMemset(d, 0, (size_t)max_blocks * 64));
  int slot; const int64_t* pinned; int rc = replay_push_idx(r, idx_host, &slot, &pinned); if (rc) return rc;
  PrepArgs p = prep_args(h, r, pinned); HIPCHK(launch_prep(p, g_stream));
  StepArgs a = step_args(h); a.from_ring = 1; a.src = r->d_ring; a.idx = h->d_idx;
  HIPCHK(hipStreamSynchronize(g_stream));
  HIPCHK(set_timing_buffer(d));
  // kernel ids >= 100: round-3 variants — 100 conv1 on bf16 MFMA (warm: third launch on the same indexes), 101 the same, ONE launch
  // (frames never touched before: HBM + TLB cold, what a train step sees), 102 conv3_fwd on 36-deep chunks
  for (int rep = 0; rep < (kernel == 101 ? 1 : 3); ++rep) {                         // last launch's stamps survive
    if (kernel == K_HEAD) { HeadArgs hd = head_args(h, 1); HIPCHK(launch_head(a, hd, g_stream)); }
    else if (kernel == 100 || kernel == 101) { h->host_idx_cur = idx_host; const hipError_t le = launch_tuned(h, K_CONV1_FWD, a, g_stream, 0, 4); h->host_idx_cur = nullptr; HIPCHK(le); }
    else if (kernel == 102) HIPCHK(launch_tuned(h, K_CONV3_FWD, a, g_stream, 0, 2));
#ifdef SDQN_EXPERIMENTS
    else if (kernel == 103) {              // the fused update + conv1 launch (the update applies whatever the slabs hold: timing only)
      UpdateArgs u = make_update_args(h, a); u.mode = 0; u.bsz = (float)h->B; u.skip_fc4 = 1; u.w1_ctr = h->w1_ctr;
      h->w1_epochs += 1;
      HIPCHK(launch_upd_conv1(u, a, idx_host, h->w1_ctr, 64u * h->w1_epochs, h->w1_ctr + 1, h->r3_xcd & 1, g_stream));
    }
#endif
    else if (kernel == 104) { UpdateArgs u = make_update_args(h, a); u.mode = 0; u.bsz = (float)h->B; u.skip_fc4 = 1; HIPCHK(launch_update(u, g_stream)); }
    else HIPCHK(launch_tuned(h, kernel, a, g_stream));
  }
  HIPCHK(hipStreamSynchronize(g_stream));
  HIPCHK(set_timing_buffer(nullptr));
  HIPCHK(hipMemcpy(out, d, (size_t)max_blocks * 64, hipMemcpyDeviceToHost));
  hipFree(d);
  return replay_release_idx(r, slot);
}
#endif
