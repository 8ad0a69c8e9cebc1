// sdqn_api_replay.hip — ReplayMemory handles: pinned ring + HBM mirror, add / sample / gather (replay_memory.py:6-79)
#include "api_internal.h"

std::vector<sdqn_replay_s*> g_replays;
// ---- replay -----------------------------------------------------------------------------------------

int replay_free(sdqn_replay_s* r) {
  if (!r) return SDQN_OK;
  for (size_t i = 0; i < g_replays.size(); ++i) if (g_replays[i] == r) { g_replays.erase(g_replays.begin() + i); break; }
  if (g_stream) hipStreamSynchronize(g_stream);
  if (!(r->flags & SDQN_REPLAY_ZERO_COPY)) { hipFree(r->d_ring); hipFree(r->d_meta); }
  hipFree(r->d_pre); hipFree(r->d_rew);                   // (d_post / d_act / d_term live inside these two blocks, likewise on the host)
  hipHostFree(r->screens); hipHostFree(r->actions); hipHostFree(r->rewards); hipHostFree(r->terminals);
  hipHostFree(r->h_meta); hipHostFree(r->h_pre); hipHostFree(r->h_rew); hipHostFree(r->h_idx);
  for (int i = 0; i < NSLOT; ++i) if (r->slot_ev[i]) hipEventDestroy(r->slot_ev[i]);
  if (r->mb_upload_ev) hipEventDestroy(r->mb_upload_ev);
  free(r->mb_snap);
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
  r->mb_snap = (uint8_t*)malloc((size_t)batch * 10);
  if (!r->mb_snap) { set_error("out of host memory (minibatch metadata snapshot)"); replay_free(r); return SDQN_ERR_HIP; }
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
int replay_flush_pending(sdqn_replay_s* r) {          // one event for every slot released since the last one
  if (r->npending == 0) return SDQN_OK;
  const int last = r->pending[r->npending - 1];
  HIPCHK(hipEventRecord(r->slot_ev[last], g_stream));
  for (int i = 0; i < r->npending; ++i) { r->slot_cover[r->pending[i]] = last; r->slot_busy[r->pending[i]] = true; }
  r->npending = 0;
  return SDQN_OK;
}
int replay_push_idx(sdqn_replay_s* r, const int64_t* idx, int* slot_out, const int64_t** dev) {
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
int replay_release_idx(sdqn_replay_s* r, int slot) {
  HIPCHK(hipEventRecord(r->slot_ev[slot], g_stream)); r->slot_busy[slot] = true; r->slot_cover[slot] = slot; return SDQN_OK;
}
// The train paths release their index slots in batches: an event record is a packet of its own in the dependent launch chain and
// costs ~2.8 us of GPU time — one per step took 3.9 % off the step rate (12 998 -> 13 500 steps/s, tools/exp/README.md) and made every
// call's first ~20 steps slow.  One record per SLOT_BATCH releases covers all the slots used
// since the previous one; a slot is reused NSLOT = 64 pushes after its use, so its covering event is recorded long before.
static const int SLOT_BATCH = 16;
int replay_release_idx_batched(sdqn_replay_s* r, int slot, bool flush) {
  r->pending[r->npending++] = slot;
  if (flush || r->npending >= SLOT_BATCH) return replay_flush_pending(r);
  return SDQN_OK;
}

GatherArgs gather_args(sdqn_replay_s* r, const int64_t* didx) {
  r->mb_dev_gen++;                                  // (every launch built from these arguments overwrites the device minibatch)
  GatherArgs g; g.ring = r->d_ring; g.meta = r->d_meta; g.idx = didx; g.pre = r->d_pre; g.post = r->d_post;
  g.actions = r->d_act; g.rewards = r->d_rew; g.terminals = r->d_term; g.B = r->B; return g;
}
int replay_gather_generic(sdqn_replay_s* r, const int64_t* didx) {      // any geometry (generic_net.hip)
  r->mb_dev_gen++;
  GatherGenericArgs g; g.ring = r->d_ring; g.meta = r->d_meta; g.idx = didx; g.pre = r->d_pre; g.post = r->d_post;
  g.actions = r->d_act; g.rewards = r->d_rew; g.terminals = r->d_term; g.B = r->B; g.hist = r->hist; g.frame = r->frame;
  HIPCHK(launch_gather_generic(g, g_stream));
  return SDQN_OK;
}
// what the gather being enqueued will leave in d_rew | d_act | d_term: the packed metadata the device mirror was uploaded FROM
// (h_meta == d_meta in stream order; sdqn_net_train_host compares a tuple's small arrays with this)
static void snapshot_small(sdqn_replay_s* r, const int64_t* idx) {
  const size_t B = (size_t)r->B;
  int64_t* rew = reinterpret_cast<int64_t*>(r->mb_snap); uint8_t* act = r->mb_snap + B * 8; uint8_t* term = act + B;
  for (size_t i = 0; i < B; ++i) { const MetaRec& m = r->h_meta[idx[i]]; rew[i] = m.reward; act[i] = m.action; term[i] = m.terminal; }
  r->mb_snap_gen = r->mb_dev_gen;
}
extern "C" int sdqn_replay_gather(sdqn_replay_t r, const int64_t* idx_host) {
  ARGCHK(r && idx_host, "NULL argument");
  for (int i = 0; i < r->B; ++i)
    ARGCHK(idx_host[i] >= r->hist && idx_host[i] < r->count, "index %lld out of range (count %lld)", (long long)idx_host[i], (long long)r->count);
  if (!r->tuned_geom) {
    int slot; const int64_t* didx; int rc = replay_push_idx(r, idx_host, &slot, &didx); if (rc) return rc;
    rc = replay_gather_generic(r, didx); if (rc) return rc;
    snapshot_small(r, idx_host); r->mb_gather_gen = r->mb_dev_gen;
    return replay_release_idx(r, slot);
  }
  if (r->B <= 256) {               // the indexes ride in the kernel arguments: no pinned slot, no release event (sdqn_kernels.hip)
    HIPCHK(launch_gather(gather_args(r, nullptr), g_stream, idx_host));
    snapshot_small(r, idx_host); r->mb_gather_gen = r->mb_dev_gen;
    return SDQN_OK;
  }
  int slot; const int64_t* didx; int rc = replay_push_idx(r, idx_host, &slot, &didx); if (rc) return rc;
  HIPCHK(launch_gather(gather_args(r, didx), g_stream));
  snapshot_small(r, idx_host); r->mb_gather_gen = r->mb_dev_gen;
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
  if (gen == UINT64_MAX) {
    // "the minibatch of the last sdqn_replay_gather, whose host copy I have NOT fetched": if anything has overwritten the device
    // minibatch since (the generic path's fused gather, a bench gather), those states exist nowhere any more — an error, never a
    // silent step on another gather's states or on host buffers that were never filled
    if (r->mb_gather_gen != r->mb_dev_gen && r->mb_host_gen != r->mb_gather_gen) {
      set_error("the device minibatch of the last sdqn_replay_gather (generation %llu) was overwritten (now %llu) before its states were fetched",
                (unsigned long long)r->mb_gather_gen, (unsigned long long)r->mb_dev_gen);
      return SDQN_ERR_STATE;
    }
    gen = r->mb_gather_gen;
  }
  r->mb_clean_declared = true; r->mb_clean_on_device = gen == 0 || gen == r->mb_dev_gen;    // (a stale generation: the host buffers are uploaded as always)
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
