// sdqn_api_dp.hip — data parallel: RCCL resolved at run time, communicators, start-up probe, overlap form
#include "api_internal.h"

Rccl g_rccl;
// ---- RCCL (resolved at run time from the library the process already uses) ------------------------------
int rccl_load(const char* path) {
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
