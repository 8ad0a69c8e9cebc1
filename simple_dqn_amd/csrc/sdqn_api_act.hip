// sdqn_api_act.hip — acting path: device-resident StateBuffer, one-launch forward, speculative act step (agent.py:48-85, state_buffer.py)
#include "api_internal.h"

uint64_t next_statebuf_gen() { static uint64_t n = 0; return (++n << 40) | 1; }
// ---- device-resident StateBuffer --------------------------------------------------------------------------
// A ring of SB_SLOTS frame slots in HBM; frame t goes to slot `pos`, the current state is the contiguous window
// of the `hist` slots ending there.  When the ring is full the last hist-1 frames are copied back to the start
// (one 21 KB D2D every SB_SLOTS-hist+1 adds), so an add is ONE 7 KB H2D from a pinned staging slot.
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
const uint8_t* statebuf_window(sdqn_statebuf_s* s) { return s->d + (size_t)(s->pos - s->hist + 1) * s->frame; }
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
int predict_state_enqueue(sdqn_net_s* h, sdqn_statebuf_s* sb) {
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
bool act_trace() { static const bool on = getenv("SDQN_ACT_TRACE") != nullptr; return on; }
static double g_act_enq_ns = 0, g_act_wait_ns = 0; static long g_act_calls = 0;
int predict_state_collect(sdqn_net_s* h, float* q_out) {
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
