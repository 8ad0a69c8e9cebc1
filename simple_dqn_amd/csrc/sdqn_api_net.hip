// sdqn_api_net.hip — DeepQNetwork handles: create / destroy, weights, profiler, options, small read-backs (deepqnetwork.py:16-105,188-192)
#include "api_internal.h"

// One step structure per (B regime, datatype, batch-norm, data-parallel form): DESIGN.md 12.  The step structures that were built, tested
// bit-identical and measured SLOWER in rounds 1-4 (hoist, f4w_early, fuse_upd, head_f4d, two_streams, fwd_rb, bwd_order, rb:<id>, bt_x,
// bt_planes, the ping-pong / stream-K / direct-to-LDS block-tile routines, the XCC-local chain probe) left the product tree in round 5:
// tools/exp/experiments_r04.patch re-creates them on top of the commit named there, tools/exp/README.md holds their measurements.
static const char* const RETIRED_OPTIONS[] = {"hoist", "two_streams", "fuse_dbg", "fwd_rb", "head_f4d", "fuse_upd", "f4w_early", "bt_planes", "bt_x", "bwd_order", nullptr};
#define RETIRED_OPTION_REFUSED(NAME) do { set_error("option %s belonged to an experiment that was measured slower than the default step and has been removed from " \
                                                    "the library (tools/exp/README.md; tools/exp/experiments_r04.patch)", NAME); return SDQN_ERR_ARG; } while (0)
// ---- network -------------------------------------------------------------------------------------------

int dalloc(sdqn_net_s* h, void** p, size_t bytes, bool zero) {
  HIPCHK(hipMalloc(p, bytes)); h->allocs.push_back(*p);
  if (zero) HIPCHK(hipMemsetAsync(*p, 0, bytes, g_stream));
  return SDQN_OK;
}
int net_free(sdqn_net_s* h) {
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
  if (c->datatype == 0) {                  // (all-zero planes == all-zero W1, which is what the zeroed theta holds until set_weights)
    NCHK(dalloc(h, (void**)&h->w1p[0], (size_t)3 * W1P_PLANE * 2));
    if (c->target_enabled) NCHK(dalloc(h, (void**)&h->w1p[1], (size_t)3 * W1P_PLANE * 2)); else h->w1p[1] = h->w1p[0];
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

float* which_buf(sdqn_net_s* h, int which) {
  switch (which) { case 0: return h->theta; case 1: return h->theta_t; case 2: return h->state; case 3: return h->g;
                   case 4: return h->state2; default: return nullptr; }
}
// BatchNorm pseudo-layers 5..8 (batch_norm only): [beta | gamma] at NPW + bn_off(l); which 5 / 6 = running statistics
bool bn_layer_span(sdqn_net_s* h, int which, int layer, float** base, int64_t* n) {
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
int gen_set(sdqn_net_s* h, int which, int layer, const void* w, int64_t n, bool f64) {
  ARGCHK(layer >= 0 && layer < 5 && which >= 0 && which <= 4 && which != 3, "bad arguments (which %d, layer %d)", which, layer);
  ARGCHK(which != 4 || h->cfg.optimizer != 0, "this optimizer has no second state");
  ARGCHK(n == h->gen->layer_size(layer), "layer %d holds %lld values, got %lld", layer, (long long)h->gen->layer_size(layer), (long long)n);
  GENCHK(h->gen->set_param(which, layer, w, f64));
  return SDQN_OK;
}
int gen_get(sdqn_net_s* h, int which, int layer, void* w, int64_t n, bool f64) {
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
int prof_collect(sdqn_net_s* h) {
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
int prof_event(sdqn_net_s* h, hipEvent_t* e) {
  if (!h->prof_free.empty()) { *e = h->prof_free.back(); h->prof_free.pop_back(); return SDQN_OK; }
  HIPCHK(hipEventCreate(e)); return SDQN_OK;
}
// profile_mode 1 (default): the launch itself records its dispatch packet's begin / end timestamps into the pair (launch.h:
// what rocprofv3 --kernel-trace reports, nothing added to the queue); 0, and always for launches that are not ONE kernel
// (RCCL, BatchNorm's two passes): hipEventRecord markers around the launch (adds ~2.6 us of packet processing to the figure)

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
extern "C" int sdqn_net_update_target(sdqn_net_t h) {
  ARGCHK(h, "NULL handle");
  if (h->gen) { GENCHK(h->gen->update_target()); return SDQN_OK; }
  { int rc = join_comm(h); if (rc) return rc; }
  if (h->theta_t != h->theta) {
    HIPCHK(hipMemcpyAsync(h->theta_t, h->theta, (size_t)h->NP * 4, hipMemcpyDeviceToDevice, g_stream));   // deepqnetwork.py:102-105
    if (h->w1p[0] && h->w1p[1] != h->w1p[0])
      HIPCHK(hipMemcpyAsync(h->w1p[1], h->w1p[0], (size_t)3 * W1P_PLANE * 2, hipMemcpyDeviceToDevice, g_stream));
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
  bool retired = !strncmp(name, "rb:", 3) || !strncmp(name, "btx:", 4);            // (switching one of them OFF stays a no-op, as before)
  for (int i = 0; RETIRED_OPTIONS[i]; ++i) retired = retired || !strcmp(name, RETIRED_OPTIONS[i]);
  if (retired) { if (value) RETIRED_OPTION_REFUSED(name); return SDQN_OK; }
  if (!strcmp(name, "keep_gradients")) h->keep_grads = value != 0;
  else if (!strcmp(name, "grad_only")) h->grad_only = value != 0;
  else if (!strcmp(name, "h16_wgrad_mfma")) h->h16_wgrad_mfma = value != 0;
  else if (!strcmp(name, "c1w_in_wgrads")) { if (value < 0 || value > 2) { set_error("bad c1w_in_wgrads (0..2)"); return SDQN_ERR_ARG; } h->c1w_in_wgrads = value; }
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
  else if (!strcmp(name, "fused_launches")) h->fused_launches = value != 0;
  else if (!strcmp(name, "conv1_bf16")) h->conv1_bf16 = value != 0;     // 0: conv1_fwd on the fp32-MFMA engine (round-2 kernel)
  else if (!strcmp(name, "wt")) h->wt = value;
  else if (!strcmp(name, "prep_inline")) h->prep_inline = value != 0;
  else if (!strcmp(name, "r3_xcd")) h->r3_xcd = value;
  else if (!strcmp(name, "conv1w_bf16")) h->conv1w_bf16 = value;   // 0: conv1_wgrad on the fp32-MFMA engine (round-2 kernel)
  else if (!strcmp(name, "conv3_c36")) h->conv3_c36 = value != 0;       // 0: conv3_fwd on the engine's 32-deep chunks (round-2 kernel)
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
  else if (!strncmp(name, "bt:", 3)) {                     // block-tile engine: menu entry of kernel id (0 built-in, -1 latency engine)
    int id = atoi(name + 3);
    if (id < 0 || id >= K_COUNT || value < -1 || value > (id == K_WGRADS ? 2 : 8)) { set_error("bad bt override"); return SDQN_ERR_ARG; }
    h->bt[id] = value;
  }
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
    else if (kernel == 100 || kernel == 101) { h->host_idx_cur = idx_host; const hipError_t le = launch_tuned(h, K_CONV1_FWD, a, g_stream, 4); h->host_idx_cur = nullptr; HIPCHK(le); }
    else if (kernel == 102) HIPCHK(launch_tuned(h, K_CONV3_FWD, a, g_stream, 2));
    else if (kernel == 104) { UpdateArgs u = make_update_args(h, a); u.mode = 0; u.bsz = (float)h->B; u.skip_fc4 = 1; HIPCHK(launch_update(u, g_stream)); }
    else HIPCHK(launch_tuned(h, kernel, a, g_stream));
  }
  HIPCHK(hipStreamSynchronize(g_stream));
  HIPCHK(set_timing_buffer(nullptr));
  HIPCHK(hipMemcpy(out, d, (size_t)max_blocks * 64, hipMemcpyDeviceToHost));
  hipFree(d);
  return replay_release_idx(r, slot);
}
// round 5: ONE real train step (ring path) with per-wave stamps around the launches of kernel id `kernel` (latency engine's tile routine):
// out_waves [max_blocks][16 waves][4 phases] cycles + out_xcc [max_blocks] (gemm_engine.h: SDQN_WSTAMP).  `steps_before` ordinary steps run
// first, so the stamped launch sits in a warmed-up dependent chain.
extern "C" int sdqn_debug_time_step_waves(sdqn_net_t h, sdqn_replay_t r, uint32_t* mt, int kernel, int steps_before, unsigned long long* out_waves,
                                          unsigned long long* out_xcc, int max_blocks) {
  ARGCHK(h && r && mt && out_waves && out_xcc && max_blocks > 0, "bad arguments");
  const size_t n = (size_t)max_blocks * 64 + max_blocks;
  unsigned long long* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, n * 8));
  HIPCHK(hipMemset(d, 0, n * 8));
  if (!h->wt_words) HIPCHK(hipHostMalloc((void**)&h->wt_words, 32, hipHostMallocDefault));
  h->wt_words[0] = (unsigned long long)(uintptr_t)d; h->wt_words[1] = 0; h->wt_words[2] = (unsigned long long)(unsigned)max_blocks;
  int rc = sdqn_net_train_many(h, r, mt, steps_before, nullptr); if (rc) { hipFree(d); return rc; }
  h->wt_kid = kernel;
  rc = sdqn_net_train_many(h, r, mt, 1, nullptr);
  h->wt_kid = -1;
  if (rc) { hipFree(d); return rc; }
  HIPCHK(hipStreamSynchronize(g_stream));
  std::vector<unsigned long long> tmp(n);
  HIPCHK(hipMemcpy(tmp.data(), d, n * 8, hipMemcpyDeviceToHost));
  memcpy(out_waves, tmp.data(), (size_t)max_blocks * 64 * 8);
  memcpy(out_xcc, tmp.data() + (size_t)max_blocks * 64, (size_t)max_blocks * 8);
  hipFree(d);
  return SDQN_OK;
}
#endif
