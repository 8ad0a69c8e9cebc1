// sdqn_api_core.hip — library-wide state (streams, device binding, last error) and the host-only sampler entry points of the C ABI.
// No CPU fallback lives in any of these files: every device entry point needs a HIP device and fails loudly without one.
#include "api_internal.h"

thread_local std::string g_err;
void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_err = buf;
}
hipStream_t g_stream = nullptr, g_side = nullptr, g_comm = nullptr;
hipEvent_t g_ev[5];
int g_dev = -1;
// ------------------------------------------------------------------------------------------------

int ensure_stream() {
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
int sample_checked(uint32_t* mt, const uint8_t* terminals, int64_t count, int64_t current, int hist,
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
