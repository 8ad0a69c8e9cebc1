// generic_net.h — the DQN train step for the configurations the tuned path does not cover: `--datatype float64`
// (src/main.py:53 -> deepqnetwork.py:33) and screens / history lengths other than 84 x 84 x 4 (src/main.py:27-28,34 ->
// deepqnetwork.py:21-22,28,83-91: the layer stack is the same, its sizes follow the input).  Same algorithm, same C ABI,
// same Neon layouts at the boundary; every layer is an explicit im2col + a tiled FMA GEMM in the network's own
// precision (generic_net.hip).  Selected by sdqn_net_create from the configuration alone; the 84 x 84 x 4 float32 /
// float16 configurations never come here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/sdqn.h"

namespace sdqn {

struct GenericNet {
  virtual ~GenericNet() {}
  virtual bool is_f64() const = 0;
  virtual int64_t layer_size(int layer) const = 0;                       // 0..4 = conv1, conv2, conv3, fc4, fc5 (Neon layouts)
  virtual int64_t param_count() const = 0;
  // which: 0 theta, 1 theta-, 2 optimizer state, 3 last gradient sum (get only), 4 second optimizer state; host data float or double
  virtual hipError_t set_param(int which, int layer, const void* host, bool host_f64) = 0;
  virtual hipError_t get_param(int which, int layer, void* host, bool host_f64) = 0;          // synchronises
  // Q(s; theta) of n <= batch_size states already on the device ([n][hist][H][W] u8); q_host [n][A], synchronises
  virtual hipError_t predict_dev(const uint8_t* states_dev, int n, void* q_host, bool host_f64) = 0;
  virtual hipError_t predict_host(const uint8_t* states_host, int n, void* q_host, bool host_f64) = 0;
  // one train step (deepqnetwork.py:107-172) on a device-resident minibatch; asynchronous on the stream
  virtual hipError_t train_dev(const uint8_t* pre, const uint8_t* post, const uint8_t* act, const int64_t* rew,
                               const uint8_t* term, int epoch) = 0;
  virtual hipError_t train_host(const uint8_t* pre, const uint8_t* act, const int64_t* rew, const uint8_t* post,
                                const uint8_t* term, int epoch) = 0;
  // device-resident states (a ReplayMemory's gathered minibatch) + host metadata: only the 10 x B bytes of (r, a, t) are uploaded
  virtual hipError_t train_dev_host_meta(const uint8_t* pre_dev, const uint8_t* post_dev, const uint8_t* act, const int64_t* rew,
                                         const uint8_t* term, int epoch) = 0;
  virtual hipError_t read_cost(double* cost) = 0;                        // cost of the last step (synchronises)
  virtual hipError_t reset_cost_sum() = 0;                               // running sum over steps (train_many's mean)
  virtual hipError_t read_cost_sum(double* sum) = 0;
  virtual hipError_t last_q(void* preq, void* maxpostq, bool host_f64) = 0;
  virtual hipError_t update_target() = 0;                                // deepqnetwork.py:102-105
  virtual size_t state_bytes() const = 0;                                // hist * H * W
};

// nullptr + *err on failure (bad geometry, out of memory)
GenericNet* make_generic_net(const sdqn_net_cfg& c, hipStream_t stream, std::string* err);

// the standalone replay gather for any geometry (replay_memory.py:71-78): pre[k] = frames idx-hist .. idx-1, post[k] = idx-hist+1 .. idx
struct GatherGenericArgs {
  const uint8_t* ring; const void* meta /*MetaRec[]*/; const int64_t* idx /*device*/;
  uint8_t *pre, *post, *actions; int64_t* rewards; uint8_t* terminals;
  int B, hist; int64_t frame;
};
hipError_t launch_gather_generic(const GatherGenericArgs& g, hipStream_t s);

}  // namespace sdqn
