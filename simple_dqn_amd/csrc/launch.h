// launch.h — the one way step kernels are launched.  Normally a plain launch; when the host orchestration has armed a
// pair of events (sdqn_net_profile), the launch hands them to hipExtLaunchKernel, which stores the DISPATCH PACKET's own
// begin / end timestamps in them — the same two timestamps `rocprofv3 --kernel-trace` reports for the launch, with no
// marker packets added to the queue (an hipEventRecord pair around a launch adds ~2.6 us of packet processing to what it
// measures and perturbs the dependent chain it sits in).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

namespace sdqn {
struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; bool used = false; };
LaunchEvents& launch_events();             // thread-local (sdqn_kernels.hip); armed by api_internal.h's LAUNCH_ON for ONE launch
}

#define SDQN_LAUNCH(kernel, grid, block, shmem, stream, ...) do { \
  sdqn::LaunchEvents& le__ = sdqn::launch_events(); \
  if (le__.start) { \
    hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, le__.start, le__.stop, 0, __VA_ARGS__); \
    le__.start = le__.stop = nullptr; le__.used = true; \
  } else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__); \
} while (0)
