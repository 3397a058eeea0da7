// Error reporting + version for libmunit_hip.so.
#include "common.h"
#include <cstring>

namespace {
thread_local char g_err[512] = "";
}

void munit_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* munit_last_error(void) { return g_err; }
// 2: munit_conv_desc carries in_dtype / out_dtype (a caller built against version 1 passes a short struct); the hipGraph
// entry points of version 1 (munit_adam_step_graph, munit_store_floats, munit_stream_cross_wait) are gone.
// 3: + munit_comm_unique_id / _init / _allreduce / _destroy, munit_shutdown (comm.hip); nothing removed or changed.
extern "C" int munit_version(void) { return 4; }

// waiter stream waits for everything enqueued so far on signaler (both on the current device): hipEventRecord +
// hipStreamWaitEvent on one cached event per (thread, device) -- the wait captures the event's state when it is issued, so
// the event can be re-recorded right away.  The Python side forks backward-weight onto its side stream ~200 times per step;
// through torch.cuda.Stream.wait_stream that costs an Event and a Stream object each time.
extern "C" int munit_stream_wait_stream(munit_stream_t waiter, munit_stream_t signaler) {
  constexpr int MAX_DEV = 16;
  static thread_local hipEvent_t ev[MAX_DEV] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) {
    munit_set_error("stream_wait_stream: no current device");
    return MUNIT_ERR_ARG;
  }
  if (ev[dev] == nullptr && hipEventCreateWithFlags(&ev[dev], hipEventDisableTiming) != hipSuccess) {
    munit_set_error("stream_wait_stream: hipEventCreate failed");
    return MUNIT_ERR_LAUNCH;
  }
  hipEvent_t use = ev[dev];
  hipError_t e = hipEventRecord(use, (hipStream_t)signaler);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)waiter, use, 0);
  if (e != hipSuccess) {
    munit_set_error("stream_wait_stream: %s", hipGetErrorString(e));
    return MUNIT_ERR_LAUNCH;
  }
  return MUNIT_OK;
}
