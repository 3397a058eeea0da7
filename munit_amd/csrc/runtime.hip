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
extern "C" int munit_version(void) { return 1; }

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
  // Inside a stream capture every fork / join edge gets an event of its own (re-recording one event object while the
  // graph under construction still refers to its previous record crashed hipStreamEndCapture on ROCm 7.2); these few
  // hundred events per captured step are kept until the process ends.
  hipEvent_t use = ev[dev];
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)signaler, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) {
    if (hipEventCreateWithFlags(&use, hipEventDisableTiming) != hipSuccess) {
      munit_set_error("stream_wait_stream: hipEventCreate failed (capture)");
      return MUNIT_ERR_LAUNCH;
    }
  } else {
    (void)hipGetLastError();
  }
  hipError_t e = hipEventRecord(use, (hipStream_t)signaler);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)waiter, use, 0);
  if (e != hipSuccess) {
    munit_set_error("stream_wait_stream: %s", hipGetErrorString(e));
    return MUNIT_ERR_LAUNCH;
  }
  return MUNIT_OK;
}

// a and b each wait for what the other has enqueued so far: both events are recorded first, then both waits are issued
// (a record that follows a wait on the other stream builds a chained edge inside a capture, which hipStreamEndCapture of
// ROCm 7.2 did not survive when neither stream had a kernel node yet).
extern "C" int munit_stream_cross_wait(munit_stream_t a, munit_stream_t b) {
  hipEvent_t ea, eb;
  if (hipEventCreateWithFlags(&ea, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&eb, hipEventDisableTiming) != hipSuccess) {
    munit_set_error("stream_cross_wait: hipEventCreate failed");
    return MUNIT_ERR_LAUNCH;
  }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing((hipStream_t)a, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
  if (!capturing) (void)hipGetLastError();
  hipError_t e = hipEventRecord(ea, (hipStream_t)a);
  if (e == hipSuccess) e = hipEventRecord(eb, (hipStream_t)b);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)a, eb, 0);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)b, ea, 0);
  if (!capturing) {   // outside a capture the events can go at once (destruction is deferred until they complete)
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
  }
  if (e != hipSuccess) {
    munit_set_error("stream_cross_wait: %s", hipGetErrorString(e));
    return MUNIT_ERR_LAUNCH;
  }
  return MUNIT_OK;
}
