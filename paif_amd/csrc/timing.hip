// Measurement helpers for bench.py's live roofline (BASELINE.json metric): HIP events WITHOUT the system-scope fence
// (hipEventDisableSystemFence: "events that are only used to measure timing") recorded on the stream the kernels are launched on.  A
// torch.cuda.Event record performs a system-scope release between two kernels; around every launch of the dominant kernel that costs
// 2-3 % of a 3.5 ms step.  Not part of the product path: nothing in paif_amd/ calls these outside ops.KernelTimer.
#include "paif_common.h"

extern "C" int paif_timing_event_create(void** ev) {
  PAIF_REQUIRE(ev, PAIF_EINVAL, "timing_event_create: null pointer");
  hipEvent_t e;
  hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
  if (rc != hipSuccess) { paif::set_error("timing_event_create: %s", hipGetErrorString(rc)); return (int)rc; }
  *ev = reinterpret_cast<void*>(e);
  return 0;
}

extern "C" int paif_timing_event_record(void* ev, paif_stream_t stream) {
  PAIF_REQUIRE(ev, PAIF_EINVAL, "timing_event_record: null event");
  hipError_t rc = hipEventRecord(reinterpret_cast<hipEvent_t>(ev), paif::as_stream(stream));
  if (rc != hipSuccess) { paif::set_error("timing_event_record: %s", hipGetErrorString(rc)); return (int)rc; }
  return 0;
}

// milliseconds between two recorded events; both must have completed (call after a stream / device synchronize)
extern "C" int paif_timing_event_elapsed_ms(void* start, void* stop, float* ms) {
  PAIF_REQUIRE(start && stop && ms, PAIF_EINVAL, "timing_event_elapsed_ms: null pointer");
  hipError_t rc = hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop));
  if (rc != hipSuccess) { paif::set_error("timing_event_elapsed_ms: %s", hipGetErrorString(rc)); return (int)rc; }
  return 0;
}

extern "C" int paif_timing_event_destroy(void* ev) {
  if (!ev) return 0;
  hipError_t rc = hipEventDestroy(reinterpret_cast<hipEvent_t>(ev));
  if (rc != hipSuccess) { paif::set_error("timing_event_destroy: %s", hipGetErrorString(rc)); return (int)rc; }
  return 0;
}
