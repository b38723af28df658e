// HIP-event brackets that survive stream capture (section 8(d): the roofline figures are HIP-event durations of the
// dominant kernels measured inside the timed region, and the timed region replays HIP graphs).
// A plain hipEventRecord is refused while a stream captures (and hipEventRecordWithFlags(.., hipEventRecordExternal)
// is not implemented by this runtime), so during capture the record becomes an EVENT-RECORD NODE added behind the
// capture's current leaves, and the capture continues from that node: every replay of the graph then stamps the
// event at that point of the stream order, and hipEventElapsedTime between two such events is the duration of what
// was captured between them in the LAST replay.
#include "bmv_common.hpp"

namespace bmv {
static thread_local LaunchEvents g_launch_events;
LaunchEvents take_launch_events() {
  LaunchEvents e = g_launch_events;
  g_launch_events = LaunchEvents{};
  return e;
}
void set_launch_events(hipEvent_t start, hipEvent_t stop) { g_launch_events.start = start, g_launch_events.stop = stop; }
}  // namespace bmv

using namespace bmv;

extern "C" {

int bmv_bind_next_launch(bmv_event_t start, bmv_event_t stop) {
  BMV_REQUIRE((start == nullptr) == (stop == nullptr), "bmv_bind_next_launch: one event without the other");
  set_launch_events(reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop));
  return BMV_OK;
}

int bmv_launch_events_pending(void) {   // 1: the events of bmv_bind_next_launch were not taken by a launch (and are dropped)
  const LaunchEvents e = take_launch_events();
  return e.start ? 1 : 0;
}

int bmv_event_create(bmv_event_t* ev) {
  BMV_REQUIRE(ev, "bmv_event_create: null pointer");
  hipEvent_t e;
  hipError_t r = hipEventCreate(&e);
  if (r != hipSuccess) {
    set_error("bmv_event_create: %s", hipGetErrorString(r));
    return BMV_ERR_LAUNCH;
  }
  *ev = reinterpret_cast<bmv_event_t>(e);
  return BMV_OK;
}

int bmv_event_destroy(bmv_event_t ev) {
  if (!ev) return BMV_OK;
  hipError_t r = hipEventDestroy(reinterpret_cast<hipEvent_t>(ev));
  if (r != hipSuccess) {
    set_error("bmv_event_destroy: %s", hipGetErrorString(r));
    return BMV_ERR_LAUNCH;
  }
  return BMV_OK;
}

int bmv_event_record(bmv_event_t ev, bmv_stream_t stream) {
  BMV_REQUIRE(ev, "bmv_event_record: null event");
  hipEvent_t e = reinterpret_cast<hipEvent_t>(ev);
  hipStream_t s = as_stream(stream);
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  hipError_t r = hipStreamGetCaptureInfo_v2(s, &status, &id, &graph, &deps, &ndeps);
  if (r == hipSuccess && status == hipStreamCaptureStatusActive) {
    hipGraphNode_t node;
    r = hipGraphAddEventRecordNode(&node, graph, deps, ndeps, e);
    if (r == hipSuccess) r = hipStreamUpdateCaptureDependencies(s, &node, 1, hipStreamSetCaptureDependencies);
  } else if (r == hipSuccess) {
    r = hipEventRecord(e, s);
  }
  if (r != hipSuccess) {
    set_error("bmv_event_record: %s", hipGetErrorString(r));
    return BMV_ERR_LAUNCH;
  }
  return BMV_OK;
}

int bmv_event_elapsed_us(bmv_event_t start, bmv_event_t end, float* us) {
  BMV_REQUIRE(start && end && us, "bmv_event_elapsed_us: null pointer");
  float ms = 0.f;
  hipError_t r = hipEventElapsedTime(&ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(end));
  if (r != hipSuccess) {   // not recorded yet, or not complete: the caller synchronises first
    set_error("bmv_event_elapsed_us: %s", hipGetErrorString(r));
    return BMV_ERR_INVALID;
  }
  *us = ms * 1e3f;
  return BMV_OK;
}

}  // extern "C"
