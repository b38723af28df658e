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

namespace {
struct DeferredEntry {
  const void* ptr;
  DeferredPtr d;
  bool taken;
};
constexpr int kMaxDeferred = 8;
thread_local DeferredEntry g_deferred[kMaxDeferred];
thread_local int g_ndeferred = 0;
}  // namespace

DeferredPtr deferred_for(const void* ptr) {
  for (int i = 0; i < g_ndeferred; ++i)
    if (g_deferred[i].ptr == ptr) {
      g_deferred[i].taken = true;
      return g_deferred[i].d;
    }
  return DeferredPtr{};
}
int deferred_finish() {
  int left = 0;
  for (int i = 0; i < g_ndeferred; ++i) left += g_deferred[i].taken ? 0 : 1;
  g_ndeferred = 0;
  return left;
}

struct PtrTableSet {
  int n;
  int slot[16];
  const void* value[16];
};
__global__ void ptr_table_set_kernel(const void** table, PtrTableSet s) {
  const int i = threadIdx.x;
  if (i < s.n) table[s.slot[i]] = s.value[i];
}
// one launch in front of a replayed frame: the table entries of this frame AND the frame's small inputs (cameras,
// near / far: a few dozen floats each) copied into the captured buffers -- under run.py's per-frame synchronize every
// launch in front of the graph costs ~10-15 us of latency whatever it moves
struct FrameFeed {
  int n_ptr, n_copy;
  int slot[16];
  const void* value[16];
  const float* src[8];
  float* dst[8];
  int count[8];
};
__global__ void __launch_bounds__(256) frame_feed_kernel(const void** table, FrameFeed f) {
  const int t = threadIdx.x;
  if (t < f.n_ptr) table[f.slot[t]] = f.value[t];
  for (int c = 0; c < f.n_copy; ++c)
    for (int j = t; j < f.count[c]; j += 256) f.dst[c][j] = f.src[c][j];
}
// The same work as the FIRST NODE of the frame's graph, its arguments read from a ring of messages in pinned HOST memory:
// the host writes message number n (plain stores) and replays the graph -- no launch in front of the replay at all.
// `state[0]` counts the replays on the device (the host counts its posts; up to R frames may be in flight), a message
// whose sequence number is not the replay's number raises state[1] (bmv_frame_feed_ring_faults).
struct FrameFeedMsg {
  unsigned seq;
  int n_ptr, n_copy, pad;
  int slot[16];
  const void* value[16];
  const float* src[8];
  float* dst[8];
  int count[8];
};
static_assert(sizeof(FrameFeedMsg) == 368, "layout shared with boostmvsnerfs_amd/ops.py FeedRing");
__global__ void __launch_bounds__(256) frame_feed_ring_kernel(const void** table, const FrameFeedMsg* ring, unsigned* state,
                                                              int R) {
  __shared__ FrameFeedMsg m;
  const int t = threadIdx.x;
  const unsigned n = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const volatile unsigned* g = reinterpret_cast<const volatile unsigned*>(ring + (n % (unsigned)R));
  if (t < (int)(sizeof(FrameFeedMsg) / 4)) reinterpret_cast<unsigned*>(&m)[t] = g[t];
  __syncthreads();
  if (t == 0 && m.seq != n) atomicAdd(&state[1], 1u);
  if (t < m.n_ptr && t < 16) table[m.slot[t]] = m.value[t];
  const int nc = m.n_copy < 8 ? m.n_copy : 8;
  for (int c = 0; c < nc; ++c)
    for (int j = t; j < m.count[c]; j += 256) m.dst[c][j] = m.src[c][j];
  __syncthreads();
  if (t == 0) __hip_atomic_store(&state[0], n + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// src (captured buffer) -> the tensor table[slot] points at when the kernel runs: the frame's small outputs, as a node of
// the frame's own graph
__global__ void __launch_bounds__(256) copy_to_slot_kernel(const float* __restrict__ src, const void* const* table, int slot,
                                                           size_t n) {
  float* dst = static_cast<float*>(const_cast<void*>(table[slot]));
  if (dst == src) return;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// ... several of them in one launch (a frame's small outputs: one node at the end of the graph instead of one each)
struct SlotCopies {
  int n;
  const float* src[8];
  int slot[8];
  unsigned count[8];
};
__global__ void __launch_bounds__(256) copy_to_slots_kernel(const void* const* table, SlotCopies c) {
  for (int k = 0; k < c.n; ++k) {
    float* dst = static_cast<float*>(const_cast<void*>(table[c.slot[k]]));
    const float* src = c.src[k];
    if (dst == src) continue;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < c.count[k]; i += gridDim.x * 256) dst[i] = src[i];
  }
}
}  // namespace bmv

using namespace bmv;

extern "C" {

int bmv_bind_next_launch(bmv_event_t start, bmv_event_t stop) {
  BMV_REQUIRE((start == nullptr) == (stop == nullptr), "bmv_bind_next_launch: one event without the other");
  set_launch_events(reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop));
  return BMV_OK;
}

int bmv_defer_pointer(const void* ptr, const void* const* table, int slot) {
  BMV_REQUIRE(ptr && table && slot >= 0, "bmv_defer_pointer: null pointer / negative slot");
  BMV_REQUIRE(g_ndeferred < kMaxDeferred, "bmv_defer_pointer: more than %d deferrals pending", kMaxDeferred);
  g_deferred[g_ndeferred++] = DeferredEntry{ptr, DeferredPtr{table, slot}, false};
  return BMV_OK;
}

int bmv_deferred_pending(void) { return deferred_finish(); }

int bmv_ptr_table_set(void* table, int n, const int* slots, const void* const* values, bmv_stream_t stream) {
  BMV_REQUIRE(table && slots && values, "bmv_ptr_table_set: null pointer");
  BMV_REQUIRE(n > 0 && n <= 16, "bmv_ptr_table_set: 1..16 entries per call (n=%d)", n);
  PtrTableSet s;
  s.n = n;
  for (int i = 0; i < n; ++i) {
    BMV_REQUIRE(slots[i] >= 0, "bmv_ptr_table_set: negative slot");
    s.slot[i] = slots[i], s.value[i] = values[i];
  }
  hipLaunchKernelGGL(ptr_table_set_kernel, dim3(1), dim3(64), 0, as_stream(stream), static_cast<const void**>(table), s);
  BMV_LAUNCH_END("bmv_ptr_table_set");
}

int bmv_frame_feed(void* table, int n_ptr, const int* slots, const void* const* values, int n_copy,
                   const float* const* src, float* const* dst, const int* counts, bmv_stream_t stream) {
  BMV_REQUIRE(n_ptr >= 0 && n_ptr <= 16 && n_copy >= 0 && n_copy <= 8 && n_ptr + n_copy > 0,
              "bmv_frame_feed: up to 16 table entries and 8 small copies (n_ptr=%d, n_copy=%d)", n_ptr, n_copy);
  BMV_REQUIRE((n_ptr == 0 || (table && slots && values)) && (n_copy == 0 || (src && dst && counts)),
              "bmv_frame_feed: null pointer");
  FrameFeed f;
  f.n_ptr = n_ptr, f.n_copy = n_copy;
  for (int i = 0; i < n_ptr; ++i) {
    BMV_REQUIRE(slots[i] >= 0, "bmv_frame_feed: negative slot");
    f.slot[i] = slots[i], f.value[i] = values[i];
  }
  for (int i = 0; i < n_copy; ++i) {
    BMV_REQUIRE(src[i] && dst[i] && counts[i] >= 0 && counts[i] <= 65536,
                "bmv_frame_feed: small copies only (<= 65536 floats each; entry %d has %d)", i, counts[i]);
    f.src[i] = src[i], f.dst[i] = dst[i], f.count[i] = counts[i];
  }
  hipLaunchKernelGGL(frame_feed_kernel, dim3(1), dim3(256), 0, as_stream(stream), static_cast<const void**>(table), f);
  BMV_LAUNCH_END("bmv_frame_feed");
}

int bmv_frame_feed_ring(void* table, const void* ring, unsigned* state, int R, bmv_stream_t stream) {
  BMV_REQUIRE(table && ring && state && R > 0, "bmv_frame_feed_ring: bad arguments");
  hipLaunchKernelGGL(frame_feed_ring_kernel, dim3(1), dim3(256), 0, as_stream(stream), static_cast<const void**>(table),
                     static_cast<const FrameFeedMsg*>(ring), state, R);
  BMV_LAUNCH_END("bmv_frame_feed_ring");
}
int bmv_frame_feed_msg_bytes(void) { return (int)sizeof(FrameFeedMsg); }

int bmv_copy_to_slot(const float* src, const void* const* table, int slot, long n, bmv_stream_t stream) {
  BMV_REQUIRE(src && table && slot >= 0 && n > 0, "bmv_copy_to_slot: bad arguments");
  const unsigned grid = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(copy_to_slot_kernel, dim3(grid), dim3(256), 0, as_stream(stream), src, table, slot, (size_t)n);
  BMV_LAUNCH_END("bmv_copy_to_slot");
}

int bmv_copy_to_slots(int n, const float* const* src, const void* const* table, const int* slots, const long* counts,
                      bmv_stream_t stream) {
  BMV_REQUIRE(n > 0 && n <= 8 && src && table && slots && counts, "bmv_copy_to_slots: 1..8 copies (n=%d)", n);
  SlotCopies c;
  c.n = n;
  long most = 0;
  for (int i = 0; i < n; ++i) {
    BMV_REQUIRE(src[i] && slots[i] >= 0 && counts[i] > 0 && counts[i] < (1l << 31), "bmv_copy_to_slots: bad entry %d", i);
    c.src[i] = src[i], c.slot[i] = slots[i], c.count[i] = (unsigned)counts[i];
    most = counts[i] > most ? counts[i] : most;
  }
  const unsigned grid = (unsigned)((most + 255) / 256 < 2048 ? (most + 255) / 256 : 2048);
  hipLaunchKernelGGL(copy_to_slots_kernel, dim3(grid), dim3(256), 0, as_stream(stream), table, c);
  BMV_LAUNCH_END("bmv_copy_to_slots");
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
