// Device-side scratch words of the launchers: tile counters of the dynamically scheduled persistent kernels and the
// hand-off blocks of the kernels whose last block finishes a reduction (common.h declares both).  The library has no
// allocator and its launchers neither allocate nor synchronise, so the words live in two static device arrays; this
// file decides who gets which word and FAILS — with a message, never by wrapping around — when it cannot keep two
// launches that may be in flight together apart.
//
//   eager launches     one slot per (device, stream): 64 tile counters + one hand-off block.  Launches on one stream
//                      run in order, so a stream's launch k+1 (its counter memset included) starts after launch k has
//                      left the words zero / done with them; launches on different streams never share a slot.  64 slots
//                      per device; when all are taken, a slot whose stream has drained or no longer exists
//                      (hipStreamQuery != hipErrorNotReady) is handed on; if every one of the 64 streams is busy, error.
//   recorded launches  (the stream is capturing into a hipGraph) take their words from a separate region, for good: a
//                      replayed graph meets only its own words, whatever stream it is replayed on.  Nothing is ever
//                      handed out twice; when the region is used up the launcher returns an error (re-capturing
//                      thousands of graphs in one process is the only way there).
#include <mutex>
#include "common.h"

namespace tmgcn {

constexpr int kMaxDevices = 16;
constexpr int kStreamSlots = 64;                       // eager slots per device
constexpr int kGroup = 64;                             // tile counters per eager slot (the widest launch asks for 64)
constexpr int kCapturedCounters = 32768;               // tile counters for recorded launches (128 KB)
constexpr int kCapturedSync = 4032;                    // hand-off blocks for recorded launches
__device__ unsigned int g_tile_counters[kStreamSlots * kGroup + kCapturedCounters];
__device__ int g_sync_words[(kStreamSlots + kCapturedSync) * kSyncInts];

static thread_local char g_pool_err[256] = "";
const char* pool_error() { return g_pool_err; }

namespace {
struct DevicePools {
  std::mutex mu;
  unsigned int* counters = nullptr;
  int32_t* sync = nullptr;
  hipStream_t stream[kStreamSlots];
  int n_streams = 0;
  int64_t captured_counters = 0, captured_sync = 0;
};
DevicePools g_pools[kMaxDevices];

DevicePools* pools_of_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    snprintf(g_pool_err, sizeof(g_pool_err), "device ordinal %d is outside the %d devices this library keeps scratch words for", dev, kMaxDevices);
    return nullptr;
  }
  DevicePools* p = &g_pools[dev];
  std::lock_guard<std::mutex> lk(p->mu);
  if (!p->counters) {
    void *c = nullptr, *s = nullptr;
    if (hipGetSymbolAddress(&c, HIP_SYMBOL(g_tile_counters)) != hipSuccess || hipGetSymbolAddress(&s, HIP_SYMBOL(g_sync_words)) != hipSuccess) {
      (void)hipGetLastError();
      snprintf(g_pool_err, sizeof(g_pool_err), "cannot resolve the scratch-word arrays on device %d", dev);
      return nullptr;
    }
    p->counters = static_cast<unsigned int*>(c);
    p->sync = static_cast<int32_t*>(s);
  }
  return p;
}

bool capturing(hipStream_t stream) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return cs == hipStreamCaptureStatusActive;
}

// the eager slot of `stream` (caller holds p->mu); -1 with g_pool_err set when 64 other streams are all busy
int eager_slot(DevicePools* p, hipStream_t stream) {
  for (int i = 0; i < p->n_streams; ++i)
    if (p->stream[i] == stream) return i;
  if (p->n_streams < kStreamSlots) {
    p->stream[p->n_streams] = stream;
    return p->n_streams++;
  }
  for (int i = 0; i < kStreamSlots; ++i) {
    if (capturing(p->stream[i])) continue;                 // (querying a capturing stream would invalidate its capture)
    const hipError_t q = hipStreamQuery(p->stream[i]);     // drained, or destroyed (an error other than NotReady): reusable
    if (q != hipSuccess) (void)hipGetLastError();
    if (q != hipErrorNotReady) {
      p->stream[i] = stream;
      return i;
    }
  }
  snprintf(g_pool_err, sizeof(g_pool_err), "launches are in flight on %d other streams of this device: no scratch-word slot left for one more", kStreamSlots);
  return -1;
}
}  // namespace

unsigned int* acquire_tile_counters(hipStream_t stream, int n) {
  if (n < 1 || n > kGroup) {
    snprintf(g_pool_err, sizeof(g_pool_err), "%d tile counters asked for one launch (1..%d)", n, kGroup);
    return nullptr;
  }
  DevicePools* p = pools_of_current_device();
  if (!p) return nullptr;
  unsigned int* c = nullptr;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    if (capturing(stream)) {
      if (p->captured_counters + n > kCapturedCounters) {
        snprintf(g_pool_err, sizeof(g_pool_err), "the %d tile counters kept for launches recorded into hipGraphs are used up "
                 "(recorded launches keep theirs for good); capture fewer graphs per process", kCapturedCounters);
        return nullptr;
      }
      c = p->counters + kStreamSlots * kGroup + p->captured_counters;
      p->captured_counters += n;
    } else {
      const int slot = eager_slot(p, stream);
      if (slot < 0) return nullptr;
      c = p->counters + slot * kGroup;
    }
  }
  if (hipMemsetAsync(c, 0, sizeof(unsigned int) * n, stream) != hipSuccess) {
    snprintf(g_pool_err, sizeof(g_pool_err), "hipMemsetAsync of the tile counters: %s", hipGetErrorString(hipGetLastError()));
    return nullptr;
  }
  return c;
}

int32_t* acquire_sync_word(hipStream_t stream) {
  DevicePools* p = pools_of_current_device();
  if (!p) return nullptr;
  std::lock_guard<std::mutex> lk(p->mu);
  if (capturing(stream)) {
    if (p->captured_sync >= kCapturedSync) {
      snprintf(g_pool_err, sizeof(g_pool_err), "the %d hand-off blocks kept for launches recorded into hipGraphs are used up "
               "(recorded launches keep theirs for good); capture fewer graphs per process, or pass the launcher its own `sync` block", kCapturedSync);
      return nullptr;
    }
    return p->sync + (int64_t)(kStreamSlots + p->captured_sync++) * kSyncInts;
  }
  const int slot = eager_slot(p, stream);
  return slot < 0 ? nullptr : p->sync + (int64_t)slot * kSyncInts;
}

}  // namespace tmgcn

using namespace tmgcn;

extern "C" int tmgcn_pool_stats(int64_t* out, int32_t n) {
  TMGCN_REQUIRE(out && n >= 6, "pool_stats: needs room for 6 values");
  DevicePools* p = pools_of_current_device();
  TMGCN_REQUIRE(p, "pool_stats: %s", pool_error());
  static int32_t host[(kStreamSlots + kCapturedSync) * kSyncInts];
  static std::mutex host_mu;
  std::lock_guard<std::mutex> hl(host_mu);
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host, p->sync, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) {
    set_error("pool_stats: %s", hipGetErrorString(hipGetLastError()));
    return TMGCN_ERR_LAUNCH;
  }
  int64_t nonzero = 0;
  for (size_t i = 0; i < sizeof(host) / sizeof(host[0]); ++i) nonzero += host[i] != 0;
  std::lock_guard<std::mutex> lk(p->mu);
  out[0] = p->n_streams;
  out[1] = p->captured_counters;
  out[2] = p->captured_sync;
  out[3] = nonzero;
  out[4] = kCapturedCounters;
  out[5] = kCapturedSync;
  return TMGCN_OK;
}
