// Context, memory plumbing, events: the non-compute part of the C ABI (include/cosnarks_hip.h).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "common.hpp"

namespace csh {

static thread_local std::string tl_error;
static thread_local int tl_device = -1;

struct ThreadStreams {
  std::map<int, hipStream_t> by_device;
  std::map<hipStream_t, Arena> arenas;
  ~ThreadStreams() {
    // Process/thread teardown: the HIP runtime may already be gone; leak deliberately.
  }
};
static thread_local ThreadStreams tl_streams;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  tl_error = buf;
}

int ensure_device() {
  if (tl_device >= 0) return CSH_OK;
  return csh_init(0);
}

hipStream_t resolve_stream(void* s) {
  if (s) return reinterpret_cast<hipStream_t>(s);
  auto it = tl_streams.by_device.find(tl_device);
  if (it != tl_streams.by_device.end()) return it->second;
  hipStream_t st = nullptr;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr;  // fall back to the null stream
  tl_streams.by_device[tl_device] = st;
  return st;
}

int Arena::reserve(size_t bytes) {
  off = 0;
  if (bytes <= cap) return CSH_OK;
  if (base) {
    CSH_HIP(hipFree(base));
    base = nullptr;
    cap = 0;
  }
  size_t want = bytes + (bytes >> 3) + (1 << 20);
  void* p = nullptr;
  CSH_HIP(hipMalloc(&p, want));
  base = static_cast<char*>(p);
  cap = want;
  return CSH_OK;
}

Arena& arena_for(hipStream_t s) { return tl_streams.arenas[s]; }

}  // namespace csh

using namespace csh;

extern "C" {

int csh_init(int device) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    set_error("no HIP device available (hipGetDeviceCount: %s, count=%d); this library has no CPU fallback",
              hipGetErrorString(e), count);
    return CSH_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= count) {
    set_error("device %d out of range (count %d)", device, count);
    return CSH_ERR_INVALID;
  }
  CSH_HIP(hipSetDevice(device));
  tl_device = device;
  return CSH_OK;
}

int csh_shutdown(void) {
  for (auto& kv : tl_streams.arenas) {
    if (kv.second.base) (void)hipFree(kv.second.base);
    kv.second = Arena();
  }
  tl_streams.arenas.clear();
  return CSH_OK;
}

const char* csh_last_error(void) { return tl_error.c_str(); }
const char* csh_version(void) { return "cosnarks-hip 0.1.0 (gfx950)"; }

int csh_device_count(int* count) {
  CSH_REQUIRE(count, "count is NULL");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return CSH_ERR_NO_DEVICE;
  }
  return CSH_OK;
}

int csh_malloc(void** dev_ptr, size_t bytes) {
  CSH_REQUIRE(dev_ptr, "dev_ptr is NULL");
  CSH_TRY(ensure_device());
  CSH_HIP(hipMalloc(dev_ptr, bytes ? bytes : 1));
  return CSH_OK;
}
int csh_free(void* dev_ptr) {
  if (!dev_ptr) return CSH_OK;
  CSH_HIP(hipFree(dev_ptr));
  return CSH_OK;
}
int csh_memcpy_h2d(void* dev_dst, const void* host_src, size_t bytes) {
  CSH_TRY(ensure_device());
  CSH_HIP(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
  return CSH_OK;
}
int csh_memcpy_d2h(void* host_dst, const void* dev_src, size_t bytes) {
  CSH_TRY(ensure_device());
  CSH_HIP(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
  return CSH_OK;
}
int csh_sync(void* stream) {
  CSH_TRY(ensure_device());
  CSH_HIP(hipStreamSynchronize(resolve_stream(stream)));
  return CSH_OK;
}

int csh_event_create(void** ev) {
  CSH_REQUIRE(ev, "ev is NULL");
  CSH_TRY(ensure_device());
  hipEvent_t e;
  CSH_HIP(hipEventCreate(&e));
  *ev = e;
  return CSH_OK;
}
int csh_event_record(void* ev, void* stream) {
  CSH_TRY(ensure_device());
  CSH_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev), resolve_stream(stream)));
  return CSH_OK;
}
int csh_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms) {
  CSH_REQUIRE(ms, "ms is NULL");
  CSH_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(ev_stop)));
  CSH_HIP(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(ev_start), reinterpret_cast<hipEvent_t>(ev_stop)));
  return CSH_OK;
}
int csh_event_destroy(void* ev) {
  CSH_HIP(hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)));
  return CSH_OK;
}

}  // extern "C"
