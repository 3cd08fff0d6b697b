// Host-pointer entry points: what does it cost to move a caller's pageable buffer, and which staging policy wins?
//   hipcc -O2 -o /tmp/pcie_probe tools/experiments/pcie_probe.cpp -lpthread && /tmp/pcie_probe
// Policies: (A) hipMemcpy from / to pageable memory (what round 3 did), (B) hipHostRegister for the duration of the call,
// (C) a persistent pinned bounce buffer, chunked: CPU memcpy of chunk k+1 overlaps the DMA of chunk k (1 / 2 / 4 copy threads),
// (D) D2H into a freshly allocated vector (page faults included: what `vec![0; n]` + download costs).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F>
static double best_of(int reps, F f) {
  double best = 1e30;
  for (int i = 0; i < reps; ++i) {
    const double t0 = now_ms();
    f();
    const double t = now_ms() - t0;
    if (t < best) best = t;
  }
  return best;
}

static void par_memcpy(char* dst, const char* src, size_t bytes, int threads) {
  if (threads <= 1) {
    memcpy(dst, src, bytes);
    return;
  }
  std::vector<std::thread> th;
  const size_t per = (bytes / threads + 63) & ~size_t(63);
  for (int t = 0; t < threads; ++t) {
    const size_t lo = per * t, hi = lo + per < bytes ? lo + per : bytes;
    if (lo < hi) th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
  }
  for (auto& x : th) x.join();
}

int main() {
  CK(hipSetDevice(0));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const size_t CH = 4u << 20;
  char* bounce[2];
  hipEvent_t ev[2];
  for (int i = 0; i < 2; ++i) {
    CK(hipHostMalloc((void**)&bounce[i], CH, hipHostMallocDefault));
    CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
  }
  for (size_t mb : {1, 8, 32, 64, 128}) {
    const size_t bytes = mb << 20;
    char* dev;
    CK(hipMalloc((void**)&dev, bytes));
    char* host = (char*)aligned_alloc(4096, bytes);
    memset(host, 1, bytes);
    char* pinned;
    CK(hipHostMalloc((void**)&pinned, bytes, hipHostMallocDefault));
    memset(pinned, 2, bytes);
    const double pin_h2d = best_of(5, [&] { CK(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); });
    const double pin_d2h = best_of(5, [&] { CK(hipMemcpyAsync(pinned, dev, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); });
    const double a_h2d = best_of(5, [&] { CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); });
    const double a_d2h = best_of(5, [&] { CK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); });
    double reg = 0, unreg = 0, b_h2d = 0, b_d2h = 0;
    const double b_total = best_of(5, [&] {
      double t0 = now_ms();
      CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
      double t1 = now_ms();
      CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st));
      CK(hipStreamSynchronize(st));
      double t2 = now_ms();
      CK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      double t3 = now_ms();
      CK(hipHostUnregister(host));
      double t4 = now_ms();
      reg = t1 - t0, b_h2d = t2 - t1, b_d2h = t3 - t2, unreg = t4 - t3;
    });
    (void)b_total;
    double c_h2d[3], c_d2h[3];
    int ti = 0;
    for (int threads : {1, 2, 4}) {
      c_h2d[ti] = best_of(5, [&] {
        size_t off = 0;
        int k = 0;
        while (off < bytes) {
          const size_t len = bytes - off < CH ? bytes - off : CH;
          CK(hipEventSynchronize(ev[k & 1]));
          par_memcpy(bounce[k & 1], host + off, len, threads);
          CK(hipMemcpyAsync(dev + off, bounce[k & 1], len, hipMemcpyHostToDevice, st));
          CK(hipEventRecord(ev[k & 1], st));
          off += len;
          ++k;
        }
        CK(hipStreamSynchronize(st));
      });
      c_d2h[ti] = best_of(5, [&] {
        size_t off = 0, done = 0;
        int k = 0, kd = 0;
        while (done < bytes) {
          while (off < bytes && k - kd < 2) {
            const size_t len = bytes - off < CH ? bytes - off : CH;
            CK(hipMemcpyAsync(bounce[k & 1], dev + off, len, hipMemcpyDeviceToHost, st));
            CK(hipEventRecord(ev[k & 1], st));
            off += len;
            ++k;
          }
          const size_t len = bytes - done < CH ? bytes - done : CH;
          CK(hipEventSynchronize(ev[kd & 1]));
          par_memcpy(host + done, bounce[kd & 1], len, threads);
          done += len;
          ++kd;
        }
      });
      ++ti;
    }
    // (D) fresh vector + download
    const double d_page = best_of(3, [&] {
      char* f = (char*)malloc(bytes);
      CK(hipMemcpyAsync(f, dev, bytes, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      free(f);
    });
    const double d_reg = best_of(3, [&] {
      char* f = (char*)malloc(bytes);
      CK(hipHostRegister(f, bytes, hipHostRegisterDefault));
      CK(hipMemcpyAsync(f, dev, bytes, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      CK(hipHostUnregister(f));
      free(f);
    });
    const double d_zero = best_of(3, [&] {
      char* f = (char*)calloc(bytes, 1);
      volatile char s = 0;
      for (size_t i = 0; i < bytes; i += 4096) s += f[i];
      free(f);
    });
    printf("{\"mb\": %zu, \"pinned_h2d_ms\": %.3f, \"pinned_d2h_ms\": %.3f, \"pageable_h2d_ms\": %.3f, \"pageable_d2h_ms\": %.3f, "
           "\"register_ms\": %.3f, \"unregister_ms\": %.3f, \"registered_h2d_ms\": %.3f, \"registered_d2h_ms\": %.3f, "
           "\"bounce_h2d_ms_t1_t2_t4\": [%.3f, %.3f, %.3f], \"bounce_d2h_ms_t1_t2_t4\": [%.3f, %.3f, %.3f], "
           "\"fresh_vec_d2h_pageable_ms\": %.3f, \"fresh_vec_d2h_registered_ms\": %.3f, \"fresh_vec_touch_only_ms\": %.3f}\n",
           mb, pin_h2d, pin_d2h, a_h2d, a_d2h, reg, unreg, b_h2d, b_d2h, c_h2d[0], c_h2d[1], c_h2d[2], c_d2h[0], c_d2h[1], c_d2h[2], d_page, d_reg,
           d_zero);
    fflush(stdout);
    CK(hipFree(dev));
    CK(hipHostFree(pinned));
    free(host);
  }
  return 0;
}
