// D2H into memory the caller has just allocated: the DMA engine's first touch of every page costs ~0.45 us (r04_a_pcie_probe: 32 MB in
// 3.9 ms against 0.6 ms into warm pages). Which way of populating the pages first is cheapest?
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/prefault_probe tools/experiments/prefault_probe.cpp -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void populate(char* p, size_t bytes, int threads, int how) {  // how 0: madvise, 1: one store per page
  auto work = [=](size_t lo, size_t hi) {
    if (how == 0) {
      char* a = (char*)(((uintptr_t)p + lo) & ~uintptr_t(4095));
      char* e = p + hi;
      if (madvise(a, e - a, MADV_POPULATE_WRITE) != 0) for (size_t i = lo; i < hi; i += 4096) ((volatile char*)p)[i] = 0;
    } else {
      for (size_t i = lo; i < hi; i += 4096) ((volatile char*)p)[i] = 0;
    }
  };
  if (threads <= 1) return work(0, bytes);
  std::vector<std::thread> th;
  const size_t per = ((bytes / threads) + 4095) & ~size_t(4095);
  for (int t = 0; t < threads; ++t) {
    const size_t lo = per * t, hi = lo + per < bytes ? lo + per : bytes;
    if (lo < hi) th.emplace_back(work, lo, hi);
  }
  for (auto& x : th) x.join();
}

int main() {
  CK(hipSetDevice(0));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  { FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r"); char b[128] = {0}; if (f) { fgets(b, 127, f); fclose(f); } printf("{\"thp\": \"%s\"}\n", strtok(b, "\n")); }
  for (size_t mb : {32, 64}) {
    const size_t bytes = mb << 20;
    char* dev;
    CK(hipMalloc((void**)&dev, bytes));
    CK(hipMemset(dev, 7, bytes));
    for (int how : {0, 1}) for (int threads : {0, 1, 2, 4, 8, 16}) {
      double best_pop = 1e30, best_copy = 1e30, best_tot = 1e30;
      for (int rep = 0; rep < 4; ++rep) {
        char* f = (char*)malloc(bytes);
        const double t0 = now_ms();
        if (threads) populate(f, bytes, threads, how);
        const double t1 = now_ms();
        CK(hipMemcpyAsync(f, dev, bytes, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        const double t2 = now_ms();
        if (f[bytes - 1] != 7 || f[0] != 7) printf("BAD\n");
        free(f);
        if (t2 - t0 < best_tot) best_tot = t2 - t0, best_pop = t1 - t0, best_copy = t2 - t1;
      }
      printf("{\"mb\": %zu, \"how\": \"%s\", \"threads\": %d, \"populate_ms\": %.3f, \"d2h_ms\": %.3f, \"total_ms\": %.3f}\n", mb, how ? "store" : "madvise", threads, best_pop, best_copy, best_tot);
    }
    // a fresh std::vector zero-filled by one thread (what `std::vector<Fr> h(n)` / `vec![F::zero(); n]` costs before any copy)
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) { const double t0 = now_ms(); std::vector<char> v(bytes); const double t1 = now_ms(); if (v[bytes / 2]) printf("?"); if (t1 - t0 < best) best = t1 - t0; }
    printf("{\"mb\": %zu, \"fresh_vector_zero_fill_ms\": %.3f}\n", mb, best);
    // H2D from a freshly written vector (already faulted by the writer): the upload leg
    { std::vector<char> v(bytes, 3); double b2 = 1e30; for (int rep = 0; rep < 4; ++rep) { const double t0 = now_ms(); CK(hipMemcpyAsync(dev, v.data(), bytes, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); const double t1 = now_ms(); if (t1 - t0 < b2) b2 = t1 - t0; } printf("{\"mb\": %zu, \"h2d_from_written_vector_ms\": %.3f}\n", mb, b2); }
    CK(hipFree(dev));
  }
  return 0;
}
