// HIP-only reproducer attempt for the "stalled process" state of DESIGN.md 3.4 (no cosnarks code): does a process that copies straight from /
// into pageable buffers which are then freed and re-allocated start losing ~10 ms steps on later copies?
//   hipcc -O2 --offload-arch=gfx950 tools/experiments/hip_pageable_stall.cpp -o /tmp/stall && /tmp/stall [circuits] [iters] [mode]
// mode 0: as described below (the result buffer is malloc'ed and freed per proof, copies go straight from / into it);
// mode 1: ONE result buffer reused for every proof (never freed); mode 2: the result buffer is malloc'ed and freed per proof, but the
// runtime never sees it: the download lands in a hipHostMalloc'ed buffer and is memcpy'ed on, the re-upload goes the other way round;
// mode 3: mode 2 + the witness upload staged the same way (the runtime sees no pageable memory at all after the key upload).
// Per "circuit": two 64 MB temporaries are uploaded and freed (a key / matrix upload), then `iters` "proofs": upload a 32 MB witness that lives
// as long as the circuit, run a kernel over it, download 32 MB into a freshly malloc'ed result, upload that result again (the h MSM),
// free it. Prints the median / max ms of the witness upload, the download and the re-upload per circuit.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include <sys/resource.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_touch(unsigned long long* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 6364136223846793005ull + 1;
}
// STALL_VMSTAT=1: /proc/vmstat and the thread's context-switch counts around every witness upload; the non-zero deltas of uploads slower
// than 5 ms (and of the first fast one, for contrast) are printed
static std::map<std::string, long long> vmstat() {
  std::map<std::string, long long> m;
  if (FILE* f = fopen("/proc/vmstat", "r")) {
    char k[128];
    long long v;
    while (fscanf(f, "%127s %lld", k, &v) == 2) m[k] = v;
    fclose(f);
  }
  rusage ru;
  if (getrusage(RUSAGE_THREAD, &ru) == 0) {
    m["thread_voluntary_ctx_switches"] = ru.ru_nvcsw;
    m["thread_involuntary_ctx_switches"] = ru.ru_nivcsw;
    m["thread_minor_faults"] = ru.ru_minflt;
    m["thread_user_us"] = ru.ru_utime.tv_sec * 1000000LL + ru.ru_utime.tv_usec;
    m["thread_sys_us"] = ru.ru_stime.tv_sec * 1000000LL + ru.ru_stime.tv_usec;
  }
  return m;
}
static void print_delta(const char* tag, double t_ms, const std::map<std::string, long long>& a, const std::map<std::string, long long>& b) {
  printf("  [%s upload %.2f ms]", tag, t_ms);
  for (auto& kv : b) {
    auto it = a.find(kv.first);
    const long long d = kv.second - (it == a.end() ? 0 : it->second);
    if (d != 0 && kv.first.rfind("nr_", 0) != 0) printf(" %s=%+lld", kv.first.c_str(), d);
  }
  printf("\n");
}
static double ms(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
int main(int argc, char** argv) {
  const int circuits = argc > 1 ? atoi(argv[1]) : 4, iters = argc > 2 ? atoi(argv[2]) : 12, mode = argc > 3 ? atoi(argv[3]) : 0;
  const size_t W = size_t(32) << 20, T = size_t(64) << 20;
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  void *dw, *dh, *dt;
  CK(hipMalloc(&dw, W)); CK(hipMalloc(&dh, W)); CK(hipMalloc(&dt, T));
  void* pinned = nullptr;
  CK(hipHostMalloc(&pinned, W, hipHostMallocDefault));
  char* reused = (char*)malloc(W + 1);
  printf("mode %d\n", mode);
  for (int c = 0; c < circuits; ++c) {
    for (int k = 0; k < 2; ++k) {  // temporaries of a key / matrix upload
      char* tmp = (char*)malloc(T);
      memset(tmp, k + 1, T);
      CK(hipMemcpyAsync(dt, tmp, T, hipMemcpyHostToDevice, st));
      CK(hipStreamSynchronize(st));
      free(tmp);
    }
    char* wit = (char*)malloc(W);
    memset(wit, 7, W);
    std::vector<double> up, down, reup;
    const bool vm = getenv("STALL_VMSTAT") != nullptr;
    bool fast_shown = false;
    for (int it = 0; it < iters; ++it) {
      std::map<std::string, long long> v0;
      if (vm) v0 = vmstat();
      auto t0 = std::chrono::steady_clock::now();
      if (mode == 3) {
        memcpy(pinned, wit, W);
        CK(hipMemcpyAsync(dw, pinned, W, hipMemcpyHostToDevice, st));
      } else {
        CK(hipMemcpyAsync(dw, wit, W, hipMemcpyHostToDevice, st));
      }
      CK(hipStreamSynchronize(st));
      up.push_back(ms(t0));
      if (vm) {
        const auto v1 = vmstat();
        if (up.back() > 5.0) print_delta("SLOW", up.back(), v0, v1);
        else if (!fast_shown) { print_delta("fast", up.back(), v0, v1); fast_shown = true; }
      }
      hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, st, (unsigned long long*)dw, W / 8);
      CK(hipMemcpyAsync(dh, dw, W, hipMemcpyDeviceToDevice, st));
      char* h = mode == 1 ? reused : (char*)malloc(W + 1);  // mode 0 / 2: never touched before the copy, as a Vec::with_capacity result is
      CK(hipStreamSynchronize(st));
      t0 = std::chrono::steady_clock::now();
      if (mode >= 2) {
        CK(hipMemcpyAsync(pinned, dh, W, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        memcpy(h, pinned, W);
      } else {
        CK(hipMemcpyAsync(h, dh, W, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
      }
      down.push_back(ms(t0));
      t0 = std::chrono::steady_clock::now();
      if (mode >= 2) {
        memcpy(pinned, h, W);
        CK(hipMemcpyAsync(dh, pinned, W, hipMemcpyHostToDevice, st));
      } else {
        CK(hipMemcpyAsync(dh, h, W, hipMemcpyHostToDevice, st));
      }
      CK(hipStreamSynchronize(st));
      reup.push_back(ms(t0));
      if (mode != 1) free(h);
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mx = [](const std::vector<double>& v) { return *std::max_element(v.begin(), v.end()); };
    printf("circuit %d: witness upload median %.2f max %.2f ms | result download median %.2f max %.2f | result re-upload median %.2f max %.2f\n", c, med(up), mx(up),
           med(down), mx(down), med(reup), mx(reup));
    fflush(stdout);
    free(wit);
  }
  return 0;
}
