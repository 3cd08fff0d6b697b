// Context, memory plumbing, events: the non-compute part of the C ABI (include/cosnarks_hip.h).
#include <errno.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <sys/mman.h>

#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "common.hpp"

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

namespace csh {

// Populate the pages of a D2H destination from a few host threads while the device is still busy (see HostXfer in common.hpp).
// MADV_POPULATE_WRITE (Linux >= 5.14) faults the pages in writable without touching their content; on kernels without it the
// call fails with EINVAL and the copy simply pays the first touch itself, as before.
static inline int us_since(std::chrono::steady_clock::time_point t0) {
  return (int)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
}
void host_populate_begin(void* p, size_t bytes, std::vector<std::thread>& workers) {
  const int knob = tune().host_populate.load(std::memory_order_relaxed);
  const int threads = knob & 0xff;  // bit 8 (default on): ask for transparent huge pages on the range first (MADV_HUGEPAGE, a hint)
  if (!p || threads <= 0 || bytes < (size_t(4) << 20)) return;
  char* lo = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 4095) & ~uintptr_t(4095));  // whole pages inside the buffer only
  char* hi = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + bytes) & ~uintptr_t(4095));
  if (hi <= lo) return;
  const size_t span = (size_t)(hi - lo);
  // 2 MiB pages where the kernel grants them: 16 faults instead of 8192 for a 32 MB result (the hosts run THP in "madvise" mode)
  if (knob & 0x100) (void)madvise(lo, span, MADV_HUGEPAGE);
  const size_t per = ((span / (size_t)threads) + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);  // 2 MiB-granular shares
  for (int t = 0; t < threads; ++t) {
    char* a = lo + per * (size_t)t;
    if (a >= hi) break;
    const size_t len = (size_t)(hi - a) < per ? (size_t)(hi - a) : per;
    try {
      workers.emplace_back([a, len] {
        // MADV_POPULATE_WRITE faults the pages in writable without touching their content; it may stop early (EINTR / EAGAIN under
        // memory-management contention with the driver pinning the upload's source pages: seen as 11-16 ms witness maps when the copy
        // then first-touched the rest itself), so it is retried piecewise, and where the kernel lacks it (EINVAL) every page is
        // touched by storing back the byte it holds (content unchanged: the copy that fills the buffer is enqueued after the join).
        const auto t0 = std::chrono::steady_clock::now();
        size_t done = 0;
        const size_t piece = size_t(2) << 20;
        int tries = 0;
        while (done < len) {
          const size_t l = len - done < piece ? len - done : piece;
          if (madvise(a + done, l, MADV_POPULATE_WRITE) == 0) {
            done += l;
            tries = 0;
          } else if ((errno == EINTR || errno == EAGAIN) && ++tries < 64) {
            continue;
          } else {
            for (size_t i = 0; i < l; i += 4096) {  // write back what is there: in-place entry points pass their INPUT buffer here
              volatile char* q = reinterpret_cast<volatile char*>(a + done + i);
              *q = *q;
            }
            done += l;
            tries = 0;
          }
        }
        tune().stat_populate_us.fetch_add(us_since(t0), std::memory_order_relaxed);
      });
    } catch (...) {  // no thread to be had: the copy populates the rest
      break;
    }
  }
}

int* lane_d2h_slow_run();  // consecutive stalled result copies of the calling thread's lane on its device (defined with the lanes below)

// Copier pool of the staged transfers: a handful of persistent host threads (started on first use, parked on a condition variable)
// instead of seven std::thread constructions per staged copy (~40 us each on these hosts: 0.3 ms per transfer, two or three transfers per
// witness map). run(k, fn): fn runs on the caller and on up to k pool threads at once (each claims chunks from the caller's own atomic
// counter) and run returns when all of them have returned. A caller that finds the workers busy with another caller's copy does more
// of its own chunks itself AND, once its own fn() has returned, withdraws the jobs no worker has picked up yet (round 6, ADVICE r5: it
// used to wait until every queued job had been dequeued, i.e. behind workers blocked in another caller's hipEventSynchronize).
namespace {
struct CopierPool {
  static constexpr int THREADS = 8;
  std::mutex mu;
  std::condition_variable cv;
  struct Job {
    std::function<void()>* fn;
    std::atomic<int>* pending;
    std::mutex* done_mu;
    std::condition_variable* done_cv;
  };
  std::vector<Job> queue;
  int started = 0;
  void worker() {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [&] { return !queue.empty(); });
        j = queue.back();
        queue.pop_back();
      }
      (*j.fn)();
      {
        // the notification goes out UNDER the caller's mutex: once it is released the caller may return from run() and its mutex,
        // condition variable and counter (all on its stack) are gone -- nothing of the job is touched after this block
        std::lock_guard<std::mutex> g(*j.done_mu);
        j.pending->fetch_sub(1, std::memory_order_acq_rel);
        j.done_cv->notify_one();
      }
    }
  }
  void run(int helpers, std::function<void()> fn) {
    if (tune().host_copier_pool.load(std::memory_order_relaxed) == 0) {  // A/B: a std::thread per helper, as before round 5
      std::vector<std::thread> th;
      try {
        for (int t = 0; t < helpers; ++t) th.emplace_back(fn);
      } catch (...) {
      }
      fn();
      for (auto& t : th) t.join();
      return;
    }
    std::atomic<int> pending{0};
    std::mutex done_mu;
    std::condition_variable done_cv;
    if (helpers > 0) {
      std::lock_guard<std::mutex> g(mu);
      while (started < THREADS && started < helpers) {
        try {
          std::thread([this] { worker(); }).detach();  // process-lifetime threads: they only ever touch memory a caller of run() keeps alive until it returns
          ++started;
        } catch (...) {
          break;
        }
      }
      const int k = helpers < started ? helpers : started;
      pending.store(k);
      for (int i = 0; i < k; ++i) queue.push_back(Job{&fn, &pending, &done_mu, &done_cv});
    }
    if (pending.load() > 0) cv.notify_all();
    fn();
    if (pending.load(std::memory_order_acquire) > 0) {  // withdraw this call's jobs that nobody has started: they would find no chunk left anyway
      std::lock_guard<std::mutex> g(mu);
      int withdrawn = 0;
      for (size_t i = queue.size(); i-- > 0;)
        if (queue[i].fn == &fn) {
          queue.erase(queue.begin() + (ptrdiff_t)i);
          ++withdrawn;
        }
      if (withdrawn) pending.fetch_sub(withdrawn, std::memory_order_acq_rel);
    }
    std::unique_lock<std::mutex> g(done_mu);
    done_cv.wait(g, [&] { return pending.load(std::memory_order_acquire) == 0; });
  }
};
CopierPool& copiers() {
  static CopierPool* p = new CopierPool();  // never destroyed: its detached workers may outlive static destruction
  return *p;
}
}  // namespace

// Round 5 root cause of the "stalled copies" (profiles/archive/r05_l .. r05_r_trait_stall_*; reproducer tools/experiments/trait_stall_probe.py): when
// the runtime copies straight from / into caller memory it pins those pages, and once such memory has been unmapped again (a freed h
// vector, a dropped witness: from the second proof of a process on) some LATER operation of the process -- any stream, either direction,
// staged or not -- takes 10, 20 or 30 ms longer, in steps of the kernel's 10 ms tick, on about half the processes of a box. With BOTH
// directions staged (the runtime never sees caller memory) 16 of 16 probe runs were clean; with either direction direct 5 of 12 stalled.
// The condition is process-wide, so is the reaction: a stalled large copy in either direction (two in a row on one lane) sends every
// large transfer of the process through the staged paths for the next STAGE_ALL_SPELL transfers, then direct copies are tried again.
// Page-locked staging is bounded (round 6): a transfer of up to STAGE_RING_BYTES is staged whole (<= 16 chunks, as measured in round 5 for
// the 32 MB vectors of a 2^20 witness map); a larger one cycles through a ring of that size in generations of eight 2 MiB chunks.
constexpr size_t STAGE_RING_CHUNK = size_t(2) << 20;
constexpr size_t STAGE_RING_BYTES = size_t(32) << 20;
static std::atomic<int> g_stage_all_left{0};
constexpr int STAGE_ALL_SPELL = 4096;
static bool stage_all_take() {
  int v = g_stage_all_left.load(std::memory_order_relaxed);
  while (v > 0)
    if (g_stage_all_left.compare_exchange_weak(v, v - 1, std::memory_order_relaxed)) return true;
  return false;
}
static void stage_all_begin() {
  g_stage_all_left.store(STAGE_ALL_SPELL, std::memory_order_relaxed);
  tune().stat_stage_all_switches.fetch_add(1, std::memory_order_relaxed);
}

void HostXfer::join() {
  if (workers.empty()) return;
  const auto t0 = std::chrono::steady_clock::now();
  for (auto& t : workers)
    if (t.joinable()) t.join();
  workers.clear();
  tune().stat_join_wait_us.fetch_add(us_since(t0), std::memory_order_relaxed);
}
HostXfer::~HostXfer() {
  join();
  for (Staged& s : staged)
    for (hipEvent_t e : s.landed)
      if (e) (void)hipEventDestroy(e);
}
// Result copy. Direct: one DMA into the caller's pages (the runtime pins them for the duration). Staged (tune "host_d2h" = 1): the
// driver never touches the caller's memory -- chunks land in the lane's page-locked buffer and host threads move them on. On most
// boxes the direct copy of 32 MB takes 0.6 ms once the pages are present; on some (profiles/archive/r04_zd_trait_modes.log, same code, same
// sizes) the host-facing witness map took 12-24 ms instead of 2.4 in steps of ~10 ms while every device-resident path ran at
// its usual speed, i.e. the stall sits in the driver's handling of freshly populated caller pages.
int HostXfer::d2h(void* host, const void* dev, size_t bytes, hipStream_t st) {
  // tune "host_d2h": 0 = always direct, 1 (default since round 5) = always staged, 2 = direct, timed; two copies in a row on one lane that take more than
  // three times their PCIe time + 4 ms send every large transfer of the PROCESS (both directions, see g_stage_all_left) through the staged
  // paths for a spell, after which a direct copy is tried again.
  int* slow_run = lane_d2h_slow_run();
  const int mode = tune().host_d2h.load(std::memory_order_relaxed);
  const bool large = bytes >= (size_t(4) << 20);
  bool stage = large && mode == 1;
  if (large && mode == 2 && stage_all_take()) stage = true;
  // ONE staged copy per HostXfer: the lane has one page-locked slot for this purpose (pinned_for(st ^ 0x8)); a second staged copy before
  // finish() would land in the same buffer -- or free it, if it is larger -- while the first is still waiting to be moved on.
  // Every caller today copies one result per HostXfer; a second one takes the direct path.
  if (!staged.empty()) stage = false;
  void *ph = nullptr, *pd = nullptr;
  if (stage && bytes > STAGE_RING_BYTES && pinned_for((hipStream_t)((uintptr_t)st ^ 0x8), STAGE_RING_BYTES, &ph, &pd)) {
    // larger than the staging ring: nothing is enqueued here; finish() moves it through the ring's two halves (DMA of generation g
    // under the copy-out of generation g - 1) once the stream's earlier work has been waited for anyway
    Staged sg;
    sg.host = host, sg.pinned = static_cast<const char*>(ph), sg.bytes = bytes, sg.chunk = 0;
    sg.ring_dev = dev;
    staged.push_back(std::move(sg));
    tune().stat_d2h_staged.fetch_add(1, std::memory_order_relaxed);
    return CSH_OK;
  }
  if (stage && pinned_for((hipStream_t)((uintptr_t)st ^ 0x8), bytes, &ph, &pd)) {
    Staged sg;
    sg.host = host, sg.pinned = static_cast<const char*>(ph), sg.bytes = bytes;
    sg.chunk = (((bytes + 15) / 16) + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);  // <= 16 chunks, 2 MiB-granular
    for (size_t off = 0; off < bytes; off += sg.chunk) {
      const size_t l = bytes - off < sg.chunk ? bytes - off : sg.chunk;
      hipEvent_t e = nullptr;
      hipError_t rc = hipMemcpyAsync(static_cast<char*>(ph) + off, static_cast<const char*>(dev) + off, l, hipMemcpyDeviceToHost, st);
      if (rc == hipSuccess) rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
      if (rc == hipSuccess) rc = hipEventRecord(e, st);
      sg.landed.push_back(e);
      if (rc != hipSuccess) {
        staged.push_back(std::move(sg));  // the destructor releases the events
        set_error("staged result copy failed: %s", hipGetErrorString(rc));
        return CSH_ERR_HIP;
      }
    }
    staged.push_back(std::move(sg));
    tune().stat_d2h_staged.fetch_add(1, std::memory_order_relaxed);
    return CSH_OK;
  }
  if (large && mode == 2) {  // the copy alone between two stream waits (a copy into pageable memory returns when it is done anyway)
    CSH_HIP(hipStreamSynchronize(st));
    join();
    const auto t0 = std::chrono::steady_clock::now();
    CSH_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
    CSH_HIP(hipStreamSynchronize(st));
    const double ms = us_since(t0) * 1e-3;
    // consecutive stalled copies of this lane (one alone may be the process's first: runtime start-up)
    if (ms > 3.0 * (double)bytes / 20e6 + 4.0) {
      tune().stat_d2h_slow.fetch_add(1, std::memory_order_relaxed);
      if (++*slow_run >= 2) {
        stage_all_begin();
        *slow_run = 1;  // the probe after the staged spell switches back at once if it stalls again
      }
    } else {
      *slow_run = 0;
    }
    return CSH_OK;
  }
  join();
  if (bytes) CSH_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
  return CSH_OK;
}
int HostXfer::finish(hipStream_t st) {
  join();  // the destination's pages are present from here on
  if (!staged.empty()) {
    int device = 0;
    (void)hipGetDevice(&device);
    const auto t0 = std::chrono::steady_clock::now();
    std::atomic<int> failed{0};
    for (Staged& sg : staged) {
      if (sg.ring_dev) {  // through the ring: half h receives generation g while the copiers empty the other half
        const size_t half = STAGE_RING_BYTES / 2, ngen = (sg.bytes + half - 1) / half;
        hipEvent_t ev[2] = {nullptr, nullptr};
        hipError_t rc = hipEventCreateWithFlags(&ev[0], hipEventDisableTiming);
        if (rc == hipSuccess) rc = hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
        auto enqueue = [&](size_t g) {
          const size_t off = g * half, l = sg.bytes - off < half ? sg.bytes - off : half;
          hipError_t e = hipMemcpyAsync(const_cast<char*>(sg.pinned) + (g & 1) * half, static_cast<const char*>(sg.ring_dev) + off, l, hipMemcpyDeviceToHost, st);
          if (e == hipSuccess) e = hipEventRecord(ev[g & 1], st);
          return e;
        };
        if (rc == hipSuccess) rc = enqueue(0);
        for (size_t g = 0; rc == hipSuccess && g < ngen; ++g) {
          if (g + 1 < ngen) rc = enqueue(g + 1);   // into the half generation g - 1 was copied out of below
          if (rc == hipSuccess) rc = hipEventSynchronize(ev[g & 1]);
          if (rc != hipSuccess) break;
          const size_t off = g * half, l = sg.bytes - off < half ? sg.bytes - off : half;
          const size_t nch = (l + STAGE_RING_CHUNK - 1) / STAGE_RING_CHUNK;
          std::atomic<size_t> next{0};
          const char* src = sg.pinned + (g & 1) * half;
          char* dst = static_cast<char*>(sg.host) + off;
          copiers().run(nch > 1 ? 7 : 0, [&] {
            for (;;) {
              const size_t c = next.fetch_add(1, std::memory_order_relaxed);
              if (c >= nch) return;
              const size_t o = c * STAGE_RING_CHUNK, ll = l - o < STAGE_RING_CHUNK ? l - o : STAGE_RING_CHUNK;
              memcpy(dst + o, src + o, ll);
            }
          });
        }
        for (hipEvent_t e : ev)
          if (e) (void)hipEventDestroy(e);
        if (rc != hipSuccess) failed.store(1);
        continue;
      }
      const size_t nchunks = sg.landed.size();
      std::atomic<size_t> next{0};
      auto work = [&] {
        for (;;) {
          const size_t c = next.fetch_add(1, std::memory_order_relaxed);
          if (c >= nchunks) return;
          if (hipEventSynchronize(sg.landed[c]) != hipSuccess) {
            failed.store(1);
            return;
          }
          const size_t off = c * sg.chunk, l = sg.bytes - off < sg.chunk ? sg.bytes - off : sg.chunk;
          memcpy(static_cast<char*>(sg.host) + off, sg.pinned + off, l);
        }
      };
      const int extra = nchunks > 8 ? 7 : (nchunks > 1 ? 3 : 0);
      copiers().run(extra, [&, device] {
        (void)hipSetDevice(device);  // pool threads serve callers on any device
        work();
      });
    }
    tune().stat_finish_us.fetch_add(us_since(t0), std::memory_order_relaxed);
    if (failed.load()) {
      (void)hipStreamSynchronize(st);
      set_error("staged result copy: a chunk failed on the device");
      return CSH_ERR_HIP;
    }
  }
  const auto t0 = std::chrono::steady_clock::now();
  CSH_HIP(hipStreamSynchronize(st));
  if (staged.empty()) tune().stat_finish_us.fetch_add(us_since(t0), std::memory_order_relaxed);
  return CSH_OK;
}

int HostStage::down(void* host, const void* dev, size_t bytes) {
  HostXfer x;
  x.expect_d2h(host, bytes);  // overlaps whatever the stream still has queued
  CSH_TRY(x.d2h(host, dev, bytes, st));
  return x.finish(st);
}

static thread_local std::string tl_error;
static thread_local int tl_device = -1;

// A "lane" = one HIP stream plus the scratch arenas used on it. Lanes are pooled per device and leased to host
// threads: short-lived callers (the reference spawns rayon tasks / scoped threads per proof) reuse warm streams and
// already-grown arenas instead of paying hipStreamCreate + hipMalloc on every call.
struct PinnedSlot {
  void* host = nullptr;  // hipHostMalloc: page-locked, mapped into the device's address space
  void* dev = nullptr;   // the device-side alias of `host`
  size_t cap = 0;
};
struct Lane {
  int device = 0;
  hipStream_t stream = nullptr;
  // stall DETECTION is per lane (a lane is leased to one host thread at a time: plain ints, no races between concurrent callers, VERDICT r4
  // #6): consecutive stalled large copies of this lane, either direction. The REACTION is process-wide (g_stage_all_left): so is the stall.
  int d2h_slow_run = 0, h2d_slow_run = 0;
  hipEvent_t h2d_done[4] = {nullptr, nullptr, nullptr, nullptr};  // last DMA out of each page-locked upload slot
  hipEvent_t h2d_ring_ev[4][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};  // per slot: last DMA out of each half of the staging ring
  std::map<hipStream_t, Arena> arenas;
  std::map<hipStream_t, PinnedSlot> pinned;
};
static std::mutex g_lane_mu;
static std::map<int, std::vector<Lane*>> g_free_lanes;

struct LaneHolder {
  std::map<int, Lane*> by_device;  // key: device, or device + AUX_KEY for the thread's second (overlap) lane
  static constexpr int AUX_KEY = 1 << 20;
  Lane* get(int key) {
    const int device = key >= AUX_KEY ? key - AUX_KEY : key;
    auto it = by_device.find(key);
    if (it != by_device.end()) return it->second;
    Lane* l = nullptr;
    {
      std::lock_guard<std::mutex> g(g_lane_mu);
      auto& fl = g_free_lanes[device];
      if (!fl.empty()) {
        l = fl.back();
        fl.pop_back();
      }
    }
    if (!l) {
      l = new Lane();
      l->device = device;
      tune().stat_lanes.fetch_add(1, std::memory_order_relaxed);
      if (hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) != hipSuccess) l->stream = nullptr;  // fall back to the null stream
    }
    by_device[key] = l;
    return l;
  }
  ~LaneHolder() {  // thread exit: hand the lanes back (work on them has been synchronised by the API contract)
    std::lock_guard<std::mutex> g(g_lane_mu);
    for (auto& kv : by_device) g_free_lanes[kv.first >= AUX_KEY ? kv.first - AUX_KEY : kv.first].push_back(kv.second);
  }
};
static thread_local LaneHolder tl_lanes;
int* lane_d2h_slow_run() { return &tl_lanes.get(tl_device < 0 ? 0 : tl_device)->d2h_slow_run; }

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && *e) ? (int)strtol(e, nullptr, 0) : dflt;  // decimal, 0x.. or 0.. as the caller writes it
}
struct TuneEntry { const char* key; const char* env; std::atomic<int> Tune::*field; };
static const TuneEntry kTune[] = {
    {"msm_c", "CSH_MSM_C", &Tune::msm_c},
    {"msm_l", "CSH_MSM_L", &Tune::msm_l},
    {"msm_balanced", "CSH_MSM_BALANCED", &Tune::msm_balanced},
    {"msm_w", "CSH_MSM_W", &Tune::msm_w},
    {"msm_timing", "CSH_MSM_TIMING", &Tune::msm_timing},
    {"msm_no_table", "CSH_MSM_NO_TABLE", &Tune::msm_no_table},
    {"msm_multi_overlap", "CSH_MSM_MULTI_OVERLAP", &Tune::msm_multi_overlap},
    {"msm_share_uploads", "CSH_MSM_SHARE_UPLOADS", &Tune::msm_share_uploads},
    {"stat_uploads_shared", "CSH_STAT_UPLOADS_SHARED", &Tune::stat_uploads_shared},
    {"host_h2d", "CSH_HOST_H2D", &Tune::host_h2d},
    {"stat_h2d_slow", "CSH_STAT_H2D_SLOW", &Tune::stat_h2d_slow},
    {"stat_h2d_staged", "CSH_STAT_H2D_STAGED", &Tune::stat_h2d_staged},
    {"stat_stage_all_switches", "CSH_STAT_STAGE_ALL_SWITCHES", &Tune::stat_stage_all_switches},
    {"stat_pinned_kib", "CSH_STAT_PINNED_KIB", &Tune::stat_pinned_kib},
    {"host_copier_pool", "CSH_HOST_COPIER_POOL", &Tune::host_copier_pool},
    {"host_timing", "CSH_HOST_TIMING", &Tune::host_timing},
    {"stat_wm_h2d_us", "CSH_STAT_WM_H2D_US", &Tune::stat_wm_h2d_us},
    {"stat_wm_dev_us", "CSH_STAT_WM_DEV_US", &Tune::stat_wm_dev_us},
    {"stat_wm_d2h_us", "CSH_STAT_WM_D2H_US", &Tune::stat_wm_d2h_us},
    {"acc_blk", "CSH_ACC_BLK", &Tune::acc_blk},
    {"sort_two_level", "CSH_SORT_TWO_LEVEL", &Tune::sort_two_level},
    {"vec_max_blocks", "CSH_VEC_MAX_BLOCKS", &Tune::vec_max_blocks},
    {"ntt_lazy", "CSH_NTT_LAZY", &Tune::ntt_lazy},
    {"ntt_threads", "CSH_NTT_THREADS", &Tune::ntt_threads},
    {"msm_variant", "CSH_MSM_VARIANT", &Tune::msm_variant},
    {"msm_seg_buckets", "CSH_MSM_SEG_BUCKETS", &Tune::msm_seg_buckets},
    {"msm_wide_lb", "CSH_MSM_WIDE_LB", &Tune::msm_wide_lb},
    {"msm_wide_chunks", "CSH_MSM_WIDE_CHUNKS", &Tune::msm_wide_chunks},
    {"allow_unmasked_rep3", "CSH_ALLOW_UNMASKED_REP3", &Tune::allow_unmasked_rep3},
    {"ntt_variant", "CSH_NTT_VARIANT", &Tune::ntt_variant},
    {"ntt_pair", "CSH_NTT_PAIR", &Tune::ntt_pair},
    {"ntt_pair_min_log", "CSH_NTT_PAIR_MIN_LOG", &Tune::ntt_pair_min_log},
    {"h_unfused", "CSH_H_UNFUSED", &Tune::h_unfused},
    {"h_table_cache", "CSH_H_TABLE_CACHE", &Tune::h_table_cache},
    {"host_populate", "CSH_HOST_POPULATE", &Tune::host_populate},
    {"host_d2h", "CSH_HOST_D2H", &Tune::host_d2h},
    {"comm_timeout_ms", "CSH_COMM_TIMEOUT_MS", &Tune::comm_timeout_ms},
    {"comm_nonblocking", "CSH_COMM_NONBLOCKING", &Tune::comm_nonblocking},
    {"stat_arena_grows", "CSH_STAT_ARENA_GROWS", &Tune::stat_arena_grows},
    {"stat_lanes", "CSH_STAT_LANES", &Tune::stat_lanes},
    {"stat_populate_us", "CSH_STAT_POPULATE_US", &Tune::stat_populate_us},
    {"stat_join_wait_us", "CSH_STAT_JOIN_WAIT_US", &Tune::stat_join_wait_us},
    {"stat_finish_us", "CSH_STAT_FINISH_US", &Tune::stat_finish_us},
    {"stat_d2h_slow", "CSH_STAT_D2H_SLOW", &Tune::stat_d2h_slow},
    {"stat_d2h_staged", "CSH_STAT_D2H_STAGED", &Tune::stat_d2h_staged},
};
Tune& tune() {
  static Tune* t = [] {
    Tune* x = new Tune();
    for (const TuneEntry& e : kTune) {
      int v = env_int(e.env, (x->*(e.field)).load());
      // no result-changing knob is readable from the environment (an inherited variable must not make a prover emit invalid proofs):
      // the experiment bits of ntt_variant are dropped, allow_unmasked_rep3 is never taken from it
      if (!kExperiments && e.field == &Tune::ntt_variant) v &= ~NTT_VARIANT_EXPERIMENT_BITS;
      if (e.field == &Tune::allow_unmasked_rep3) v = 0;
      (x->*(e.field)).store(v);
    }
    return x;
  }();
  return *t;
}

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

int device_simds() {
  static std::atomic<int> cache[64];
  int dev = tl_device;
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return 1024;
  if (dev < 0 || dev >= 64) return 1024;
  int v = cache[dev].load(std::memory_order_relaxed);
  if (v == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    v = 4 * cus;
    cache[dev].store(v, std::memory_order_relaxed);
  }
  return v;
}

hipStream_t resolve_stream(void* s) {
  if (s) return reinterpret_cast<hipStream_t>(s);
  return tl_lanes.get(tl_device)->stream;
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
  tune().stat_arena_grows.fetch_add(1, std::memory_order_relaxed);
  void* p = nullptr;
  CSH_HIP(hipMalloc(&p, want));
  base = static_cast<char*>(p);
  cap = want;
  return CSH_OK;
}

Arena& arena_for(hipStream_t s) { return tl_lanes.get(tl_device)->arenas[s]; }

// a second pooled stream for the calling thread (overlapping two stages of one call); nullptr if none could be created
// Small page-locked result buffer bound to (calling thread's lane, stream): kernels write a call's few hundred bytes of results
// straight into host memory, so the synchronous entry points end with a stream synchronisation instead of a staged copy.
bool pinned_for(hipStream_t s, size_t bytes, void** host, void** dev) {
  PinnedSlot& slot = tl_lanes.get(tl_device)->pinned[s];
  if (!slot.host || slot.cap < bytes) {
    if (slot.host) {
      (void)hipHostFree(slot.host);
      tune().stat_pinned_kib.fetch_sub((int)(slot.cap >> 10), std::memory_order_relaxed);
    }
    slot = PinnedSlot{};
    const size_t cap = bytes < 16384 ? 16384 : bytes;
    void* h = nullptr;
    void* d = nullptr;
    if (hipHostMalloc(&h, cap, hipHostMallocDefault) != hipSuccess) return false;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
      (void)hipHostFree(h);
      return false;
    }
    slot.host = h;
    slot.dev = d;
    slot.cap = cap;
    tune().stat_pinned_kib.fetch_add((int)(cap >> 10), std::memory_order_relaxed);
  }
  *host = slot.host;
  *dev = slot.dev;
  return true;
}

int raise_lds_limit(const void* kernel, size_t bytes) {
  static thread_local std::map<std::pair<const void*, int>, size_t> raised;
  int dev = tl_device;
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
  size_t& have = raised[{kernel, dev}];
  if (have >= bytes) return CSH_OK;
  CSH_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  have = bytes;
  return CSH_OK;
}

hipStream_t resolve_aux_stream() { return tl_lanes.get(tl_device + LaneHolder::AUX_KEY)->stream; }

int upload_h2d(void* dev, const void* host, size_t bytes, hipStream_t st, int slot) {
  if (!bytes) return CSH_OK;
  const int mode = tune().host_h2d.load(std::memory_order_relaxed);
  const bool large = bytes >= (size_t(4) << 20);
  Lane* lane = tl_lanes.get(tl_device < 0 ? 0 : tl_device);
  bool stage = large && mode == 1;
  if (large && mode == 2 && stage_all_take()) stage = true;
  slot &= 3;
  void *ph = nullptr, *pd = nullptr;
  if (stage) {
    // the slot's previous DMAs must have drained before its buffer is overwritten (or regrown)
    if (lane->h2d_done[slot]) CSH_HIP(hipEventSynchronize(lane->h2d_done[slot]));
    if (!pinned_for((hipStream_t)((uintptr_t)st ^ (uintptr_t)(0x10 + 0x10 * slot)), bytes < STAGE_RING_BYTES ? bytes : STAGE_RING_BYTES, &ph, &pd)) stage = false;
  }
  if (stage && bytes > STAGE_RING_BYTES) {
    // Round 6: transfers beyond the ring size go through a FIXED page-locked ring (two halves of 8 x 2 MiB, pinned_for above
    // was asked for STAGE_RING_BYTES): generation g of eight chunks is copied into half g & 1 by the copier threads, each chunk's DMA
    // following at once, while the DMAs of generation g - 1 drain from the other half; a half is reused once the event recorded after
    // its previous generation has fired. A key or a 2^24 scalar vector used to leave its FULL size page-locked per (lane, stream, slot)
    // for the life of the process (2 GB on the witness-map thread of a 2^24 prove).
    const size_t chunk = STAGE_RING_CHUNK, half = STAGE_RING_BYTES / 2, per_gen = half / chunk;
    const size_t nchunks = (bytes + chunk - 1) / chunk, ngen = (nchunks + per_gen - 1) / per_gen;
    hipEvent_t* gen_ev = lane->h2d_ring_ev[slot];
    int device = 0;
    (void)hipGetDevice(&device);
    std::mutex enq;
    for (size_t g = 0; g < ngen; ++g) {
      const size_t h = g & 1;
      if (!gen_ev[h]) CSH_HIP(hipEventCreateWithFlags(&gen_ev[h], hipEventDisableTiming));
      else if (g >= 2) CSH_HIP(hipEventSynchronize(gen_ev[h]));
      const size_t c0 = g * per_gen, c1 = c0 + per_gen < nchunks ? c0 + per_gen : nchunks;
      std::atomic<size_t> next{c0};
      std::atomic<int> failed{0};
      auto work = [&] {
        for (;;) {
          const size_t c = next.fetch_add(1, std::memory_order_relaxed);
          if (c >= c1) return;
          const size_t off = c * chunk, l = bytes - off < chunk ? bytes - off : chunk;
          char* stage_at = static_cast<char*>(ph) + h * half + (c - c0) * chunk;
          memcpy(stage_at, static_cast<const char*>(host) + off, l);
          std::lock_guard<std::mutex> gq(enq);
          if (hipMemcpyAsync(static_cast<char*>(dev) + off, stage_at, l, hipMemcpyHostToDevice, st) != hipSuccess) failed.store(1);
        }
      };
      copiers().run(c1 - c0 > 1 ? 7 : 0, [&, device] {
        (void)hipSetDevice(device);
        work();
      });
      if (failed.load()) {
        set_error("staged upload: a chunk copy failed");
        return CSH_ERR_HIP;
      }
      CSH_HIP(hipEventRecord(gen_ev[h], st));
    }
    if (!lane->h2d_done[slot]) CSH_HIP(hipEventCreateWithFlags(&lane->h2d_done[slot], hipEventDisableTiming));
    CSH_HIP(hipEventRecord(lane->h2d_done[slot], st));
    tune().stat_h2d_staged.fetch_add(1, std::memory_order_relaxed);
    return CSH_OK;
  }
  if (stage) {
    const size_t chunk = (((bytes + 15) / 16) + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);  // <= 16 chunks, 2 MiB-granular
    const size_t nchunks = (bytes + chunk - 1) / chunk;
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    std::mutex enq;  // the chunks' DMAs are enqueued by whichever thread finished copying them (disjoint ranges: order is irrelevant)
    int device = 0;
    (void)hipGetDevice(&device);
    auto work = [&] {
      for (;;) {
        const size_t c = next.fetch_add(1, std::memory_order_relaxed);
        if (c >= nchunks) return;
        const size_t off = c * chunk, l = bytes - off < chunk ? bytes - off : chunk;
        memcpy(static_cast<char*>(ph) + off, static_cast<const char*>(host) + off, l);
        std::lock_guard<std::mutex> g(enq);
        if (hipMemcpyAsync(static_cast<char*>(dev) + off, static_cast<const char*>(ph) + off, l, hipMemcpyHostToDevice, st) != hipSuccess) failed.store(1);
      }
    };
    const int extra = nchunks > 8 ? 7 : (nchunks > 1 ? 3 : 0);  // 32 MB: eight copiers of two chunks each (~0.4 ms of host copy, the DMAs trail by one chunk)
    copiers().run(extra, [&, device] {
      (void)hipSetDevice(device);
      work();
    });
    if (failed.load()) {
      set_error("staged upload: a chunk copy failed");
      return CSH_ERR_HIP;
    }
    if (!lane->h2d_done[slot]) CSH_HIP(hipEventCreateWithFlags(&lane->h2d_done[slot], hipEventDisableTiming));
    CSH_HIP(hipEventRecord(lane->h2d_done[slot], st));
    tune().stat_h2d_staged.fetch_add(1, std::memory_order_relaxed);
    return CSH_OK;
  }
  if (large && mode == 2) {  // a copy from pageable memory returns when the source has been consumed: the call itself is what stalls
    const auto t0 = std::chrono::steady_clock::now();
    CSH_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st));
    const double ms = us_since(t0) * 1e-3;
    if (ms > 3.0 * (double)bytes / 20e6 + 4.0) {
      tune().stat_h2d_slow.fetch_add(1, std::memory_order_relaxed);
      if (++lane->h2d_slow_run >= 2) {
        stage_all_begin();
        lane->h2d_slow_run = 1;
      }
    } else {
      lane->h2d_slow_run = 0;
    }
    return CSH_OK;
  }
  CSH_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st));
  return CSH_OK;
}

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
  auto drop = [](Lane* l) {
    for (auto& kv : l->arenas)
      if (kv.second.base) (void)hipFree(kv.second.base);
    l->arenas.clear();
    for (auto& kv : l->pinned)
      if (kv.second.host) {
        (void)hipHostFree(kv.second.host);
        tune().stat_pinned_kib.fetch_sub((int)(kv.second.cap >> 10), std::memory_order_relaxed);
      }
    l->pinned.clear();
  };
  for (auto& kv : tl_lanes.by_device) drop(kv.second);
  std::lock_guard<std::mutex> g(g_lane_mu);
  for (auto& kv : g_free_lanes)
    for (Lane* l : kv.second) drop(l);
  return CSH_OK;
}

const char* csh_last_error(void) { return tl_error.c_str(); }
const char* csh_version(void) { return "cosnarks-hip 0.1.0 (gfx950)"; }

int csh_tune_set(const char* key, int value) {
  CSH_REQUIRE(key, "key is NULL");
  for (const TuneEntry& e : kTune)
    if (!strcmp(e.key, key)) {
      if (!kExperiments && ((e.field == &Tune::ntt_variant && (value & NTT_VARIANT_EXPERIMENT_BITS) != 0) ||
                            (e.field == &Tune::allow_unmasked_rep3 && value != 0))) {
        set_error("csh_tune_set: '%s' = %d selects a timing experiment that returns wrong results; it only exists in builds with -DCSH_EXPERIMENTS", key, value);
        return CSH_ERR_INVALID;
      }
      (tune().*(e.field)).store(value);
      return CSH_OK;
    }
  set_error("csh_tune_set: unknown key '%s'", key);
  return CSH_ERR_INVALID;
}
int csh_tune_get(const char* key, int* value) {
  CSH_REQUIRE(key && value, "NULL argument");
  for (const TuneEntry& e : kTune)
    if (!strcmp(e.key, key)) {
      *value = (tune().*(e.field)).load();
      return CSH_OK;
    }
  set_error("csh_tune_get: unknown key '%s'", key);
  return CSH_ERR_INVALID;
}

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

int csh_current_device(int* device) {
  CSH_REQUIRE(device, "device is NULL");
  CSH_TRY(ensure_device());
  *device = tl_device;
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
  if (bytes >= (size_t(4) << 20)) {  // large: the same staged / direct policy as every other upload from caller memory ("host_h2d")
    hipStream_t st = resolve_stream(nullptr);
    CSH_TRY(upload_h2d(dev_dst, host_src, bytes, st, 0));
    CSH_HIP(hipStreamSynchronize(st));
    return CSH_OK;
  }
  CSH_HIP(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
  return CSH_OK;
}
int csh_memcpy_d2h(void* host_dst, const void* dev_src, size_t bytes) {
  CSH_TRY(ensure_device());
  if (bytes >= (size_t(4) << 20)) {  // large: HostXfer ("host_d2h", "host_populate")
    hipStream_t st = resolve_stream(nullptr);
    CSH_HIP(hipDeviceSynchronize());  // hipMemcpy semantics: everything queued on the device before this call is visible to the copy
    HostXfer x;
    x.expect_d2h(host_dst, bytes);
    CSH_TRY(x.d2h(host_dst, dev_src, bytes, st));
    return x.finish(st);
  }
  // hipMemcpy orders itself after the NULL stream only, and the lane streams are non-blocking: work the CALLING thread queued with
  // stream = NULL (e.g. csh_util_generate_bases_dev) must have landed before the copy reads it (found in round 6: the 64-step BLS12-377
  // G2 generator kernel was still running when a 768-byte copy read its output)
  CSH_HIP(hipStreamSynchronize(resolve_stream(nullptr)));
  CSH_HIP(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
  return CSH_OK;
}
// device-to-device copy between (possibly different) GPUs; stream = NULL: synchronous, otherwise asynchronous on a stream of the
// CALLING thread's device
int csh_memcpy_peer(void* dst, int dst_device, const void* src, int src_device, size_t bytes, void* stream) {
  CSH_REQUIRE((dst && src) || bytes == 0, "NULL argument");
  CSH_TRY(ensure_device());
  if (bytes == 0) return CSH_OK;
  // Always on a stream of the calling thread (its lane stream when none is given): a device-to-device hipMemcpy[Peer] runs on the
  // NULL stream and may return before the copy has finished, and the lane streams are non-blocking -- an MSM launched right after
  // would race it (seen as wrong Rep3 proofs with three parties copying at once).
  hipStream_t st = resolve_stream(stream);
  CSH_HIP(hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, st));
  if (!stream) CSH_HIP(hipStreamSynchronize(st));
  return CSH_OK;
}
int csh_extract_component_dev(const uint64_t* shares_dev, uint32_t ncomp, uint32_t comp, size_t n, uint64_t* out_dev, void* stream) {
  CSH_REQUIRE(shares_dev && out_dev, "NULL argument");
  CSH_REQUIRE(ncomp >= 1 && ncomp <= 4 && comp < ncomp, "bad component selector");
  CSH_TRY(ensure_device());
  if (n == 0) return CSH_OK;
  CSH_HIP(hipMemcpy2DAsync(out_dev, 32, reinterpret_cast<const char*>(shares_dev) + 32 * (size_t)comp, 32 * (size_t)ncomp, 32, n,
                           hipMemcpyDeviceToDevice, resolve_stream(stream)));
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
