// Library-internal plumbing: error state, per-thread device/stream, stream-ordered scratch arenas.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cosnarks_hip.h"

namespace csh {

void set_error(const char* fmt, ...);
int ensure_device();                    // lazy csh_init(0) for the calling thread; returns csh_status
int device_simds();                     // SIMDs of the calling thread's device (4 per CU; 1024 on MI355X), cached per device
hipStream_t resolve_stream(void* s);    // NULL -> the calling thread's stream on its current device

#define CSH_HIP(call)                                                                 \
  do {                                                                                \
    hipError_t e__ = (call);                                                          \
    if (e__ != hipSuccess) {                                                          \
      csh::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      return e__ == hipErrorOutOfMemory ? CSH_ERR_OOM : CSH_ERR_HIP;                  \
    }                                                                                 \
  } while (0)

#define CSH_TRY(expr)            \
  do {                           \
    int rc__ = (expr);           \
    if (rc__ != CSH_OK) return rc__; \
  } while (0)

#define CSH_REQUIRE(cond, msg)      \
  do {                              \
    if (!(cond)) {                  \
      csh::set_error("%s", msg);    \
      return CSH_ERR_INVALID;       \
    }                               \
  } while (0)

// Scratch arena bound to (thread, stream): reuse across calls on the same stream is safe by stream
// ordering; growing frees the old block (hipFree synchronises).
struct Arena {
  char* base = nullptr;
  size_t cap = 0;
  size_t off = 0;
  int reserve(size_t bytes);  // ensure capacity, reset offset
  template <class T>
  T* take(size_t count) {
    size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
    T* p = reinterpret_cast<T*>(base + off);
    off += bytes;
    return p;
  }
  static size_t padded(size_t bytes) { return (bytes + 255) & ~size_t(255); }
};
Arena& arena_for(hipStream_t s);
hipStream_t resolve_aux_stream();  // second pooled stream of the calling thread (may be nullptr)
bool pinned_for(hipStream_t s, size_t bytes, void** host, void** dev);  // page-locked result buffer of (thread lane, stream)

// Upload of a caller's (pageable) host buffer. Round 5 (profiles/archive/r05_l .. r05_p_trait_stall_*): on some boxes, from the second circuit of a
// process on, a hipMemcpyAsync FROM caller memory of 32 MB takes 10-20 ms instead of 0.6-1.3 (steps of ~10 ms: the driver's pinning of the
// source pages retrying with a one-jiffy sleep) -- that, not the result copy, is the "stalled witness map" of round 4. tune "host_h2d":
// 0 = one copy from the caller's pages, 1 = staged (host threads copy 2 MiB chunks into the lane's page-locked buffer, each chunk's DMA
// follows at once: the driver never pins caller memory; +0.3-0.5 ms per 32 MB on a healthy box), 2 = direct and timed; two in a row
// on one lane that took more than three times their PCIe time + 4 ms switch the whole process to staged transfers in BOTH directions for a
// spell. DEFAULT = 1, and 1 for results too (host_d2h): the stalled state is sticky -- a process that has entered it keeps stalling after the
// switch (profiles/archive/r05_s_trait_stall_auto_mode.log: 23-33 ms per witness map with every transfer staged by then), whereas a process whose
// transfers were staged from its first call never entered it in 32 of 32 probe runs (r05_r, r05_t). The price is ~10 % of a trait-path
// prove at 2^20 on a box that would have stayed healthy (18.3 against 16.6 ms); 0 / 2 remain for hosts known to be fine. `slot` (0..3) names one of the lane's
// page-locked staging buffers: uploads of one call that are in flight together use different slots. Returns once `host` has been read.
int upload_h2d(void* dev, const void* host, size_t bytes, hipStream_t st, int slot);

// Host-pointer convenience path: stage inputs into a per-thread arena, run on the thread's stream, copy back.
struct HostStage {
  hipStream_t st = nullptr;
  Arena* ar = nullptr;
  int slot = 0;
  int begin(size_t total_bytes) {
    CSH_TRY(ensure_device());
    st = resolve_stream(nullptr);
    ar = &arena_for((hipStream_t)((uintptr_t)st ^ 0x1));  // distinct arena from kernel-internal scratch
    return ar->reserve(total_bytes);
  }
  template <class T>
  int up(T*& dev, const void* host, size_t bytes) {
    dev = reinterpret_cast<T*>(ar->take<char>(bytes));
    if (host && bytes) CSH_TRY(upload_h2d(dev, host, bytes, st, slot++ & 3));
    return CSH_OK;
  }
  int down(void* host, const void* dev, size_t bytes);  // capi.hip: populates a large destination's pages first (see HostXfer)
};

// Host-pointer entry points that hand the caller LARGE buffers back (h of a witness map, a transformed vector). Measured on the MI355X
// boxes (profiles/archive/r04_a_pcie_probe.jsonl, r04_b_prefault_probe.jsonl): a copy from / to pageable memory whose pages are present runs
// at the PCIe rate (32 MB in 0.60 ms = 56 GB/s, the same as from hipHostMalloc memory; hipHostRegister is a ~1 us no-op on these hosts),
// but a D2H into memory the caller has only just allocated pays the DMA engine's first touch of every page: 3.9 ms for 32 MB.
// HostXfer therefore populates the destination's pages from a helper thread (MADV_POPULATE_WRITE: no content change) WHILE the
// device works, and joins it before the copy is enqueued. Measured (profiles/archive/r04_y_populate.log, six alternating trials per setting,
// host-facing witness map at 2^20): one thread with a transparent-huge-page hint on the range 2.36-2.47 ms (= upload + device + copy:
// nothing left exposed); two / four threads with the hint 3.1-3.8 / 2.9-3.4 (they contend); WITHOUT the hint 4.3-22 ms, bimodal
// (8192 4-KiB faults interleaved with the driver's own page handling) -- the hint is what makes it reliable.
void host_populate_begin(void* p, size_t bytes, std::vector<std::thread>& workers);  // no-op below 4 MiB or with tune host_populate = 0
struct HostXfer {
  std::vector<std::thread> workers;
  // a large result staged through the lane's page-locked buffer (tune "host_d2h" = 1): chunked copies into it, an event per chunk,
  // host threads copy each chunk on into the caller's memory as it lands (finish())
  struct Staged {
    void* host = nullptr;
    const char* pinned = nullptr;
    size_t bytes = 0, chunk = 0;
    std::vector<hipEvent_t> landed;
    const void* ring_dev = nullptr;  // non-null: larger than the staging ring, moved through it in finish() (capi.hip)
  };
  std::vector<Staged> staged;
  ~HostXfer();
  void join();  // capi.hip (counts the time the caller waited: stat_join_wait_us)
  // call as early as the destination is known: the pages are populated while uploads and kernels run
  void expect_d2h(void* host, size_t bytes) { host_populate_begin(host, bytes, workers); }
  int h2d_slot = 0;
  int h2d(void* dev, const void* host, size_t bytes, hipStream_t st) {
    if (bytes) CSH_TRY(upload_h2d(dev, host, bytes, st, h2d_slot++ & 3));
    return CSH_OK;
  }
  int d2h(void* host, const void* dev, size_t bytes, hipStream_t st);  // capi.hip
  int finish(hipStream_t st);                                          // capi.hip
};

// ntt.hip internals reused by the fused Groth16 pipeline
struct Domain;
int ntt_run(const Domain* d, uint64_t* data, uint32_t ncomp, bool dif, hipStream_t st);
int ntt_coset_table(const Domain* d, const uint64_t* shift, uint64_t* out_dev, hipStream_t st);
bool ntt_scale_table_supported(const Domain* d);
int ntt_run_dif_table(const Domain* d, uint64_t* data, uint32_t ncomp, const uint64_t* scale_table, hipStream_t st);
int ntt_run_pair_table(const Domain* d, uint64_t* data, uint32_t ncomp, const uint64_t* scale_table, hipStream_t st);  // ifft (scaled by the table) + fft, tile passes fused
int ntt_coset_table_scaled(const Domain* d, const uint64_t* shift, uint64_t* out_dev, hipStream_t st);
int ntt_coset_table_scaled_cached(const Domain* d, const uint64_t* shift, uint64_t* scratch, hipStream_t st, const uint64_t** table);
// out = a * b - c (plain / Shamir) and out = rep3_local_mul(a, b) + mask - c: the last step of a witness map in one sweep
int vec_mul_sub_dev(csh_curve_t f, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t* out, size_t n, hipStream_t st);
int rep3_local_mul_sub_dev(csh_curve_t f, const uint64_t* a, const uint64_t* b, const uint64_t* mask, const uint64_t* c, uint64_t* out, size_t n, hipStream_t st);
int ntt_bit_reverse(csh_curve_t c, uint64_t* data, uint32_t log_n, uint32_t ncomp, hipStream_t st);
size_t domain_size_of(const Domain* d);
csh_curve_t domain_curve_of(const Domain* d);

// Process-wide tuning knobs (csh_tune_set / csh_tune_get): read once from the environment at load (CSH_MSM_C, ...),
// changed at run time through the C ABI -- no getenv on any call path.
struct Tune {
  std::atomic<int> msm_c{0};             // forced window width (0 = cost model)
  std::atomic<int> msm_l{0};             // forced entries per accumulate lane (0 = from the launch width)
  std::atomic<int> msm_balanced{1};      // balanced windows (msm_impl.hpp choose_windows): 0 = uniform c-bit windows as in rounds 1-4
  std::atomic<int> msm_w{0};             // balanced windows: forced number of windows (0 = cost model)
  std::atomic<int> msm_timing{0};        // record per-stage HIP events (csh_msm_last_timing)
  std::atomic<int> msm_no_table{0};      // ignore fixed-base tables
  std::atomic<int> stat_arena_grows{0};  // counters (read with csh_tune_get): scratch arenas (re)allocated, lanes (streams) created
  std::atomic<int> stat_lanes{0};
  std::atomic<int> stat_populate_us{0}, stat_join_wait_us{0}, stat_finish_us{0}, stat_d2h_slow{0}, stat_d2h_staged{0};  // page population: worker time, caller's wait for it, final stream wait
  std::atomic<int> msm_multi_overlap{1}; // alternate bucket stages of csh_msm_multi_dev between two streams
  std::atomic<int> msm_share_uploads{2}; // csh_msm: concurrent calls handed the same host scalar slice share one upload (1) and run as ONE multi-MSM of the uploading call (2, default) (msm.hip SharedUpload)
  std::atomic<int> stat_uploads_shared{0};  // counter: csh_msm calls that reused a concurrent call's upload
  std::atomic<int> host_h2d{1};  // uploads of >= 4 MiB from pageable caller memory: 0 direct, 1 (default) staged through page-locked chunks, 2 direct + timed, staged after stalls (upload_h2d)
  std::atomic<int> stat_h2d_slow{0}, stat_h2d_staged{0}, stat_stage_all_switches{0};
  std::atomic<int> stat_pinned_kib{0};  // page-locked host memory currently held by the lanes' staging / result slots, KiB
  std::atomic<int> host_copier_pool{1};  // staged transfers: 1 = persistent copier pool, 0 = a std::thread per helper and copy (A/B)
  std::atomic<int> host_timing{0};  // diagnostics: 1 = the host-facing witness map synchronises after its upload, its kernels and its result copy and adds the three times to the counters below
  std::atomic<int> stat_wm_h2d_us{0}, stat_wm_dev_us{0}, stat_wm_d2h_us{0};
  std::atomic<int> acc_blk{0};           // accumulate workgroup size (0 = default)
  std::atomic<int> sort_two_level{-1};   // -1 auto, 0 / 1 forced
  std::atomic<int> vec_max_blocks{65536};
  std::atomic<int> ntt_lazy{1};
  std::atomic<int> ntt_threads{1024};
  std::atomic<int> msm_variant{0};       // experimental kernel variants (A/B runs)
  std::atomic<int> msm_wide_lb{0};       // wide sort (msm_sort_wide.hip): log2 buckets per level-2 partition, 8 .. 11 (0 = default)
  std::atomic<int> msm_wide_chunks{0};   // wide sort: level-1 blocks aimed at (0 = default 1024)
  std::atomic<int> msm_seg_buckets{0};   // buckets per window-reduction segment (0 = as many segments as fit one round)
  std::atomic<int> allow_unmasked_rep3{0};  // Rep3 products without the re-randomising masks: refused unless set (tests)
  std::atomic<int> ntt_variant{0};
  std::atomic<int> ntt_pair{1};           // witness maps: fuse the last pass of each inverse transform with the first pass of the forward one (0 = two launches, A/B)
  std::atomic<int> ntt_pair_min_log{20};  // ... for domains of at least 2^this many points (measured, Rep3 witness map device time, profiles/r06_n_pair_ab.log:
                                          // 2^20 1.583 -> 1.542 ms, 2^22 6.64 -> 6.49 ms; 2^16 +3.5 %, 2^12 +6 %: below 2^20 the radix-2 passes it replaces are the faster ones)
  std::atomic<int> h_table_cache{1};      // Groth16 h pipeline: keep the scaled coset table with the domain (0 = rebuild it per witness map, A/B)
  std::atomic<int> h_unfused{0};          // Groth16 h pipeline: 1 = the unfused step-by-step sequence (A/B, tests)
  std::atomic<int> host_d2h{1};  // large results to pageable memory: 0 = one copy straight into the caller's pages, 1 (default since round 5) = staged
                                 // through a page-locked buffer + host threads, 2 = direct, timed, staged for a while after a copy that stalled (HostXfer::d2h)
  std::atomic<int> comm_timeout_ms{120000};  // deadline of an RCCL communicator's construction (0 = on the calling thread, no deadline)
  std::atomic<int> comm_nonblocking{0};  // csh_comm_init_rank: 1 = ncclCommInitRankConfig(blocking = 0) + polling instead of ncclCommInitRank
  std::atomic<int> host_populate{0x101};  // low byte: threads populating a large D2H destination's pages before the copy (0 = off); bit 8: huge-page hint
};
Tune& tune();

// A Rep3 local multiplication without its correlated mask is not a valid sharing step: once opened, the products leak
// cross terms (rep3/arithmetic.rs:132-146 always adds masking_field_elements_vec). NULL masks / seeds are therefore an
// error unless the caller has opted in explicitly with csh_tune_set("allow_unmasked_rep3", 1) (unit tests of the arithmetic).
// Round 6: the override only exists in builds with -DCSH_EXPERIMENTS (never the product build of build.py).
inline int require_rep3_masks(bool have_masks, const char* what) {
  if (have_masks) return CSH_OK;
#ifdef CSH_EXPERIMENTS
  if (tune().allow_unmasked_rep3.load(std::memory_order_relaxed)) return CSH_OK;
#endif
  set_error("%s: Rep3 (protocol 1) needs its masks / ChaCha12 seeds; unmasked products leak cross terms when opened", what);
  return CSH_ERR_INVALID;
}

// Knob values that make a kernel skip work (timing experiments: WRONG RESULTS). They exist only in builds with -DCSH_EXPERIMENTS;
// the product library refuses them in csh_tune_set and never reads them from the environment.
constexpr int NTT_VARIANT_EXPERIMENT_BITS = 0x3000 | 0xF0000;  // bits 12-13: run one pass of the plan; bits 16-19: skip butterflies / loads / stores / canonicalisation
#ifdef CSH_EXPERIMENTS
constexpr bool kExperiments = true;
#else
constexpr bool kExperiments = false;
#endif

// hipFuncSetAttribute(.., hipFuncAttributeMaxDynamicSharedMemorySize, bytes) once per (calling thread, device, kernel): the attribute belongs
// to the kernel ON A DEVICE, and one host thread may drive several devices (csh_msm_split drives every GPU of the node from one thread), so
// a per-thread "already raised" flag alone (rounds 1-5) would skip the call on every device after the first. capi.hip.
int raise_lds_limit(const void* kernel, size_t bytes);

inline int grid_for(size_t n, int block, int max_blocks = 256 * 16) {
  size_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > (size_t)max_blocks) g = max_blocks;
  return (int)g;
}

}  // namespace csh
