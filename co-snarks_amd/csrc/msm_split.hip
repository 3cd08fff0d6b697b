// One MSM split over the GPUs of a node (BASELINE config 5), entirely behind the C ABI.
//
// The points are cut into contiguous ranges, one per GPU (bases pre-sharded at csh_bases_upload). Every GPU runs the
// full Pippenger pipeline on its range down to W window sums (msm_partial_t: header + W XYZZ points, a few KiB). The one
// exchange step moves those partial buffers -- RCCL ncclAllGather over xGMI, or hipMemcpyPeer to a root device, or a plain
// device-to-host copy per GPU -- and the host folds them (fold_partials_t: window-wise sums, one Horner pass, one
// inversion). RCCL has no elliptic-curve reduction operator, so an all-reduce is not applicable; the payload is
// latency-bound (<= 48 KiB per rank), not link-bound.
//
// Two host shapes, both without torch:
//   * one process (or thread) per GPU:   csh_comm_unique_id -> csh_comm_init_rank -> csh_msm_split_rank_dev
//   * one thread driving every GPU:      csh_msm_split (modes PEER / HOST; RCCL with comms from csh_comm_init_all)
// RCCL is bound with dlopen/dlsym (librccl.so.1; the copy already loaded by a host such as torch is reused), so the
// library has no link-time dependency on it and hosts that never split an MSM never load it.
//
// Reference seam: the five MSM closures of co-circom/co-groth16/src/groth16.rs:227-294 (msm_public_points_hs ->
// msm_unchecked); a Rust host calls these entry points from the same place it calls csh_msm_dev.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "msm_impl.hpp"

namespace csh {

CSH_MSM_INSTANTIATE(extern, Bn254G1Cfg)
CSH_MSM_INSTANTIATE(extern, Bn254G2Cfg)
CSH_MSM_INSTANTIATE(extern, Bls381G1Cfg)
CSH_MSM_INSTANTIATE(extern, Bls381G2Cfg)
CSH_MSM_INSTANTIATE(extern, Bls377G1Cfg)
CSH_MSM_INSTANTIATE(extern, Bls377G2Cfg)
CSH_MSM_INSTANTIATE(extern, GrumpkinG1Cfg)

// ---- RCCL binding (restated from the public nccl.h ABI: opaque communicator, 128-byte unique id by value) --------------
struct NcclId {
  char internal[128];
};
static_assert(sizeof(NcclId) == CSH_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
constexpr int NCCL_UINT8 = 1;  // ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
  // optional (RCCL >= 2.14 semantics): non-blocking communicator construction
  int (*CommInitRankConfig)(void**, int, NcclId, int, void*) = nullptr;
  int (*CommGetAsyncError)(void*, int*) = nullptr;
  int (*CommAbort)(void*) = nullptr;
  int (*GetVersion)(int*) = nullptr;
  int (*CommInitAll)(void**, int, const int*) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string why;
};

static Rccl* rccl() {
  static Rccl* r = [] {
    Rccl* x = new Rccl();
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      x->lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (x->lib) break;
    }
    if (!x->lib) {
      const char* e = dlerror();
      x->why = std::string("dlopen(librccl.so.1) failed: ") + (e ? e : "?");
      return x;
    }
    bool ok = true;
    auto sym = [&](const char* name) {
      void* p = dlsym(x->lib, name);
      if (!p) {
        ok = false;
        x->why = std::string("librccl lacks ") + name;
      }
      return p;
    };
    x->GetUniqueId = reinterpret_cast<decltype(x->GetUniqueId)>(sym("ncclGetUniqueId"));
    x->CommInitRank = reinterpret_cast<decltype(x->CommInitRank)>(sym("ncclCommInitRank"));
    x->CommInitAll = reinterpret_cast<decltype(x->CommInitAll)>(sym("ncclCommInitAll"));
    x->CommDestroy = reinterpret_cast<decltype(x->CommDestroy)>(sym("ncclCommDestroy"));
    x->AllGather = reinterpret_cast<decltype(x->AllGather)>(sym("ncclAllGather"));
    x->GroupStart = reinterpret_cast<decltype(x->GroupStart)>(sym("ncclGroupStart"));
    x->GroupEnd = reinterpret_cast<decltype(x->GroupEnd)>(sym("ncclGroupEnd"));
    x->GetErrorString = reinterpret_cast<decltype(x->GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) {
      dlclose(x->lib);
      x->lib = nullptr;
      return x;
    }
    x->CommInitRankConfig = reinterpret_cast<decltype(x->CommInitRankConfig)>(dlsym(x->lib, "ncclCommInitRankConfig"));
    x->CommGetAsyncError = reinterpret_cast<decltype(x->CommGetAsyncError)>(dlsym(x->lib, "ncclCommGetAsyncError"));
    x->CommAbort = reinterpret_cast<decltype(x->CommAbort)>(dlsym(x->lib, "ncclCommAbort"));
    x->GetVersion = reinterpret_cast<decltype(x->GetVersion)>(dlsym(x->lib, "ncclGetVersion"));
    return x;
  }();
  return r;
}

#define CSH_RCCL_LOADED(R)                                   \
  do {                                                       \
    if (!(R)->lib) {                                         \
      set_error("RCCL unavailable: %s", (R)->why.c_str());   \
      return CSH_ERR_HIP;                                    \
    }                                                        \
  } while (0)
#define CSH_NCCL(R, call)                                                                                 \
  do {                                                                                                    \
    const int rc__ = (call);                                                                              \
    if (rc__ != 0) {                                                                                      \
      set_error("%s failed: %s (%s:%d)", #call, (R)->GetErrorString(rc__), __FILE__, __LINE__);           \
      return CSH_ERR_HIP;                                                                                 \
    }                                                                                                     \
  } while (0)

// the largest partial buffer of any group (BLS12-381 G2): communicator buffers are sized once for it
constexpr size_t MAX_PARTIAL_BYTES = sizeof(PartialHeader) + 2 * 192 * (size_t)MAX_WINDOWS;

// ncclConfig_t as the public header declares it (rccl.h, ncclConfig_v22700; fields are only ever appended and the library reads `size`
// bytes, padding the rest with its defaults): used for ONE attribute, blocking = 0.
struct NcclConfig {
  size_t size;
  unsigned int magic, version;
  int blocking, cgaClusterSize, minCTAs, maxCTAs;
  const char* netName;
  int splitShare, trafficClass;
  const char* commName;
  int collnetEnable, CTAPolicy, shrinkShare, nvlsCTAs;
};
constexpr int NCCL_IN_PROGRESS = 7;          // ncclResult_t ncclInProgress
constexpr int NCCL_UNDEF_INT = INT32_MIN;    // NCCL_CONFIG_UNDEF_INT

struct Comm {
  bool nonblocking = false;  // built with blocking = 0: every RCCL call on it may return ncclInProgress and is then polled
  void* nccl = nullptr;  // ncclComm_t; nullptr for a 1-rank communicator made without RCCL
  int rank = 0, nranks = 1, device = 0;
  char* part_dev = nullptr;    // this rank's partial
  char* gather_dev = nullptr;  // nranks partials
  char* gather_host = nullptr; // pinned
};

static int comm_alloc(Comm* c) {
  CSH_HIP(hipMalloc(reinterpret_cast<void**>(&c->part_dev), MAX_PARTIAL_BYTES));
  CSH_HIP(hipMalloc(reinterpret_cast<void**>(&c->gather_dev), MAX_PARTIAL_BYTES * (size_t)c->nranks));
  CSH_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->gather_host), MAX_PARTIAL_BYTES * (size_t)c->nranks, hipHostMallocDefault));
  return CSH_OK;
}
static void comm_release(Comm* c) {
  if (c->part_dev) (void)hipFree(c->part_dev);
  if (c->gather_dev) (void)hipFree(c->gather_dev);
  if (c->gather_host) (void)hipHostFree(c->gather_host);
  delete c;
}

// Poll a non-blocking communicator until its pending operation has finished, failed or `timeout_ms` has passed (<= 0: no deadline).
// -> the final ncclResult_t, or ncclInProgress on a timeout.
static int comm_wait(Rccl* R, void* nccl, long timeout_ms) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    int st = 0;
    const int rc = R->CommGetAsyncError(nccl, &st);
    if (rc != 0) return rc;
    if (st != NCCL_IN_PROGRESS) return st;
    if (timeout_ms > 0 && std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms) return NCCL_IN_PROGRESS;
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
}
// ncclAllGather on a communicator of either kind
static int comm_all_gather(Rccl* R, Comm* c, const void* send, void* recv, size_t bytes, hipStream_t st) {
  int rc = R->AllGather(send, recv, bytes, NCCL_UINT8, c->nccl, st);
  if (rc == NCCL_IN_PROGRESS && c->nonblocking) rc = comm_wait(R, c->nccl, tune().comm_timeout_ms.load(std::memory_order_relaxed));
  return rc;
}

static int split_args(const Bases* B, size_t offset, size_t n, const void* scalars) {
  CSH_REQUIRE(B, "bases is NULL");
  CSH_REQUIRE(offset <= B->n && n <= B->n - offset, "offset + n exceeds the uploaded bases");
  CSH_REQUIRE(scalars || n == 0, "scalars is NULL");
  return CSH_OK;
}

static int partial_async(const Bases* B, size_t offset, size_t n, const uint64_t* scalars_dev, int mont, void* out_dev, hipStream_t st) {
  CURVE_DISPATCH(B->curve, B->group, (msm_partial_t<Cfg>(B, offset, n, scalars_dev, mont, out_dev, st, false)));
}
static int fold(csh_curve_t curve, csh_group_t group, const void* partials_host, size_t nparts, void* out_jacobian) {
  CURVE_DISPATCH(curve, group, (fold_partials_t<Cfg>(partials_host, nparts, out_jacobian)));
}

}  // namespace csh

using namespace csh;

extern "C" {

int csh_comm_unique_id(uint8_t id[CSH_COMM_ID_BYTES]) {
  CSH_REQUIRE(id, "id is NULL");
  Rccl* R = rccl();
  CSH_RCCL_LOADED(R);
  NcclId nid;
  CSH_NCCL(R, R->GetUniqueId(&nid));
  memcpy(id, &nid, sizeof nid);
  return CSH_OK;
}

int csh_comm_init_rank(const uint8_t id[CSH_COMM_ID_BYTES], int nranks, int rank, csh_comm_t* out) {
  CSH_REQUIRE(out, "out is NULL");
  CSH_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank / nranks");
  CSH_REQUIRE(id || nranks == 1, "id is NULL");
  CSH_TRY(ensure_device());
  Comm* c = new Comm();
  c->rank = rank;
  c->nranks = nranks;
  if (hipGetDevice(&c->device) != hipSuccess) c->device = 0;
  if (nranks > 1 || id) {  // a 1-rank communicator with id == NULL needs no RCCL at all
    Rccl* R = rccl();
    if (!R->lib) {
      delete c;
      set_error("RCCL unavailable: %s", R->why.c_str());
      return CSH_ERR_HIP;
    }
    NcclId nid;
    memcpy(&nid, id, sizeof nid);
    // ncclCommInitRank is collective and cannot be interrupted: a rank whose peers never arrive (or a fabric bootstrap that wedges)
    // would hang its host for good. The construction therefore runs on a helper thread and the caller waits for it against a deadline
    // -- tune "comm_timeout_ms", default 120 s, 0 = call it on this thread as before. Past the deadline the caller gets an error it
    // can act on (fall back to its own exchange) and the helper is left behind: if the bootstrap ever completes it tears the
    // communicator down itself. (RCCL's own non-blocking construction -- ncclCommInitRankConfig with blocking = 0 polled through
    // ncclCommGetAsyncError -- buys nothing on this stack (RCCL 2.27.7, ROCm 7.2): that call itself does not return while a peer is
    // missing, with or without ncclCommAbort, profiles/archive/r04_s_rccl_deadline.log. The helper therefore makes the plain blocking call,
    // the path every RCCL user exercises; tune "comm_nonblocking" = 1 selects the other one. One rank cannot wait for anybody and is
    // built inline.)
    const long timeout_ms = tune().comm_timeout_ms.load(std::memory_order_relaxed);
    struct Job {
      std::mutex mu;
      std::condition_variable cv;
      bool done = false, abandoned = false;
      int rc = 0;
      void* nccl = nullptr;
      bool nonblocking = false;
    };
    auto build = [R, nid, nranks, rank](void** out_comm, bool* nonblocking) -> int {
      int ver = 0;
      if (tune().comm_nonblocking.load(std::memory_order_relaxed) != 0 && R->CommInitRankConfig && R->CommGetAsyncError && R->GetVersion && R->GetVersion(&ver) == 0) {
        NcclConfig cfg;
        cfg.size = sizeof(NcclConfig);
        cfg.magic = 0xcafebeef;
        cfg.version = (unsigned)ver;
        cfg.blocking = 0;
        cfg.cgaClusterSize = cfg.minCTAs = cfg.maxCTAs = NCCL_UNDEF_INT;
        cfg.netName = nullptr;
        cfg.splitShare = cfg.trafficClass = NCCL_UNDEF_INT;
        cfg.commName = nullptr;
        cfg.collnetEnable = cfg.CTAPolicy = cfg.shrinkShare = cfg.nvlsCTAs = NCCL_UNDEF_INT;
        int rc = R->CommInitRankConfig(out_comm, nranks, nid, rank, &cfg);
        // (the deadline again: on the timeout_ms <= 0 path nobody else watches this poll)
        if (rc == NCCL_IN_PROGRESS || (rc == 0 && *out_comm)) rc = comm_wait(R, *out_comm, tune().comm_timeout_ms.load(std::memory_order_relaxed));
        *nonblocking = rc == 0;
        return rc;
      }
      *nonblocking = false;
      return R->CommInitRank(out_comm, nranks, nid, rank);
    };
    int rc = 0;
    if (timeout_ms <= 0 || nranks == 1) {
      rc = build(&c->nccl, &c->nonblocking);
    } else {
      auto job = std::make_shared<Job>();
      const int device = c->device;
      std::thread([job, build, device, R] {
        void* comm = nullptr;
        bool nb = false;
        int r = hipSetDevice(device) == hipSuccess ? build(&comm, &nb) : 1;
        std::unique_lock<std::mutex> lk(job->mu);
        if (job->abandoned) {  // nobody is waiting any more: the communicator is ours to take down
          lk.unlock();
          if (r == 0 && comm) (void)R->CommDestroy(comm);
          return;
        }
        job->rc = r, job->nccl = comm, job->nonblocking = nb, job->done = true;
        job->cv.notify_all();
      }).detach();
      std::unique_lock<std::mutex> lk(job->mu);
      if (!job->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return job->done; })) {
        job->abandoned = true;
        lk.unlock();
        delete c;
        set_error("csh_comm_init_rank(rank %d of %d): the communicator did not come up within %ld ms (csh_tune_set(\"comm_timeout_ms\", ..)); "
                  "the construction was abandoned", rank, nranks, timeout_ms);
        return CSH_ERR_HIP;
      }
      rc = job->rc;
      c->nccl = job->nccl;
      c->nonblocking = job->nonblocking;
    }
    if (rc != 0) {
      delete c;
      set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, R->GetErrorString(rc));
      return CSH_ERR_HIP;
    }
  }
  const int rc = comm_alloc(c);
  if (rc != CSH_OK) {
    if (c->nccl) (void)rccl()->CommDestroy(c->nccl);
    comm_release(c);
    return rc;
  }
  *out = reinterpret_cast<csh_comm_t>(c);
  return CSH_OK;
}

int csh_comm_init_all(const int* devices, int ndev, csh_comm_t* out) {
  CSH_REQUIRE(devices && out && ndev >= 1 && ndev <= 64, "bad arguments");
  for (int i = 0; i < ndev; ++i)
    for (int j = 0; j < i; ++j) CSH_REQUIRE(devices[i] != devices[j], "csh_comm_init_all: a device appears twice");
  int saved = 0;
  CSH_TRY(csh_current_device(&saved));
  Rccl* R = rccl();
  CSH_RCCL_LOADED(R);
  std::vector<void*> nccl(ndev, nullptr);
  CSH_NCCL(R, R->CommInitAll(nccl.data(), ndev, devices));
  int rc = CSH_OK;
  std::vector<Comm*> made;
  for (int i = 0; i < ndev && rc == CSH_OK; ++i) {
    Comm* c = new Comm();
    c->nccl = nccl[i];
    c->rank = i;
    c->nranks = ndev;
    c->device = devices[i];
    made.push_back(c);
    rc = csh_init(devices[i]);
    if (rc == CSH_OK) rc = comm_alloc(c);
  }
  (void)csh_init(saved);
  if (rc != CSH_OK) {
    for (int i = 0; i < ndev; ++i) (void)R->CommDestroy(nccl[i]);
    for (Comm* c : made) comm_release(c);
    return rc;
  }
  for (int i = 0; i < ndev; ++i) out[i] = reinterpret_cast<csh_comm_t>(made[i]);
  return CSH_OK;
}

int csh_comm_info(csh_comm_t comm, int* rank, int* nranks, int* device) {
  CSH_REQUIRE(comm, "comm is NULL");
  const Comm* c = reinterpret_cast<const Comm*>(comm);
  if (rank) *rank = c->rank;
  if (nranks) *nranks = c->nranks;
  if (device) *device = c->device;
  return CSH_OK;
}

int csh_comm_destroy(csh_comm_t comm) {
  if (!comm) return CSH_OK;
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (c->nccl) (void)rccl()->CommDestroy(c->nccl);
  comm_release(c);
  return CSH_OK;
}

int csh_msm_split_rank_dev(csh_comm_t comm, csh_bases_t bases, size_t offset, size_t n, const uint64_t* scalars_dev, int mont,
                           void* out_jacobian, void* stream) {
  CSH_REQUIRE(comm && out_jacobian, "NULL argument");
  Comm* c = reinterpret_cast<Comm*>(comm);
  const Bases* B = reinterpret_cast<const Bases*>(bases);
  CSH_TRY(split_args(B, offset, n, scalars_dev));
  CSH_TRY(ensure_device());
  int cur = -1;
  CSH_HIP(hipGetDevice(&cur));
  if (cur != c->device || cur != B->device) {
    set_error("split MSM: the calling thread is bound to device %d, the communicator to %d, the bases to %d", cur, c->device, B->device);
    return CSH_ERR_INVALID;
  }
  hipStream_t st = resolve_stream(stream);
  const size_t pb = partial_bytes_of(B->curve, B->group);
  // A rank whose own range fails (out of memory in the sort arena, ...) must still enter the collective, or its peers wait in
  // ncclAllGather forever: it contributes a zeroed record (no magic), every rank's fold then refuses the gathered set, and this
  // rank reports its original error.
  const int rc_local = partial_async(B, offset, n, scalars_dev, mont, c->part_dev, st);
  std::string local_err;
  if (rc_local != CSH_OK) {
    local_err = csh_last_error();
    (void)hipMemsetAsync(c->part_dev, 0, pb, st);
  }
  const char* src = c->part_dev;
  if (c->nranks > 1) {
    Rccl* R = rccl();
    CSH_NCCL(R, comm_all_gather(R, c, c->part_dev, c->gather_dev, pb, st));
    src = c->gather_dev;
  }
  CSH_HIP(hipMemcpyAsync(c->gather_host, src, pb * (size_t)c->nranks, hipMemcpyDeviceToHost, st));
  CSH_HIP(hipStreamSynchronize(st));
  if (rc_local != CSH_OK) {
    set_error("%s", local_err.c_str());
    return rc_local;
  }
  return fold(B->curve, B->group, c->gather_host, (size_t)c->nranks, out_jacobian);
}

int csh_msm_split(const csh_bases_t* bases, const size_t* offsets, const size_t* counts, const uint64_t* const* scalars_dev, size_t k,
                  int mont, int mode, const csh_comm_t* comms, void* out_jacobian) {
  CSH_REQUIRE(bases && offsets && counts && scalars_dev && out_jacobian, "NULL argument");
  CSH_REQUIRE(k >= 1 && k <= 64, "split MSM: 1..64 parts");
  CSH_REQUIRE(mode == CSH_SPLIT_PEER || mode == CSH_SPLIT_HOST || mode == CSH_SPLIT_RCCL, "split MSM: unknown exchange mode");
  const Bases* B0 = reinterpret_cast<const Bases*>(bases[0]);
  CSH_REQUIRE(B0, "bases[0] is NULL");
  for (size_t i = 0; i < k; ++i) {
    const Bases* B = reinterpret_cast<const Bases*>(bases[i]);
    CSH_TRY(split_args(B, offsets[i], counts[i], scalars_dev[i]));
    CSH_REQUIRE(B->curve == B0->curve && B->group == B0->group, "split MSM: all parts must be of one curve and group");
  }
  if (mode == CSH_SPLIT_RCCL) {
    CSH_REQUIRE(comms, "split MSM: RCCL mode needs the communicators of csh_comm_init_all");
    for (size_t i = 0; i < k; ++i) {
      const Comm* c = reinterpret_cast<const Comm*>(comms[i]);
      CSH_REQUIRE(c && c->nranks == (int)k && c->rank == (int)i, "split MSM: communicator i must be rank i of k");
      CSH_REQUIRE(c->device == reinterpret_cast<const Bases*>(bases[i])->device, "split MSM: communicator and bases of part i live on different devices");
    }
  }
  int saved = 0;
  CSH_TRY(csh_current_device(&saved));
  const size_t pb = partial_bytes_of(B0->curve, B0->group);
  struct Part {
    int dev;
    hipStream_t st;
    char* buf;
  };
  std::vector<Part> parts(k);
  // partial buffers: one region per device (parts sharing a device run back to back on its stream)
  std::vector<int> devs;
  for (size_t i = 0; i < k; ++i) {
    parts[i].dev = reinterpret_cast<const Bases*>(bases[i])->device;
    if (std::find(devs.begin(), devs.end(), parts[i].dev) == devs.end()) devs.push_back(parts[i].dev);
  }
  int rc = CSH_OK;
  std::vector<hipEvent_t> peer_events;
  auto body = [&]() -> int {
    for (int d : devs) {
      CSH_TRY(csh_init(d));
      hipStream_t st = resolve_stream(nullptr);
      size_t cnt = 0;
      for (size_t i = 0; i < k; ++i) cnt += parts[i].dev == d;
      Arena& pa = arena_for((hipStream_t)((uintptr_t)st ^ 0x8));
      CSH_TRY(pa.reserve(Arena::padded(pb) * (cnt + (d == devs[0] ? k : 0))));  // + the gather region on the root device
      for (size_t i = 0; i < k; ++i)
        if (parts[i].dev == d) {
          parts[i].st = st;
          parts[i].buf = mode == CSH_SPLIT_RCCL ? reinterpret_cast<Comm*>(comms[i])->part_dev : pa.take<char>(pb);
        }
    }
    // 1. every range down to its window sums, asynchronously on its device's stream
    for (size_t i = 0; i < k; ++i) {
      CSH_TRY(csh_init(parts[i].dev));
      CSH_TRY(partial_async(reinterpret_cast<const Bases*>(bases[i]), offsets[i], counts[i], scalars_dev[i], mont, parts[i].buf, parts[i].st));
    }
    // 2. the exchange
    std::vector<char> host_plain;
    const char* host = nullptr;
    if (mode == CSH_SPLIT_HOST) {
      host_plain.resize(pb * k);
      for (size_t i = 0; i < k; ++i) {
        CSH_TRY(csh_init(parts[i].dev));
        CSH_HIP(hipMemcpyAsync(host_plain.data() + pb * i, parts[i].buf, pb, hipMemcpyDeviceToHost, parts[i].st));
      }
      for (int d : devs) {
        CSH_TRY(csh_init(d));
        CSH_HIP(hipStreamSynchronize(resolve_stream(nullptr)));
      }
      host = host_plain.data();
    } else if (mode == CSH_SPLIT_PEER) {
      const int root = devs[0];
      std::vector<hipEvent_t>& evs = peer_events;  // destroyed by the caller on every path (error returns included)
      for (size_t i = 0; i < k; ++i) {
        if (parts[i].dev == root) continue;
        CSH_TRY(csh_init(parts[i].dev));
        hipEvent_t e;
        CSH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        evs.push_back(e);
        CSH_HIP(hipEventRecord(e, parts[i].st));
      }
      CSH_TRY(csh_init(root));
      hipStream_t rs = resolve_stream(nullptr);
      Arena& pa = arena_for((hipStream_t)((uintptr_t)rs ^ 0x8));
      char* gather = pa.take<char>(pb * k);
      size_t ei = 0;
      for (size_t i = 0; i < k; ++i) {
        if (parts[i].dev == root) {
          CSH_HIP(hipMemcpyAsync(gather + pb * i, parts[i].buf, pb, hipMemcpyDeviceToDevice, rs));
        } else {
          CSH_HIP(hipStreamWaitEvent(rs, evs[ei++], 0));
          CSH_HIP(hipMemcpyPeerAsync(gather + pb * i, root, parts[i].buf, parts[i].dev, pb, rs));
        }
      }
      host_plain.resize(pb * k);
      CSH_HIP(hipMemcpyAsync(host_plain.data(), gather, pb * k, hipMemcpyDeviceToHost, rs));
      CSH_HIP(hipStreamSynchronize(rs));
      host = host_plain.data();
    } else {
      Rccl* R = rccl();
      CSH_RCCL_LOADED(R);
      if (k > 1) {
        CSH_NCCL(R, R->GroupStart());
        for (size_t i = 0; i < k; ++i) {
          Comm* c = reinterpret_cast<Comm*>(comms[i]);
          // inside a group nothing progresses before GroupEnd: no polling here, "in progress" from a non-blocking communicator is fine
          int rcg = R->AllGather(c->part_dev, c->gather_dev, pb, NCCL_UINT8, c->nccl, parts[i].st);
          if (rcg == NCCL_IN_PROGRESS && c->nonblocking) rcg = 0;
          if (rcg != 0) {
            (void)R->GroupEnd();
            set_error("ncclAllGather(part %zu) failed: %s", i, R->GetErrorString(rcg));
            return CSH_ERR_HIP;
          }
        }
        {
          // communicators built non-blocking (tune comm_nonblocking): inside a group the AllGather calls return at once and GroupEnd
          // itself reports ncclInProgress -- not a failure: every communicator is polled to completion against the deadline
          const int rce = R->GroupEnd();
          bool any_nb = false;
          for (size_t i = 0; i < k; ++i) any_nb = any_nb || reinterpret_cast<Comm*>(comms[i])->nonblocking;
          if (rce == NCCL_IN_PROGRESS && any_nb) {
            for (size_t i = 0; i < k; ++i) {
              Comm* c = reinterpret_cast<Comm*>(comms[i]);
              if (!c->nonblocking) continue;
              const int rw = comm_wait(R, c->nccl, tune().comm_timeout_ms.load(std::memory_order_relaxed));
              if (rw != 0) {
                set_error("grouped ncclAllGather (non-blocking communicator %zu) did not complete: %s", i, R->GetErrorString(rw));
                return CSH_ERR_HIP;
              }
            }
          } else {
            CSH_NCCL(R, rce);
          }
        }
      }
      Comm* c0 = reinterpret_cast<Comm*>(comms[0]);
      CSH_TRY(csh_init(parts[0].dev));
      CSH_HIP(hipMemcpyAsync(c0->gather_host, k > 1 ? c0->gather_dev : c0->part_dev, pb * k, hipMemcpyDeviceToHost, parts[0].st));
      for (int d : devs) {  // every rank's collective must have drained before its buffers are reused
        CSH_TRY(csh_init(d));
        CSH_HIP(hipStreamSynchronize(resolve_stream(nullptr)));
      }
      host = c0->gather_host;
    }
    // 3. host fold: window-wise sums, one Horner pass, one inversion
    return fold(B0->curve, B0->group, host, k, out_jacobian);
  };
  rc = body();
  std::string keep = rc != CSH_OK ? std::string(csh_last_error()) : std::string();
  if (rc != CSH_OK) {
    // An error return must not leave kernels or copies of the earlier parts in flight: they write arena and communicator
    // buffers the next call reuses. Drain the lane stream of every device this call touched before handing control back.
    for (int d : devs)
      if (csh_init(d) == CSH_OK) (void)hipStreamSynchronize(resolve_stream(nullptr));
  }
  for (hipEvent_t e : peer_events) (void)hipEventDestroy(e);
  (void)csh_init(saved);
  if (rc != CSH_OK) set_error("%s", keep.c_str());
  return rc;
}

}  // extern "C"
