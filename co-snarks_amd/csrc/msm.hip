// Pippenger bucket MSM on gfx950 (BN254 / BLS12-381 G1 and G2, Grumpkin G1).
//
// Pipeline (all on one HIP stream, no host round-trip until the final W window sums):
//   1. k_msm_digits   scalar -> canonical -> signed c-bit digit codes (u16 per point and window)
//   2. sort stage (msm_sort.hip): LDS histogram per (window, chunk), prefix over chunks, per-window scan (lanes per
//                     window = ceil(entries / L)), counting-sort scatter of (point index | sign) into bucket order --
//                     single level with LDS cursors for small n, two-level LDS tile sort for n >= 2^20; no global atomics
//   3. k_msm_accum    one lane per run of L sorted entries (equal work per lane, whatever the bucket sizes):
//                     gather affine bases, XYZZ mixed additions in registers in the signed lazy field, one partial per
//                     touched bucket, stored as the lazy value
//      k_msm_merge    per-bucket fold of its partials -> dense bucket sums (block-wide tree for giant buckets)
//   4. k_msm_reduce   segments of buckets: running-sum  sum_b b*B_b  per segment
//      k_msm_fold     pairwise tree over the segment results -> one sum per window (all still lazy-field XYZZ),
//      k_msm_gather_windows converts the W window sums back to the arkworks encoding
//   host: Horner over the W window sums (W*c doublings) in 64-bit limbs, one inversion, Jacobian (x, y, 1) out.
//
// Replaces taceo_ark_algebra::msm::{msm_unchecked, msm_bigint} (see include/cosnarks_hip.h for the call
// sites). The result is a group element; it is bit-identical to the reference after affine normalisation.
#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "msm_impl.hpp"

namespace csh {

thread_local float tl_msm_timing[6] = {0, 0, 0, 0, 0, 0};
thread_local uint32_t tl_msm_params[4] = {0, 0, 0, 0};

CSH_MSM_INSTANTIATE(extern, Bn254G1Cfg)
CSH_MSM_INSTANTIATE(extern, Bn254G2Cfg)
CSH_MSM_INSTANTIATE(extern, Bls381G1Cfg)
CSH_MSM_INSTANTIATE(extern, Bls381G2Cfg)
CSH_MSM_INSTANTIATE(extern, GrumpkinG1Cfg)
CSH_MSM_INSTANTIATE(extern, Bls377G1Cfg)
CSH_MSM_INSTANTIATE(extern, Bls377G2Cfg)
// The accumulate kernels live ONLY in msm_accum_*.hip (pinned multiply-add order). This unit names them too (accum_occupancy asks the
// runtime about them), and without these declarations it instantiated its own UNPINNED copies of the G1 kernels, which the runtime then
// launched instead of the pinned ones (two code objects registering one host stub: the copy of this unit won; 146 against 144 VGPRs on
// BN254 G1, 248 against 217 on BLS12-381 G1 -- every G1 accumulate since the occupancy query of round 4 ran 3-4 % slower than the kernel the
// profiles of round 3 describe; found in round 6 when an experimental variant that existed in one unit only beat it by exactly that margin
// (profiles/r06_s_ahead_groups_ab.log; the variant itself -- the gather two entries ahead -- is worth nothing: r06_t_ahead_clean_ab.log).
CSH_MSM_ACCUM_INSTANTIATE(extern, Bn254G1Cfg)
CSH_MSM_ACCUM_INSTANTIATE(extern, Bn254G2Cfg)
CSH_MSM_ACCUM_INSTANTIATE(extern, Bls381G1Cfg)
CSH_MSM_ACCUM_INSTANTIATE(extern, Bls381G2Cfg)
CSH_MSM_ACCUM_INSTANTIATE(extern, GrumpkinG1Cfg)
CSH_MSM_ACCUM_INSTANTIATE(extern, Bls377G1Cfg)
CSH_MSM_ACCUM_INSTANTIATE(extern, Bls377G2Cfg)
CSH_MSM_ACCUM_PAIR_INSTANTIATE(extern, Bn254G2Cfg)
CSH_MSM_ACCUM_PAIR_INSTANTIATE(extern, Bls381G2Cfg)
CSH_MSM_ACCUM_PAIR_INSTANTIATE(extern, Bls377G2Cfg)

static int repack_bases(Bases* B, hipStream_t st);

}  // namespace csh

using namespace csh;

// ---- one upload for concurrent host-scalar calls over the same slice ----------------------------------------------------------------
// The reference issues the five MSMs of a proof concurrently (rayon_join5) and four of them read the same `aux_assignment` slice: behind
// an unchanged reference that is four csh_msm calls with the same host pointer at the same time, i.e. four 32 MB uploads sharing the
// PCIe link before the first kernel of any of them can start (2^20: ~3 ms during which the GPU idles). A call that finds another call
// IN FLIGHT with the same (device, pointer, length) waits for that call's upload (a stream-to-stream event wait) and reads its device
// copy instead. Sound by the ordinary contract of a synchronous entry point -- the caller must not modify an input while a call that
// was handed it is running -- and only by that: nothing is reused once the last call using it has returned, no content is compared, no
// copy outlives its callers. Device buffers come from a small per-device pool (no hipMalloc in the steady state).
namespace {
struct SharedUpload {
  int device = 0;
  const void* host = nullptr;
  size_t bytes = 0, cap = 0;
  void* dev = nullptr;
  hipEvent_t ready = nullptr;
  int refs = 0;
  std::mutex m;  // guards `state` and the batch
  std::condition_variable cv;
  int state = 0;  // 0 = the owner has not recorded `ready` yet, 1 = recorded, -1 = the owner's upload failed
  // Round 6: calls that arrive while the owner is still uploading do not only share the copy -- they hand the owner their (bases,
  // offset, out) and wait: the owner runs ALL of them as one csh_msm_multi_dev (ONE digit sort for the handles that share length and
  // offset, bucket stages alternating between two streams), i.e. the four aux-assignment MSMs of an unchanged reference's rayon_join5
  // take the device-resident prover's path. tune "msm_share_uploads" = 2 (default); 1 = share the upload only.
  struct Req {
    csh_bases_t bases = nullptr;
    size_t offset = 0;
    void* out = nullptr;
    int rc = CSH_OK;
    std::string err;
  };
  std::vector<Req*> batch;  // joiners' requests (the owner's own is not in here)
  bool sealed = false;      // the owner has taken the batch: later calls share the upload only
  bool batch_done = false;  // results (or errors) of the batch are in place
  int curve = -1, mont = -1;  // what the owner's call runs with: only calls of the same curve and scalar encoding join
};
std::mutex g_up_mu;
std::vector<SharedUpload*> g_up_live;                       // entries with refs > 0
std::vector<std::pair<int, std::pair<size_t, void*>>> g_up_pool;  // (device, (capacity, buffer)) free device buffers
constexpr size_t UP_POOL_MAX = 8;
constexpr size_t UP_POOL_MAX_BYTES = size_t(1) << 30;  // the free buffers together (a 2^24 scalar vector is 512 MB): beyond it a buffer is freed

struct SharedUploadRef {
  SharedUpload* e = nullptr;
  hipStream_t stream = nullptr;
  const void* dev() const { return e->dev; }
  bool is_owner = false;
  // req != nullptr: the caller is willing to be run by the owner (see SharedUpload::batch); *joined = true then means req->rc / req->err hold
  // its result when acquire returns and it must not run anything itself
  int acquire(int device, const void* host, size_t bytes, hipStream_t st, SharedUpload::Req* req = nullptr, int curve = -1, int mont = -1, bool* joined = nullptr) {
    stream = st;
    bool owner = false;
    {
      std::lock_guard<std::mutex> g(g_up_mu);
      for (SharedUpload* x : g_up_live)
        if (x->device == device && x->host == host && x->bytes == bytes) {
          bool failed;
          {
            std::lock_guard<std::mutex> gs(x->m);
            failed = x->state < 0;
          }
          if (failed) continue;  // an entry whose owner's upload failed only waits for its last waiter: a NEW call uploads for itself
          e = x;
          break;
        }
      if (e) {
        ++e->refs;
      } else {
        e = new SharedUpload();
        e->device = device, e->host = host, e->bytes = bytes, e->refs = 1;
        e->curve = curve, e->mont = mont;
        size_t best = (size_t)-1;
        for (size_t i = 0; i < g_up_pool.size(); ++i)
          if (g_up_pool[i].first == device && g_up_pool[i].second.first >= bytes && (best == (size_t)-1 || g_up_pool[i].second.first < g_up_pool[best].second.first)) best = i;
        if (best != (size_t)-1) {
          e->cap = g_up_pool[best].second.first;
          e->dev = g_up_pool[best].second.second;
          g_up_pool.erase(g_up_pool.begin() + (ptrdiff_t)best);
        }
        g_up_live.push_back(e);
        owner = true;
      }
    }
    is_owner = owner;
    if (owner) {
      hipError_t rc = hipSuccess;
      if (!e->dev) {
        e->cap = (bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
        rc = hipMalloc(&e->dev, e->cap);
        if (rc != hipSuccess) e->dev = nullptr;
      }
      if (rc == hipSuccess) rc = hipEventCreateWithFlags(&e->ready, hipEventDisableTiming);
      if (rc == hipSuccess && upload_h2d(e->dev, host, bytes, st, 3) != CSH_OK) rc = hipErrorUnknown;
      if (rc == hipSuccess) rc = hipEventRecord(e->ready, st);
      {
        std::lock_guard<std::mutex> g(e->m);
        e->state = rc == hipSuccess ? 1 : -1;
      }
      e->cv.notify_all();
      if (rc != hipSuccess) {
        set_error("csh_msm: uploading the scalars failed: %s", hipGetErrorString(rc));
        return rc == hipErrorOutOfMemory ? CSH_ERR_OOM : CSH_ERR_HIP;
      }
      return CSH_OK;
    }
    {
      std::unique_lock<std::mutex> g(e->m);
      if (req && joined && !e->sealed && e->curve == curve && e->mont == mont && curve >= 0 && e->batch.size() < 15) {
        e->batch.push_back(req);
        *joined = true;
        e->cv.wait(g, [&] { return e->batch_done || e->state < 0; });
        if (!e->batch_done) {  // the owner's upload failed before it took the batch: withdraw the request (it lives on this call's stack)
          e->batch.erase(std::find(e->batch.begin(), e->batch.end(), req));
          set_error("csh_msm: the concurrent call that was uploading this scalar slice failed");
          return CSH_ERR_HIP;
        }
        tune().stat_uploads_shared.fetch_add(1, std::memory_order_relaxed);
        return CSH_OK;
      }
      e->cv.wait(g, [&] { return e->state != 0; });
      if (e->state < 0) {
        set_error("csh_msm: the concurrent call that was uploading this scalar slice failed");
        return CSH_ERR_HIP;
      }
    }
    CSH_HIP(hipStreamWaitEvent(st, e->ready, 0));
    tune().stat_uploads_shared.fetch_add(1, std::memory_order_relaxed);
    return CSH_OK;
  }
  // owner only, after a successful acquire: closes the batch and returns the joiners' requests (empty: nobody joined)
  std::vector<SharedUpload::Req*> seal() {
    std::lock_guard<std::mutex> g(e->m);
    e->sealed = true;
    return e->batch;
  }
  void finish_batch() {
    {
      std::lock_guard<std::mutex> g(e->m);
      e->batch_done = true;
    }
    e->cv.notify_all();
  }
  ~SharedUploadRef() {
    if (!e) return;
    if (is_owner) {  // whatever path the owner left on: joiners must not wait for ever (their requests then carry the error set below)
      bool wake = false;
      {
        std::lock_guard<std::mutex> g(e->m);
        if (!e->batch_done) {
          e->sealed = true;
          for (SharedUpload::Req* r : e->batch)
            if (r->rc == CSH_OK && r->err.empty()) {
              r->rc = CSH_ERR_HIP;
              r->err = "csh_msm: the concurrent call that ran this batch failed before it reached this request";
            }
          e->batch_done = true;
          wake = true;
        }
      }
      if (wake) e->cv.notify_all();
    }
    (void)hipStreamSynchronize(stream);  // a no-op after a successful (synchronous) MSM; after a failed one nothing queued on this stream may still read the copy
    SharedUpload* dead = nullptr;
    {
      std::lock_guard<std::mutex> g(g_up_mu);
      if (--e->refs == 0) {
        g_up_live.erase(std::find(g_up_live.begin(), g_up_live.end(), e));
        dead = e;
        size_t pooled = 0;
        for (auto& b : g_up_pool) pooled += b.second.first;
        if (dead->dev && g_up_pool.size() < UP_POOL_MAX && pooled + dead->cap <= UP_POOL_MAX_BYTES) {
          g_up_pool.push_back({dead->device, {dead->cap, dead->dev}});
          dead->dev = nullptr;
        }
      }
    }
    if (dead) {  // every call that used the copy has synchronised its stream (csh_msm_dev is synchronous) or failed before launching
      if (dead->ready) (void)hipEventDestroy(dead->ready);
      if (dead->dev) (void)hipFree(dead->dev);
      delete dead;
    }
  }
};
}  // namespace

static int csh::repack_bases(Bases* B, hipStream_t st) {
  CURVE_DISPATCH(B->curve, B->group, (repack_bases_t<Cfg>(B, st)));
}

static int valid_cg(csh_curve_t c, csh_group_t g) {
  CSH_REQUIRE(c == CSH_BN254 || c == CSH_BLS12_381 || c == CSH_GRUMPKIN || c == CSH_BLS12_377, "unknown curve");
  CSH_REQUIRE(g == CSH_G1 || (g == CSH_G2 && c != CSH_GRUMPKIN), "unknown group");
  return CSH_OK;
}

extern "C" {

static int bases_upload_common(csh_curve_t curve, csh_group_t group, const void* pts, size_t n, size_t stride, bool src_dev, void* stream,
                               csh_bases_t* out) {
  CSH_REQUIRE(out, "out is NULL");
  CSH_TRY(valid_cg(curve, group));
  CSH_REQUIRE(n < (size_t(1) << 31), "at most 2^31-1 bases per handle");
  CSH_REQUIRE(pts || n == 0, "points is NULL");
  CSH_TRY(ensure_device());
  const size_t pb = point_bytes_of(curve, group);
  if (stride == 0) stride = pb;
  CSH_REQUIRE(stride >= pb, "stride_bytes smaller than a packed affine point");
  Bases* B = new Bases();
  B->curve = curve;
  B->group = group;
  B->n = n;
  B->point_bytes = pb;
  B->points = nullptr;
  if (hipGetDevice(&B->device) != hipSuccess) B->device = 0;
  hipError_t e = hipMalloc(&B->points, n ? n * pb : 1);
  if (e != hipSuccess) {
    delete B;
    set_error("hipMalloc(%zu bytes) for bases failed: %s", n * pb, hipGetErrorString(e));
    return CSH_ERR_OOM;
  }
  if (n) {
    hipStream_t st = resolve_stream(stream);
    const hipMemcpyKind kind = src_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (!src_dev && stride == pb) {  // packed host points: the "host_h2d" policy (staged by default: the runtime does not pin the caller's key)
      if (upload_h2d(B->points, pts, n * pb, st, 0) != CSH_OK) e = hipErrorUnknown;
    } else {
      e = stride == pb ? hipMemcpyAsync(B->points, pts, n * pb, kind, st) : hipMemcpy2DAsync(B->points, pb, pts, stride, pb, n, kind, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
      (void)hipFree(B->points);
      delete B;
      set_error("bases upload failed: %s", hipGetErrorString(e));
      return CSH_ERR_HIP;
    }
    int rc = repack_bases(B, st);
    if (rc != CSH_OK) {
      (void)hipFree(B->points);
      delete B;
      return rc;
    }
  }
  *out = reinterpret_cast<csh_bases_t>(B);
  return CSH_OK;
}

int csh_bases_upload(csh_curve_t curve, csh_group_t group, const void* pts, size_t n, size_t stride, csh_bases_t* out) {
  return bases_upload_common(curve, group, pts, n, stride, false, nullptr, out);
}
int csh_bases_upload_dev(csh_curve_t curve, csh_group_t group, const void* pts, size_t n, size_t stride, void* stream, csh_bases_t* out) {
  return bases_upload_common(curve, group, pts, n, stride, true, stream, out);
}
int csh_bases_len(csh_bases_t bases, size_t* n) {
  CSH_REQUIRE(bases && n, "NULL argument");
  *n = reinterpret_cast<Bases*>(bases)->n;
  return CSH_OK;
}
// A copy of a handle, or of the range [offset, offset + n) of it (points and the matching columns of every fixed-base table row, already
// in the stored encoding) on another GPU: device-to-device, no host staging and no re-encoding. The calling thread's device binding is
// restored. A range clone is its own handle of n bases: point i of the clone is point offset + i of the source.
int csh_bases_clone_range(csh_bases_t src, size_t offset, size_t n, int device, csh_bases_t* out) {
  CSH_REQUIRE(src && out, "NULL argument");
  const Bases* S = reinterpret_cast<const Bases*>(src);
  CSH_REQUIRE(offset <= S->n && n <= S->n - offset, "csh_bases_clone_range: offset + n exceeds the number of bases");
  int ndev = 0, cur = 0;
  CSH_HIP(hipGetDeviceCount(&ndev));
  CSH_REQUIRE(device >= 0 && device < ndev, "device out of range");
  CSH_HIP(hipGetDevice(&cur));
  Bases* B = new Bases(*S);
  B->device = device;
  B->n = n;
  B->points = nullptr;
  B->table = nullptr;
  const size_t pb = S->point_bytes, pbytes = n * pb, rows = S->table ? (size_t)S->table_W : 0;
  if (!rows) B->table_c = B->table_W = 0;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess && pbytes) e = hipMalloc(&B->points, pbytes);
  if (e == hipSuccess && pbytes && rows) e = hipMalloc(&B->table, pbytes * rows);
  if (e == hipSuccess && pbytes) e = hipMemcpyPeer(B->points, device, static_cast<const char*>(S->points) + offset * pb, S->device, pbytes);
  for (size_t k = 0; e == hipSuccess && pbytes && k < rows; ++k)  // row k of the table: table[k * n + i]
    e = hipMemcpyPeer(static_cast<char*>(B->table) + k * pbytes, device, static_cast<const char*>(S->table) + (k * S->n + offset) * pb, S->device, pbytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess && S->device != device) {  // and the source side of the copies
    e = hipSetDevice(S->device);
    if (e == hipSuccess) e = hipDeviceSynchronize();
  }
  (void)hipSetDevice(cur);
  if (e != hipSuccess) {
    const bool oom = e == hipErrorOutOfMemory;
    if (B->points) (void)hipFree(B->points);
    if (B->table) (void)hipFree(B->table);
    delete B;
    set_error("csh_bases_clone_range to device %d failed: %s", device, hipGetErrorString(e));
    return oom ? CSH_ERR_OOM : CSH_ERR_HIP;
  }
  *out = reinterpret_cast<csh_bases_t>(B);
  return CSH_OK;
}
int csh_bases_clone(csh_bases_t src, int device, csh_bases_t* out) {
  CSH_REQUIRE(src && out, "NULL argument");
  return csh_bases_clone_range(src, 0, reinterpret_cast<const Bases*>(src)->n, device, out);
}
int csh_bases_free(csh_bases_t bases) {
  if (!bases) return CSH_OK;
  Bases* B = reinterpret_cast<Bases*>(bases);
  if (B->points) (void)hipFree(B->points);
  if (B->table) (void)hipFree(B->table);
  delete B;
  return CSH_OK;
}

// Fixed-base window tables for a set of bases that is reused across MSMs (a proving key query): see msm_impl.hpp. c = 0
// picks the window width (16 from 2^17 points on, narrower below); handles of fewer than 1024 points stay as they are.
static int bases_precompute(csh_bases_t bases, int c, int groups);
int csh_bases_table_policy(size_t key_points, int* c_out, int* rows_out) {
  CSH_REQUIRE(c_out && rows_out, "c_out / rows_out is NULL");
  *c_out = 0;
  *rows_out = 0;
  if (key_points < (size_t(1) << 14) || key_points > (size_t(1) << 26)) return CSH_OK;
  if (key_points < (size_t(1) << 15)) {
    // 2^14 .. 2^15: one row per window at c <= 16 (the plain sort stage); measured a tie with the plain handle (profiles/r06_f_policy_fullmerge_small.log)
    int c = 16;
    while (c > 10 && (size_t(1) << (c + 1)) > key_points) --c;
    *c_out = c;
    *rows_out = 16;
    return CSH_OK;
  }
  // Round 6: ONE bucket set, one table row per window, windows wider than 16 bits (msm_sort_wide.hip). Interleaved with the plain handle
  // and with every c = 12 .. 18 / 17 .. 22 on the same box (profiles/r06_e_policy_*.log, r06_f_policy_fullmerge_small.log): c = 17 is the
  // best or within 1 % of it from 2^15 to 2^21 points (2^16 0.355 against 0.40 ms plain, 2^18 0.56 / 0.71, 2^20 +15 %, 2^21 +13 %), c = 20
  // from 2^22 on (2^22 +10 %, 2^23 +12 %, 2^24 +15 .. 17 %; c = 22 has the fewest additions but its 2^21 buckets cost more tail and
  // sort than they save). rows = "at least the windows of either scalar field": the precompute clamps to W = ceil((bits + 1) / c).
  if (key_points <= (size_t(3) << 20)) {
    *c_out = 17;
    *rows_out = 16;
  } else {
    *c_out = 20;
    *rows_out = 13;
  }
  return CSH_OK;
}
int csh_bases_drop_tables(csh_bases_t bases) {
  CSH_REQUIRE(bases, "bases is NULL");
  Bases* B = reinterpret_cast<Bases*>(bases);
  if (B->table) {
    (void)hipFree(B->table);
    B->table = nullptr;
    B->table_c = B->table_W = 0;
  }
  return CSH_OK;
}
int csh_bases_precompute(csh_bases_t bases, int c) { return bases_precompute(bases, c, 0); }
int csh_bases_precompute_grouped(csh_bases_t bases, int c, int groups) {
  CSH_REQUIRE(groups >= 2 && groups <= MAX_WINDOWS, "groups must be in [2, 128]");
  return bases_precompute(bases, c, groups);
}
static int bases_precompute(csh_bases_t bases, int c, int groups) {
  CSH_REQUIRE(bases, "bases is NULL");
  CSH_TRY(ensure_device());
  Bases* B = reinterpret_cast<Bases*>(bases);
  CSH_REQUIRE(c == 0 || (c >= 4 && c <= 22), "window width must be 0 (auto) or in [4, 22]");
  if (B->table) {
    (void)hipFree(B->table);
    B->table = nullptr;
  }
  if (B->n < 1024) return CSH_OK;
  if (c == 0) {
    c = 16;
    while (c > 10 && (size_t(1) << (c + 1)) > B->n) --c;  // ~2 points per bucket and window at least
  }
  {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != B->device) {
      set_error("bases were uploaded on device %d but the calling thread is bound to device %d (csh_init)", B->device, cur);
      return CSH_ERR_INVALID;
    }
  }
  hipStream_t st = resolve_stream(nullptr);
  CURVE_DISPATCH(B->curve, B->group, (precompute_table_t<Cfg>(B, c, groups, st)));
}

static int msm_args(csh_bases_t bases, size_t offset, size_t n, const void* scalars, const void* out) {
  CSH_REQUIRE(bases, "bases is NULL");
  CSH_REQUIRE(out, "out is NULL");
  Bases* B = reinterpret_cast<Bases*>(bases);
  CSH_REQUIRE(offset <= B->n && n <= B->n - offset, "offset + n exceeds the uploaded bases");
  CSH_REQUIRE(scalars || n == 0, "scalars is NULL");
  int cur = -1;
  if (hipGetDevice(&cur) == hipSuccess && cur != B->device) {
    set_error("bases were uploaded on device %d but the calling thread is bound to device %d (csh_init): upload a copy per device", B->device, cur);
    return CSH_ERR_INVALID;
  }
  return CSH_OK;
}

int csh_msm_dev(csh_bases_t bases, size_t offset, size_t n, const uint64_t* scalars_dev, int mont, void* out_host, void* stream) {
  CSH_TRY(msm_args(bases, offset, n, scalars_dev, out_host));
  CSH_TRY(ensure_device());
  Bases* B = reinterpret_cast<Bases*>(bases);
  hipStream_t st = resolve_stream(stream);
  CURVE_DISPATCH(B->curve, B->group, (msm_t<Cfg>(B, offset, n, scalars_dev, mont, out_host, st)));
}

int csh_msm(csh_bases_t bases, size_t offset, size_t n, const uint64_t* scalars, int mont, void* out) {
  CSH_TRY(msm_args(bases, offset, n, scalars, out));
  const size_t bytes = 32 * n;
  if (bytes >= (size_t(1) << 20) && tune().msm_share_uploads.load(std::memory_order_relaxed) != 0) {
    // Concurrent calls over ONE host slice (the reference's rayon_join5, groth16.rs:227-294: four of the five MSM closures of a proof
    // read the same aux_assignment) share one upload: see SharedUpload above.
    CSH_TRY(ensure_device());
    hipStream_t st = resolve_stream(nullptr);
    int device = 0;
    (void)hipGetDevice(&device);
    const bool batching = tune().msm_share_uploads.load(std::memory_order_relaxed) >= 2;
    const Bases* B = reinterpret_cast<const Bases*>(bases);
    SharedUpload::Req req;
    req.bases = bases, req.offset = offset, req.out = out;
    bool joined = false;
    SharedUploadRef up;
    CSH_TRY(up.acquire(device, scalars, bytes, st, batching ? &req : nullptr, batching ? (int)B->curve : -1, mont != 0, &joined));
    if (joined) {  // the call that was uploading this slice ran this MSM with its own (csh_msm_multi_dev): the result is in `out`
      if (req.rc != CSH_OK) set_error("%s", req.err.c_str());
      return req.rc;
    }
    const uint64_t* dsc = reinterpret_cast<const uint64_t*>(up.dev());
    if (!up.is_owner || !batching) return csh_msm_dev(bases, offset, n, dsc, mont, out, st);
    const std::vector<SharedUpload::Req*> others = up.seal();
    if (others.empty()) {
      up.finish_batch();
      return csh_msm_dev(bases, offset, n, dsc, mont, out, st);
    }
    // G2 handles first: their host fold (Horner over Fp2 windows) then runs under the G1 bucket stages that follow (as the mirror's prover orders them)
    std::vector<SharedUpload::Req*> all;
    all.push_back(&req);
    for (SharedUpload::Req* r : others) all.push_back(r);
    std::stable_sort(all.begin(), all.end(), [](const SharedUpload::Req* a, const SharedUpload::Req* b) {
      return reinterpret_cast<const Bases*>(a->bases)->group > reinterpret_cast<const Bases*>(b->bases)->group;
    });
    std::vector<csh_bases_t> hs;
    std::vector<size_t> offs;
    std::vector<void*> outs;
    for (SharedUpload::Req* r : all) hs.push_back(r->bases), offs.push_back(r->offset), outs.push_back(r->out);
    const int rc = csh_msm_multi_dev(hs.data(), offs.data(), all.size(), n, dsc, mont, outs.data(), st);
    const std::string err = rc == CSH_OK ? std::string() : std::string(csh_last_error());
    for (SharedUpload::Req* r : others) r->rc = rc, r->err = err;
    up.finish_batch();
    return rc;
  }
  HostStage h;
  CSH_TRY(h.begin(Arena::padded(bytes)));
  uint64_t* ds;
  CSH_TRY(h.up(ds, scalars, bytes));
  return csh_msm_dev(bases, offset, n, ds, mont, out, h.st);
}

// One MSM per share component over the same bases: the Rep3PointShare {a, b} of pointshare::msm_public_points. The AoS share vector
// crosses PCIe once; each component is cut out on the device (a strided device copy) and runs the ordinary pipeline.
int csh_msm_shares(csh_bases_t bases, size_t offset, size_t n, const uint64_t* shares, uint32_t ncomp, int mont, void* const* outs) {
  CSH_REQUIRE(ncomp >= 1 && ncomp <= 2 && outs, "msm_shares: ncomp must be 1 or 2 and outs non-NULL");
  for (uint32_t c = 0; c < ncomp; ++c) CSH_REQUIRE(outs[c], "msm_shares: NULL output");
  CSH_TRY(msm_args(bases, offset, n, shares, outs[0]));
  if (ncomp == 1) return csh_msm(bases, offset, n, shares, mont, outs[0]);
  HostStage h;
  CSH_TRY(h.begin(Arena::padded(32 * (size_t)ncomp * n) + Arena::padded(32 * n)));
  uint64_t *dsh, *dcomp;
  CSH_TRY(h.up(dsh, shares, 32 * (size_t)ncomp * n));
  CSH_TRY(h.up(dcomp, nullptr, 32 * n));
  for (uint32_t c = 0; c < ncomp; ++c) {
    CSH_TRY(csh_extract_component_dev(dsh, ncomp, c, n, dcomp, h.st));
    CSH_TRY(csh_msm_dev(bases, offset, n, dcomp, mont, outs[c], h.st));  // synchronous: dcomp is free again when it returns
  }
  return CSH_OK;
}

int csh_msm_partial_bytes(csh_curve_t curve, csh_group_t group, size_t* bytes) {
  CSH_REQUIRE(bytes, "bytes is NULL");
  CSH_TRY(valid_cg(curve, group));
  *bytes = partial_bytes_of(curve, group);
  return CSH_OK;
}

int csh_msm_partial_dev(csh_bases_t bases, size_t offset, size_t n, const uint64_t* scalars_dev, int mont, void* out_dev, void* stream) {
  CSH_TRY(msm_args(bases, offset, n, scalars_dev, out_dev));
  CSH_TRY(ensure_device());
  Bases* B = reinterpret_cast<Bases*>(bases);
  hipStream_t st = resolve_stream(stream);
  CURVE_DISPATCH(B->curve, B->group, (msm_partial_t<Cfg>(B, offset, n, scalars_dev, mont, out_dev, st, true)));
}

int csh_msm_fold_partials(csh_curve_t curve, csh_group_t group, const void* partials_host, size_t nparts, void* out_jacobian) {
  CSH_REQUIRE(partials_host && out_jacobian, "NULL argument");
  CURVE_DISPATCH(curve, group, (fold_partials_t<Cfg>(partials_host, nparts, out_jacobian)));
}

// k MSMs over ONE scalar vector (the four aux-assignment MSMs of a Groth16 proof: A, B in G1, B in G2, L; groth16.rs:237-284):
// the digit decomposition and the bucket sort depend on the scalars only and are done once; the bucket stage runs per set
// of bases. All handles must belong to the same curve (same scalar field); every MSM uses n points from its offset.
int csh_msm_multi_dev(const csh_bases_t* bases, const size_t* offsets, size_t k, size_t n, const uint64_t* scalars_dev, int mont,
                      void* const* outs_host, void* stream) {
  CSH_REQUIRE(bases && offsets && outs_host && k >= 1 && k <= 16, "msm_multi: bad arguments");
  CSH_REQUIRE(scalars_dev || n == 0, "scalars is NULL");
  CSH_REQUIRE(n < (size_t(1) << 31), "n too large");
  CSH_TRY(ensure_device());
  const Bases* B0 = reinterpret_cast<const Bases*>(bases[0]);
  CSH_REQUIRE(B0, "bases[0] is NULL");
  for (size_t i = 0; i < k; ++i) {
    const Bases* B = reinterpret_cast<const Bases*>(bases[i]);
    CSH_REQUIRE(B && outs_host[i], "msm_multi: NULL handle or output");
    CSH_REQUIRE(B->curve == B0->curve, "msm_multi: all bases must belong to one curve");
    CSH_REQUIRE(offsets[i] <= B->n && n <= B->n - offsets[i], "msm_multi: offset + n exceeds the number of bases");
    CSH_REQUIRE(B->device == B0->device, "msm_multi: bases live on different devices");
  }
  {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != B0->device) {
      set_error("bases were uploaded on device %d but the calling thread is bound to device %d (csh_init)", B0->device, cur);
      return CSH_ERR_INVALID;
    }
  }
  hipStream_t st = resolve_stream(stream);
  struct Ops {
    size_t (*bytes)(const MsmParams*);
    int (*bucket)(const void*, const MsmParams*, const SortOut*, hipStream_t, Arena*, void*, hipEvent_t*);
    void (*fold)(const void*, int, int, int, void*);
    size_t xyzz_bytes;
    int (*occ)();  // accumulate waves of the group's kernel that fit one SIMD
  };
  auto ops_of = [](const Bases* B, Ops* o) -> bool {
#define CSH_OPS(CFG) *o = Ops{msm_bucket_bytes<CFG>, msm_bucket_stage<CFG>, fold_windows_erased<CFG>, sizeof(XYZZ<CFG::Fq>), accum_occupancy<CFG>}; return true
    if (B->curve == CSH_BN254 && B->group == CSH_G1) { CSH_OPS(Bn254G1Cfg); }
    if (B->curve == CSH_BN254 && B->group == CSH_G2) { CSH_OPS(Bn254G2Cfg); }
    if (B->curve == CSH_BLS12_381 && B->group == CSH_G1) { CSH_OPS(Bls381G1Cfg); }
    if (B->curve == CSH_BLS12_381 && B->group == CSH_G2) { CSH_OPS(Bls381G2Cfg); }
    if (B->curve == CSH_GRUMPKIN && B->group == CSH_G1) { CSH_OPS(GrumpkinG1Cfg); }
    if (B->curve == CSH_BLS12_377 && B->group == CSH_G1) { CSH_OPS(Bls377G1Cfg); }
    if (B->curve == CSH_BLS12_377 && B->group == CSH_G2) { CSH_OPS(Bls377G2Cfg); }
#undef CSH_OPS
    return false;
  };
  std::vector<Ops> ops(k);
  for (size_t i = 0; i < k; ++i) CSH_REQUIRE(ops_of(reinterpret_cast<const Bases*>(bases[i]), &ops[i]), "msm_multi: unknown curve/group");
  if (n == 0) {
    for (size_t i = 0; i < k; ++i) ops[i].fold(nullptr, 0, 2, 0, outs_host[i]);
    return CSH_OK;
  }
  const int bits = scalar_bits_of(B0->curve);
  // merged-window mode needs every handle to carry tables of one window width (the digit codes are shared)
  bool merged = true;
  for (size_t i = 0; i < k; ++i) {
    const Bases* B = reinterpret_cast<const Bases*>(bases[i]);
    merged = merged && msm_use_table(B, n) && B->table_c == B0->table_c && B->table_W == B0->table_W;
  }
  int occ = 8;  // the shared plan's lane length suits the group with the fewest co-resident accumulate waves (a G2 handle: 1)
  for (auto& o : ops) occ = std::min(occ, o.occ());
  const MsmParams pdig = merged ? msm_plan_merged(n, bits, mont, B0->table_c, B0->table_W, B0->n, 0, occ).dig : msm_plan(n, bits, mont, occ);
  MsmParams p = merged ? msm_plan_merged(n, bits, mont, B0->table_c, B0->table_W, B0->n, 0, occ).srt : pdig;
  size_t bucket_max = 0, win_bytes = 0;
  for (auto& o : ops) {
    bucket_max = std::max(bucket_max, o.bytes(&p));
    win_bytes += Arena::padded(o.xyzz_bytes * MAX_WINDOWS);
  }
  Arena& ar = arena_for(st);
  // merged mode: the remap (table stride, offset) differs per handle, so the scatter runs per handle on shared digit
  // codes; otherwise one sort serves all
  // two bucket-stage scratch regions: consecutive bucket stages alternate between the caller's stream and a second one, so
  // the latency-bound bucket reduction of MSM i overlaps the throughput-bound accumulation of MSM i + 1
  CSH_TRY(ar.reserve(msm_sort_bytes(p, pdig) + 2 * Arena::padded(bucket_max)));
  Arena& wa = arena_for((hipStream_t)((uintptr_t)st ^ 0x2));
  CSH_TRY(wa.reserve(win_bytes));
  auto sort_stage = [&](const MsmParams& ps, SortOut* so) -> int {
    if (B0->curve == CSH_BLS12_381) return msm_sort_stage<Bls381Fr>(ps, pdig, scalars_dev, st, ar, so, nullptr);
    if (B0->curve == CSH_GRUMPKIN) return msm_sort_stage<Bn254Fq>(ps, pdig, scalars_dev, st, ar, so, nullptr);
    if (B0->curve == CSH_BLS12_377) return msm_sort_stage<Bls377Fr>(ps, pdig, scalars_dev, st, ar, so, nullptr);
    return msm_sort_stage<Bn254Fr>(ps, pdig, scalars_dev, st, ar, so, nullptr);
  };
  SortOut so;
  std::vector<char*> win_dev(k);
  hipStream_t aux = nullptr;
  std::vector<hipStream_t> stage_stream(k, st);
  if (!merged) {
    const bool overlap = tune().msm_multi_overlap.load(std::memory_order_relaxed) != 0;
    if (overlap && k > 1) aux = resolve_aux_stream();  // pooled with the thread's lane: no stream creation per call
    const bool two = overlap && k > 1 && aux != nullptr;
    CSH_TRY(sort_stage(p, &so));
    const size_t mark = ar.off;
    hipEvent_t sorted_ev = nullptr;
    if (two) {
      CSH_HIP(hipEventCreateWithFlags(&sorted_ev, hipEventDisableTiming));
      CSH_HIP(hipEventRecord(sorted_ev, st));
      CSH_HIP(hipStreamWaitEvent(aux, sorted_ev, 0));
    }
    for (size_t i = 0; i < k; ++i) {
      const bool on_aux = two && (i & 1);
      stage_stream[i] = on_aux ? aux : st;
      ar.off = mark + (on_aux ? Arena::padded(bucket_max) : 0);  // per-stream scratch region: stages on one stream are ordered
      const Bases* B = reinterpret_cast<const Bases*>(bases[i]);
      win_dev[i] = wa.take<char>(ops[i].xyzz_bytes * MAX_WINDOWS);
      CSH_TRY(ops[i].bucket(static_cast<const char*>(B->points) + offsets[i] * B->point_bytes, &p, &so, stage_stream[i], &ar, win_dev[i], nullptr));
    }
    if (sorted_ev) (void)hipEventDestroy(sorted_ev);
  } else {
    // Handles that share (table stride, offset) share the sorted index list; a different pair needs its own scatter. Round 6: the bucket
    // stages alternate between the caller's stream and the lane's second stream here too (each with its own scratch region), so the
    // latency-bound bucket reduction of MSM i runs under the accumulation of MSM i + 1 -- with ONE bucket set per MSM the reduction is
    // 0.2-0.4 ms of a 1.4 ms G1 MSM. A re-sort waits for the other stream's readers of the previous list.
    const bool overlap = tune().msm_multi_overlap.load(std::memory_order_relaxed) != 0;
    if (overlap && k > 1) aux = resolve_aux_stream();
    const bool two = overlap && k > 1 && aux != nullptr;
    size_t last_stride = (size_t)-1, last_off = (size_t)-1;
    size_t mark = 0;
    hipEvent_t sorted_ev = nullptr, aux_done = nullptr;
    if (two) {
      CSH_HIP(hipEventCreateWithFlags(&sorted_ev, hipEventDisableTiming));
      CSH_HIP(hipEventCreateWithFlags(&aux_done, hipEventDisableTiming));
    }
    bool aux_used = false;
    for (size_t i = 0; i < k; ++i) {
      const Bases* B = reinterpret_cast<const Bases*>(bases[i]);
      if (B->n != last_stride || offsets[i] != last_off) {
        if (two && aux_used) {  // bucket stages on the second stream still read the list this sort overwrites
          CSH_HIP(hipEventRecord(aux_done, aux));
          CSH_HIP(hipStreamWaitEvent(st, aux_done, 0));
        }
        ar.off = 0;
        p.remap_stride = (uint32_t)B->n;
        p.remap_off = (uint32_t)offsets[i];
        CSH_TRY(sort_stage(p, &so));
        mark = ar.off;
        last_stride = B->n;
        last_off = offsets[i];
        if (two) {
          CSH_HIP(hipEventRecord(sorted_ev, st));
          CSH_HIP(hipStreamWaitEvent(aux, sorted_ev, 0));
        }
      }
      const bool on_aux = two && (i & 1);
      stage_stream[i] = on_aux ? aux : st;
      aux_used = aux_used || on_aux;
      ar.off = mark + (on_aux ? Arena::padded(bucket_max) : 0);
      win_dev[i] = wa.take<char>(ops[i].xyzz_bytes * MAX_WINDOWS);
      CSH_TRY(ops[i].bucket(B->table, &p, &so, stage_stream[i], &ar, win_dev[i], nullptr));
    }
    if (sorted_ev) (void)hipEventDestroy(sorted_ev);
    if (aux_done) (void)hipEventDestroy(aux_done);
  }
  // window sums of all k results through one page-locked buffer of the lane (DMA copies, no staging), pageable fallback
  size_t slice = 0;
  for (size_t i = 0; i < k; ++i) slice = std::max(slice, Arena::padded(ops[i].xyzz_bytes * MAX_WINDOWS));
  void *pin_host = nullptr, *pin_dev = nullptr;
  std::vector<char> pageable;
  char* wins = nullptr;
  if (pinned_for((hipStream_t)((uintptr_t)st ^ 0x4), slice * k, &pin_host, &pin_dev)) {
    wins = static_cast<char*>(pin_host);
  } else {
    pageable.resize(slice * k);
    wins = pageable.data();
  }
  // each result is folded on the host (Horner over its windows: ~80 us on G1, ~250 us on G2) as soon as ITS window sums have
  // arrived, while the device is still busy with the later bucket stages
  std::vector<hipEvent_t> arrived(k, nullptr);
  auto drop_events = [&] {
    for (hipEvent_t e : arrived)
      if (e) (void)hipEventDestroy(e);
  };
  for (size_t i = 0; i < k; ++i) {
    hipError_t e = hipMemcpyAsync(wins + slice * i, win_dev[i], ops[i].xyzz_bytes * p.W, hipMemcpyDeviceToHost, stage_stream[i]);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&arrived[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(arrived[i], stage_stream[i]);
    if (e != hipSuccess) {
      drop_events();
      set_error("msm_multi: queuing the window sums of result %zu failed: %s", i, hipGetErrorString(e));
      return CSH_ERR_HIP;
    }
  }
  for (size_t i = 0; i < k; ++i) {
    const hipError_t e = hipEventSynchronize(arrived[i]);
    if (e != hipSuccess) {
      (void)hipStreamSynchronize(st);
      if (aux) (void)hipStreamSynchronize(aux);
      drop_events();
      set_error("msm_multi: result %zu failed on the device: %s", i, hipGetErrorString(e));
      return CSH_ERR_HIP;
    }
    ops[i].fold(wins + slice * i, p.W, p.c, p.wide, outs_host[i]);
  }
  drop_events();
  CSH_HIP(hipStreamSynchronize(st));
  if (aux) CSH_HIP(hipStreamSynchronize(aux));
  return CSH_OK;
}

int csh_msm_plan(csh_curve_t curve, size_t n, uint32_t out[6]) {
  CSH_REQUIRE(out, "out is NULL");
  CSH_REQUIRE(curve == CSH_BN254 || curve == CSH_BLS12_381 || curve == CSH_GRUMPKIN || curve == CSH_BLS12_377, "unknown curve");
  CSH_REQUIRE(n >= 1 && n < (size_t(1) << 31), "n out of range");
  const int bits = scalar_bits_of(curve);
  uint32_t keep[4];
  for (int i = 0; i < 4; ++i) keep[i] = tl_msm_params[i];  // planning must not disturb csh_msm_last_params
  const MsmParams p = msm_plan(n, bits, 1);
  for (int i = 0; i < 4; ++i) tl_msm_params[i] = keep[i];
  const uint64_t lanes = (n + p.L - 1) / p.L;
  out[0] = (uint32_t)p.c;
  out[1] = (uint32_t)p.W;
  out[2] = p.L;
  out[3] = p.S;
  out[4] = (uint32_t)((uint64_t)p.W * ((lanes + ACC_BLK - 1) / ACC_BLK) * (ACC_BLK / 64));
  out[5] = (uint32_t)device_simds();
  return CSH_OK;
}

int csh_msm_last_params(uint32_t out[4]) {
  CSH_REQUIRE(out, "out is NULL");
  for (int i = 0; i < 4; ++i) out[i] = tl_msm_params[i];
  return CSH_OK;
}

int csh_msm_last_timing(float out_ms[6]) {
  CSH_REQUIRE(out_ms, "out_ms is NULL");
  for (int i = 0; i < 6; ++i) out_ms[i] = tl_msm_timing[i];
  return CSH_OK;
}

}  // extern "C"
