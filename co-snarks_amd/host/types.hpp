// Host-side mirror of the reference's Groth16 types (C++ above the C ABI; the reference is Rust).
//
// Mirrors: ark-groth16 `ProvingKey` / `Proof` as the prover reads them (co-circom/co-groth16/src/groth16.rs:
// 219-225, 237-290, 333-337), taceo-groth16 `ConstraintMatrices` (co-groth16/src/lib.rs:268-279),
// `SharedWitness{public_inputs, witness}` (co-circom/co-circom-types/src/lib.rs:205-218), and the zkey/wtns
// ingest the reference delegates to taceo-circom-types (co-circom/src/bin/co-circom.rs:1005-1006).
// Field / curve arithmetic on the host reuses the kernels' own templates (csrc/field.hpp, curve.hpp).
#pragma once
#include <sched.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/cosnarks_hip.h"
#include "../csrc/curve.hpp"
#include "../csrc/host_fp64.hpp"

namespace cosnarks {

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

inline void check(int rc, const char* what) {
  if (rc != CSH_OK) throw Error(std::string(what) + ": " + csh_last_error());
}

struct Bn254 {
  static constexpr csh_curve_t ID = CSH_BN254;
  using Fr = csh::Bn254Fr;
  using Fq = csh::Bn254Fq;
  using Fq2 = csh::Bn254Fq2;
  static constexpr uint64_t FR_GENERATOR = 5;  // ark_bn254::Fr::GENERATOR
  static const char* name() { return "bn128"; }
  static const uint32_t* g1_generator_words() { return csh::Bn254G1Gen; }
};
struct Bls12_381 {
  static constexpr csh_curve_t ID = CSH_BLS12_381;
  using Fr = csh::Bls381Fr;
  using Fq = csh::Bls381Fq;
  using Fq2 = csh::Bls381Fq2;
  static constexpr uint64_t FR_GENERATOR = 7;  // ark_bls12_381::Fr::GENERATOR
  static const char* name() { return "bls12381"; }
  static const uint32_t* g1_generator_words() { return csh::Bls381G1Gen; }
};

// the curve of the reference's LibSnarkReduction fixtures (co-groth16/src/lib.rs:231-300): witness maps and, since round 6, the
// whole plain proof (cog16_prove_libsnark: ark ProvingKey + Matrix blobs + wtns in, ark Proof out)
struct Bls12_377 {
  static constexpr csh_curve_t ID = CSH_BLS12_377;
  using Fr = csh::Bls377Fr;
  using Fq = csh::Bls377Fq;
  using Fq2 = csh::Bls377Fq2;
  static constexpr uint64_t FR_GENERATOR = 22;  // ark_bls12_377::Fr::GENERATOR
  static const char* name() { return "bls12377"; }
  static const uint32_t* g1_generator_words() { return csh::Bls377G1Gen; }
};

// Tracing spans mirroring the reference's `tracing::debug_span!` names (groth16.rs:229-331, reduction.rs:97-191):
// COG16_TRACE=1 prints "<span> took <ms>" on close, like the CLI's FmtSpan::CLOSE layer (co-circom.rs:580-599).
struct Span {
  const char* name;
  std::chrono::steady_clock::time_point t0;
  bool on;
  explicit Span(const char* n) : name(n), t0(std::chrono::steady_clock::now()), on(getenv("COG16_TRACE") != nullptr) {}
  ~Span() {
    if (on) fprintf(stderr, "[cog16] %s took %.3f ms\n", name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
};

// rayon-style parallel for over [0, n) in contiguous chunks (the reference uses par_iter().with_min_len(k))
// threads worth starting: the CPUs this process may run on (a cgroup / affinity mask on a 256-CPU host often grants 16), as rayon's
// default pool size does (available_parallelism)
inline size_t host_threads() {
  static const size_t n = [] {
    size_t k = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) k = (size_t)CPU_COUNT(&set);
    if (k == 0) k = std::thread::hardware_concurrency();
    if (k == 0) k = 1;
    // a cgroup CPU quota below the affinity count (16 CPUs' worth of time on a 256-CPU host): more runnable threads than the quota only
    // get the whole group throttled for the rest of the scheduler period (the mask draw of a Rep3 party took 67 ms with 64 threads on
    // such a box, profiles/archive/r05_d_bench_20_5.log)
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      long long quota = 0, period = 0;
      char q[32] = {0};
      if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
        quota = atoll(q);
        const size_t cpus = (size_t)((quota + period - 1) / period);
        if (cpus >= 1 && cpus < k) k = cpus;
      }
      fclose(f);
    }
    return k > 64 ? (size_t)64 : k;
  }();
  return n;
}
template <class Fn>
inline void parallel_for(size_t n, size_t min_len, Fn fn) {
  size_t nt = host_threads();
  size_t chunks = std::min(nt, std::max<size_t>(1, n / std::max<size_t>(1, min_len)));
  if (chunks <= 1) {
    fn(0, n);
    return;
  }
  std::vector<std::thread> th;
  const size_t per = (n + chunks - 1) / chunks;
  for (size_t c = 0; c < chunks; ++c) {
    const size_t lo = c * per, hi = std::min(n, lo + per);
    if (lo >= hi) break;
    th.emplace_back([=] { fn(lo, hi); });
  }
  for (auto& t : th) t.join();
}

// Process-wide device list for one prover's MSM placement (cog16_set_prover_devices): keys built afterwards spread their five
// queries over these GPUs (ProvingKey::place); entry 0 stands for the key's home GPU whatever its value. Empty / one entry: off.
struct ProverDevices {
  std::mutex mu;
  std::vector<int> devices;
  static ProverDevices& get() {
    static ProverDevices d;
    return d;
  }
  int mode = 0;  // ProvingKey::PLACE_AUTO / PLACE_BY_QUERY / PLACE_BY_RANGE
  std::vector<int> snapshot(int* mode_out = nullptr) {
    std::lock_guard<std::mutex> g(mu);
    if (mode_out) *mode_out = mode;
    return devices;
  }
};

// wall-clock phases of the calling thread's last prove_inner (host clock around synchronous device work): witness upload + map,
// the five MSM groups, the finish (openings, a few point operations)
// Allocated, never zero-filled host memory for a result that is written in full: what `Vec::with_capacity(n)` + `set_len(n)` (or a
// `.collect()` into a fresh Vec) is in Rust. (A value-initialised std::vector of 32 MB costs 4.7 ms of page faults and zero fill on the
// GPU hosts, profiles/archive/r04_b_prefault_probe.jsonl; the library populates the pages of its results from helper threads while the device works.)
template <class E>
struct UninitBuf {
  E* p = nullptr;
  size_t n = 0;
  UninitBuf() = default;
  explicit UninitBuf(size_t count) : p(static_cast<E*>(malloc(count * sizeof(E) + 1))), n(count) {
    if (!p) throw Error("out of host memory");
  }
  UninitBuf(UninitBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr, o.n = 0; }
  UninitBuf& operator=(UninitBuf&& o) noexcept {
    if (this != &o) {
      free(p);
      p = o.p, n = o.n, o.p = nullptr, o.n = 0;
    }
    return *this;
  }
  UninitBuf(const UninitBuf&) = delete;
  UninitBuf& operator=(const UninitBuf&) = delete;
  ~UninitBuf() { free(p); }
  E* data() { return p; }
  const E* data() const { return p; }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  E& operator[](size_t i) { return p[i]; }
  const E& operator[](size_t i) const { return p[i]; }
};

struct ProveTimes {
  double witness_ms = 0, msm_ms = 0, finish_ms = 0;
  double mask_ms = 0;  // trait path, Rep3: the two host mask draws (or the seed draw) inside witness_ms
  double half_ms = 0;  // trait path, Rep3: to_half_share over the witness (groth16.rs:159-163), between witness_ms and msm_ms
};
inline ProveTimes& last_prove_times() {
  static thread_local ProveTimes t;
  return t;
}
inline double ms_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// scalars uploaded once and shared by several MSMs (the four queries that consume aux_assignment)
// Device buffers are recycled across proofs: hipMalloc / hipFree cost 0.1-1 ms each (hipFree also drains the device),
// a prover asks for the same few sizes every time.
struct DevicePool {
  std::mutex mu;
  std::map<int, std::multimap<size_t, void*>> free_;  // per device: a buffer is only valid on the GPU it was allocated on
  static DevicePool& get() {
    static DevicePool p;
    return p;
  }
  static int device() {
    int d = 0;
    (void)csh_current_device(&d);
    return d;
  }
  void* take(size_t bytes, size_t* cap, int* dev_out) {
    const int dev = device();
    *dev_out = dev;
    {
      std::lock_guard<std::mutex> g(mu);
      auto& fl = free_[dev];
      auto it = fl.lower_bound(bytes);
      if (it != fl.end() && it->first <= 2 * bytes + (size_t(1) << 20)) {
        void* p = it->second;
        *cap = it->first;
        fl.erase(it);
        return p;
      }
    }
    void* p = nullptr;
    check(csh_malloc(&p, bytes), "csh_malloc");
    *cap = bytes;
    return p;
  }
  void give(void* p, size_t cap, int dev) {
    std::lock_guard<std::mutex> g(mu);
    free_[dev].emplace(cap, p);
  }
  void trim() {
    std::lock_guard<std::mutex> g(mu);
    for (auto& d : free_)
      for (auto& kv : d.second) csh_free(kv.second);
    free_.clear();
  }
};

// n field elements (32 * comps bytes each) on the device; the buffer returns to the pool on destruction, so all work
// reading it must have completed (csh_msm_dev is synchronous; async producers are followed by csh_sync).
struct DeviceScalars {
  void* dev = nullptr;
  size_t n = 0;
  size_t cap = 0;
  int device = 0;
  DeviceScalars() = default;
  explicit DeviceScalars(size_t count, size_t comps = 1) : n(count) { dev = DevicePool::get().take(count * comps * 32 + 32, &cap, &device); }
  DeviceScalars(const void* host, size_t count, size_t comps = 1) : DeviceScalars(count, comps) {
    if (count) check(csh_memcpy_h2d(dev, host, count * comps * 32), "csh_memcpy_h2d");
  }
  DeviceScalars(const DeviceScalars&) = delete;
  DeviceScalars& operator=(const DeviceScalars&) = delete;
  DeviceScalars(DeviceScalars&& o) noexcept : dev(o.dev), n(o.n), cap(o.cap), device(o.device) { o.dev = nullptr; }
  ~DeviceScalars() {
    if (dev) DevicePool::get().give(dev, cap, device);
  }
};

// ---- group helpers on the host (Projective = XYZZ internally; results leave as affine) --------------------
template <class F>
using AffineT = csh::Affine<F>;
template <class F>
using Proj = csh::XYZZ<F>;

template <class F>
inline Proj<F> into_group(const AffineT<F>& a) { return Proj<F>::from_affine(a); }
template <class F>
inline AffineT<F> into_affine(const Proj<F>& p) { return csh::xyzz_to_affine(p); }
template <class F>
inline Proj<F> point_add(Proj<F> a, const Proj<F>& b) {
  csh::xyzz_add(a, b);
  return a;
}
template <class F>
inline Proj<F> point_neg(const Proj<F>& a) { return csh::xyzz_neg(a); }
// scalar given in Montgomery form (an Fr element). Round 6: the double-and-add runs on 64-bit limbs (csrc/host_fp64.hpp: the same bytes, one
// __int128 product where the 32-bit-limb code has four): the host multiples of a proof (delta r, delta s, delta_2 s, g_a s, r s_g1, delta rs)
// were ~0.45 ms of a 12 ms prove.
template <class F, class Fr>
inline Proj<F> point_mul(const Proj<F>& p, const Fr& k_mont) {
  using H = typename csh::Host64<F>::type;
  static_assert(sizeof(csh::XYZZ<H>) == sizeof(Proj<F>), "the 64-bit view must alias the 32-bit encoding");
  const Fr k = k_mont.from_mont();
  csh::XYZZ<H> ph, acc = csh::XYZZ<H>::inf();
  memcpy((void*)&ph, &p, sizeof ph);
  bool started = false;
  for (int i = Fr::N - 1; i >= 0; --i)
    for (int b = 31; b >= 0; --b) {
      if (started) acc = csh::xyzz_dbl(acc);
      if ((k.l[i] >> b) & 1) {
        csh::xyzz_add(acc, ph);
        started = true;
      }
    }
  Proj<F> out;
  memcpy((void*)&out, &acc, sizeof out);
  return out;
}

// Device-resident query (uploaded once per proving key) + the host copy for the tiny public-input MSM
template <class F>
struct Query {
  std::vector<AffineT<F>> host;  // full copy, or (synthetic keys) only the first few entries the host reads
  size_t len = 0;                // number of points in the query
  csh_bases_t dev = nullptr;
  size_t size() const { return len ? len : host.size(); }
  // `lead` padding points sit in front of the query on the device (point i of the query is point lead + i of `dev`). The l query is
  // uploaded with lead = 1 + n_public: its handle then has the length of the a / b queries and takes the aux scalars at the same offset, so
  // csh_msm_multi_dev sorts the digits ONCE for all four aux MSMs (a fixed-base table's row stride is the handle's length: a handle of a
  // different length needed its own sort, 0.2 ms of a 2^20 prove). The padding is a copy of the query's first points and is never read.
  size_t lead = 0;
  void upload(csh_curve_t curve, csh_group_t group, size_t lead_pad = 0) {
    len = host.size();
    lead = host.empty() ? 0 : lead_pad;
    if (!lead) {
      check(csh_bases_upload(curve, group, host.data(), host.size(), 0, &dev), "csh_bases_upload");
      return;
    }
    std::vector<AffineT<F>> padded(lead + host.size());
    for (size_t i = 0; i < lead; ++i) padded[i] = host[i % host.size()];
    memcpy((void*)(padded.data() + lead), host.data(), host.size() * sizeof(AffineT<F>));
    check(csh_bases_upload(curve, group, padded.data(), padded.size(), 0, &dev), "csh_bases_upload");
  }
  // straight from a file image (zkey sections hold packed Montgomery little-endian points, the C ABI's own layout):
  // the points go to the device from where they lie; the host keeps only the first `keep_host` entries it reads
  void upload_from(csh_curve_t curve, csh_group_t group, const uint8_t* packed, size_t count, size_t keep_host, bool to_device, size_t lead_pad = 0) {
    len = count;
    const size_t k = to_device ? (keep_host < count ? keep_host : count) : count;
    host.resize(k);
    if (k) memcpy((void*)host.data(), packed, k * sizeof(AffineT<F>));
    lead = to_device && count ? lead_pad : 0;
    if (to_device && !lead) check(csh_bases_upload(curve, group, packed, count, 0, &dev), "csh_bases_upload");
    if (to_device && lead) {
      const size_t pb = sizeof(AffineT<F>);
      std::vector<uint8_t> padded((lead + count) * pb);
      for (size_t i = 0; i < lead; ++i) memcpy(padded.data() + i * pb, packed + (i % count) * pb, pb);
      memcpy(padded.data() + lead * pb, packed, count * pb);
      check(csh_bases_upload(curve, group, padded.data(), lead + count, 0, &dev), "csh_bases_upload");
    }
  }
  void release() {
    if (dev) csh_bases_free(dev);
    dev = nullptr;
  }
};

// The planning half of ProvingKey::place, free of handles and devices (CPU-testable through cog16_placement_plan): which slot takes
// which query (by query: longest-processing-time-first with the measured G2 : G1 cost ratio 2.5; slot 0 = the home GPU), and the
// effective mode (AUTO -> by query up to two slots, by range from three on).
enum { PLACE_AUTO = 0, PLACE_BY_QUERY = 1, PLACE_BY_RANGE = 2 };
enum { Q_A = 0, Q_B1 = 1, Q_B2 = 2, Q_L = 3, Q_H = 4 };
inline int plan_placement(const size_t sizes[5], size_t nslots, int mode, int slot_out[5]) {
  for (int q = 0; q < 5; ++q) slot_out[q] = 0;
  if (nslots < 2) return PLACE_BY_QUERY;
  const int eff = mode == PLACE_AUTO ? (nslots <= 2 ? PLACE_BY_QUERY : PLACE_BY_RANGE) : mode;
  if (eff == PLACE_BY_RANGE) return eff;
  std::vector<double> load(nslots, 0.0);
  int order[5] = {Q_B2, Q_A, Q_B1, Q_L, Q_H};
  auto weight = [&](int q) { return (q == Q_B2 ? 2.5 : 1.0) * (double)sizes[q]; };
  std::stable_sort(order, order + 5, [&](int x, int y) { return weight(x) > weight(y); });
  for (int q : order) {
    if (sizes[q] == 0) continue;
    size_t best = 0;
    for (size_t sl = 1; sl < nslots; ++sl)
      if (load[sl] < load[best]) best = sl;
    load[best] += weight(q);
    slot_out[q] = (int)best;
  }
  return eff;
}
// the k-th of ns contiguous ranges of n entries (by range)
inline void plan_range(size_t n, size_t ns, size_t slot, size_t* lo, size_t* hi) {
  *lo = n / ns * slot + std::min(slot, n % ns);
  *hi = *lo + n / ns + (slot < n % ns ? 1 : 0);
}

template <class P>
struct ProvingKey {
  using G1 = AffineT<typename P::Fq>;
  using G2 = AffineT<typename P::Fq2>;
  G1 alpha_g1, beta_g1, delta_g1;
  G2 beta_g2, gamma_g2, delta_g2;
  Query<typename P::Fq> a_query, b_g1_query, l_query, h_query;
  Query<typename P::Fq2> b_g2_query;
  std::vector<G1> ic;
  // Fixed-base tables on the five queries (csh_bases_precompute_grouped): rows x the key memory on the device, built once per
  // key. With g rows, windows w, w + W', ..., w + (g - 1) W' share a bucket set, so an MSM reduces W' = ceil(W / g) windows and
  // the host folds W' window sums (BN254 2^20, 4 rows: G2 query 4.7 -> 4.3 ms, G1 queries -1..3 %). Round 6: the policy is one row per
  // window with 17- / 20-bit windows (ONE bucket set, 15 / 13 additions per point instead of 17 / 16). COG16_TABLES=0 disables,
  // COG16_TABLES=g sets the row count. All queries get the same (c, rows): csh_msm_multi_dev shares one digit pass over them.
  void build_tables() {
    size_t big = 0;
    for (size_t n : {a_query.size(), b_g1_query.size(), l_query.size(), h_query.size(), b_g2_query.size()}) big = n > big ? n : big;
    // the policy lives in the library (csh_bases_table_policy), shared with the Rust bases cache; profiles/r06_e_policy_*.log
    int c = 0, rows = 0;
    check(csh_bases_table_policy(big, &c, &rows), "csh_bases_table_policy");
    if (const char* e = getenv("COG16_TABLES")) {
      rows = atoi(e);
      if (c > 16 && rows < 16) c = 16;  // fewer rows than windows: the grouped form of rounds 2-5 (c <= 16)
    }
    if (rows < 2 || c == 0) return;
    for (csh_bases_t h : {a_query.dev, b_g1_query.dev, l_query.dev, h_query.dev, b_g2_query.dev}) {
      if (!h) continue;
      const int rc = csh_bases_precompute_grouped(h, c, rows);
      if (rc == CSH_ERR_OOM) {  // tables are an optimisation: a key that does not leave room for them proves from the plain points
        for (csh_bases_t g : {a_query.dev, b_g1_query.dev, l_query.dev, h_query.dev, b_g2_query.dev})
          if (g) (void)csh_bases_drop_tables(g);
        return;
      }
      check(rc, "csh_bases_precompute_grouped");
    }
  }
  // ---- per-commitment device placement ---------------------------------------------------------------------------------
  // The five query MSMs of one proof are independent (rayon_join5, groth16.rs:227-294); on a node with several GPUs each can run
  // on its own device (north star: "independent MSM instances ... per polynomial commitment shard across the 8 GPUs"). `place`
  // assigns the queries to the slots of `devices` (slot 0 = the key's home GPU, where witness map and scalars live; a physical GPU
  // may appear more than once -- tests fold every slot onto one GPU) by longest-processing-time-first with the measured cost
  // ratio of a G2 to a G1 MSM (2.5), and clones each non-home query onto its GPU (csh_bases_clone: device to device, tables
  // included). create_proof_device then ships the scalars by peer copy (32 bytes per entry) and runs the groups concurrently.
  enum { Q_A = cosnarks::Q_A, Q_B1 = cosnarks::Q_B1, Q_B2 = cosnarks::Q_B2, Q_L = cosnarks::Q_L, Q_H = cosnarks::Q_H };
  // Two shapes of the same idea. BY_QUERY: whole queries are assigned to slots (LPT); the G2 query bounds the gain (N >= 3 GPUs:
  // ~2.5 G1-units on its slot). BY_RANGE: every slot takes the k-th contiguous range of EVERY query (the four aux queries of a
  // range still share one digit sort on their slot; the host adds the N partial results per query: N - 1 point additions), so
  // the work per slot is total / N whatever N is, at the price of smaller MSMs per slot. AUTO = BY_QUERY up to two GPUs,
  // BY_RANGE from three on.
  enum { PLACE_AUTO = cosnarks::PLACE_AUTO, PLACE_BY_QUERY = cosnarks::PLACE_BY_QUERY, PLACE_BY_RANGE = cosnarks::PLACE_BY_RANGE };
  struct Placement {
    std::vector<int> devices;
    int mode = PLACE_BY_QUERY;
    int slot[5] = {0, 0, 0, 0, 0};                    // BY_QUERY: the slot of each query
    std::vector<std::array<csh_bases_t, 5>> handles;  // [slot][query]; slot 0 = the home handles (not owned)
    // BY_RANGE: a slot's clone holds only the slot's range of the query: base[slot][q] = index (in the home query) of the clone's
    // first point, count[slot][q] = its length (slot 0 and BY_QUERY clones: 0 and the whole query)
    std::vector<std::array<size_t, 5>> base, count, lead;
  } placement;
  size_t query_size(int q) const {
    return q == Q_A ? a_query.size() : q == Q_B1 ? b_g1_query.size() : q == Q_B2 ? b_g2_query.size() : q == Q_L ? l_query.size() : h_query.size();
  }
  size_t query_lead(int q) const { return q == Q_L ? l_query.lead : 0; }  // Query::lead: only the l query is padded
  csh_bases_t home_handle(int q) const {
    return q == Q_A ? a_query.dev : q == Q_B1 ? b_g1_query.dev : q == Q_B2 ? b_g2_query.dev : q == Q_L ? l_query.dev : h_query.dev;
  }
  bool placed() const { return placement.devices.size() > 1; }
  size_t slots() const { return placed() ? placement.devices.size() : 1; }
  csh_bases_t handle_for(size_t slot, int q) const { return slot == 0 || !placed() ? home_handle(q) : placement.handles[slot][q]; }
  // offset of home-query point `index` inside the handle handle_for(slot, q); throws if the slot's clone does not hold [index, index + n)
  size_t offset_in(size_t slot, int q, size_t index, size_t n) const {
    if (slot == 0 || !placed() || placement.base.empty()) return index + query_lead(q);
    const size_t b = placement.base[slot][q], c = placement.count[slot][q];
    if (index < b || index + n > b + c) throw Error("placement: the slot's clone of the query does not hold the requested range");
    return index - b + placement.lead[slot][q];  // (a whole clone keeps the home handle's padding in front, a range clone has none)
  }
  // does `slot` work on query q, and on which part [lo, hi) of an index space of n entries?
  bool slot_has(size_t slot, int q) const {
    if (!placed()) return slot == 0;
    return placement.mode == PLACE_BY_RANGE ? true : placement.slot[q] == (int)slot;
  }
  void slot_range(size_t slot, size_t n, size_t* lo, size_t* hi) const {
    if (!placed() || placement.mode != PLACE_BY_RANGE) {
      *lo = 0;
      *hi = n;
      return;
    }
    plan_range(n, placement.devices.size(), slot, lo, hi);
  }
  void unplace() {
    for (size_t sl = 1; sl < placement.handles.size(); ++sl)
      for (auto& c : placement.handles[sl])
        if (c) csh_bases_free(c);
    placement.handles.clear();
    placement.base.clear();
    placement.count.clear();
    placement.lead.clear();
    for (auto& sl : placement.slot) sl = 0;
    placement.devices.clear();
  }
  void place(const std::vector<int>& devices, int mode = PLACE_AUTO) {
    unplace();
    if (devices.size() < 2) return;
    const size_t ns = devices.size();
    size_t sizes[5];
    for (int q = 0; q < 5; ++q) sizes[q] = home_handle(q) ? query_size(q) : 0;
    placement.mode = plan_placement(sizes, ns, mode, placement.slot);
    placement.handles.assign(ns, std::array<csh_bases_t, 5>{nullptr, nullptr, nullptr, nullptr, nullptr});
    placement.base.assign(ns, std::array<size_t, 5>{0, 0, 0, 0, 0});
    placement.count.assign(ns, std::array<size_t, 5>{sizes[0], sizes[1], sizes[2], sizes[3], sizes[4]});
    placement.lead.assign(ns, std::array<size_t, 5>{query_lead(0), query_lead(1), query_lead(2), query_lead(3), query_lead(4)});
    for (int q = 0; q < 5; ++q) placement.handles[0][q] = home_handle(q);
    placement.devices = devices;  // (set before the clones: unplace() on a failure below frees what was made)
    // BY_RANGE: slot sl works on the sl-th range of the aux index space (the n_aux = |l_query| private-witness entries; a / b_g1 / b_g2
    // hold 1 + n_public points in front of them, groth16.rs:179-203) and on the sl-th range of h: it gets those points only -- 1/N of the
    // key per GPU instead of N copies of all of it. Keys whose queries do not have that shape are cloned whole.
    const size_t n_aux = sizes[Q_L];
    const bool aux_shape = sizes[Q_A] >= n_aux && sizes[Q_A] == sizes[Q_B1] && sizes[Q_A] == sizes[Q_B2] && n_aux > 0;
    const size_t lead = aux_shape ? sizes[Q_A] - n_aux : 0;
    bool oom = false;
    try {
      for (size_t sl = 1; sl < ns && !oom; ++sl)
        for (int q = 0; q < 5 && !oom; ++q) {
          const bool wanted = placement.mode == PLACE_BY_RANGE ? true : placement.slot[q] == (int)sl;
          if (!wanted || !sizes[q]) continue;
          int rc = CSH_OK;
          if (placement.mode == PLACE_BY_RANGE && aux_shape) {
            size_t lo = 0, hi = 0;
            plan_range(q == Q_H ? sizes[Q_H] : n_aux, ns, sl, &lo, &hi);
            const size_t first = (q == Q_H || q == Q_L ? 0 : lead) + lo;
            placement.base[sl][q] = first;
            placement.count[sl][q] = hi - lo;
            placement.lead[sl][q] = 0;
            if (hi > lo) rc = csh_bases_clone_range(home_handle(q), query_lead(q) + first, hi - lo, devices[sl], &placement.handles[sl][q]);
          } else {
            rc = csh_bases_clone(home_handle(q), devices[sl], &placement.handles[sl][q]);
          }
          if (rc == CSH_ERR_OOM) oom = true;  // does not fit next to what that GPU already holds
          else check(rc, "csh_bases_clone");
        }
    } catch (...) {
      unplace();
      throw;
    }
    if (oom) unplace();  // prove unplaced rather than fail to load the key (build_tables falls back the same way)
  }
  void place_default() {
    int mode = PLACE_AUTO;
    const std::vector<int> d = ProverDevices::get().snapshot(&mode);
    place(d, mode);
  }
  ~ProvingKey() {
    unplace();
    a_query.release();
    b_g1_query.release();
    l_query.release();
    h_query.release();
    b_g2_query.release();
  }
};

template <class P>
struct ConstraintMatrices {
  size_t num_instance_variables = 0;
  size_t num_witness_variables = 0;
  size_t num_constraints = 0;
  std::vector<std::vector<std::pair<typename P::Fr, size_t>>> a, b, c;  // c: only LibSnarkReduction reads it (a zkey has no C)
  // device-resident CSR copies (uploaded once per circuit) for the on-device constraint evaluation
  csh_matrix_t a_dev = nullptr, b_dev = nullptr, c_dev = nullptr;
  void upload() {
    auto up = [&](const std::vector<std::vector<std::pair<typename P::Fr, size_t>>>& m, csh_matrix_t* out) {
      std::vector<uint64_t> row_ptr(m.size() + 1, 0);
      std::vector<uint32_t> col;
      std::vector<typename P::Fr> val;
      for (size_t i = 0; i < m.size(); ++i) {
        for (auto& [c, idx] : m[i]) {
          col.push_back((uint32_t)idx);
          val.push_back(c);
        }
        row_ptr[i + 1] = col.size();
      }
      check(csh_matrix_upload(P::ID, row_ptr.data(), col.data(), (const uint64_t*)val.data(), m.size(), col.size(), out), "csh_matrix_upload");
    };
    up(a, &a_dev);
    up(b, &b_dev);
    if (!c.empty()) up(c, &c_dev);
  }
  ConstraintMatrices() = default;
  ConstraintMatrices(const ConstraintMatrices&) = delete;
  ConstraintMatrices& operator=(const ConstraintMatrices&) = delete;
  ~ConstraintMatrices() {
    if (a_dev) csh_matrix_free(a_dev);
    if (b_dev) csh_matrix_free(b_dev);
    if (c_dev) csh_matrix_free(c_dev);
  }
};

template <class P>
struct Proof {
  AffineT<typename P::Fq> a, c;
  AffineT<typename P::Fq2> b;
};

template <class P, class Share>
struct SharedWitness {
  std::vector<typename P::Fr> public_inputs;  // includes the constant 1
  std::vector<Share> witness;
};

// ---- decimal printing (proof JSON in the circom.proof schema) -----------------------------------------------
template <class F>
inline std::string to_decimal(const F& mont) {
  F c = mont.from_mont();
  uint32_t w[F::N];
  for (int i = 0; i < F::N; ++i) w[i] = c.l[i];
  std::string out;
  bool nz = true;
  while (nz) {
    uint64_t rem = 0;
    nz = false;
    for (int i = F::N - 1; i >= 0; --i) {
      uint64_t cur = (rem << 32) | w[i];
      w[i] = (uint32_t)(cur / 1000000000u);
      rem = cur % 1000000000u;
      if (w[i]) nz = true;
    }
    char buf[16];
    snprintf(buf, sizeof buf, nz ? "%09u" : "%u", (unsigned)rem);
    out = std::string(buf) + out;
  }
  return out;
}

template <class P>
inline std::string proof_to_json(const Proof<P>& pr) {
  auto g1 = [](const AffineT<typename P::Fq>& p) {
    if (p.is_inf()) return std::string("[\"0\", \"1\", \"0\"]");
    return "[\"" + to_decimal(p.x) + "\", \"" + to_decimal(p.y) + "\", \"1\"]";
  };
  std::string b;
  if (pr.b.is_inf())
    b = "[[\"0\", \"0\"], [\"1\", \"0\"], [\"0\", \"0\"]]";
  else
    b = "[[\"" + to_decimal(pr.b.x.c0) + "\", \"" + to_decimal(pr.b.x.c1) + "\"], [\"" + to_decimal(pr.b.y.c0) + "\", \"" +
        to_decimal(pr.b.y.c1) + "\"], [\"1\", \"0\"]]";
  return "{\"pi_a\": " + g1(pr.a) + ", \"pi_b\": " + b + ", \"pi_c\": " + g1(pr.c) +
         ", \"protocol\": \"groth16\", \"curve\": \"" + P::name() + "\"}";
}

}  // namespace cosnarks
