// C entry points of the host mirror (for the Python tests / bench and as a template for the Rust shim):
//   cog16_prove_plain  == Groth16::<P>::plain_prove::<CircomReduction>            (groth16.rs:484-490)
//   cog16_prove_rep3   == three parties in one process, each running Rep3CoGroth16::prove::<_, CircomReduction>
//                         (groth16.rs:360-379) over LocalNetwork pairs -- the layout of the reference's own
//                         e2e test (tests/tests/circom/e2e_tests/rep3.rs:57-69): asserts the three proofs agree.
// curve: 0 BN254, 1 BLS12-381. r/s: canonical little-endian 4 x u64, or NULL to draw them with T::rand.
#include <atomic>
#include <chrono>
#include <random>

#include "groth16.hpp"
#include "arkwire.hpp"
#include "sharefile.hpp"
#include "plonk_honk.hpp"
#include "zkey.hpp"

#include <malloc.h>
#include <sys/random.h>

namespace {
// The prover's large host vectors (32 MB masks, h, half shares at 2^20) are malloc'ed and freed once per proof; glibc serves blocks of
// that size with mmap and returns them with munmap, so every proof pays tens of thousands of page faults and two or three multi-ms
// unmaps that stall every other thread of the process behind the address-space lock (round 6: freeing the two mask vectors of one Rep3
// witness map cost ~8 ms on the calling thread; moved to a background thread the same cost reappeared in the page faults of the next
// allocation, profiles/r06_m_retain_ab.log). The mirror therefore asks glibc to keep such blocks in its heap (no mmap per block, no
// trim): what jemalloc / mimalloc -- the allocators a Rust prover usually links -- do by default. Measured at 2^20 (same log): plain
// trait-path prove 16.8-18.2 -> 14.7-14.9 ms, a seeded Rep3 party 19.2 -> 16.8-17.0 ms, three of them on one GPU 64 -> 40-47 ms; the
// host-mask party 40-42 -> 39 ms (its mask draw gets SLOWER on retained pages, 12 -> 20 ms, while everything around it gets faster).
// COG16_MALLOC_RETAIN=0 leaves glibc alone.
struct MallocRetain {
  MallocRetain() {
    const char* e = getenv("COG16_MALLOC_RETAIN");
    if (e && atoi(e) == 0) return;
    (void)mallopt(M_MMAP_THRESHOLD, 1 << 30);
    (void)mallopt(M_TRIM_THRESHOLD, (int)((1u << 31) - 1));
    (void)mallopt(M_TOP_PAD, 64 << 20);
  }
} g_malloc_retain;
}  // namespace

namespace cosnarks {
void secure_random_bytes(void* out, size_t n) {
  uint8_t* p = static_cast<uint8_t*>(out);
  size_t got = 0;
  while (got < n) {
    const ssize_t r = getrandom(p + got, n - got, 0);
    if (r > 0) {
      got += (size_t)r;
      continue;
    }
    break;
  }
  if (got < n) {  // pre-3.17 kernels / seccomp without getrandom
    FILE* f = fopen("/dev/urandom", "rb");
    if (f) {
      got += fread(p + got, 1, n - got, f);
      fclose(f);
    }
  }
  if (got < n) throw Error("no OS entropy source (getrandom and /dev/urandom both failed)");
}
}  // namespace cosnarks

using namespace cosnarks;

namespace {

thread_local std::string g_err;

template <class P>
typename P::Fr fr_from_canonical(const uint64_t v[4]) {
  typename P::Fr f;
  memcpy(&f, v, 32);
  return f.to_mont();
}

// the root cause first: parties that merely saw the abort of a failing peer report "network aborted"
void throw_first_party_error(const std::string* errs, int n) {
  for (int pass = 0; pass < 2; ++pass)
    for (int p = 0; p < n; ++p)
      if (!errs[p].empty() && (pass == 1 || errs[p].find("network aborted") == std::string::npos))
        throw Error("party " + std::to_string(p) + ": " + errs[p]);
}

int write_out(const std::string& json, char* out, size_t cap) {
  if (json.size() + 1 > cap) {
    g_err = "output buffer too small";
    return -1;
  }
  memcpy(out, json.c_str(), json.size() + 1);
  return 0;
}

template <class P>
int prove_plain_t(const uint8_t* zkey, size_t zlen, const uint8_t* wtns, size_t wlen, const uint64_t* r, const uint64_t* s, char* out,
                  size_t cap, uint64_t* h_out, size_t h_cap) {
  using T = PlainGroth16Driver<P>;
  using Fr = typename P::Fr;
  ProvingKey<P> pk;
  ConstraintMatrices<P> m;
  parse_zkey<P>(zkey, zlen, pk, m);
  UnitState st0, st1;
  Fr rr, ss;
  if (r) rr = fr_from_canonical<P>(r);
  if (s) ss = fr_from_canonical<P>(s);
  std::vector<Fr> h;
  Proof<P> pr;
  if (getenv("COG16_HOST_WTNS") || trait_path_flag().load()) {  // witness values converted on the host, SharedWitness as in the reference's CLI
    std::vector<Fr> w = parse_wtns<P>(wtns, wlen);
    SharedWitness<P, Fr> sw;
    sw.public_inputs.assign(w.begin(), w.begin() + m.num_instance_variables);
    sw.witness.assign(w.begin() + m.num_instance_variables, w.end());
    pr = CoGroth16<P, T>::template prove_inner<CircomReduction>(nullptr, nullptr, st0, st1, pk, m, sw, r ? &rr : nullptr, s ? &ss : nullptr, &h);
  } else {
    // zkey queries and the wtns values go to the device straight from the file images; only the public
    // inputs (a handful of values) are decoded on the host
    std::vector<Fr> pub = parse_wtns_prefix<P>(wtns, wlen, m.num_instance_variables);
    const DeviceScalars wit_dev = parse_wtns_to_device<P>(wtns, wlen, m.num_instance_variables);
    pr = CoGroth16<P, T>::template prove_inner_device_witness<CircomReduction>(nullptr, nullptr, st0, st1, pk, m, pub, wit_dev, r ? &rr : nullptr,
                                                                               s ? &ss : nullptr, &h);
  }
  if (h_out) {
    if (h.size() > h_cap) throw Error("h_out too small");
    memcpy(h_out, h.data(), h.size() * 32);
  }
  return write_out(proof_to_json(pr), out, cap);
}

// share_field_elements (mpc-core/src/protocols/rep3.rs:281-292, 375-389) with a seeded RNG
template <class Fr>
struct SeededSharer {
  ShareRng gen;  // seed 0 = OS entropy; a non-zero seed is for reproducible tests only
  explicit SeededSharer(uint64_t seed, uint64_t domain = 1) : gen(seed, domain) {}
  Fr rnd() {
    uint8_t b[32];
    for (int i = 0; i < 4; ++i) {
      uint64_t v = gen();
      memcpy(b + 8 * i, &v, 8);
    }
    return from_be_bytes_mod_order<Fr>(b);
  }
  void share(const Fr& val, Rep3PrimeFieldShare<Fr> out3[3]) {
    Fr a = rnd(), b = rnd();
    Fr c = Fr::sub(Fr::sub(val, a), b);
    out3[0] = {a, c};
    out3[1] = {b, a};
    out3[2] = {c, b};
  }
};

// CompressedRep3SharedWitness::share_rep3 (co-circom-types/src/lib.rs:279-333). compression as the reference's enum
// (lib.rs:150-161): 0 None (replicated), 1 HalfShares (additive), 2 SeededShares, 3 SeededHalfShares -- the seeded
// levels keep one explicit share vector and two ChaCha12 seeds (rep3.rs:455-533).
template <class P>
void split_witness_rep3(const std::vector<typename P::Fr>& w, size_t npub, int compression, uint64_t seed,
                        sharefile::CompressedRep3SharedWitness<P> out[3]) {
  using Fr = typename P::Fr;
  if (npub > w.size()) throw Error("num_inputs exceeds the witness length");
  if (compression < 0 || compression > 3) throw Error("compression must be 0 (none), 1 (half shares), 2 (seeded shares) or 3 (seeded half shares)");
  SeededSharer<Fr> sh(seed);
  static const sharefile::Rep3Variant kinds[4] = {sharefile::REPLICATED, sharefile::ADDITIVE, sharefile::SEEDED_REPLICATED, sharefile::SEEDED_ADDITIVE};
  for (int p = 0; p < 3; ++p) {
    out[p].public_inputs.assign(w.begin(), w.begin() + npub);
    out[p].kind = kinds[compression];
  }
  if (compression >= 2) {
    sharefile::SeededShare<Fr> a, b, c;
    b.is_seed = c.is_seed = true;
    b.len = c.len = w.size() - npub;
    for (int i = 0; i < 4; ++i) {
      uint64_t v = sh.gen(), u = sh.gen();
      memcpy(b.seed + 8 * i, &v, 8);
      memcpy(c.seed + 8 * i, &u, 8);
    }
    const std::vector<Fr> bv = b.expand(), cv = c.expand();
    for (size_t i = npub; i < w.size(); ++i) a.shares.push_back(Fr::sub(Fr::sub(w[i], bv[i - npub]), cv[i - npub]));
    if (compression == 3) {  // share_field_elements_additive_seeded: [a, b, c]
      out[0].sa = a;
      out[1].sa = b;
      out[2].sa = c;
    } else {  // share_field_elements_seeded: {a, c}, {b, a}, {c, b}
      out[0].sa = a; out[0].sb = c;
      out[1].sa = b; out[1].sb = a;
      out[2].sa = c; out[2].sb = b;
    }
    return;
  }
  for (size_t i = npub; i < w.size(); ++i) {
    Rep3PrimeFieldShare<Fr> t[3];
    sh.share(w[i], t);
    for (int p = 0; p < 3; ++p) {
      if (compression == 0) out[p].replicated.push_back(t[p]);
      else out[p].additive.push_back(t[p].a);
    }
  }
}

// Rep3CoGroth16::prove (groth16.rs:360-379) with three in-process parties. `shares[p]` is what party p read from its
// `.shared` file (co-circom.rs:1014-1016); additive half shares are completed inside the party's thread.
template <class P>
int prove_rep3_core(const uint8_t* zkey, size_t zlen, sharefile::CompressedRep3SharedWitness<P> shares[3], uint64_t seed, const uint64_t* r,
                    const uint64_t* s, char* out, size_t cap, uint64_t* h_shares_out, size_t h_cap) {
  using T = Rep3Groth16Driver<P>;
  using Fr = typename P::Fr;
  using Share = Rep3PrimeFieldShare<Fr>;
  int ndev = 1;
  csh_device_count(&ndev);
  // One GPU per party when the node has them (BASELINE config 4): device memory is not shared between GPUs, so every
  // party then holds its own copy of the proving key and matrices on its device; on one GPU a single copy serves all.
  const bool per_party_keys = ndev > 1;
  ProvingKey<P> pk_shared;
  ConstraintMatrices<P> m_shared;
  parse_zkey<P>(zkey, zlen, pk_shared, m_shared, /*upload=*/!per_party_keys);
  // checked up front for all parties: a party that bails out alone would leave the other two waiting on the network
  for (int p = 0; p < 3; ++p) {
    if (shares[p].public_inputs.size() != m_shared.num_instance_variables) throw Error("witness share: public input count does not match the proving key");
    if (shares[p].length() != m_shared.num_witness_variables) throw Error("witness share: the amount of private witness variables does not match the proving key");
  }
  SeededSharer<Fr> rs_sharer(seed, /*domain=*/11);
  Share r3[3], s3[3];
  if (r) rs_sharer.share(fr_from_canonical<P>(r), r3);
  if (s) rs_sharer.share(fr_from_canonical<P>(s), s3);
  auto nets0 = LocalNetwork::new_parties(3), nets1 = LocalNetwork::new_parties(3);
  Proof<P> proofs[3];
  std::vector<Fr> hs[3];
  std::string errs[3];
  std::vector<std::thread> th;
  for (int p = 0; p < 3; ++p) {
    th.emplace_back([&, p] {
      try {
        check(csh_init(ndev > 0 ? p % ndev : 0), "csh_init");  // one GPU per party when the node has them (BASELINE config 4)
        ProvingKey<P> pk_own;
        ConstraintMatrices<P> m_own;
        if (per_party_keys) parse_zkey<P>(zkey, zlen, pk_own, m_own);  // uploads onto this thread's device
        const ProvingKey<P>& pk = per_party_keys ? pk_own : pk_shared;
        const ConstraintMatrices<P>& m = per_party_keys ? m_own : m_shared;
        uint8_t my_seed[32];
        ShareRng(seed, 100 + p).fill(my_seed, 32);  // Rep3State::new draws it from the OS rng (rep3.rs:56-76); seed != 0: tests
        Rep3State state0 = Rep3State::create(nets0[p], my_seed);  // groth16.rs:371
        Rep3State state1 = state0.fork(0);                         // :372
        SharedWitness<P, Share> sw = sharefile::uncompress<P>(std::move(shares[p]), nets0[p]);  // co-circom.rs:1016
        proofs[p] = CoGroth16<P, T>::template prove_inner<CircomReduction>(&nets0[p], &nets1[p], state0, state1, pk, m, sw,
                                                                            r ? &r3[p] : nullptr, s ? &s3[p] : nullptr, &hs[p]);
      } catch (const std::exception& e) {
        errs[p] = e.what();
        nets0[p].abort();  // let the other parties unwind instead of waiting on this one
        nets1[p].abort();  // let the other parties unwind instead of waiting on this one
      }
    });
  }
  for (auto& t : th) t.join();
  throw_first_party_error(errs, 3);
  std::string j0 = proof_to_json(proofs[0]);
  if (j0 != proof_to_json(proofs[1]) || j0 != proof_to_json(proofs[2])) throw Error("the three parties disagree on the proof");
  if (h_shares_out) {
    const size_t n = hs[0].size();
    if (3 * n > h_cap) throw Error("h_shares_out too small");
    for (int p = 0; p < 3; ++p) memcpy(h_shares_out + 4 * n * p, hs[p].data(), 32 * n);
  }
  return write_out(j0, out, cap);
}

template <class P>
int prove_rep3_t(const uint8_t* zkey, size_t zlen, const uint8_t* wtns, size_t wlen, uint64_t seed, const uint64_t* r, const uint64_t* s,
                 char* out, size_t cap, uint64_t* h_shares_out, size_t h_cap) {
  std::vector<typename P::Fr> w = parse_wtns<P>(wtns, wlen);
  sharefile::CompressedRep3SharedWitness<P> shares[3];
  split_witness_rep3<P>(w, zkey_num_instance_variables<P>(zkey, zlen), 0, seed, shares);
  return prove_rep3_core<P>(zkey, zlen, shares, seed, r, s, out, cap, h_shares_out, h_cap);
}

// shamir::share (shamir.rs:359-376): random degree-`deg` polynomial, shares = evaluations at 1..n
template <class Fr>
struct SeededShamirSharer {
  ShareRng gen;  // seed 0 = OS entropy; a non-zero seed is for reproducible tests only
  int n;
  SeededShamirSharer(uint64_t seed, int parties, uint64_t domain = 2) : gen(seed, domain), n(parties) {}
  Fr rnd() {
    uint8_t b[32];
    for (int i = 0; i < 4; ++i) {
      uint64_t v = gen();
      memcpy(b + 8 * i, &v, 8);
    }
    return from_be_bytes_mod_order<Fr>(b);
  }
  std::vector<Fr> share(const Fr& secret, int deg) {
    std::vector<Fr> coeffs{secret};
    for (int i = 0; i < deg; ++i) coeffs.push_back(rnd());
    std::vector<Fr> out(n);
    for (int p = 0; p < n; ++p) {
      Fr x = Fr::from_u64(p + 1), e = Fr::zero();
      for (size_t k = coeffs.size(); k-- > 0;) e = Fr::add(Fr::mul(e, x), coeffs[k]);
      out[p] = e;
    }
    return out;
  }
};

// SharedWitness::share_shamir (co-circom-types/src/lib.rs:362-381)
template <class P>
std::vector<SharedWitness<P, typename P::Fr>> split_witness_shamir(const std::vector<typename P::Fr>& w, size_t npub, int t, int n, uint64_t seed) {
  using Fr = typename P::Fr;
  if (npub > w.size()) throw Error("num_inputs exceeds the witness length");
  if (n < 2 * t + 1 || t < 1) throw Error("num_parties must be at least 2 * threshold + 1");
  SeededShamirSharer<Fr> sh(seed, n);
  std::vector<SharedWitness<P, Fr>> sw(n);
  for (int p = 0; p < n; ++p) sw[p].public_inputs.assign(w.begin(), w.begin() + npub);
  for (size_t i = npub; i < w.size(); ++i) {
    auto v = sh.share(w[i], t);
    for (int p = 0; p < n; ++p) sw[p].witness.push_back(v[p]);
  }
  return sw;
}

// ShamirCoGroth16::prove (groth16.rs:439-463) with n in-process parties, each holding the share it read from its
// `.shared` file (co-circom.rs:1031-1035). Preprocessing (ShamirPreprocessing::new, DN07 double sharings) is replaced by a
// dealer that hands every party its (degree-t, degree-2t) share pairs; with r/s given, the first two pairs share exactly
// r and s.
template <class P>
int prove_shamir_core(const uint8_t* zkey, size_t zlen, std::vector<SharedWitness<P, typename P::Fr>>& sw, int t, uint64_t seed, const uint64_t* r,
                      const uint64_t* s, char* out, size_t cap) {
  using T = ShamirGroth16Driver<P>;
  using Fr = typename P::Fr;
  const int n = (int)sw.size();
  if (n < 2 * t + 1 || t < 1) throw Error("num_parties must be at least 2 * threshold + 1");
  int ndev = 1;
  csh_device_count(&ndev);
  // device memory is not shared between GPUs: with one GPU per party every party uploads its own copy of the proving key
  // and matrices onto its device (as prove_rep3_core does); on one GPU a single copy serves all
  const bool per_party_keys = ndev > 1;
  ProvingKey<P> pk_shared;
  ConstraintMatrices<P> m_shared;
  parse_zkey<P>(zkey, zlen, pk_shared, m_shared, /*upload=*/!per_party_keys);
  for (auto& w : sw) {
    if (w.public_inputs.size() != m_shared.num_instance_variables) throw Error("witness share: public input count does not match the proving key");
    if (w.witness.size() != m_shared.num_witness_variables) throw Error("witness share: the amount of private witness variables does not match the proving key");
  }
  SeededShamirSharer<Fr> dealer(seed, n, /*domain=*/12);
  // dealer: three double sharings per party (two rand calls + one scalar_mul: groth16.rs:448-449)
  std::vector<std::deque<std::pair<Fr, Fr>>> pairs(n);
  for (int k = 0; k < 3; ++k) {
    Fr v = dealer.rnd();
    if (k == 0 && r) v = fr_from_canonical<P>(r);
    if (k == 1 && s) v = fr_from_canonical<P>(s);
    auto st = dealer.share(v, t), s2t = dealer.share(v, 2 * t);
    for (int p = 0; p < n; ++p) pairs[p].push_back({st[p], s2t[p]});
  }
  auto nets0 = LocalNetwork::new_parties(n), nets1 = LocalNetwork::new_parties(n);
  std::vector<Proof<P>> proofs(n);
  std::vector<std::string> errs(n);
  std::vector<std::thread> th;
  for (int p = 0; p < n; ++p) {
    th.emplace_back([&, p] {
      try {
        check(csh_init(ndev > 0 ? p % ndev : 0), "csh_init");
        ProvingKey<P> pk_own;
        ConstraintMatrices<P> m_own;
        if (per_party_keys) parse_zkey<P>(zkey, zlen, pk_own, m_own);  // uploads onto this thread's device
        const ProvingKey<P>& pk = per_party_keys ? pk_own : pk_shared;
        const ConstraintMatrices<P>& m = per_party_keys ? m_own : m_shared;
        auto state0 = ShamirState<Fr>::create(p, n, t, pairs[p]);
        auto state1 = state0.fork(1);
        proofs[p] = CoGroth16<P, T>::template prove_inner<CircomReduction>(&nets0[p], &nets1[p], state0, state1, pk, m, sw[p], nullptr, nullptr);
      } catch (const std::exception& e) {
        errs[p] = e.what();
        nets0[p].abort();  // let the other parties unwind instead of waiting on this one
        nets1[p].abort();  // let the other parties unwind instead of waiting on this one
      }
    });
  }
  for (auto& x : th) x.join();
  throw_first_party_error(errs.data(), n);
  std::string j0 = proof_to_json(proofs[0]);
  for (int p = 1; p < n; ++p)
    if (j0 != proof_to_json(proofs[p])) throw Error("the parties disagree on the proof");
  return write_out(j0, out, cap);
}

// From a plain witness: ShamirCoGroth16::prove after share_shamir, or Rep3CoGroth16::prove_with_shamir_bridge
// (groth16.rs:394-417) when `bridge` is set.
template <class P>
int prove_shamir_t(const uint8_t* zkey, size_t zlen, const uint8_t* wtns, size_t wlen, int n, int t, uint64_t seed, const uint64_t* r,
                   const uint64_t* s, int bridge, char* out, size_t cap) {
  using Fr = typename P::Fr;
  if (n < 2 * t + 1 || t < 1) throw Error("num_parties must be at least 2 * threshold + 1");
  if (bridge && (n != 3 || t != 1)) throw Error("the Rep3 -> Shamir bridge is the 3-party, threshold-1 case");
  std::vector<Fr> w = parse_wtns<P>(wtns, wlen);
  const size_t npub = zkey_num_instance_variables<P>(zkey, zlen);
  if (!bridge) {
    auto sw = split_witness_shamir<P>(w, npub, t, n, seed);
    return prove_shamir_core<P>(zkey, zlen, sw, t, seed, r, s, out, cap);
  }
  if (npub > w.size()) throw Error("num_inputs exceeds the witness length");
  // Rep3 shares first (rep3.rs:281-292), then translate_primefield_repshare_vec on the device (bridges/rep3_to_shamir.rs:43-62)
  SeededSharer<Fr> sh(seed);
  std::vector<SharedWitness<P, Fr>> sw(3);
  std::vector<Rep3PrimeFieldShare<Fr>> rs[3];
  for (size_t i = npub; i < w.size(); ++i) {
    Rep3PrimeFieldShare<Fr> t3[3];
    sh.share(w[i], t3);
    for (int p = 0; p < 3; ++p) rs[p].push_back(t3[p]);
  }
  for (int p = 0; p < 3; ++p) {
    sw[p].public_inputs.assign(w.begin(), w.begin() + npub);
    const uint64_t e = p + 1, z1 = p == 0 ? 3 : p, z2 = p == 2 ? 1 : p + 2;  // get_translation_points (:14-28): f(X) = 1 - X/z
    Fr x = Fr::sub(Fr::one(), Fr::mul(Fr::from_u64(e), Fr::inv(Fr::from_u64(z1))));
    Fr y = Fr::sub(Fr::one(), Fr::mul(Fr::from_u64(e), Fr::inv(Fr::from_u64(z2))));
    sw[p].witness.resize(rs[p].size());
    check(csh_rep3_to_shamir_vec(P::ID, (const uint64_t*)rs[p].data(), (const uint64_t*)&x, (const uint64_t*)&y, (uint64_t*)sw[p].witness.data(),
                                 rs[p].size()), "csh_rep3_to_shamir_vec");
  }
  return prove_shamir_core<P>(zkey, zlen, sw, t, seed, r, s, out, cap);
}

// ---- synthetic large circuit with a known-trapdoor-style key: every query point is
// k_i * G with k_i = splitmix64(seed + i) | 1, so A, B, C have closed-form discrete logs and are checked with
// three scalar multiplications instead of a pairing. Constraints: w[j+1] * w[j+2] = w[j+3].
template <class F>
void synth_query(csh_curve_t curve, csh_group_t group, uint64_t seed, size_t n, Query<F>& q, size_t lead = 0) {
  void* dev = nullptr;
  const size_t pb = sizeof(AffineT<F>);
  if (lead > n) lead = 0;
  check(csh_malloc(&dev, (lead + n) * pb), "csh_malloc");
  char* pts = static_cast<char*>(dev) + lead * pb;
  check(csh_util_generate_bases_dev(curve, group, seed, n, pts, nullptr), "csh_util_generate_bases_dev");
  check(csh_sync(nullptr), "csh_sync");
  int cur = 0;
  check(csh_current_device(&cur), "csh_current_device");
  if (lead) check(csh_memcpy_peer(dev, cur, pts, cur, lead * pb, nullptr), "csh_memcpy_peer");  // Query::lead: padding = the first points again
  check(csh_sync(nullptr), "csh_sync");
  check(csh_bases_upload_dev(curve, group, dev, lead + n, 0, nullptr, &q.dev), "csh_bases_upload_dev");
  q.len = n;
  q.lead = lead;
  q.host.resize(n < 4 ? n : 4);
  check(csh_memcpy_d2h(q.host.data(), pts, q.host.size() * sizeof(AffineT<F>)), "csh_memcpy_d2h");
  csh_free(dev);
}

template <class Fr>
struct Rep3Sharer {
  ShareRng gen;
  explicit Rep3Sharer(uint64_t seed) : gen(seed, 3) {}
  Fr rnd() {
    uint8_t b[32];
    for (int i = 0; i < 4; ++i) {
      uint64_t v = gen();
      memcpy(b + 8 * i, &v, 8);
    }
    return from_be_bytes_mod_order<Fr>(b);
  }
  void share(const std::vector<Fr>& vals, std::vector<Rep3PrimeFieldShare<Fr>> out[3]) {
    for (auto& v : vals) {
      Fr a = rnd(), b = rnd();
      Fr c = Fr::sub(Fr::sub(v, a), b);
      out[0].push_back({a, c});
      out[1].push_back({b, a});
      out[2].push_back({c, b});
    }
  }
};

// The synthetic circuit with its known-dlog proving key resident on the device: built once, proved many times (bench.py times
// K calls of prove() between barriers), checked against the closed form.
struct SynthBase {
  virtual ~SynthBase() = default;
  virtual void prove(bool want_h) = 0;
  virtual bool closed_form() = 0;
  ProveTimes phases;
  double key_ms = 0;
};
template <class P>
struct SynthCircuit : SynthBase {
  using T = PlainGroth16Driver<P>;
  using Fr = typename P::Fr;
  using Fq = typename P::Fq;
  using Fq2 = typename P::Fq2;
  static constexpr uint64_t SA = 0x1000000000ull, SB1 = 0x2000000000ull, SB2 = 0x3000000000ull, SL = 0x4000000000ull, SH = 0x5000000000ull;
  static constexpr uint64_t d_alpha = 0x1111, d_beta = 0x2222, d_delta = 0x3333;
  size_t domain, nc, n_vars;
  AffineT<Fq> g1;
  AffineT<Fq2> g2;
  ProvingKey<P> pk;
  ConstraintMatrices<P> m;
  std::vector<Fr> w, h;
  SharedWitness<P, Fr> sw;
  Fr r, s;
  UnitState st0, st1;
  Proof<P> proof;
  SynthCircuit(int log_domain, const uint32_t* g1_words, const uint32_t* g2_words) {
    domain = size_t(1) << log_domain;
    nc = domain - 2;
    n_vars = nc + 3;
    memcpy(&g1, g1_words, sizeof g1);
    memcpy(&g2, g2_words, sizeof g2);
    auto t0 = std::chrono::steady_clock::now();
    synth_query<Fq>(P::ID, CSH_G1, SA, n_vars, pk.a_query);
    synth_query<Fq>(P::ID, CSH_G1, SB1, n_vars, pk.b_g1_query);
    synth_query<Fq2>(P::ID, CSH_G2, SB2, n_vars, pk.b_g2_query);
    synth_query<Fq>(P::ID, CSH_G1, SL, n_vars - 2, pk.l_query, 2);
    synth_query<Fq>(P::ID, CSH_G1, SH, domain, pk.h_query);
    pk.build_tables();
    pk.place_default();
    auto kG1 = [&](uint64_t k) { return into_affine(point_mul(into_group(g1), Fr::from_u64(k))); };
    auto kG2 = [&](uint64_t k) { return into_affine(point_mul(into_group(g2), Fr::from_u64(k))); };
    pk.alpha_g1 = kG1(d_alpha);
    pk.beta_g1 = kG1(d_beta);
    pk.beta_g2 = kG2(d_beta);
    pk.delta_g1 = kG1(d_delta);
    pk.delta_g2 = kG2(d_delta);
    key_ms = ms_since(t0);
    m.num_instance_variables = 2;
    m.num_witness_variables = n_vars - 2;
    m.num_constraints = nc;
    m.a.resize(nc);
    m.b.resize(nc);
    const Fr one = Fr::one();
    for (size_t j = 0; j < nc; ++j) {
      m.a[j].push_back({one, j + 1});
      m.b[j].push_back({one, j + 2});
    }
    m.upload();
    w.resize(n_vars);
    w[0] = one;
    w[1] = Fr::from_u64(3);
    w[2] = Fr::from_u64(5);
    for (size_t j = 0; j < nc; ++j) w[j + 3] = Fr::mul(w[j + 1], w[j + 2]);
    sw.public_inputs.assign(w.begin(), w.begin() + 2);
    sw.witness.assign(w.begin() + 2, w.end());
    r = Fr::from_u64(123456789);
    s = Fr::from_u64(987654321);
  }
  // h is only fetched (for the closed-form check) when asked: a prover does not need it on the host
  void prove(bool want_h) override {
    proof = CoGroth16<P, T>::template prove_inner<CircomReduction>(nullptr, nullptr, st0, st1, pk, m, sw, &r, &s, want_h ? &h : nullptr);
    phases = last_prove_times();
  }
  static bool eq1(const AffineT<Fq>& x, const AffineT<Fq>& y) { return x.x == y.x && x.y == y.y; }
  bool closed_form() override {
    if (h.size() != domain) prove(true);
    auto dl = [](uint64_t seed, size_t i) { return Fr::from_u64(csh_util_splitmix64(seed + i) | 1ull); };
    Fr sa = Fr::zero(), sb1 = Fr::zero(), sb2 = Fr::zero(), sl = Fr::zero(), sh = Fr::zero();
    for (size_t i = 0; i < n_vars; ++i) {
      sa = Fr::add(sa, Fr::mul(w[i], dl(SA, i)));
      sb1 = Fr::add(sb1, Fr::mul(w[i], dl(SB1, i)));
      sb2 = Fr::add(sb2, Fr::mul(w[i], dl(SB2, i)));
    }
    for (size_t j = 0; j + 2 < n_vars; ++j) sl = Fr::add(sl, Fr::mul(w[j + 2], dl(SL, j)));
    for (size_t i = 0; i < domain; ++i) sh = Fr::add(sh, Fr::mul(h[i], dl(SH, i)));
    const Fr dd = Fr::from_u64(d_delta);
    Fr dA = Fr::add(Fr::add(Fr::mul(r, dd), Fr::from_u64(d_alpha)), sa);
    Fr dB1 = Fr::add(Fr::add(Fr::mul(s, dd), Fr::from_u64(d_beta)), sb1);
    Fr dB2 = Fr::add(Fr::add(Fr::mul(s, dd), Fr::from_u64(d_beta)), sb2);
    Fr dC = Fr::add(Fr::add(Fr::sub(Fr::add(Fr::mul(s, dA), Fr::mul(r, dB1)), Fr::mul(Fr::mul(r, s), dd)), sl), sh);
    AffineT<Fq> wa = into_affine(point_mul(into_group(g1), dA)), wc = into_affine(point_mul(into_group(g1), dC));
    AffineT<Fq2> wb = into_affine(point_mul(into_group(g2), dB2));
    return eq1(wa, proof.a) && eq1(wc, proof.c) && wb.x == proof.b.x && wb.y == proof.b.y;
  }
};

// median of a sample and the index of the element that realises it (the phases reported beside a median are those of THAT run)
static size_t median_index(const std::vector<double>& v) {
  std::vector<size_t> idx(v.size());
  for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
  std::sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return v[x] < v[y]; });
  return idx[idx.size() / 2];
}

// Every figure is the MEDIAN of `iters` runs after warm-up runs (median of >= 20; rounds 1-4 reported the minimum, which hid
// a 17.7 -> 40 ms spread on one box, profiles/archive/r04_zd_trait_modes.log); *_min entries keep the minimum beside it.
// rep3_trait_out (nullable, 2 x 13 doubles: [0..12] host masks = the shim's default, [13..25] seeded device masks = its opt-in
// all-GPU-parties mode): {three parties on this GPU: wall ms median, min; party 0 of the median run: mask draw, witness map (incl. masks),
// to_half_share, five MSMs, finish; proofs equal the plain proof (1 / 0); ONE party alone on the GPU, no peers (witness map + to_half_share
// + five MSMs, the finish's three curve points left out): ms median, mask draw, witness map, to_half_share, five MSMs}.
template <class P>
int bench_synth_t(int log_domain, int iters, double* out_ms, int* check_ok, const uint32_t* g1_words, const uint32_t* g2_words, bool with_rep3,
                  double* phases_out = nullptr, double* trait_out = nullptr, double* rep3_trait_out = nullptr, double* mins_out = nullptr) {
  using T = PlainGroth16Driver<P>;
  using Fr = typename P::Fr;
  using Fq = typename P::Fq;
  // COG16_LEAK_BENCH_CIRCUIT=1 (diagnostics, tools/experiments/trait_stall_probe.py): the circuit of this call is never destroyed, so none of
  // the host memory the runtime has copied from / to is unmapped while later calls run
  std::unique_ptr<SynthCircuit<P>> sc_owner(new SynthCircuit<P>(log_domain, g1_words, g2_words));
  SynthCircuit<P>& sc = *sc_owner;
  struct Leak {
    std::unique_ptr<SynthCircuit<P>>& o;
    ~Leak() {
      if (getenv("COG16_LEAK_BENCH_CIRCUIT")) (void)o.release();
    }
  } leak{sc_owner};
  ProvingKey<P>& pk = sc.pk;
  ConstraintMatrices<P>& m = sc.m;
  SharedWitness<P, Fr>& sw = sc.sw;
  const Fr r = sc.r, s = sc.s;
  out_ms[3] = sc.key_ms;
  std::vector<double> hs, totals;
  std::vector<ProveTimes> phs;
  for (int it = 0; it < iters + 2; ++it) {  // two warm-up rounds (streams, arenas and pooled buffers of the worker threads)
    auto a0 = std::chrono::steady_clock::now();
    std::vector<Fr> hh = CircomReduction::witness_map_from_matrices<P, T>(sc.st0, m, sw.public_inputs, sw.witness);
    auto a1 = std::chrono::steady_clock::now();
    sc.prove(it + 1 == iters + 2);
    const double total = ms_since(a1);
    if (it < 2) continue;
    hs.push_back(std::chrono::duration<double, std::milli>(a1 - a0).count());
    totals.push_back(total);
    phs.push_back(sc.phases);
  }
  const size_t mi = median_index(totals);
  out_ms[0] = hs[median_index(hs)];  // host-facing witness_map_from_matrices alone (witness up, device pipeline, h down)
  out_ms[1] = phs[mi].msm_ms;        // the five MSM groups of the median prove (create_proof_device up to the join), host clock
  out_ms[2] = totals[mi];            // Groth16 prove (prove_inner), key resident on the device
  if (mins_out) mins_out[0] = *std::min_element(totals.begin(), totals.end());
  if (phases_out) {
    phases_out[0] = phs[mi].witness_ms;  // witness upload + device-resident witness map inside that prove
    phases_out[1] = phs[mi].msm_ms;
    phases_out[2] = phs[mi].finish_ms;
  }
  *check_ok = sc.closed_form();
  if (trait_out) {
    // The same circuit, key and witness through the "trait path" (groth16.hpp): the sequence rust/co-groth16-hip drives behind the unchanged
    // reference -- one host-slice witness-map call, h on the host, five concurrent host-scalar MSMs. trait_out = {prove ms (median of iters),
    // witness map ms, five MSMs ms, finish ms (of that prove), closed-form check of the trait-path proof (1 / 0)}.
    struct Restore {
      int prev = trait_path_flag().exchange(1);
      ~Restore() { trait_path_flag().store(prev); }
    } restore;
    std::vector<double> tt;
    std::vector<ProveTimes> tp;
    for (int it = 0; it < iters + 2; ++it) {
      auto a1 = std::chrono::steady_clock::now();
      sc.prove(false);
      if (it < 2) continue;  // warm the lanes of the five MSM threads
      tt.push_back(ms_since(a1));
      tp.push_back(sc.phases);
    }
    sc.prove(true);
    const size_t ti = median_index(tt);
    trait_out[0] = tt[ti];
    trait_out[1] = tp[ti].witness_ms;
    trait_out[2] = tp[ti].msm_ms;
    trait_out[3] = tp[ti].finish_ms;
    trait_out[4] = sc.closed_form() ? 1.0 : 0.0;
    if (mins_out) mins_out[1] = *std::min_element(tt.begin(), tt.end());
  }
  const Proof<P>& proof = sc.proof;
  auto eq1 = [](const AffineT<Fq>& x, const AffineT<Fq>& y) { return x.x == y.x && x.y == y.y; };

  // BASELINE config 4 at scale: three in-process Rep3 parties prove the same circuit with the same r, s; they share this GPU (or take
  // one GPU each when the node has several, with key copies per device). Wall time of the whole three-party run; the agreed proof
  // must equal the plain one. Three ways to drive a party: the device-resident mirror (device ChaCha12 masks from the party's own keys --
  // reachable only with an upstream edit, the generators of Rep3Rand are private), and the two modes of the zero-upstream-edit trait path.
  out_ms[4] = 0;
  out_ms[5] = 0;
  if (with_rep3) {
    using T3 = Rep3Groth16Driver<P>;
    using Share = Rep3PrimeFieldShare<Fr>;
    Rep3Sharer<Fr> sharer(99);
    std::vector<Share> wsh[3];
    sharer.share(sw.witness, wsh);
    Share r3[3], s3[3];
    {
      std::vector<Share> t[3];
      sharer.share({r, s}, t);
      for (int p = 0; p < 3; ++p) {
        r3[p] = t[p][0];
        s3[p] = t[p][1];
      }
    }
    SharedWitness<P, Share> sw3[3];
    for (int p = 0; p < 3; ++p) {
      sw3[p].public_inputs = sw.public_inputs;
      sw3[p].witness = std::move(wsh[p]);
    }
    const int iters3 = iters < 7 ? iters : 7;  // 40-60 ms per round
    // mode 0: device-resident mirror; 1: trait path with host masks; 2: trait path with seeded device masks
    auto three_parties = [&](int mode, double* med_out, double* min_out, ProveTimes* party0, bool* ok_out) {
      struct Restore {
        int prev;
        explicit Restore(int m) : prev(trait_path_flag().exchange(m)) {}
        ~Restore() { trait_path_flag().store(prev); }
      } restore(mode);
      std::vector<double> walls;
      std::vector<ProveTimes> p0;
      Proof<P> proofs[3];
      for (int it = 0; it < iters3 + 1; ++it) {  // one warm-up round
        auto nets0 = LocalNetwork::new_parties(3), nets1 = LocalNetwork::new_parties(3);
        std::string errs[3];
        ProveTimes times[3];
        auto b0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int p = 0; p < 3; ++p) {
          th.emplace_back([&, p] {
            try {
              check(csh_init(0), "csh_init");
              uint8_t my_seed[32];
              ShareRng(4242ull + it, 100 + p).fill(my_seed, 32);
              Rep3State state0 = Rep3State::create(nets0[p], my_seed);
              Rep3State state1 = state0.fork(0);
              proofs[p] = CoGroth16<P, T3>::template prove_inner<CircomReduction>(&nets0[p], &nets1[p], state0, state1, pk, m, sw3[p], &r3[p], &s3[p]);
              times[p] = last_prove_times();
            } catch (const std::exception& e) {
              errs[p] = e.what();
              nets0[p].abort();  // let the other parties unwind instead of waiting on this one
              nets1[p].abort();
            }
          });
        }
        for (auto& t : th) t.join();
        const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - b0).count();
        for (int p = 0; p < 3; ++p)
          if (!errs[p].empty()) throw Error("rep3 party " + std::to_string(p) + ": " + errs[p]);
        if (it == 0) continue;
        walls.push_back(wall);
        p0.push_back(times[0]);
      }
      const size_t wi = median_index(walls);
      *med_out = walls[wi];
      *min_out = *std::min_element(walls.begin(), walls.end());
      if (party0) *party0 = p0[wi];
      bool ok3 = true;
      for (int p = 0; p < 3; ++p)
        ok3 = ok3 && eq1(proofs[p].a, proof.a) && eq1(proofs[p].c, proof.c) && proofs[p].b.x == proof.b.x && proofs[p].b.y == proof.b.y;
      *ok_out = ok3;
    };
    double med = 0, mn = 0;
    bool ok3 = false;
    three_parties(0, &med, &mn, nullptr, &ok3);
    out_ms[4] = med;
    out_ms[5] = ok3 ? 1.0 : 0.0;
    if (mins_out) mins_out[2] = mn;
    if (rep3_trait_out) {
      for (int mode = 1; mode <= 2; ++mode) {
        double* o = rep3_trait_out + 13 * (mode - 1);
        ProveTimes t0;
        three_parties(mode, &o[0], &o[1], &t0, &ok3);
        o[2] = t0.mask_ms;
        o[3] = t0.witness_ms;
        o[4] = t0.half_ms;
        o[5] = t0.msm_ms;
        o[6] = t0.finish_ms;
        o[7] = ok3 ? 1.0 : 0.0;
        // ONE party with the GPU to itself: what a party of a one-GPU-per-party deployment computes between its network rounds
        struct Restore {
          int prev;
          explicit Restore(int m) : prev(trait_path_flag().exchange(m)) {}
          ~Restore() { trait_path_flag().store(prev); }
        } restore(mode);
        uint8_t sd1[32], sd2[32];
        ShareRng(777, 1).fill(sd1, 32);
        ShareRng(777, 2).fill(sd2, 32);
        Rep3State st{0, Rep3Rand(sd1, sd2)};
        std::vector<double> alone;
        std::vector<ProveTimes> ap;
        for (int it = 0; it < iters + 2; ++it) {
          auto a0 = std::chrono::steady_clock::now();
          ProveTimes pt;
          UninitBuf<Fr> h = CircomReduction::witness_map_trait_path<P, T3>(st, m, sw3[0].public_inputs, sw3[0].witness);
          pt.mask_ms = last_prove_times().mask_ms;
          pt.witness_ms = ms_since(a0);
          auto a1 = std::chrono::steady_clock::now();
          UninitBuf<Fr> half(sw3[0].witness.size());
          parallel_for(half.size(), 1 << 16, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) half.data()[i] = T3::to_half_share(sw3[0].witness[i]);
          });
          pt.half_ms = ms_since(a1);
          (void)CoGroth16<P, T3>::msm_groups_trait_path(0, pk, r3[0], s3[0], h.data(), h.size(), sw3[0].public_inputs, half.data(), half.size());
          pt.msm_ms = last_prove_times().msm_ms;
          if (it < 2) continue;
          alone.push_back(ms_since(a0));
          ap.push_back(pt);
        }
        const size_t ai = median_index(alone);
        o[8] = alone[ai];
        o[9] = ap[ai].mask_ms;
        o[10] = ap[ai].witness_ms;
        o[11] = ap[ai].half_ms;
        o[12] = ap[ai].msm_ms;
      }
    }
  }
  return 0;
}

// BASELINE config 4 at scale with ONE GPU PER PARTY: three in-process Rep3 parties prove the synthetic 2^log_domain circuit, party p bound
// to devices[p] with its own copy of the proving key and matrices on that GPU (independent instances: no data-path collective, the
// parties exchange three curve points over the in-process network). devices may repeat (the three parties folded onto fewer GPUs).
// out = {wall ms of the best of `iters` three-party proves, proofs agree and equal the plain proof (1 / 0), key setup ms per device}.
template <class P>
int bench_rep3_party_per_gpu_t(int log_domain, int iters, const int devices[3], double* out, const uint32_t* g1_words, const uint32_t* g2_words) {
  using T3 = Rep3Groth16Driver<P>;
  using Fr = typename P::Fr;
  using Fq = typename P::Fq;
  using Share = Rep3PrimeFieldShare<Fr>;
  int home = 0;
  (void)csh_current_device(&home);
  // one circuit (key + matrices + witness) per distinct device, built on that device
  std::map<int, std::unique_ptr<SynthCircuit<P>>> circuits;
  double key_ms = 0;
  for (int p = 0; p < 3; ++p) {
    if (circuits.count(devices[p])) continue;
    std::string err;
    Joined th([&] {
      check(csh_init(devices[p]), "csh_init");
      circuits[devices[p]].reset(new SynthCircuit<P>(log_domain, g1_words, g2_words));
    });
    th.join();
    key_ms = std::max(key_ms, circuits[devices[p]]->key_ms);
  }
  SynthCircuit<P>& c0 = *circuits[devices[0]];
  const Fr r = c0.r, s = c0.s;
  Rep3Sharer<Fr> sharer(99);
  std::vector<Share> wsh[3];
  sharer.share(c0.sw.witness, wsh);
  Share r3[3], s3[3];
  {
    std::vector<Share> t[3];
    sharer.share({r, s}, t);
    for (int p = 0; p < 3; ++p) r3[p] = t[p][0], s3[p] = t[p][1];
  }
  SharedWitness<P, Share> sw3[3];
  for (int p = 0; p < 3; ++p) {
    sw3[p].public_inputs = c0.sw.public_inputs;
    sw3[p].witness = std::move(wsh[p]);
  }
  double best3 = 1e30;
  Proof<P> proofs[3];
  for (int it = 0; it < iters + 1; ++it) {  // + 1 warm-up round (lanes, arenas and pooled buffers of every party thread)
    auto nets0 = LocalNetwork::new_parties(3), nets1 = LocalNetwork::new_parties(3);
    std::string errs[3];
    auto b0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int p = 0; p < 3; ++p) {
      th.emplace_back([&, p] {
        try {
          check(csh_init(devices[p]), "csh_init");
          SynthCircuit<P>& c = *circuits[devices[p]];
          uint8_t my_seed[32];
          ShareRng(4242ull + it, 100 + p).fill(my_seed, 32);
          Rep3State state0 = Rep3State::create(nets0[p], my_seed);
          Rep3State state1 = state0.fork(0);
          proofs[p] = CoGroth16<P, T3>::template prove_inner<CircomReduction>(&nets0[p], &nets1[p], state0, state1, c.pk, c.m, sw3[p], &r3[p], &s3[p]);
        } catch (const std::exception& e) {
          errs[p] = e.what();
          nets0[p].abort();
          nets1[p].abort();
        }
      });
    }
    for (auto& t : th) t.join();
    if (it) best3 = std::min(best3, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - b0).count());
    for (int p = 0; p < 3; ++p)
      if (!errs[p].empty()) throw Error("rep3 party " + std::to_string(p) + ": " + errs[p]);
  }
  // the plain proof of the same circuit with the same r, s, on the home device's circuit
  check(csh_init(devices[0]), "csh_init");
  c0.prove(false);
  auto eq1 = [](const AffineT<Fq>& x, const AffineT<Fq>& y) { return x.x == y.x && x.y == y.y; };
  bool ok3 = true;
  for (int p = 0; p < 3; ++p)
    ok3 = ok3 && eq1(proofs[p].a, c0.proof.a) && eq1(proofs[p].c, c0.proof.c) && proofs[p].b.x == c0.proof.b.x && proofs[p].b.y == c0.proof.b.y;
  for (auto& kv : circuits) {  // free every device's key on a thread bound to that device
    Joined th([&] {
      check(csh_init(kv.first), "csh_init");
      kv.second.reset();
    });
    th.join();
  }
  check(csh_init(home), "csh_init");
  out[0] = best3;
  out[1] = ok3 ? 1.0 : 0.0;
  out[2] = key_ms;
  return 0;
}

// witness_map_from_matrices of either reduction on caller-supplied matrices (CSR) and a full witness: plain (mode 0) or
// three in-process Rep3 parties (mode 1; h_out receives the three parties' half-share vectors back to back).
template <class P>
int witness_map_t(int reduction, int mode, const uint64_t* const row_ptr[3], const uint32_t* const col[3], const uint64_t* const coef[3],
                  size_t n_rows, size_t n_instance, const uint64_t* witness_full, size_t n_vars, uint64_t seed, uint64_t* h_out, size_t h_cap) {
  using Fr = typename P::Fr;
  ConstraintMatrices<P> m;
  m.num_instance_variables = n_instance;
  m.num_witness_variables = n_vars - n_instance;
  m.num_constraints = n_rows;
  auto fill = [&](int k, std::vector<std::vector<std::pair<Fr, size_t>>>& dst) {
    if (!row_ptr[k]) return;
    dst.resize(n_rows);
    for (size_t i = 0; i < n_rows; ++i)
      for (uint64_t e = row_ptr[k][i]; e < row_ptr[k][i + 1]; ++e) {
        Fr c;
        memcpy(&c, coef[k] + 4 * e, 32);
        dst[i].push_back({c, (size_t)col[k][e]});
      }
  };
  fill(0, m.a);
  fill(1, m.b);
  fill(2, m.c);
  m.upload();
  std::vector<Fr> w(n_vars);
  memcpy(w.data(), witness_full, 32 * n_vars);
  std::vector<Fr> pub(w.begin(), w.begin() + n_instance);
  auto run = [&](auto driver_tag, auto& state, const auto& wit) {
    using T = decltype(driver_tag);
    if (reduction == 0) return CircomReduction::witness_map_from_matrices<P, T>(state, m, pub, wit);
    return LibSnarkReduction::witness_map_from_matrices<P, T>(state, m, pub, wit);
  };
  if (mode == 0) {
    UnitState st;
    std::vector<Fr> wit(w.begin() + n_instance, w.end());
    std::vector<Fr> h = run(PlainGroth16Driver<P>{}, st, wit);
    if (h.size() > h_cap) throw Error("h_out too small");
    memcpy(h_out, h.data(), 32 * h.size());
    return (int)h.size();
  }
  using Share = Rep3PrimeFieldShare<Fr>;
  ShareRng gen(seed, 4);
  auto rnd = [&] {
    uint8_t b[32];
    for (int i = 0; i < 4; ++i) {
      uint64_t v = gen();
      memcpy(b + 8 * i, &v, 8);
    }
    return from_be_bytes_mod_order<Fr>(b);
  };
  std::vector<Share> wit[3];
  for (size_t i = n_instance; i < n_vars; ++i) {
    Fr a = rnd(), b = rnd();
    Fr c = Fr::sub(Fr::sub(w[i], a), b);
    wit[0].push_back({a, c});
    wit[1].push_back({b, a});
    wit[2].push_back({c, b});
  }
  auto nets = LocalNetwork::new_parties(3);
  std::vector<Fr> hs[3];
  std::string errs[3];
  std::vector<std::thread> th;
  for (int p = 0; p < 3; ++p) {
    th.emplace_back([&, p] {
      try {
        check(csh_init(0), "csh_init");
        uint8_t my_seed[32];
        ShareRng(seed, 100 + p).fill(my_seed, 32);  // Rep3State::new draws it from the OS rng (rep3.rs:56-76); seed != 0: tests
        Rep3State state = Rep3State::create(nets[p], my_seed);
        hs[p] = run(Rep3Groth16Driver<P>{}, state, wit[p]);
      } catch (const std::exception& e) {
        errs[p] = e.what();
        nets[p].abort();  // let the other parties unwind instead of waiting on this one
      }
    });
  }
  for (auto& t : th) t.join();
  throw_first_party_error(errs, 3);
  const size_t n = hs[0].size();
  if (3 * n > h_cap) throw Error("h_out too small");
  for (int p = 0; p < 3; ++p) memcpy(h_out + 4 * n * p, hs[p].data(), 32 * n);
  return (int)n;
}

// ---- PLONK / UltraHonk driver call sites (plonk_honk.hpp) on caller data: plain values in, shared inside for Rep3 --------
template <class Fn>
void run_three_parties(uint64_t seed, Fn fn) {  // fn(party, Rep3State&)
  auto nets = LocalNetwork::new_parties(3);
  std::string errs[3];
  std::vector<std::thread> th;
  for (int p = 0; p < 3; ++p) {
    th.emplace_back([&, p] {
      try {
        check(csh_init(0), "csh_init");
        uint8_t my_seed[32];
        ShareRng(seed, 100 + p).fill(my_seed, 32);  // Rep3State::new draws it from the OS rng (rep3.rs:56-76); seed != 0: tests
        Rep3State state = Rep3State::create(nets[p], my_seed);
        fn(p, state);
      } catch (const std::exception& e) {
        errs[p] = e.what();
        nets[p].abort();  // let the other parties unwind instead of waiting on this one
      }
    });
  }
  for (auto& t : th) t.join();
  throw_first_party_error(errs, 3);
}

template <class P>
int driver_fft_t(int driver, int inverse, int snarkjs, const uint64_t* data, size_t n_in, size_t domain_size, uint64_t seed, uint64_t* out) {
  using Fr = typename P::Fr;
  EvaluationDomain<P> d = snarkjs ? EvaluationDomain<P>::snarkjs(domain_size) : EvaluationDomain<P>::arkworks(domain_size);
  std::vector<Fr> v(n_in);
  memcpy((void*)v.data(), data, 32 * n_in);
  if (driver != 1) {
    std::vector<Fr> r = driver == 0 ? (inverse ? PlainPlonkDriver<P>::ifft(v, d) : PlainPlonkDriver<P>::fft(v, d))
                                    : (inverse ? ShamirPlonkDriver<P>::ifft(v, d) : ShamirPlonkDriver<P>::fft(v, d));
    memcpy(out, r.data(), 32 * r.size());
    return (int)r.size();
  }
  std::vector<Rep3PrimeFieldShare<Fr>> sh[3];
  Rep3Sharer<Fr>(seed).share(v, sh);
  for (int p = 0; p < 3; ++p) {  // local operation: no network, no randomness
    auto r = inverse ? Rep3PlonkDriver<P>::ifft(sh[p], d) : Rep3PlonkDriver<P>::fft(sh[p], d);
    memcpy(out + 8 * d.size * p, r.data(), 64 * r.size());
  }
  return (int)d.size;
}

template <class P>
int driver_mul_t(int driver, const uint64_t* a, const uint64_t* b, size_t n, uint64_t seed, uint64_t* out) {
  using Fr = typename P::Fr;
  std::vector<Fr> va(n), vb(n);
  memcpy((void*)va.data(), a, 32 * n);
  memcpy((void*)vb.data(), b, 32 * n);
  UnitState us;
  if (driver == 0 || driver == 2) {
    std::vector<Fr> r = driver == 0 ? PlainPlonkDriver<P>::local_mul_vec(va, vb, us) : ShamirPlonkDriver<P>::local_mul_vec(va, vb, us);
    memcpy(out, r.data(), 32 * n);
    return 0;
  }
  std::vector<Rep3PrimeFieldShare<Fr>> sa[3], sb[3];
  Rep3Sharer<Fr> sharer(seed);
  sharer.share(va, sa);
  sharer.share(vb, sb);
  run_three_parties(seed, [&](int p, Rep3State& st) {
    std::vector<Fr> r = Rep3PlonkDriver<P>::local_mul_vec(sa[p], sb[p], st);
    memcpy(out + 4 * n * p, r.data(), 32 * n);
  });
  return 0;
}

template <class C, class P>
int driver_msm_t(int driver, const void* points, size_t n_points, const uint64_t* scalars, size_t n_scalars, uint64_t seed, void* out) {
  using Fr = typename C::Fr;
  using Fq = typename C::Fq;
  std::vector<AffineT<Fq>> pts(n_points);
  memcpy((void*)pts.data(), points, sizeof(AffineT<Fq>) * n_points);
  std::vector<Fr> sc(n_scalars);
  memcpy((void*)sc.data(), scalars, 32 * n_scalars);
  auto put = [&](size_t slot, const Proj<Fq>& pt) {
    AffineT<Fq> a = into_affine(pt);
    memcpy(static_cast<char*>(out) + slot * sizeof a, &a, sizeof a);
  };
  if (driver == 3) {  // HonkCurve::fast_msm (BN254 G1 or Grumpkin)
    put(0, fast_msm<C>(pts, sc));
    return 0;
  }
  if constexpr (!std::is_same<P, void>::value) {
    if (driver == 0) put(0, PlainPlonkDriver<P>::msm_public_points_g1(pts, sc));
    else if (driver == 2) put(0, ShamirPlonkDriver<P>::msm_public_points(pts, sc));
    else {
      std::vector<Rep3PrimeFieldShare<Fr>> sh[3];
      Rep3Sharer<Fr>(seed).share(sc, sh);
      for (int p = 0; p < 3; ++p) {
        auto r = Rep3PlonkDriver<P>::msm_public_points_g1(pts, sh[p]);
        put(2 * p, r.a);
        put(2 * p + 1, r.b);
      }
    }
    return 0;
  }
  throw Error("driver not available for this curve");
}

// LibSnarkReduction straight from the reference's file formats (co-groth16/src/lib.rs:243-262): ark-serialize Matrix<F> blobs
// for a, b, c and a wtns container. h_out: plain h (Montgomery limbs); returns the domain size or -1.
template <class P>
static int libsnark_from_files_t(const uint8_t* const mats[3], const size_t lens[3], const uint8_t* wtns, size_t wlen, size_t n_instance,
                                 uint64_t* h_out, size_t h_cap) {
  using Fr = typename P::Fr;
  ConstraintMatrices<P> m;
  ark::Reader ra(mats[0], lens[0]), rb(mats[1], lens[1]), rc(mats[2], lens[2]);
  m.a = ark::read_matrix<Fr>(ra);
  m.b = ark::read_matrix<Fr>(rb);
  m.c = ark::read_matrix<Fr>(rc);
  if (!ra.done() || !rb.done() || !rc.done()) throw Error("trailing bytes after Matrix");
  if (m.a.size() != m.b.size() || m.a.size() != m.c.size()) throw Error("matrices disagree on the number of constraints");
  std::vector<Fr> w = ark::read_wtns_positional<Fr>(wtns, wlen);
  if (n_instance > w.size()) throw Error("more instance variables than witness values");
  m.num_instance_variables = n_instance;
  m.num_witness_variables = w.size() - n_instance;
  m.num_constraints = m.a.size();
  m.upload();
  UnitState st;
  std::vector<Fr> pub(w.begin(), w.begin() + n_instance), wit(w.begin() + n_instance, w.end());
  std::vector<Fr> h = LibSnarkReduction::witness_map_from_matrices<P, PlainGroth16Driver<P>>(st, m, pub, wit);
  if (h.size() > h_cap) throw Error("h_out too small");
  memcpy(h_out, h.data(), 32 * h.size());
  return (int)h.size();
}

// Groth16::<P>::plain_prove::<LibSnarkReduction> from the files the reference's test reads (co-groth16/src/lib.rs:231-290): ark
// ProvingKey (deserialize_uncompressed_unchecked), Matrix<F> blobs a / b / c, a wtns container; ConstraintMatrices filled in as :268-279
// does (num_instance_variables = b_g1_query.len() - l_query.len(), ...). r, s: canonical limbs, nullable (drawn otherwise).
// out: ark-serialize Proof {a, b, c}, uncompressed; returns its length.
template <class P>
struct LibsnarkFiles {
  const uint8_t* const* mats;
  const size_t* lens;
  const uint8_t* pkey;
  size_t pklen;
  // key + matrices onto the calling thread's device
  void load(ProvingKey<P>& pk, ConstraintMatrices<P>& m) const {
    using Fr = typename P::Fr;
    ark::Reader rp(pkey, pklen);
    ark::read_proving_key<P>(rp, pk);
    if (!rp.done()) throw Error("trailing bytes after ProvingKey");
    if (pk.b_g1_query.host.size() < pk.l_query.host.size() || pk.a_query.host.size() != pk.b_g1_query.host.size() ||
        pk.b_g2_query.host.size() != pk.b_g1_query.host.size())
      throw Error("ProvingKey: query lengths are inconsistent");
    ark::Reader ra(mats[0], lens[0]), rb(mats[1], lens[1]), rc(mats[2], lens[2]);
    m.a = ark::read_matrix<Fr>(ra);
    m.b = ark::read_matrix<Fr>(rb);
    m.c = ark::read_matrix<Fr>(rc);
    if (!ra.done() || !rb.done() || !rc.done()) throw Error("trailing bytes after Matrix");
    if (m.a.size() != m.b.size() || m.a.size() != m.c.size()) throw Error("matrices disagree on the number of constraints");
    m.num_instance_variables = pk.b_g1_query.host.size() - pk.l_query.host.size();   // lib.rs:269
    m.num_witness_variables = pk.a_query.host.size() - m.num_instance_variables;     // lib.rs:270-271
    m.num_constraints = m.a.size();
    m.upload();
    pk.a_query.upload(P::ID, CSH_G1);
    pk.b_g1_query.upload(P::ID, CSH_G1);
    pk.l_query.upload(P::ID, CSH_G1, pk.b_g1_query.host.size() - pk.l_query.host.size());
    pk.h_query.upload(P::ID, CSH_G1);
    pk.b_g2_query.upload(P::ID, CSH_G2);
    pk.build_tables();
  }
};

template <class P>
static int finish_libsnark(const Proof<P>& pr, uint8_t* out, size_t cap) {
  const std::vector<uint8_t> bytes = ark::write_proof<P>(pr);
  if (bytes.size() > cap) throw Error("output buffer too small");
  memcpy(out, bytes.data(), bytes.size());
  return (int)bytes.size();
}

template <class P>
static int prove_libsnark_t(const LibsnarkFiles<P>& files, const uint8_t* wtns, size_t wlen, const uint64_t* r, const uint64_t* s, uint8_t* out,
                            size_t cap, uint64_t* h_out, size_t h_cap) {
  using T = PlainGroth16Driver<P>;
  using Fr = typename P::Fr;
  ProvingKey<P> pk;
  ConstraintMatrices<P> m;
  files.load(pk, m);
  std::vector<Fr> w = ark::read_wtns_positional<Fr>(wtns, wlen);
  SharedWitness<P, Fr> sw;
  if (m.num_instance_variables > w.size()) throw Error("more instance variables than witness values");
  sw.public_inputs.assign(w.begin(), w.begin() + m.num_instance_variables);        // lib.rs:283-287
  sw.witness.assign(w.begin() + m.num_instance_variables, w.end());
  UnitState st0, st1;
  Fr rr, ss;
  if (r) rr = fr_from_canonical<P>(r);
  if (s) ss = fr_from_canonical<P>(s);
  std::vector<Fr> h;
  const Proof<P> pr = CoGroth16<P, T>::template prove_inner<LibSnarkReduction>(nullptr, nullptr, st0, st1, pk, m, sw, r ? &rr : nullptr,
                                                                                s ? &ss : nullptr, h_out ? &h : nullptr);
  if (h_out) {
    if (h.size() > h_cap) throw Error("h_out too small");
    memcpy(h_out, h.data(), h.size() * 32);
  }
  return finish_libsnark<P>(pr, out, cap);
}

// Rep3CoGroth16::prove::<LibSnarkReduction> (groth16.rs:360-379 with R = LibSnarkReduction) with three in-process parties, as the reference's
// Rep3 tests run theirs (tests/tests/circom/e2e_tests/rep3.rs:57-69): the witness is shared with share_field_elements semantics from `seed`,
// every party runs prove_inner on its shares (one GPU per party when the node has them), the three proofs must agree. h_out (nullable):
// the three parties' h half-shares, one after the other.
template <class P>
static int prove_libsnark_rep3_t(const LibsnarkFiles<P>& files, const uint8_t* wtns, size_t wlen, uint64_t seed, const uint64_t* r, const uint64_t* s,
                                 uint8_t* out, size_t cap, uint64_t* h_out, size_t h_cap) {
  using T = Rep3Groth16Driver<P>;
  using Fr = typename P::Fr;
  using Share = Rep3PrimeFieldShare<Fr>;
  int ndev = 1;
  csh_device_count(&ndev);
  const bool per_party_keys = ndev > 1;
  ProvingKey<P> pk_shared;
  ConstraintMatrices<P> m_shared;
  if (!per_party_keys) files.load(pk_shared, m_shared);
  std::vector<Fr> w = ark::read_wtns_positional<Fr>(wtns, wlen);
  size_t n_instance = 0;
  {
    ark::Reader rp(files.pkey, files.pklen);   // the instance count without uploading anything: ic = gamma_abc_g1
    ProvingKey<P> probe;
    ark::read_proving_key<P>(rp, probe);
    n_instance = probe.b_g1_query.host.size() - probe.l_query.host.size();
  }
  if (n_instance > w.size()) throw Error("more instance variables than witness values");
  sharefile::CompressedRep3SharedWitness<P> shares[3];
  split_witness_rep3<P>(w, n_instance, 0, seed, shares);
  SeededSharer<Fr> rs_sharer(seed, /*domain=*/11);
  Share r3[3], s3[3];
  if (r) rs_sharer.share(fr_from_canonical<P>(r), r3);
  if (s) rs_sharer.share(fr_from_canonical<P>(s), s3);
  auto nets0 = LocalNetwork::new_parties(3), nets1 = LocalNetwork::new_parties(3);
  Proof<P> proofs[3];
  std::vector<Fr> hs[3];
  std::string errs[3];
  std::vector<std::thread> th;
  for (int p = 0; p < 3; ++p) {
    th.emplace_back([&, p] {
      try {
        check(csh_init(ndev > 0 ? p % ndev : 0), "csh_init");
        ProvingKey<P> pk_own;
        ConstraintMatrices<P> m_own;
        if (per_party_keys) files.load(pk_own, m_own);
        const ProvingKey<P>& pk = per_party_keys ? pk_own : pk_shared;
        const ConstraintMatrices<P>& m = per_party_keys ? m_own : m_shared;
        uint8_t my_seed[32];
        ShareRng(seed, 100 + p).fill(my_seed, 32);
        Rep3State state0 = Rep3State::create(nets0[p], my_seed);
        Rep3State state1 = state0.fork(0);
        SharedWitness<P, Share> sw = sharefile::uncompress<P>(std::move(shares[p]), nets0[p]);
        proofs[p] = CoGroth16<P, T>::template prove_inner<LibSnarkReduction>(&nets0[p], &nets1[p], state0, state1, pk, m, sw, r ? &r3[p] : nullptr,
                                                                              s ? &s3[p] : nullptr, h_out ? &hs[p] : nullptr);
      } catch (const std::exception& e) {
        errs[p] = e.what();
        nets0[p].abort();
        nets1[p].abort();
      }
    });
  }
  for (auto& t : th) t.join();
  throw_first_party_error(errs, 3);
  const std::vector<uint8_t> b0 = ark::write_proof<P>(proofs[0]);
  if (b0 != ark::write_proof<P>(proofs[1]) || b0 != ark::write_proof<P>(proofs[2])) throw Error("the three parties disagree on the proof");
  if (h_out) {
    const size_t n = hs[0].size();
    if (3 * n > h_cap) throw Error("h_out too small");
    for (int p = 0; p < 3; ++p) memcpy(h_out + 4 * n * p, hs[p].data(), 32 * n);
  }
  return finish_libsnark<P>(proofs[0], out, cap);
}

// ShamirCoGroth16::prove::<LibSnarkReduction> (groth16.rs:439-463 with the LibSnark reduction): n in-process parties, threshold t, the witness
// shared with share_shamir semantics from `seed`, a dealer standing in for the DN07 preprocessing (as prove_shamir_core); with r / s given the
// first two double sharings share exactly r and s, so the proof equals the plain one for the same r, s.
template <class P>
static int prove_libsnark_shamir_t(const LibsnarkFiles<P>& files, const uint8_t* wtns, size_t wlen, int n, int t, uint64_t seed, const uint64_t* r,
                                   const uint64_t* s, uint8_t* out, size_t cap) {
  using T = ShamirGroth16Driver<P>;
  using Fr = typename P::Fr;
  if (n < 2 * t + 1 || t < 1) throw Error("num_parties must be at least 2 * threshold + 1");
  int ndev = 1;
  csh_device_count(&ndev);
  const bool per_party_keys = ndev > 1;
  ProvingKey<P> pk_shared;
  ConstraintMatrices<P> m_shared;
  if (!per_party_keys) files.load(pk_shared, m_shared);
  size_t n_instance = 0;
  {
    ark::Reader rp(files.pkey, files.pklen);
    ProvingKey<P> probe;
    ark::read_proving_key<P>(rp, probe);
    if (probe.b_g1_query.host.size() < probe.l_query.host.size()) throw Error("ProvingKey: query lengths are inconsistent");
    n_instance = probe.b_g1_query.host.size() - probe.l_query.host.size();
  }
  std::vector<Fr> w = ark::read_wtns_positional<Fr>(wtns, wlen);
  if (n_instance > w.size()) throw Error("more instance variables than witness values");
  auto sw = split_witness_shamir<P>(w, n_instance, t, n, seed);
  SeededShamirSharer<Fr> dealer(seed, n, /*domain=*/12);
  std::vector<std::deque<std::pair<Fr, Fr>>> pairs(n);
  for (int k = 0; k < 3; ++k) {  // two rand calls + one scalar_mul (groth16.rs:448-449)
    Fr v = dealer.rnd();
    if (k == 0 && r) v = fr_from_canonical<P>(r);
    if (k == 1 && s) v = fr_from_canonical<P>(s);
    auto st = dealer.share(v, t), s2t = dealer.share(v, 2 * t);
    for (int p = 0; p < n; ++p) pairs[p].push_back({st[p], s2t[p]});
  }
  auto nets0 = LocalNetwork::new_parties(n), nets1 = LocalNetwork::new_parties(n);
  std::vector<Proof<P>> proofs(n);
  std::vector<std::string> errs(n);
  std::vector<std::thread> th;
  for (int p = 0; p < n; ++p) {
    th.emplace_back([&, p] {
      try {
        check(csh_init(ndev > 0 ? p % ndev : 0), "csh_init");
        ProvingKey<P> pk_own;
        ConstraintMatrices<P> m_own;
        if (per_party_keys) files.load(pk_own, m_own);
        const ProvingKey<P>& pk = per_party_keys ? pk_own : pk_shared;
        const ConstraintMatrices<P>& m = per_party_keys ? m_own : m_shared;
        auto state0 = ShamirState<Fr>::create(p, n, t, pairs[p]);
        auto state1 = state0.fork(1);
        proofs[p] = CoGroth16<P, T>::template prove_inner<LibSnarkReduction>(&nets0[p], &nets1[p], state0, state1, pk, m, sw[p], nullptr, nullptr);
      } catch (const std::exception& e) {
        errs[p] = e.what();
        nets0[p].abort();
        nets1[p].abort();
      }
    });
  }
  for (auto& x : th) x.join();
  throw_first_party_error(errs.data(), n);
  const std::vector<uint8_t> b0 = ark::write_proof<P>(proofs[0]);
  for (int p = 1; p < n; ++p)
    if (b0 != ark::write_proof<P>(proofs[p])) throw Error("the parties disagree on the proof");
  return finish_libsnark<P>(proofs[0], out, cap);
}

// mode: 0 plain, 1 Rep3 (three parties), 2 + 256 * parties + 65536 * threshold: Shamir
template <class P>
static int prove_libsnark_any(int mode, const uint8_t* const mats[3], const size_t lens[3], const uint8_t* wtns, size_t wlen, const uint8_t* pkey,
                              size_t pklen, uint64_t seed, const uint64_t* r, const uint64_t* s, uint8_t* out, size_t cap, uint64_t* h_out, size_t h_cap) {
  const LibsnarkFiles<P> files{mats, lens, pkey, pklen};
  if ((mode & 255) == 2) return prove_libsnark_shamir_t<P>(files, wtns, wlen, (mode >> 8) & 255, (mode >> 16) & 255, seed, r, s, out, cap);
  return mode ? prove_libsnark_rep3_t<P>(files, wtns, wlen, seed, r, s, out, cap, h_out, h_cap)
              : prove_libsnark_t<P>(files, wtns, wlen, r, s, out, cap, h_out, h_cap);
}
static int prove_libsnark_entry(int curve, int mode, const uint8_t* a, size_t alen, const uint8_t* b, size_t blen, const uint8_t* c, size_t clen,
                                const uint8_t* wtns, size_t wlen, const uint8_t* pkey, size_t pklen, uint64_t seed, const uint64_t* r, const uint64_t* s,
                                uint8_t* out, size_t cap, uint64_t* h_out, size_t h_cap_elems) {
  try {
    const uint8_t* mats[3] = {a, b, c};
    const size_t lens[3] = {alen, blen, clen};
    if (curve == 0) return prove_libsnark_any<Bn254>(mode, mats, lens, wtns, wlen, pkey, pklen, seed, r, s, out, cap, h_out, h_cap_elems);
    if (curve == 1) return prove_libsnark_any<Bls12_381>(mode, mats, lens, wtns, wlen, pkey, pklen, seed, r, s, out, cap, h_out, h_cap_elems);
    if (curve == 3) return prove_libsnark_any<Bls12_377>(mode, mats, lens, wtns, wlen, pkey, pklen, seed, r, s, out, cap, h_out, h_cap_elems);
    g_err = "unknown curve";
    return -1;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
}  // namespace

extern "C" {

// PLONK / UltraHonk driver methods on caller data. driver: 0 plain, 1 Rep3 (three in-process parties, values shared with
// `seed`), 2 Shamir (the vector is one party's share vector), 3 (msm only) HonkCurve::fast_msm; curve 2 = Grumpkin (msm).
int cog16_driver_fft(int curve, int driver, int inverse, int snarkjs, const uint64_t* data, size_t n_in, size_t domain_size, uint64_t seed,
                     uint64_t* out) {
  try {
    if (curve == 0) return driver_fft_t<Bn254>(driver, inverse, snarkjs, data, n_in, domain_size, seed, out);
    if (curve == 1) return driver_fft_t<Bls12_381>(driver, inverse, snarkjs, data, n_in, domain_size, seed, out);
    g_err = "unknown curve";
    return -1;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
int cog16_driver_local_mul_vec(int curve, int driver, const uint64_t* a, const uint64_t* b, size_t n, uint64_t seed, uint64_t* out) {
  try {
    if (curve == 0) return driver_mul_t<Bn254>(driver, a, b, n, seed, out);
    if (curve == 1) return driver_mul_t<Bls12_381>(driver, a, b, n, seed, out);
    g_err = "unknown curve";
    return -1;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
int cog16_driver_msm(int curve, int driver, const void* points, size_t n_points, const uint64_t* scalars, size_t n_scalars, uint64_t seed,
                     void* out) {
  try {
    if (curve == 0) return driver_msm_t<G1Of<Bn254>, Bn254>(driver, points, n_points, scalars, n_scalars, seed, out);
    if (curve == 1) return driver_msm_t<G1Of<Bls12_381>, Bls12_381>(driver, points, n_points, scalars, n_scalars, seed, out);
    if (curve == 2) return driver_msm_t<GrumpkinCurve, void>(driver, points, n_points, scalars, n_scalars, seed, out);
    g_err = "unknown curve";
    return -1;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

int cog16_libsnark_from_files(int curve, const uint8_t* a, size_t alen, const uint8_t* b, size_t blen, const uint8_t* c, size_t clen,
                              const uint8_t* wtns, size_t wlen, size_t n_instance, uint64_t* h_out, size_t h_cap_elems) {
  try {
    const uint8_t* const mats[3] = {a, b, c};
    const size_t lens[3] = {alen, blen, clen};
    if (curve == 0) return libsnark_from_files_t<Bn254>(mats, lens, wtns, wlen, n_instance, h_out, h_cap_elems);
    if (curve == 1) return libsnark_from_files_t<Bls12_381>(mats, lens, wtns, wlen, n_instance, h_out, h_cap_elems);
    if (curve == 3) return libsnark_from_files_t<Bls12_377>(mats, lens, wtns, wlen, n_instance, h_out, h_cap_elems);
    g_err = "unknown curve";
    return -1;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// proof_libsnark_penumbra_bls12_377 (co-groth16/src/lib.rs:231-290) and its counterparts on the other curves: see prove_libsnark_t;
// cog16_prove_libsnark_rep3: the same circuit through three in-process Rep3 parties (prove_libsnark_rep3_t). Return the proof's byte
// length or -1.
int cog16_prove_libsnark(int curve, const uint8_t* a, size_t alen, const uint8_t* b, size_t blen, const uint8_t* c, size_t clen, const uint8_t* wtns,
                         size_t wlen, const uint8_t* pkey, size_t pklen, const uint64_t* r, const uint64_t* s, uint8_t* out, size_t cap,
                         uint64_t* h_out, size_t h_cap_elems) {
  return prove_libsnark_entry(curve, 0, a, alen, b, blen, c, clen, wtns, wlen, pkey, pklen, 0, r, s, out, cap, h_out, h_cap_elems);
}
// ShamirCoGroth16::prove::<LibSnarkReduction> with num_parties in-process parties and the given threshold (prove_libsnark_shamir_t)
int cog16_prove_libsnark_shamir(int curve, const uint8_t* a, size_t alen, const uint8_t* b, size_t blen, const uint8_t* c, size_t clen,
                                const uint8_t* wtns, size_t wlen, const uint8_t* pkey, size_t pklen, int num_parties, int threshold, uint64_t seed,
                                const uint64_t* r, const uint64_t* s, uint8_t* out, size_t cap) {
  if (num_parties < 3 || num_parties > 255 || threshold < 1 || threshold > 127) {
    g_err = "num_parties / threshold out of range";
    return -1;
  }
  return prove_libsnark_entry(curve, 2 + 256 * num_parties + 65536 * threshold, a, alen, b, blen, c, clen, wtns, wlen, pkey, pklen, seed, r, s, out, cap,
                              nullptr, 0);
}
int cog16_prove_libsnark_rep3(int curve, const uint8_t* a, size_t alen, const uint8_t* b, size_t blen, const uint8_t* c, size_t clen,
                              const uint8_t* wtns, size_t wlen, const uint8_t* pkey, size_t pklen, uint64_t seed, const uint64_t* r, const uint64_t* s,
                              uint8_t* out, size_t cap, uint64_t* h_shares_out, size_t h_cap_elems) {
  return prove_libsnark_entry(curve, 1, a, alen, b, blen, c, clen, wtns, wlen, pkey, pklen, seed, r, s, out, cap, h_shares_out, h_cap_elems);
}

// ark-serialize round trips of the host mirror (arkwire.hpp): mode 0 Vec<Fr> (in: Montgomery limbs of n elements ->
// out: serialized bytes, then parsed back and compared), mode 1 G1 affine points (n points, C-ABI layout). Returns the
// number of bytes written to `out`, or -1.
int cog16_ark_roundtrip(int curve, int mode, const void* in, size_t n, uint8_t* out, size_t cap) {
  try {
    std::vector<uint8_t> buf;
    auto run = [&](auto tag) {
      using P = decltype(tag);
      using Fr = typename P::Fr;
      using Fq = typename P::Fq;
      if (mode == 0) {
        std::vector<Fr> v(n);
        memcpy((void*)v.data(), in, sizeof(Fr) * n);
        ark::write_vec(buf, v);
        ark::Reader r(buf.data(), buf.size());
        std::vector<Fr> back = ark::read_vec<Fr>(r);
        if (!r.done() || back.size() != n || memcmp(back.data(), v.data(), sizeof(Fr) * n) != 0) throw Error("Vec<F> round trip mismatch");
      } else {
        std::vector<AffineT<Fq>> pts(n);
        memcpy((void*)pts.data(), in, sizeof(AffineT<Fq>) * n);
        for (auto& pt : pts) ark::write_g1(buf, pt);
        ark::Reader r(buf.data(), buf.size());
        for (auto& pt : pts) {
          AffineT<Fq> q = ark::read_g1<Fq>(r);
          if (memcmp(&q, &pt, sizeof q) != 0) throw Error("G1 round trip mismatch");
        }
        if (!r.done()) throw Error("trailing bytes");
      }
    };
    if (curve == 0) run(Bn254{});
    else if (curve == 1) run(Bls12_381{});
    else throw Error("unknown curve");
    if (buf.size() > cap) throw Error("output buffer too small");
    memcpy(out, buf.data(), buf.size());
    return (int)buf.size();
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// Rep3NetworkExt::{send_many, recv_many} payloads (arkwire.hpp). kind 0: field elements (Fr, Montgomery limbs on the C-ABI side), kind 1: G1
// affine points (C-ABI layout). op 0 = send_many: `in` holds n items -> message bytes in `out`, returns the message length; op 1 =
// recv_many: `in` holds a message of n BYTES -> items in `out` (cap in bytes), returns the item count. -1 on error.
long cog16_rep3_wire(int curve, int op, int kind, const void* in, size_t n, void* out, size_t cap) {
  try {
    long result = -1;
    auto run = [&](auto tag) {
      using P = decltype(tag);
      using Fr = typename P::Fr;
      using Fq = typename P::Fq;
      if (op == 0) {
        std::vector<uint8_t> msg;
        if (kind == 0) {
          std::vector<Fr> v(n);
          if (n) memcpy((void*)v.data(), in, sizeof(Fr) * n);
          msg = ark::send_many_fields(v);
        } else {
          std::vector<AffineT<Fq>> v(n);
          if (n) memcpy((void*)v.data(), in, sizeof(AffineT<Fq>) * n);
          msg = ark::send_many_g1(v);
        }
        if (msg.size() > cap) throw Error("output buffer too small");
        memcpy(out, msg.data(), msg.size());
        result = (long)msg.size();
      } else {
        if (kind == 0) {
          std::vector<Fr> v = ark::recv_many_fields<Fr>(static_cast<const uint8_t*>(in), n);
          if (v.size() * sizeof(Fr) > cap) throw Error("output buffer too small");
          if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(Fr));
          result = (long)v.size();
        } else {
          std::vector<AffineT<Fq>> v = ark::recv_many_g1<Fq>(static_cast<const uint8_t*>(in), n);
          if (v.size() * sizeof(AffineT<Fq>) > cap) throw Error("output buffer too small");
          if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(AffineT<Fq>));
          result = (long)v.size();
        }
      }
    };
    if (op != 0 && op != 1) throw Error("op must be 0 (send_many) or 1 (recv_many)");
    if (kind != 0 && kind != 1) throw Error("kind must be 0 (field elements) or 1 (G1 affine points)");
    if (curve == 0) run(Bn254{});
    else if (curve == 1) run(Bls12_381{});
    else throw Error("unknown curve");
    return result;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// Returns the domain size (entries per party in h_out) or -1. reduction: 0 CircomReduction, 1 LibSnarkReduction (needs
// the C matrix); mode: 0 plain, 1 three Rep3 parties. CSR triples for a, b, c (c may be NULL for reduction 0).
int cog16_witness_map(int curve, int reduction, int mode, const uint64_t* const row_ptr[3], const uint32_t* const col[3],
                      const uint64_t* const coef[3], size_t n_rows, size_t n_instance, const uint64_t* witness_full, size_t n_vars, uint64_t seed,
                      uint64_t* h_out, size_t h_cap_elems) {
  try {
    if (curve == 0) return witness_map_t<Bn254>(reduction, mode, row_ptr, col, coef, n_rows, n_instance, witness_full, n_vars, seed, h_out, h_cap_elems);
    if (curve == 1) return witness_map_t<Bls12_381>(reduction, mode, row_ptr, col, coef, n_rows, n_instance, witness_full, n_vars, seed, h_out, h_cap_elems);
    if (curve == 3) return witness_map_t<Bls12_377>(reduction, mode, row_ptr, col, coef, n_rows, n_instance, witness_full, n_vars, seed, h_out, h_cap_elems);
    g_err = "unknown curve";
    return -1;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// `co-circom split-witness` (co-circom.rs:660-740): protocol 0 = Rep3 (three files; compression 0 none, 1 half shares),
// 2 seeded shares, 3 seeded half shares = what the reference's CLI writes), 1 = Shamir (num_parties files of threshold `threshold`). The files are written back to back into `out`; sizes[i] is
// the length of party i's file. num_inputs counts the public inputs and the constant 1 (r1cs.num_inputs). Host-only
// (no device call). Returns the number of files or -1.
int cog16_split_witness(int curve, int protocol, const uint8_t* wtns, size_t wlen, size_t num_inputs, int compression, int threshold,
                        int num_parties, uint64_t seed, uint8_t* out, size_t cap, size_t* sizes) {
  try {
    std::vector<std::vector<uint8_t>> files;
    auto run = [&](auto tag) {
      using P = decltype(tag);
      std::vector<typename P::Fr> w = parse_wtns<P>(wtns, wlen);
      if (protocol == 0) {
        sharefile::CompressedRep3SharedWitness<P> sh[3];
        split_witness_rep3<P>(w, num_inputs, compression, seed, sh);
        for (int p = 0; p < 3; ++p) files.push_back(sharefile::write_rep3<P>(sh[p]));
      } else if (protocol == 1) {
        for (auto& sw : split_witness_shamir<P>(w, num_inputs, threshold, num_parties, seed)) files.push_back(sharefile::write_shamir<P>(sw));
      } else {
        throw Error("protocol must be 0 (Rep3) or 1 (Shamir)");
      }
    };
    if (curve == 0) run(Bn254{});
    else if (curve == 1) run(Bls12_381{});
    else throw Error("unknown curve");
    size_t at = 0;
    for (size_t i = 0; i < files.size(); ++i) {
      if (at + files[i].size() > cap) throw Error("output buffer too small");
      memcpy(out + at, files[i].data(), files[i].size());
      sizes[i] = files[i].size();
      at += files[i].size();
    }
    return (int)files.size();
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// Parse one witness-share file and write it again (must reproduce the input byte for byte); reports the variant
// (Rep3: 0 replicated, 1 seeded replicated, 2 additive, 3 seeded additive; Shamir: 0) and the element counts. Host-only. Returns bytes written or -1.
int cog16_share_file_roundtrip(int curve, int protocol, const uint8_t* file, size_t len, uint8_t* out, size_t cap, uint32_t* variant,
                               size_t* n_public, size_t* n_witness) {
  try {
    std::vector<uint8_t> buf;
    auto run = [&](auto tag) {
      using P = decltype(tag);
      if (protocol == 0) {
        auto w = sharefile::read_rep3<P>(file, len);
        *variant = w.kind;
        *n_public = w.public_inputs.size();
        *n_witness = w.length();
        buf = sharefile::write_rep3<P>(w);
      } else if (protocol == 1) {
        auto w = sharefile::read_shamir<P>(file, len);
        *variant = 0;
        *n_public = w.public_inputs.size();
        *n_witness = w.witness.size();
        buf = sharefile::write_shamir<P>(w);
      } else {
        throw Error("protocol must be 0 (Rep3) or 1 (Shamir)");
      }
    };
    if (curve == 0) run(Bn254{});
    else if (curve == 1) run(Bls12_381{});
    else throw Error("unknown curve");
    if (buf.size() > cap) throw Error("output buffer too small");
    memcpy(out, buf.data(), buf.size());
    return (int)buf.size();
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// The public-input file `generate-proof` writes next to the proof (co-circom.rs:1120-1140): the share file's public
// inputs without the leading constant 1, as a compact JSON array of decimal strings. Host-only. Returns 0 or -1.
int cog16_public_inputs_json(int curve, int protocol, const uint8_t* file, size_t len, char* out_json, size_t cap) {
  try {
    std::string j = "[";
    auto run = [&](auto tag) {
      using P = decltype(tag);
      std::vector<typename P::Fr> pub;
      if (protocol == 0) pub = sharefile::read_rep3<P>(file, len).public_inputs;
      else if (protocol == 1) pub = sharefile::read_shamir<P>(file, len).public_inputs;
      else throw Error("protocol must be 0 (Rep3) or 1 (Shamir)");
      for (size_t i = 1; i < pub.size(); ++i) {
        if (i > 1) j += ",";
        j += "\"" + to_decimal(pub[i]) + "\"";
      }
    };
    if (curve == 0) run(Bn254{});
    else if (curve == 1) run(Bls12_381{});
    else throw Error("unknown curve");
    j += "]";
    return write_out(j, out_json, cap);
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// `co-circom generate-proof` from the parties' `.shared` files (co-circom.rs:1008-1050) with in-process parties:
// protocol 0 = Rep3 (exactly three files), 1 = Shamir (n files, threshold t).
int cog16_prove_from_shares(int curve, int protocol, const uint8_t* zkey, size_t zlen, const uint8_t* const* files, const size_t* lens,
                            int num_parties, int threshold, uint64_t seed, const uint64_t* r, const uint64_t* s, char* out_json, size_t cap) {
  try {
    auto run = [&](auto tag) -> int {
      using P = decltype(tag);
      if (protocol == 0) {
        if (num_parties != 3 || threshold != 1) throw Error("REP3 only allows three parties and the threshold to be 1");
        sharefile::CompressedRep3SharedWitness<P> sh[3];
        for (int p = 0; p < 3; ++p) sh[p] = sharefile::read_rep3<P>(files[p], lens[p]);
        if (sh[0].kind != sh[1].kind || sh[0].kind != sh[2].kind) throw Error("the parties' share files use different compression");
        return prove_rep3_core<P>(zkey, zlen, sh, seed, r, s, out_json, cap, nullptr, 0);
      }
      if (protocol != 1) throw Error("protocol must be 0 (Rep3) or 1 (Shamir)");
      std::vector<SharedWitness<P, typename P::Fr>> sw;
      for (int p = 0; p < num_parties; ++p) sw.push_back(sharefile::read_shamir<P>(files[p], lens[p]));
      return prove_shamir_core<P>(zkey, zlen, sw, threshold, seed, r, s, out_json, cap);
    };
    if (curve == 0) return run(Bn254{});
    if (curve == 1) return run(Bls12_381{});
    throw Error("unknown curve");
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// `co-circom translate-witness` (co-circom.rs:925-962, co_circom::translate_witness lib.rs:93-135): the three parties'
// Rep3 `.shared` files -> their 3-party, threshold-1 Shamir `.shared` files. Replicated shares are translated locally
// by translate_primefield_repshare_vec on the device (bridges/rep3_to_shamir.rs:43-62). Additive half shares are first
// completed with one reshare_vec round and then translated the same way -- the reference instead runs a degree
// reduction with fresh Shamir preprocessing (rep3_to_shamir.rs:79-95); both yield valid degree-1 sharings of the same
// witness, the share values are random in either case. Files written back to back into `out`; returns 3 or -1.
int cog16_translate_witness(int curve, const uint8_t* const* files, const size_t* lens, uint8_t* out, size_t cap, size_t* sizes) {
  try {
    std::vector<std::vector<uint8_t>> outs(3);
    auto run = [&](auto tag) {
      using P = decltype(tag);
      using Fr = typename P::Fr;
      sharefile::CompressedRep3SharedWitness<P> sh[3];
      for (int p = 0; p < 3; ++p) sh[p] = sharefile::read_rep3<P>(files[p], lens[p]);
      size_t len[3];
      for (int p = 0; p < 3; ++p) len[p] = sh[p].length();
      if (sh[0].kind != sh[1].kind || sh[0].kind != sh[2].kind) throw Error("the parties' share files use different compression");
      if (len[0] != len[1] || len[0] != len[2]) throw Error("the parties' share files differ in length");
      auto nets = LocalNetwork::new_parties(3);
      std::string errs[3];
      std::vector<std::thread> th;
      int dev = 0;
      check(csh_current_device(&dev), "csh_current_device");
      for (int p = 0; p < 3; ++p) {
        th.emplace_back([&, p] {
          try {
            check(csh_init(dev), "csh_init");
            auto rep = sharefile::uncompress<P>(std::move(sh[p]), nets[p]);
            SharedWitness<P, Fr> sw;
            sw.public_inputs = rep.public_inputs;
            sw.witness.resize(rep.witness.size());
            const uint64_t e = p + 1, z1 = p == 0 ? 3 : p, z2 = p == 2 ? 1 : p + 2;  // get_translation_points (:14-28): f(X) = 1 - X/z
            Fr x = Fr::sub(Fr::one(), Fr::mul(Fr::from_u64(e), Fr::inv(Fr::from_u64(z1))));
            Fr y = Fr::sub(Fr::one(), Fr::mul(Fr::from_u64(e), Fr::inv(Fr::from_u64(z2))));
            if (!rep.witness.empty())
              check(csh_rep3_to_shamir_vec(P::ID, (const uint64_t*)rep.witness.data(), (const uint64_t*)&x, (const uint64_t*)&y,
                                           (uint64_t*)sw.witness.data(), rep.witness.size()), "csh_rep3_to_shamir_vec");
            outs[p] = sharefile::write_shamir<P>(sw);
          } catch (const std::exception& ex) {
            errs[p] = ex.what();
          }
        });
      }
      for (auto& t : th) t.join();
      for (int p = 0; p < 3; ++p)
        if (!errs[p].empty()) throw Error("party " + std::to_string(p) + ": " + errs[p]);
    };
    if (curve == 0) run(Bn254{});
    else if (curve == 1) run(Bls12_381{});
    else throw Error("unknown curve");
    size_t at = 0;
    for (int p = 0; p < 3; ++p) {
      if (at + outs[p].size() > cap) throw Error("output buffer too small");
      memcpy(out + at, outs[p].data(), outs[p].size());
      sizes[p] = outs[p].size();
      at += outs[p].size();
    }
    return 3;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

int cog16_prove_shamir(int curve, const uint8_t* zkey, size_t zlen, const uint8_t* wtns, size_t wlen, int num_parties, int threshold,
                       uint64_t seed, const uint64_t* r, const uint64_t* s, int bridge, char* out_json, size_t cap) {
  try {
    if (curve == 0) return prove_shamir_t<Bn254>(zkey, zlen, wtns, wlen, num_parties, threshold, seed, r, s, bridge, out_json, cap);
    if (curve == 1) return prove_shamir_t<Bls12_381>(zkey, zlen, wtns, wlen, num_parties, threshold, seed, r, s, bridge, out_json, cap);
    g_err = "unknown curve";
    return -1;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// out_ms[6] = {witness_map ms, create_proof ms, total prove ms, key generation+upload ms, three-party Rep3 prove wall ms,
// Rep3 proofs == plain proof (1/0)}; best of `iters`.
int cog16_bench_synthetic2(int curve, int log_domain, int iters, double* out_ms /* 6 entries */, int* check_ok, int with_rep3,
                           double* phases_out /* 3 entries: witness, msm, finish ms of the best prove; nullable */);
int cog16_bench_synthetic3(int curve, int log_domain, int iters, double* out_ms, int* check_ok, int with_rep3, double* phases_out, double* trait_out);
int cog16_bench_synthetic4(int curve, int log_domain, int iters, double* out_ms, int* check_ok, int with_rep3, double* phases_out, double* trait_out,
                           double* rep3_trait_out, double* mins_out);
int cog16_bench_synthetic(int curve, int log_domain, int iters, double* out_ms /* 6 entries */, int* check_ok, int with_rep3) {
  return cog16_bench_synthetic2(curve, log_domain, iters, out_ms, check_ok, with_rep3, nullptr);
}
int cog16_bench_synthetic2(int curve, int log_domain, int iters, double* out_ms, int* check_ok, int with_rep3, double* phases_out) {
  return cog16_bench_synthetic3(curve, log_domain, iters, out_ms, check_ok, with_rep3, phases_out, nullptr);
}
// ... and trait_out (5 entries, nullable): the same prove through the trait path (see bench_synth_t)
int cog16_bench_synthetic3(int curve, int log_domain, int iters, double* out_ms, int* check_ok, int with_rep3, double* phases_out, double* trait_out) {
  return cog16_bench_synthetic4(curve, log_domain, iters, out_ms, check_ok, with_rep3, phases_out, trait_out, nullptr, nullptr);
}
// + rep3_trait_out (26 doubles, layout at bench_synth_t) and mins_out (3 doubles: minimum of prove, trait path, three Rep3 parties)
int cog16_bench_synthetic4(int curve, int log_domain, int iters, double* out_ms, int* check_ok, int with_rep3, double* phases_out, double* trait_out,
                           double* rep3_trait_out, double* mins_out) {
  try {
    if (iters < 1) throw Error("cog16_bench_synthetic: iters < 1");
    if (curve == 0) return bench_synth_t<Bn254>(log_domain, iters, out_ms, check_ok, csh::Bn254G1Gen, csh::Bn254G2Gen, with_rep3 != 0, phases_out, trait_out, rep3_trait_out, mins_out);
    if (curve == 1) return bench_synth_t<Bls12_381>(log_domain, iters, out_ms, check_ok, csh::Bls381G1Gen, csh::Bls381G2Gen, with_rep3 != 0, phases_out, trait_out, rep3_trait_out, mins_out);
    g_err = "unknown curve";
    return -1;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}


int cog16_bench_rep3_party_per_gpu(int curve, int log_domain, int iters, const int devices[3], double* out /* 3 entries */) {
  try {
    if (!devices || !out || iters < 1) throw Error("cog16_bench_rep3_party_per_gpu: bad arguments");
    if (curve == 0) return bench_rep3_party_per_gpu_t<Bn254>(log_domain, iters, devices, out, csh::Bn254G1Gen, csh::Bn254G2Gen);
    if (curve == 1) return bench_rep3_party_per_gpu_t<Bls12_381>(log_domain, iters, devices, out, csh::Bls381G1Gen, csh::Bls381G2Gen);
    throw Error("unknown curve");
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

const char* cog16_last_error(void) { return g_err.c_str(); }

// The mirror's Rep3 randomness on its own (host only, no device): three parties with PRF keys k_p = keys[32 p ..] (party p holds its
// own key as rng1 and its predecessor's as rng2, rep3.rs:71-76). Per party, in this order: a mask vector of n elements
// (masking_field_elements_vec, rngs.rs:137-156; drawn in parallel over the host's cores), one random_seeds() pair (rngs.rs:233), a second
// mask vector of n2 elements (the generators must continue where a serial draw would have left them).
// masks_out: 3 x (n + n2) x 4 limbs (Montgomery); seeds_out: 3 x 64 bytes (seed1 || seed2).
int cog16_rep3_rand_selftest(int curve, const uint8_t keys[96], size_t n, size_t n2, uint64_t* masks_out, uint8_t* seeds_out) {
  try {
    if (!keys || !masks_out || !seeds_out) throw Error("cog16_rep3_rand_selftest: NULL argument");
    auto run = [&](auto tag) {
      using Fr = decltype(tag);
      for (int p = 0; p < 3; ++p) {
        Rep3Rand r(keys + 32 * p, keys + 32 * ((p + 2) % 3));
        const UninitBuf<Fr> a = r.masking_field_elements_vec<Fr>(n);
        r.random_seeds(seeds_out + 64 * p, seeds_out + 64 * p + 32);
        const UninitBuf<Fr> b = r.masking_field_elements_vec<Fr>(n2);
        uint64_t* o = masks_out + 4 * (n + n2) * (size_t)p;
        if (n) memcpy(o, a.data(), 32 * n);
        if (n2) memcpy(o + 4 * n, b.data(), 32 * n2);
      }
    };
    if (curve == 0) run(Bn254::Fr{});
    else if (curve == 1) run(Bls12_381::Fr{});
    else throw Error("unknown curve");
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// 1: every prove_inner of this process runs the "trait path" (groth16.hpp: the sequence rust/co-groth16-hip drives behind the unchanged
// reference -- host slices at every seam, one witness-map call, five concurrent host-scalar MSMs); 0: the device-resident prove.
// 2: the trait path with the shim's opt-in seeded Rep3 masks (groth16.hpp witness_map_trait_path).
int cog16_set_trait_path(int on) {
  trait_path_flag().store(on == 2 ? 2 : (on ? 1 : 0));
  return 0;
}
int cog16_get_trait_path(void) { return trait_path_flag().load(); }

// The synthetic circuit as an object: open (key + matrices + witness resident), prove (one plain prove_inner, phases out),
// check (closed form), close. bench.py's `--workload groth16_prove` times K cog16_synth_prove calls between barriers.
int cog16_synth_open(int curve, int log_domain, void** out) {
  try {
    if (!out || log_domain < 3 || log_domain > 26) throw Error("cog16_synth_open: bad arguments");
    if (curve == 0) *out = static_cast<SynthBase*>(new SynthCircuit<Bn254>(log_domain, csh::Bn254G1Gen, csh::Bn254G2Gen));
    else if (curve == 1) *out = static_cast<SynthBase*>(new SynthCircuit<Bls12_381>(log_domain, csh::Bls381G1Gen, csh::Bls381G2Gen));
    else throw Error("unknown curve");
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
int cog16_synth_prove(void* h, double* phases_out /* witness, msm, finish ms; nullable */, double* key_ms /* nullable */) {
  try {
    if (!h) throw Error("cog16_synth_prove: NULL handle");
    SynthBase* c = static_cast<SynthBase*>(h);
    c->prove(false);
    if (phases_out) {
      phases_out[0] = c->phases.witness_ms;
      phases_out[1] = c->phases.msm_ms;
      phases_out[2] = c->phases.finish_ms;
    }
    if (key_ms) *key_ms = c->key_ms;
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
int cog16_synth_check(void* h, int* ok) {
  try {
    if (!h || !ok) throw Error("cog16_synth_check: NULL argument");
    SynthBase* c = static_cast<SynthBase*>(h);
    c->prove(true);
    *ok = c->closed_form() ? 1 : 0;
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
int cog16_synth_close(void* h) {
  delete static_cast<SynthBase*>(h);
  return 0;
}

// The placement plan for a key whose five queries (a, b_g1, b_g2, l, h) have `sizes` points, on `nslots` GPUs: host-only (no device
// needed). slots_out[q] = the slot of query q (by query), ranges_out[2 * slot + {0, 1}] = [lo, hi) of n_range entries (by range);
// returns the effective mode (1 by query, 2 by range) or -1.
int cog16_placement_plan(const size_t sizes[5], int nslots, int mode, size_t n_range, int slots_out[5], size_t* ranges_out) {
  try {
    if (!sizes || !slots_out || nslots < 1 || nslots > 64 || mode < 0 || mode > 2) throw Error("cog16_placement_plan: bad arguments");
    const int eff = plan_placement(sizes, (size_t)nslots, mode, slots_out);
    if (ranges_out)
      for (int sl = 0; sl < nslots; ++sl) {
        if (eff == PLACE_BY_RANGE && nslots > 1) plan_range(n_range, (size_t)nslots, (size_t)sl, &ranges_out[2 * sl], &ranges_out[2 * sl + 1]);
        else {
          ranges_out[2 * sl] = 0;
          ranges_out[2 * sl + 1] = n_range;
        }
      }
    return eff;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// One prover's five query MSMs over several GPUs: keys built after this call clone their queries onto `devices` (entry 0 = the
// key's home GPU; a GPU may be listed more than once, which is how the single-GPU tests exercise the path). n <= 1 switches it off.
int cog16_set_prover_devices_mode(const int* devices, int n, int mode);
int cog16_set_prover_devices(const int* devices, int n) { return cog16_set_prover_devices_mode(devices, n, 0); }
// mode: 0 = automatic (whole queries per GPU up to two GPUs, ranges of every query from three on), 1 = whole queries, 2 = ranges
int cog16_set_prover_devices_mode(const int* devices, int n, int mode) {
  try {
    if (mode < 0 || mode > 2) throw Error("cog16_set_prover_devices: mode must be 0 (auto), 1 (by query) or 2 (by range)");
    int ndev = 0;
    check(csh_device_count(&ndev), "csh_device_count");
    std::vector<int> d;
    for (int i = 0; i < n; ++i) {
      if (!devices || devices[i] < 0 || devices[i] >= (ndev > 0 ? ndev : 1)) throw Error("cog16_set_prover_devices: device out of range");
      d.push_back(devices[i]);
    }
    std::lock_guard<std::mutex> g(ProverDevices::get().mu);
    ProverDevices::get().devices = d.size() > 1 ? d : std::vector<int>();
    ProverDevices::get().mode = mode;
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

int cog16_prove_plain(int curve, const uint8_t* zkey, size_t zlen, const uint8_t* wtns, size_t wlen, const uint64_t* r, const uint64_t* s,
                      char* out_json, size_t cap, uint64_t* h_out, size_t h_cap_elems) {
  try {
    if (curve == 0) return prove_plain_t<Bn254>(zkey, zlen, wtns, wlen, r, s, out_json, cap, h_out, h_cap_elems);
    if (curve == 1) return prove_plain_t<Bls12_381>(zkey, zlen, wtns, wlen, r, s, out_json, cap, h_out, h_cap_elems);
    g_err = "unknown curve";
    return -1;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

int cog16_prove_rep3(int curve, const uint8_t* zkey, size_t zlen, const uint8_t* wtns, size_t wlen, uint64_t seed, const uint64_t* r,
                     const uint64_t* s, char* out_json, size_t cap, uint64_t* h_shares_out, size_t h_cap_elems) {
  try {
    if (curve == 0) return prove_rep3_t<Bn254>(zkey, zlen, wtns, wlen, seed, r, s, out_json, cap, h_shares_out, h_cap_elems);
    if (curve == 1) return prove_rep3_t<Bls12_381>(zkey, zlen, wtns, wlen, seed, r, s, out_json, cap, h_shares_out, h_cap_elems);
    g_err = "unknown curve";
    return -1;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

}  // extern "C"
