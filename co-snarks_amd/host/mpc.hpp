// Drivers mirroring `CircomGroth16Prover<P>` (co-circom/co-groth16/src/mpc.rs:22-138) and its three
// implementors PlainGroth16Driver (mpc/plain.rs:13-134), Rep3Groth16Driver (mpc/rep3.rs:16-162) and
// ShamirGroth16Driver (mpc/shamir.rs:14-148, local methods). Same method names and argument meaning; the hot
// methods (local_mul_vec, distribute_powers_and_mul_by_const, msm_public_points_hs and the NTTs inside the
// reduction) call the C ABI of include/cosnarks_hip.h, everything else is host code as in the reference.
#pragma once
#include <deque>
#include <random>

#include "network.hpp"
#include "types.hpp"

namespace cosnarks {

// A slice of a device-resident query: the mirror of `&query[lo..]`
struct BasesView {
  csh_bases_t bases;
  size_t offset, len;
};

template <class F>
inline Proj<F> msm_device(const BasesView& v, const void* scalars_mont, size_t n) {
  csh::Jac<F> out;
  const size_t cnt = n < v.len ? n : v.len;  // msm_unchecked: the shorter of the two slices
  check(csh_msm(v.bases, v.offset, cnt, reinterpret_cast<const uint64_t*>(scalars_mont), 1, &out), "csh_msm");
  if (out.is_inf()) return Proj<F>::inf();
  return Proj<F>::from_affine(AffineT<F>{out.x, out.y});
}

// scalars already in HBM, given as a raw device pointer (a range of a resident vector)
template <class F>
inline Proj<F> msm_device_resident_ptr(const BasesView& v, const void* scalars_dev, size_t n) {
  csh::Jac<F> out;
  const size_t cnt = n < v.len ? n : v.len;
  check(csh_msm_dev(v.bases, v.offset, cnt, reinterpret_cast<const uint64_t*>(scalars_dev), 1, &out, nullptr), "csh_msm_dev");
  if (out.is_inf()) return Proj<F>::inf();
  return Proj<F>::from_affine(AffineT<F>{out.x, out.y});
}
template <class F>
inline Proj<F> msm_device_resident(const BasesView& v, const DeviceScalars& s) { return msm_device_resident_ptr<F>(v, s.dev, s.n); }

// ---- Rep3 types (mpc-core/src/protocols/rep3/arithmetic/types.rs:21-28, rngs.rs:83-187) -----------------------
template <class Fr>
struct Rep3PrimeFieldShare {
  Fr a, b;
};

struct Rep3Rand {
  ChaCha12 rng1, rng2;  // own key, previous party's key (rep3.rs:71-76 setup_prf)
  Rep3Rand(const uint8_t s1[32], const uint8_t s2[32]) : rng1(s1), rng2(s2) {}
  // rngs.rs:137-156. The reference fills its two byte vectors under rayon::join and converts them with par_chunks(..).with_min_len(512)
  // into a freshly collected Vec. The keystream is counter-mode, so the mirror cuts BOTH streams into the same element ranges over the
  // host's cores (same bytes, same elements, the generators end where a serial draw would leave them) and writes into memory that
  // is not zero-filled first -- this is the host cost a Rep3 party pays per local_mul_vec on the zero-upstream-edit path
  // (bench: rep3_trait_path.mask_draw_ms).
  template <class Fr>
  UninitBuf<Fr> masking_field_elements_vec(size_t len) {
    UninitBuf<Fr> out(len);
    const uint64_t p1 = rng1.byte_pos(), p2 = rng2.byte_pos();
    const ChaCha12 &r1 = rng1, &r2 = rng2;
    Fr* o = out.data();
    parallel_for(len, 1 << 13, [&r1, &r2, p1, p2, o](size_t lo, size_t hi) {
      ChaCha12 a = r1, b = r2;
      a.seek(p1 + 32 * (uint64_t)lo);
      b.seek(p2 + 32 * (uint64_t)lo);
      uint8_t x[32], y[32];
      for (size_t i = lo; i < hi; ++i) {
        a.fill_bytes(x, 32);
        b.fill_bytes(y, 32);
        o[i] = mask_element_from_be_bytes<Fr>(x, y);
      }
    });
    rng1.seek(p1 + 32 * (uint64_t)len);
    rng2.seek(p2 + 32 * (uint64_t)len);
    return out;
  }
  template <class Fr>
  std::pair<Fr, Fr> random_fes() {  // rngs.rs:109-113 (sampling restated as one 32-byte mod-order draw per stream)
    uint8_t a[32], b[32];
    rng1.fill_bytes(a, 32);
    rng2.fill_bytes(b, 32);
    return {from_be_bytes_mod_order<Fr>(a), from_be_bytes_mod_order<Fr>(b)};
  }
  // Hand a run of `n` mask elements to the device generator (csh_rep3_masks / csh_groth16_h_rep3_seeded): returns the
  // chunk offsets of the run in both streams and advances both generators by 32*n bytes, exactly what
  // masking_field_elements_vec(n) would have consumed.
  struct DeviceRun {
    uint8_t seed1[32], seed2[32];
    uint64_t off1, off2;
  };
  DeviceRun take_device_run(size_t n) {
    DeviceRun r;
    memcpy(r.seed1, rng1.key, 32);
    memcpy(r.seed2, rng2.key, 32);
    const uint64_t p1 = rng1.byte_pos(), p2 = rng2.byte_pos();
    if (p1 % 32 || p2 % 32) throw Error("Rep3Rand: stream position not aligned to a field-element chunk");
    r.off1 = p1 / 32;
    r.off2 = p2 / 32;
    rng1.seek(p1 + 32 * (uint64_t)n);
    rng2.seek(p2 + 32 * (uint64_t)n);
    return r;
  }
  // `rng.gen::<[u8; 32]>()` of rand 0.8 (the reference pins rand = "0.8", Cargo.toml:70; un-vendored): Standard samples an array element
  // by element and a u8 as `next_u32() as u8`, so a 32-byte seed consumes 32 keystream WORDS (128 bytes), one low byte each. Restated
  // from the published crate; no reference fixture pins it (it only matters if a mirror party ever shared a session with a reference
  // party after a fork -- the Rust shim runs the reference's own code here).
  static void gen_seed(ChaCha12& rng, uint8_t out[32]) {
    for (int i = 0; i < 32; ++i) {
      uint8_t w[4];
      rng.fill_bytes(w, 4);
      out[i] = w[0];  // little-endian word: the low byte
    }
  }
  // rngs.rs:233-237: one fresh seed from each generator. Party i's rng1 is party i+1's rng2 (rep3.rs:71-76), so the seed party i draws
  // from rng1 equals the one party i+1 draws from rng2: streams keyed with them are correlated exactly like the parents.
  void random_seeds(uint8_t s1[32], uint8_t s2[32]) {
    gen_seed(rng1, s1);
    gen_seed(rng2, s2);
  }
  Rep3Rand fork() {  // rngs.rs:96-100
    uint8_t s1[32], s2[32];
    random_seeds(s1, s2);
    return Rep3Rand(s1, s2);
  }
};

struct Rep3State {
  int id;
  Rep3Rand rand;
  // Rep3State::new (rep3.rs:56-76): draw a seed, send it to the next party, receive the previous party's
  static Rep3State create(const LocalNetwork& net, const uint8_t my_seed[32]) {
    struct Seed { uint8_t b[32]; } s, prev;
    memcpy(s.b, my_seed, 32);
    prev = net.reshare(s);
    return Rep3State{net.id(), Rep3Rand(s.b, prev.b)};
  }
  Rep3State fork(size_t) { return Rep3State{id, rand.fork()}; }
};

struct UnitState {
  int id = 0;
  UnitState fork(size_t) { return *this; }
};

// ================================= PlainGroth16Driver (mpc/plain.rs) =================================
template <class P>
struct PlainGroth16Driver {
  using Fr = typename P::Fr;
  using ArithmeticShare = Fr;
  using ArithmeticHalfShare = Fr;
  using State = UnitState;
  using Net = LocalNetwork;
  static constexpr int PROTOCOL = 0;   // csh_groth16_h protocol id
  static constexpr uint32_t NCOMP = 1;
  static constexpr bool DEVICE_MASKS = false;

  static ArithmeticShare rand(const Net*, State&) {  // mpc/plain.rs:23-26
    uint8_t b[32];
    secure_random_bytes(b, 32);  // thread_rng() in the reference: a CSPRNG
    return from_be_bytes_mod_order<Fr>(b);
  }
  static ArithmeticShare evaluate_constraint(int, const std::vector<std::pair<Fr, size_t>>& lhs, const std::vector<Fr>& pub,
                                             const std::vector<ArithmeticShare>& wit) {  // mpc/plain.rs:28-43
    Fr acc = Fr::zero();
    for (auto& [coeff, index] : lhs)
      acc = Fr::add(acc, Fr::mul(coeff, index < pub.size() ? pub[index] : wit[index - pub.size()]));
    return acc;
  }
  static std::vector<ArithmeticShare> promote_to_trivial_shares(int, const std::vector<Fr>& v) { return v; }
  static std::vector<Fr> local_mul_vec(const std::vector<ArithmeticShare>& a, const std::vector<ArithmeticShare>& b, State&) {
    std::vector<Fr> out(a.size());  // mpc/plain.rs:83-89
    check(csh_vec_mul(P::ID, (const uint64_t*)a.data(), (const uint64_t*)b.data(), (uint64_t*)out.data(), a.size()), "csh_vec_mul");
    return out;
  }
  static void distribute_powers_and_mul_by_const(std::vector<ArithmeticShare>& c, const std::vector<Fr>& roots) {
    check(csh_vec_mul_table(P::ID, (uint64_t*)c.data(), (const uint64_t*)roots.data(), c.size(), 1), "csh_vec_mul_table");
  }
  static ArithmeticHalfShare to_half_share(const ArithmeticShare& a) { return a; }
  static UninitBuf<Fr> masks(State&, size_t) { return {}; }
  template <class F>
  static Proj<F> msm_public_points_hs(const BasesView& pts, const std::vector<ArithmeticHalfShare>& s) {  // mpc/plain.rs:66-74
    return msm_device<F>(pts, s.data(), s.size());
  }
  template <class F>
  static Proj<F> msm_public_points_hs(const BasesView& pts, const DeviceScalars& s) {  // same MSM, scalars already in HBM
    return msm_device_resident<F>(pts, s);
  }
  template <class F>
  static Proj<F> scalar_mul_public_point_hs(const Proj<F>& a, const ArithmeticHalfShare& b) { return point_mul(a, b); }
  template <class F>
  static void add_assign_points_public_hs(int, Proj<F>& a, const Proj<F>& b) { a = point_add(a, b); }
  template <class F>
  static Proj<F> open_half_point(const Proj<F>& a, const Net*, State&) { return a; }
  template <class F>
  static Proj<F> scalar_mul(const Proj<F>& a, const ArithmeticShare& b, const Net*, State&) { return point_mul(a, b); }
};

// ================================= Rep3Groth16Driver (mpc/rep3.rs) =================================
template <class P>
struct Rep3Groth16Driver {
  using Fr = typename P::Fr;
  using ArithmeticShare = Rep3PrimeFieldShare<Fr>;
  using ArithmeticHalfShare = Fr;
  using State = Rep3State;
  using Net = LocalNetwork;
  static constexpr int PROTOCOL = 1;
  static constexpr uint32_t NCOMP = 2;
  static constexpr bool DEVICE_MASKS = true;  // masks of the big local_mul_vec calls are generated on the GPU ("next" row f2)

  static ArithmeticShare rand(const Net*, State& st) {  // mpc/rep3.rs:27-29 -> arithmetic::rand (arithmetic.rs:357-360)
    auto [a, b] = st.rand.template random_fes<Fr>();
    return {a, b};
  }
  static ArithmeticShare evaluate_constraint(int id, const std::vector<std::pair<Fr, size_t>>& lhs, const std::vector<Fr>& pub,
                                             const std::vector<ArithmeticShare>& wit) {  // mpc/rep3.rs:31-49
    ArithmeticShare acc{Fr::zero(), Fr::zero()};
    for (auto& [coeff, index] : lhs) {
      if (index < pub.size()) {
        Fr m = Fr::mul(pub[index], coeff);  // add_assign_public: party 0 -> a, party 1 -> b (arithmetic.rs:52-58)
        if (id == 0) acc.a = Fr::add(acc.a, m);
        else if (id == 1) acc.b = Fr::add(acc.b, m);
      } else {
        const ArithmeticShare& w = wit[index - pub.size()];
        acc.a = Fr::add(acc.a, Fr::mul(w.a, coeff));
        acc.b = Fr::add(acc.b, Fr::mul(w.b, coeff));
      }
    }
    return acc;
  }
  static std::vector<ArithmeticShare> promote_to_trivial_shares(int id, const std::vector<Fr>& v) {  // types.rs:69-82
    std::vector<ArithmeticShare> out(v.size(), {Fr::zero(), Fr::zero()});
    for (size_t i = 0; i < v.size(); ++i) {
      if (id == 0) out[i].a = v[i];
      else if (id == 1) out[i].b = v[i];
    }
    return out;
  }
  static UninitBuf<Fr> masks(State& st, size_t n) { return st.rand.template masking_field_elements_vec<Fr>(n); }
  static std::vector<Fr> local_mul_vec(const std::vector<ArithmeticShare>& a, const std::vector<ArithmeticShare>& b, State& st) {
    const UninitBuf<Fr> mask = masks(st, a.size());  // arithmetic.rs:132-146
    std::vector<Fr> out(a.size());
    check(csh_rep3_local_mul_vec(P::ID, (const uint64_t*)a.data(), (const uint64_t*)b.data(), (const uint64_t*)mask.data(),
                                 (uint64_t*)out.data(), a.size()), "csh_rep3_local_mul_vec");
    return out;
  }
  static void distribute_powers_and_mul_by_const(std::vector<ArithmeticShare>& c, const std::vector<Fr>& roots) {  // mpc/rep3.rs:95-106
    check(csh_vec_mul_table(P::ID, (uint64_t*)c.data(), (const uint64_t*)roots.data(), c.size(), 2), "csh_vec_mul_table");
  }
  static ArithmeticHalfShare to_half_share(const ArithmeticShare& a) { return a.a; }  // mpc/rep3.rs:120-122
  template <class F>
  static Proj<F> msm_public_points_hs(const BasesView& pts, const std::vector<ArithmeticHalfShare>& s) {  // mpc/rep3.rs:124-132
    return msm_device<F>(pts, s.data(), s.size());
  }
  template <class F>
  static Proj<F> msm_public_points_hs(const BasesView& pts, const DeviceScalars& s) {  // same MSM, scalars already in HBM
    return msm_device_resident<F>(pts, s);
  }
  template <class F>
  static Proj<F> scalar_mul_public_point_hs(const Proj<F>& a, const ArithmeticHalfShare& b) { return point_mul(a, b); }
  template <class F>
  static void add_assign_points_public_hs(int id, Proj<F>& a, const Proj<F>& b) {  // mpc/rep3.rs:108-118
    if (id == 0) a = point_add(a, b);
  }
  template <class F>
  static Proj<F> open_half_point(const Proj<F>& a, const Net* net, State&) {  // rep3/pointshare.rs:152-155
    AffineT<F> mine = into_affine(a);
    auto [pb, pc] = net->broadcast(mine);
    return point_add(point_add(a, into_group(pb)), into_group(pc));
  }
  template <class F>
  static Proj<F> scalar_mul(const Proj<F>& a, const ArithmeticShare& b, const Net* net, State& st) {  // mpc/rep3.rs:152-161
    AffineT<F> mine = into_affine(a);
    Proj<F> a_hs = into_group(net->reshare(mine));
    // b * point: rhs.a*self.a + rhs.b*self.a + rhs.a*self.b (rep3/pointshare/ops.rs:95-102) + masking_ec_element
    Proj<F> r = point_add(point_add(point_mul(a, b.a), point_mul(a_hs, b.a)), point_mul(a, b.b));
    // + masking_ec_element = C::rand(rng1) - C::rand(rng2) (rngs.rs:177-187, pointshare.rs:124): re-randomises the half share
    // before it is opened. A uniform group element per stream = a uniform multiple of the generator: (m1 - m2) * G, the three
    // parties' masks summing to the identity exactly as the reference's do.
    auto [m1, m2] = st.rand.template random_fes<Fr>();
    AffineT<F> gen;
    static_assert(sizeof(gen) == sizeof(AffineT<typename P::Fq>), "scalar_mul is only used on G1 (mpc.rs:131-137)");
    memcpy(&gen, P::g1_generator_words(), sizeof gen);
    return point_add(r, point_mul(into_group(gen), Fr::sub(m1, m2)));
  }
};

// ================================= Shamir (mpc-core/src/protocols/shamir*.rs) =================================
// lagrange_from_coeff (shamir.rs:442-460)
template <class Fr>
inline std::vector<Fr> lagrange_from_coeff(const std::vector<size_t>& coeffs) {
  std::vector<Fr> res;
  for (size_t i : coeffs) {
    Fr num = Fr::one(), den = Fr::one();
    const Fr fi = Fr::from_u64(i);
    for (size_t j : coeffs)
      if (i != j) {
        const Fr fj = Fr::from_u64(j);
        num = Fr::mul(num, fj);
        den = Fr::mul(den, Fr::sub(fj, fi));
      }
    res.push_back(Fr::mul(num, Fr::inv(den)));
  }
  return res;
}

template <class Fr>
struct ShamirState {
  int id = 0;
  size_t num_parties = 0, threshold = 0;
  std::vector<Fr> open_lagrange_t, open_lagrange_2t, mul_lagrange_2t;
  std::vector<Fr> mul_reconstruct_with_zeros;       // q(X): q(0) = 1, q(j) = 0 for j in num_non_zero+1..n
  std::deque<std::pair<Fr, Fr>> rng_buffer;         // (r_t, r_2t) double sharings (shamir/rngs.rs:12-60; dealt here)
  // From<ShamirPreprocessing> (shamir.rs:66-103)
  static ShamirState create(int id, size_t n, size_t t, std::deque<std::pair<Fr, Fr>> pairs) {
    ShamirState s;
    s.id = id;
    s.num_parties = n;
    s.threshold = t;
    auto ring = [&](size_t cnt) {
      std::vector<size_t> c;
      for (size_t i = 0; i < cnt; ++i) c.push_back((id + n - i) % n + 1);
      return c;
    };
    s.open_lagrange_t = lagrange_from_coeff<Fr>(ring(t + 1));
    s.open_lagrange_2t = lagrange_from_coeff<Fr>(ring(2 * t + 1));
    std::vector<size_t> first;
    for (size_t i = 1; i <= 2 * t + 1; ++i) first.push_back(i);
    s.mul_lagrange_2t = lagrange_from_coeff<Fr>(first);
    // interpolation_poly_from_zero_points (shamir.rs:568-586)
    const size_t num_non_zero = n - t;
    std::vector<Fr> num{Fr::one()};
    Fr d = Fr::one();
    for (size_t j = num_non_zero + 1; j <= n; ++j) {
      const Fr fj = Fr::from_u64(j);
      num.insert(num.begin(), Fr::zero());  // poly_times_root_inplace: num = num * (X - j)
      for (size_t k = 1; k < num.size(); ++k) num[k - 1] = Fr::sub(num[k - 1], Fr::mul(num[k], fj));
      d = Fr::mul(d, Fr::neg(fj));
    }
    const Fr c = Fr::inv(d);
    for (auto& x : num) x = Fr::mul(x, c);
    s.mul_reconstruct_with_zeros = num;
    s.rng_buffer = std::move(pairs);
    return s;
  }
  ShamirState fork(size_t amount) {  // ShamirState::fork: hands `amount` pairs to the forked state
    ShamirState s = *this;
    s.rng_buffer.clear();
    for (size_t i = 0; i < amount; ++i) {
      s.rng_buffer.push_back(rng_buffer.back());
      rng_buffer.pop_back();
    }
    return s;
  }
  std::pair<Fr, Fr> get_pair() {
    if (rng_buffer.empty()) throw Error("Shamir: out of preprocessed double sharings");
    auto p = rng_buffer.front();
    rng_buffer.pop_front();
    return p;
  }
};

template <class P>
struct ShamirGroth16Driver {
  using Fr = typename P::Fr;
  using ArithmeticShare = Fr;        // ShamirPrimeFieldShare is repr(transparent)
  using ArithmeticHalfShare = Fr;    // degree-2t sharing
  using State = ShamirState<Fr>;
  using Net = LocalNetwork;
  static constexpr int PROTOCOL = 0;
  static constexpr uint32_t NCOMP = 1;
  static constexpr bool DEVICE_MASKS = false;

  static ArithmeticShare rand(const Net*, State& st) { return st.get_pair().first; }  // ShamirState::rand: the degree-t half of a pair
  static ArithmeticShare evaluate_constraint(int, const std::vector<std::pair<Fr, size_t>>& lhs, const std::vector<Fr>& pub,
                                             const std::vector<ArithmeticShare>& wit) {  // mpc/shamir.rs:29-49: public values add to every share
    return PlainGroth16Driver<P>::evaluate_constraint(0, lhs, pub, wit);
  }
  static std::vector<ArithmeticShare> promote_to_trivial_shares(int, const std::vector<Fr>& v) { return v; }  // shamir/arithmetic.rs:229-231
  static UninitBuf<Fr> masks(State&, size_t) { return {}; }
  static std::vector<Fr> local_mul_vec(const std::vector<Fr>& a, const std::vector<Fr>& b, State&) {  // shamir/arithmetic.rs:73-79
    std::vector<Fr> out(a.size());
    check(csh_vec_mul(P::ID, (const uint64_t*)a.data(), (const uint64_t*)b.data(), (uint64_t*)out.data(), a.size()), "csh_vec_mul");
    return out;
  }
  static void distribute_powers_and_mul_by_const(std::vector<ArithmeticShare>& c, const std::vector<Fr>& roots) {  // mpc/shamir.rs:85-96
    check(csh_vec_mul_table(P::ID, (uint64_t*)c.data(), (const uint64_t*)roots.data(), c.size(), 1), "csh_vec_mul_table");
  }
  static ArithmeticHalfShare to_half_share(const ArithmeticShare& a) { return a; }  // mpc/shamir.rs:107-109
  template <class F>
  static Proj<F> msm_public_points_hs(const BasesView& pts, const std::vector<ArithmeticHalfShare>& s) {  // mpc/shamir.rs:111-119
    return msm_device<F>(pts, s.data(), s.size());
  }
  template <class F>
  static Proj<F> msm_public_points_hs(const BasesView& pts, const DeviceScalars& s) { return msm_device_resident<F>(pts, s); }
  template <class F>
  static Proj<F> scalar_mul_public_point_hs(const Proj<F>& a, const ArithmeticHalfShare& b) { return point_mul(a, b); }
  template <class F>
  static void add_assign_points_public_hs(int, Proj<F>& a, const Proj<F>& b) { a = point_add(a, b); }  // mpc/shamir.rs:98-105
  // shamir/pointshare.rs:102-110 + network.rs:96-126 broadcast_next(n, 2t+1)
  template <class F>
  static Proj<F> open_half_point(const Proj<F>& a, const Net* net, State& st) {
    const size_t n = st.num_parties, num = 2 * st.threshold + 1;
    AffineT<F> mine = into_affine(a);
    Bytes b(sizeof mine);
    memcpy(b.data(), &mine, sizeof mine);
    for (size_t s = 1; s < num; ++s) net->send((int)((st.id + s) % n), b);
    Proj<F> res = point_mul(a, st.open_lagrange_2t[0]);
    for (size_t r = 1; r < num; ++r) {
      Bytes rb = net->recv((int)((st.id + n - r) % n));
      AffineT<F> o;
      memcpy(&o, rb.data(), sizeof o);
      res = point_add(res, point_mul(into_group(o), st.open_lagrange_2t[r]));
    }
    return res;
  }
  // mpc/shamir.rs:139-147: degree_reduce_point (shamir/network.rs:246-309, king = party 0) then (b * a).a
  template <class F>
  static Proj<F> scalar_mul(const Proj<F>& a, const ArithmeticShare& b, const Net* net, State& st, const AffineT<F>& generator) {
    const size_t n = st.num_parties, t = st.threshold, num_non_zero = n - t;
    auto [r_t, r_2t] = st.get_pair();
    const Proj<F> g = into_group(generator);
    Proj<F> input = point_add(a, point_mul(g, r_2t));
    Proj<F> my_share = Proj<F>::inf();
    auto send_pt = [&](int to, const Proj<F>& p) {
      AffineT<F> af = into_affine(p);
      Bytes bb(sizeof af);
      memcpy(bb.data(), &af, sizeof af);
      net->send(to, std::move(bb));
    };
    auto recv_pt = [&](int from) {
      Bytes rb = net->recv(from);
      AffineT<F> o;
      memcpy(&o, rb.data(), sizeof o);
      return into_group(o);
    };
    if (st.id == 0) {
      Proj<F> acc = point_mul(input, st.mul_lagrange_2t[0]);
      for (size_t other = 1; other < st.mul_lagrange_2t.size(); ++other) acc = point_add(acc, point_mul(recv_pt((int)other), st.mul_lagrange_2t[other]));
      for (size_t id = 0; id < num_non_zero; ++id) {
        // evaluate acc * q(X) at X = id + 1 (poly_with_zeros_from_precomputed_point + evaluate_poly_point)
        Fr x = Fr::from_u64(id + 1), e = Fr::zero();
        for (size_t k = st.mul_reconstruct_with_zeros.size(); k-- > 0;) e = Fr::add(Fr::mul(e, x), st.mul_reconstruct_with_zeros[k]);
        Proj<F> val = point_mul(acc, e);
        if (id == 0) my_share = val;
        else send_pt((int)id, val);
      }
    } else {
      if ((size_t)st.id <= 2 * t) send_pt(0, input);
      if ((size_t)st.id < num_non_zero) my_share = recv_pt(0);
    }
    Proj<F> reduced = point_add(my_share, point_neg(point_mul(g, r_t)));
    return point_mul(reduced, b);  // scalar_mul_local (shamir/pointshare.rs:97-99)
  }
};

}  // namespace cosnarks
