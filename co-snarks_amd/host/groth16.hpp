// Mirror of co-circom/co-groth16/src/groth16.rs (CoGroth16::{prove_inner, calculate_coeff,
// create_proof_with_assignment}, groth16_roots_of_unity) and groth16/reduction.rs (R1CSToQAP,
// CircomReduction) above the C ABI. Same control flow, line-cited; the hot calls go to the device.
#pragma once
#include <array>
#include <atomic>
#include <memory>
#include <map>
#include <tuple>
#include <type_traits>
#include <mutex>
#include <thread>

#include "mpc.hpp"

namespace cosnarks {

// groth16.rs:60-100: snarkjs roots. q = smallest quadratic non-residue, z = q^TRACE, repeated squaring, reversed.
template <class Fr>
inline void groth16_roots_of_unity(size_t pow, Fr& group_gen, Fr& coset_shift) {
  constexpr int S = Fr::Params::TWO_ADICITY;
  // (p-1)/2 and TRACE as limb arrays
  uint32_t pm1[Fr::N], half[Fr::N], trace[Fr::N];
  for (int i = 0; i < Fr::N; ++i) pm1[i] = Fr::Params::MOD[i];
  pm1[0] -= 1;
  auto shr = [&](const uint32_t* in, int s, uint32_t* out) {
    for (int i = 0; i < Fr::N; ++i) {
      const int wi = i + s / 32;
      const uint64_t lo = wi < Fr::N ? in[wi] : 0;
      const uint64_t hi = wi + 1 < Fr::N ? in[wi + 1] : 0;
      out[i] = (uint32_t)((lo | (hi << 32)) >> (s % 32));
    }
  };
  shr(pm1, 1, half);
  shr(pm1, S, trace);
  Fr one = Fr::one();
  Fr minus_one = Fr::neg(one);
  Fr q = one;
  while (!(Fr::pow_limbs(q, half, Fr::N) == minus_one)) q = Fr::add(q, one);  // legendre(q) == QNR
  std::vector<Fr> roots(S + 1);
  roots[0] = Fr::pow_limbs(q, trace, Fr::N);
  for (int i = 1; i <= S; ++i) roots[i] = Fr::sqr(roots[i - 1]);
  std::vector<Fr> rev(roots.rbegin(), roots.rend());
  group_gen = rev[pow];
  coset_shift = ((size_t)S == pow) ? Fr::sqr(q) : rev[pow + 1];
}

// Domains (device twiddle tables) are reused across proofs: building one costs a hipMalloc + a kernel, a proof of the
// same circuit needs the same (curve, size, generator) every time.
struct DomainCache {
  std::mutex mu;
  // (device, curve, log size, generator limbs; all-zero = arkworks' default root): twiddle tables live on one GPU, and two
  // reductions of the same size with different roots (snarkjs' vs arkworks') must not share one
  std::map<std::tuple<int, int, uint32_t, std::array<uint64_t, 4>>, csh_domain_t> doms;
  static DomainCache& get() {
    static DomainCache c;
    return c;
  }
  csh_domain_t lookup(csh_curve_t curve, uint32_t log_n, const uint64_t* gen, int* rc_out) {
    std::lock_guard<std::mutex> g(mu);
    int dev = 0;
    (void)csh_current_device(&dev);
    std::array<uint64_t, 4> gen_limbs{};
    if (gen) memcpy(gen_limbs.data(), gen, 32);
    auto key = std::make_tuple(dev, (int)curve, log_n, gen_limbs);
    auto it = doms.find(key);
    if (it != doms.end()) {
      *rc_out = CSH_OK;
      return it->second;
    }
    csh_domain_t d = nullptr;
    *rc_out = csh_domain_create(curve, log_n, gen, &d);
    if (*rc_out == CSH_OK) doms[key] = d;
    return d;
  }
};

// The "trait path": the mirror driven exactly as rust/co-groth16-hip drives the C ABI from the UNCHANGED reference -- every seam call
// takes and returns host slices (one csh_groth16_witness_map_masks per witness map, h back on the host; to_half_share on the host; five
// concurrent csh_msm calls with host scalars, rayon_join5 of groth16.rs:227-294). Off by default: the mirror's own prove keeps the
// witness and h on the device. cog16_set_trait_path(1) / COG16_TRAIT_PATH=1 switch every prove_inner of the process over; 2 = the same
// with the shim's opt-in seeded Rep3 masks (one Rep3Rand::random_seeds() draw per witness map, both mask vectors generated on the device).
inline std::atomic<int>& trait_path_flag() {
  static std::atomic<int> f{getenv("COG16_TRAIT_PATH") ? atoi(getenv("COG16_TRAIT_PATH")) : 0};
  return f;
}

// ---- R1CSToQAP: CircomReduction::witness_map_from_matrices (reduction.rs:77-193) ------------------------------
struct CircomReduction {
  static constexpr bool HAS_DEVICE_MAP = true;
  template <class P>
  static bool device_map_available(const ConstraintMatrices<P>& m) { return m.a_dev && m.b_dev && !getenv("COG16_HOST_EVAL"); }

  // witness_map_from_matrices (reduction.rs:77-193) with the witness shares already uploaded and h left on the device:
  // the h_query MSM (groth16.rs:286-292) consumes it in place. Same draws from the party's randomness as the host-facing
  // variant below.
  template <class P, class T>
  static DeviceScalars witness_map_device(typename T::State& state, const ConstraintMatrices<P>& matrices,
                                          const std::vector<typename P::Fr>& public_inputs, const DeviceScalars& witness_dev) {
    using Fr = typename P::Fr;
    const size_t num_constraints = matrices.num_constraints;
    const size_t num_inputs = matrices.num_instance_variables;
    size_t domain_size = 1, power = 0;
    while (domain_size < num_constraints + num_inputs) {
      domain_size <<= 1;
      ++power;
    }
    if (power > (size_t)Fr::Params::TWO_ADICITY) throw Error("Polynomial Degree too large");  // :87-89
    Span span_all("witness map from matrices (device resident)");
    Fr group_gen, coset_shift;
    groth16_roots_of_unity<Fr>(power, group_gen, coset_shift);                                  // :92
    int rc = CSH_OK;
    csh_domain_t domain = DomainCache::get().lookup(P::ID, (uint32_t)power, (const uint64_t*)&group_gen, &rc);  // :93
    if (rc == CSH_ERR_DOMAIN) throw Error("Polynomial Degree too large");
    check(rc, "csh_domain_create");
    DeviceScalars h(domain_size);
    if constexpr (T::DEVICE_MASKS) {
      auto run = state.rand.take_device_run(2 * domain_size);
      rc = csh_groth16_witness_map_dev(domain, (const uint64_t*)&coset_shift, T::PROTOCOL, state.id, matrices.a_dev, matrices.b_dev, num_constraints,
                                       (const uint64_t*)public_inputs.data(), num_inputs, (const uint64_t*)witness_dev.dev, witness_dev.n, run.seed1, run.off1,
                                       run.seed2, run.off2, (uint64_t*)h.dev, nullptr);
    } else {
      rc = csh_groth16_witness_map_dev(domain, (const uint64_t*)&coset_shift, T::PROTOCOL, state.id, matrices.a_dev, matrices.b_dev, num_constraints,
                                       (const uint64_t*)public_inputs.data(), num_inputs, (const uint64_t*)witness_dev.dev, witness_dev.n, nullptr, 0, nullptr, 0,
                                       (uint64_t*)h.dev, nullptr);
    }
    check(rc, "csh_groth16_witness_map_dev");
    check(csh_sync(nullptr), "csh_sync");  // the MSM threads run on other streams
    return h;
  }

  // What rust/co-groth16-hip/src/hip_reduction.rs::HipCircomReduction does: host slices in, ONE library call, h on the host. The two
  // mask vectors are drawn through the driver's public surface in the reference's order (T::masks = masking_field_elements_vec,
  // rngs.rs:137-156 -- what T::local_mul_vec of two zero vectors returns), so the party's generators advance as in the reference.
  template <class P, class T>
  static UninitBuf<typename T::ArithmeticHalfShare> witness_map_trait_path(typename T::State& state, const ConstraintMatrices<P>& matrices,
                                                                          const std::vector<typename P::Fr>& public_inputs,
                                                                          const std::vector<typename T::ArithmeticShare>& private_witness) {
    using Fr = typename P::Fr;
    const size_t num_constraints = matrices.num_constraints;
    const size_t num_inputs = matrices.num_instance_variables;
    size_t domain_size = 1, power = 0;
    while (domain_size < num_constraints + num_inputs) {
      domain_size <<= 1;
      ++power;
    }
    if (power > (size_t)Fr::Params::TWO_ADICITY) throw Error("Polynomial Degree too large");  // :87-89
    if (!matrices.a_dev || !matrices.b_dev) throw Error("trait path: constraint matrices are not on the device");
    Span span_all("witness map from matrices (trait path: one call, host slices)");
    Fr group_gen, coset_shift;
    groth16_roots_of_unity<Fr>(power, group_gen, coset_shift);                                  // :92
    int rc = CSH_OK;
    csh_domain_t domain = DomainCache::get().lookup(P::ID, (uint32_t)power, (const uint64_t*)&group_gen, &rc);  // :93
    if (rc == CSH_ERR_DOMAIN) throw Error("Polynomial Degree too large");
    check(rc, "csh_domain_create");
    UninitBuf<typename T::ArithmeticHalfShare> h(domain_size);
    const auto t_mask0 = std::chrono::steady_clock::now();
    if constexpr (T::DEVICE_MASKS) {
      if (trait_path_flag().load(std::memory_order_relaxed) == 2) {
        // Opt-in "all parties on GPUs" mode of the shim (hip_reduction.rs, COSNARKS_HIP_SEEDED_MASKS): instead of two host mask vectors
        // the party draws ONE pair of fresh correlated seeds through the public Rep3Rand::random_seeds() (rngs.rs:233) and the device
        // generates both mask vectors from them (ChaCha12, chunks [0, n) = "c", [n, 2n) = "ab"). Every party consumes the same 2 x 128
        // keystream bytes, so the pairwise correlation survives and the three masks still cancel; the mask VALUES differ from what a
        // reference CPU party would draw, so all three parties of a session must run this mode (not wire-compatible with a CPU party).
        uint8_t s1[32], s2[32];
        state.rand.random_seeds(s1, s2);
        last_prove_times().mask_ms = ms_since(t_mask0);
        rc = csh_groth16_witness_map(domain, (const uint64_t*)&coset_shift, T::PROTOCOL, state.id, matrices.a_dev, matrices.b_dev, num_constraints,
                                     (const uint64_t*)public_inputs.data(), num_inputs, (const uint64_t*)private_witness.data(), private_witness.size(),
                                     s1, 0, s2, 0, (uint64_t*)h.data());
        check(rc, "csh_groth16_witness_map (seeded)");
        return h;
      }
    }
    const UninitBuf<Fr> mask_c = T::masks(state, domain_size);   // "c: local_mul_vec" (:160)
    const UninitBuf<Fr> mask_ab = T::masks(state, domain_size);  // "ab" (:182)
    last_prove_times().mask_ms = ms_since(t_mask0);
    const auto t_call0 = std::chrono::steady_clock::now();
    rc = csh_groth16_witness_map_masks(domain, (const uint64_t*)&coset_shift, T::PROTOCOL, state.id, matrices.a_dev, matrices.b_dev, num_constraints,
                                       (const uint64_t*)public_inputs.data(), num_inputs, (const uint64_t*)private_witness.data(), private_witness.size(),
                                       mask_c.empty() ? nullptr : (const uint64_t*)mask_c.data(), mask_ab.empty() ? nullptr : (const uint64_t*)mask_ab.data(),
                                       (uint64_t*)h.data());
    check(rc, "csh_groth16_witness_map_masks");
    if (getenv("COG16_TRACE_WM")) fprintf(stderr, "[wm trait] draw %.3f ms, call %.3f ms\n", last_prove_times().mask_ms, ms_since(t_call0));
    return h;
  }

  template <class P, class T>
  static std::vector<typename T::ArithmeticHalfShare> witness_map_from_matrices(typename T::State& state, const ConstraintMatrices<P>& matrices,
                                                                                 const std::vector<typename P::Fr>& public_inputs,
                                                                                 const std::vector<typename T::ArithmeticShare>& private_witness) {
    using Fr = typename P::Fr;
    using Share = typename T::ArithmeticShare;
    const size_t num_constraints = matrices.num_constraints;
    const size_t num_inputs = matrices.num_instance_variables;
    size_t domain_size = 1, power = 0;
    while (domain_size < num_constraints + num_inputs) {
      domain_size <<= 1;
      ++power;
    }
    if (power > (size_t)Fr::Params::TWO_ADICITY) throw Error("Polynomial Degree too large");  // :87-89
    Span span_all("witness map from matrices");
    Fr group_gen, coset_shift;
    {
      Span sp("root of unity");
      groth16_roots_of_unity<Fr>(power, group_gen, coset_shift);                               // :92
    }
    csh_domain_t domain = nullptr;
    Span* sp_dom = new Span("domain create (twiddles)");
    int rc = CSH_OK;
    domain = DomainCache::get().lookup(P::ID, (uint32_t)power, (const uint64_t*)&group_gen, &rc);  // :93 Domain::with_group_gen (snarkjs root)
    if (rc == CSH_ERR_DOMAIN) throw Error("Polynomial Degree too large");
    delete sp_dom;
    check(rc, "csh_domain_create");
    const int id = state.id;
    if (matrices.a_dev && matrices.b_dev && !getenv("COG16_HOST_EVAL")) {
      // "next" row f3: rows evaluated on the device, only the witness shares cross PCIe (reduction.rs:99-192 in one call)
      Span sp_dev("evaluate constraints + a/b/c pipeline (device)");
      std::vector<Fr> hd(domain_size);
      if constexpr (T::DEVICE_MASKS) {
        auto run = state.rand.take_device_run(2 * domain_size);
        rc = csh_groth16_witness_map(domain, (const uint64_t*)&coset_shift, T::PROTOCOL, id, matrices.a_dev, matrices.b_dev, num_constraints,
                                     (const uint64_t*)public_inputs.data(), num_inputs, (const uint64_t*)private_witness.data(),
                                     private_witness.size(), run.seed1, run.off1, run.seed2, run.off2, (uint64_t*)hd.data());
      } else {
        rc = csh_groth16_witness_map(domain, (const uint64_t*)&coset_shift, T::PROTOCOL, id, matrices.a_dev, matrices.b_dev, num_constraints,
                                     (const uint64_t*)public_inputs.data(), num_inputs, (const uint64_t*)private_witness.data(),
                                     private_witness.size(), nullptr, 0, nullptr, 0, (uint64_t*)hd.data());
      }
      check(rc, "csh_groth16_witness_map");
      return hd;
    }
    Span* sp_eval = new Span("evaluate constraints");

    // :99-130 evaluate constraints (sparse rows on the host; "next" row f3 moves this to the device)
    auto evaluate = [&](const std::vector<std::vector<std::pair<Fr, size_t>>>& m) {  // par_iter().with_min_len(256), :196-210
      std::vector<Share> res(domain_size, Share{});
      parallel_for(m.size(), 256, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) res[i] = T::evaluate_constraint(id, m[i], public_inputs, private_witness);
      });
      return res;
    };
    std::vector<Share> a = evaluate(matrices.a);
    std::vector<Share> promoted = T::promote_to_trivial_shares(id, public_inputs);
    for (size_t i = 0; i < num_inputs; ++i) a[num_constraints + i] = promoted[i];              // :111-113
    std::vector<Share> b = evaluate(matrices.b);
    delete sp_eval;

    // :135-192 on the device in one call: 6 NTTs, 2 local_mul_vec, 3 coset-table multiplications, 1 subtraction.
    // The two mask vectors are drawn in the reference's order: "c: local_mul_vec" (:160) then "ab" (:182).
    std::vector<Fr> h(domain_size);
    Span sp_h("a/b/c: ifft, distribute powers, fft, local_mul_vec, sub (device)");
    if constexpr (T::DEVICE_MASKS) {
      // both mask vectors come straight from the party's ChaCha12 keys on the device; the host generators only skip ahead
      auto run = state.rand.take_device_run(2 * domain_size);
      rc = csh_groth16_h_rep3_seeded(domain, (const uint64_t*)&coset_shift, (uint64_t*)a.data(), (uint64_t*)b.data(), run.seed1, run.off1,
                                     run.seed2, run.off2, (uint64_t*)h.data());
    } else {
      const UninitBuf<Fr> mask_c = T::masks(state, domain_size);
      const UninitBuf<Fr> mask_ab = T::masks(state, domain_size);
      rc = csh_groth16_h(domain, (const uint64_t*)&coset_shift, T::PROTOCOL, (uint64_t*)a.data(), (uint64_t*)b.data(),
                         mask_c.empty() ? nullptr : (const uint64_t*)mask_c.data(), mask_ab.empty() ? nullptr : (const uint64_t*)mask_ab.data(),
                         (uint64_t*)h.data());
    }
    check(rc, "csh_groth16_h");
    return h;
  }
};

// ---- R1CSToQAP: LibSnarkReduction::witness_map_from_matrices (reduction.rs:237-342) ---------------------------------
// The arkworks / libsnark QAP witness map: Domain::new (arkworks' 2-adic root), coset = F::GENERATOR, H = (AB - C)/Z
// as natural-order coefficients. Needs the C matrix (ConstraintMatrices::c); no fixture in the reference tree pins it
// (the Penumbra BLS12-377 keys are absent), so parity rests on the oracle restatement + the polynomial identity.
struct LibSnarkReduction {
  static constexpr bool HAS_DEVICE_MAP = false;
  template <class P, class T>
  static std::vector<typename T::ArithmeticHalfShare> witness_map_from_matrices(typename T::State& state, const ConstraintMatrices<P>& matrices,
                                                                                 const std::vector<typename P::Fr>& public_inputs,
                                                                                 const std::vector<typename T::ArithmeticShare>& private_witness) {
    using Fr = typename P::Fr;
    const size_t num_constraints = matrices.num_constraints;
    const size_t num_inputs = matrices.num_instance_variables;
    size_t domain_size = 1, power = 0;
    while (domain_size < num_constraints + num_inputs) {  // Domain::new(num_constraints + num_inputs) (:249)
      domain_size <<= 1;
      ++power;
    }
    if (power > (size_t)Fr::Params::TWO_ADICITY) throw Error("Polynomial Degree too large");  // :250
    if (!matrices.a_dev || !matrices.b_dev || !matrices.c_dev) throw Error("LibSnarkReduction: constraint matrices a, b, c must be uploaded");
    Span span_all("witness map from matrices (libsnark)");
    csh_domain_t domain = nullptr;
    check(csh_domain_create(P::ID, (uint32_t)power, nullptr, &domain), "csh_domain_create");  // arkworks default root
    const Fr gen = Fr::from_u64(P::FR_GENERATOR);                                              // P::ScalarField::GENERATOR (:255)
    std::vector<Fr> h(domain_size);
    int rc;
    if constexpr (T::DEVICE_MASKS) {
      auto run = state.rand.take_device_run(domain_size);  // the one local_mul_vec (:289)
      rc = csh_groth16_witness_map_libsnark(domain, (const uint64_t*)&gen, T::PROTOCOL, state.id, matrices.a_dev, matrices.b_dev, matrices.c_dev,
                                            num_constraints, (const uint64_t*)public_inputs.data(), num_inputs, (const uint64_t*)private_witness.data(),
                                            private_witness.size(), run.seed1, run.off1, run.seed2, run.off2, (uint64_t*)h.data());
    } else {
      rc = csh_groth16_witness_map_libsnark(domain, (const uint64_t*)&gen, T::PROTOCOL, state.id, matrices.a_dev, matrices.b_dev, matrices.c_dev,
                                            num_constraints, (const uint64_t*)public_inputs.data(), num_inputs, (const uint64_t*)private_witness.data(),
                                            private_witness.size(), nullptr, 0, nullptr, 0, (uint64_t*)h.data());
    }
    csh_domain_free(domain);
    check(rc, "csh_groth16_witness_map_libsnark");
    return h;
  }
};

// ---- CoGroth16<P, T> (groth16.rs:103-338) ------------------------------------------------------------------------
template <class P, class T>
struct CoGroth16 {
  using Fr = typename P::Fr;
  using Fq = typename P::Fq;
  using Fq2 = typename P::Fq2;
  using Share = typename T::ArithmeticShare;
  using Half = typename T::ArithmeticHalfShare;
  using Net = typename T::Net;
  using State = typename T::State;

  // T::scalar_mul (mpc.rs:131-137). The Shamir driver's degree_reduce_point needs C::generator() (network.rs:261-262).
  static Proj<Fq> scalar_mul_dispatch(const Proj<Fq>& a, const Share& b, const Net* net, State& st) {
    if constexpr (std::is_same<T, ShamirGroth16Driver<P>>::value) {
      AffineT<Fq> gen;
      memcpy(&gen, P::g1_generator_words(), sizeof gen);
      return T::template scalar_mul<Fq>(a, b, net, st, gen);
    } else {
      return T::template scalar_mul<Fq>(a, b, net, st);
    }
  }

  template <class Fn>
  static void net_leg(const Net* net0, const Net* net1, Fn fn) {
    try {
      fn();
    } catch (...) {
      if (net0) net0->abort();
      if (net1) net1->abort();
      throw;
    }
  }

  // the part of calculate_coeff (groth16.rs:179-203) around the private-input MSM, given that MSM's result
  template <class F>
  static Proj<F> finish_coeff(int id, Proj<F> initial, const Query<F>& query, const AffineT<F>& vk_param, const std::vector<Fr>& input_assignment,
                              const Proj<F>& priv_acc) {
    const size_t pub_len = input_assignment.size();
    Proj<F> pub_acc = Proj<F>::inf();  // msm_unchecked(&query[1..=pub_len], input_assignment): tiny, on the host (:194)
    for (size_t i = 0; i < pub_len; ++i) pub_acc = point_add(pub_acc, point_mul(into_group(query.host[1 + i]), input_assignment[i]));
    Proj<F> res = initial;
    T::template add_assign_points_public_hs<F>(id, res, into_group(query.host[0]));
    T::template add_assign_points_public_hs<F>(id, res, into_group(vk_param));
    T::template add_assign_points_public_hs<F>(id, res, pub_acc);
    return point_add(res, priv_acc);
  }

  // groth16.rs:179-203
  template <class F>
  static Proj<F> calculate_coeff(int id, Proj<F> initial, const Query<F>& query, const AffineT<F>& vk_param,
                                 const std::vector<Fr>& input_assignment, const DeviceScalars& aux_assignment) {
    const size_t pub_len = input_assignment.size();
    Proj<F> priv_acc = T::template msm_public_points_hs<F>(BasesView{query.dev, 1 + pub_len, query.size() - 1 - pub_len}, aux_assignment);
    Proj<F> pub_acc = Proj<F>::inf();  // msm_unchecked(&query[1..=pub_len], input_assignment): tiny, on the host (:194)
    for (size_t i = 0; i < pub_len; ++i) pub_acc = point_add(pub_acc, point_mul(into_group(query.host[1 + i]), input_assignment[i]));
    Proj<F> res = initial;
    T::template add_assign_points_public_hs<F>(id, res, into_group(query.host[0]));
    T::template add_assign_points_public_hs<F>(id, res, into_group(vk_param));
    T::template add_assign_points_public_hs<F>(id, res, pub_acc);
    return point_add(res, priv_acc);
  }

  // groth16.rs:207-338
  static Proof<P> create_proof_with_assignment(const Net* net0, const Net* net1, State& state0, State& state1, const ProvingKey<P>& pkey,
                                               const Share& r, const Share& s, const std::vector<Half>& h,
                                               const std::vector<Fr>& input_assignment, const std::vector<Half>& aux_assignment) {
    // aux_assignment feeds four MSMs (A, B1, B2, L): upload it once; h feeds one
    Span* sp_up = new Span("upload aux_assignment + h");
    const DeviceScalars aux_dev(aux_assignment.data(), aux_assignment.size());
    const DeviceScalars h_dev(h.data(), h.size());
    delete sp_up;
    return create_proof_device(net0, net1, state0, state1, pkey, r, s, h_dev, input_assignment, aux_dev);
  }

  // groth16.rs:296-337: everything after the five MSM groups
  static Proof<P> finish_proof(const Net* net0, const Net* net1, State& state0, State& state1, const ProvingKey<P>& pkey, const Share& r, const Share& s,
                               const Proj<Fq>& r_g1, const Proj<Fq>& s_g1, const Proj<Fq2>& s_g2, const Proj<Fq>& l_acc, const Proj<Fq>& h_acc) {
    const Proj<Fq> delta_g1 = into_group(pkey.delta_g1);
    const auto t_fin0 = std::chrono::steady_clock::now();
    Span sp_fin("finish - open two points and some adds");

    Half rs = T::local_mul_vec({r}, {s}, state0).back();                                     // :297
    Proj<Fq> r_s_delta_g1 = T::template scalar_mul_public_point_hs<Fq>(delta_g1, rs);        // :298
    Proj<Fq> g_a_opened, r_g1_b;
    {  // mpc_net::join (:305-308): two network legs
      // a leg that throws aborts both networks first, so that the other leg (and the peers) unwind instead of waiting
      Joined n1([&] { net_leg(net0, net1, [&] { r_g1_b = scalar_mul_dispatch(s_g1, r, net1, state1); }); });
      net_leg(net0, net1, [&] { g_a_opened = T::template open_half_point<Fq>(r_g1, net0, state0); });
      n1.join();
    }
    Proj<Fq> g_c = T::template scalar_mul_public_point_hs<Fq>(g_a_opened, T::to_half_share(s));  // :313-314
    g_c = point_add(g_c, r_g1_b);
    g_c = point_add(g_c, point_neg(r_s_delta_g1));
    g_c = point_add(g_c, l_acc);
    g_c = point_add(g_c, h_acc);
    Proj<Fq> g_c_opened;
    Proj<Fq2> g2_b_opened;
    {  // :325-328
      Joined n1([&] { net_leg(net0, net1, [&] { g2_b_opened = T::template open_half_point<Fq2>(s_g2, net1, state1); }); });
      net_leg(net0, net1, [&] { g_c_opened = T::template open_half_point<Fq>(g_c, net0, state0); });
      n1.join();
    }
    Proof<P> proof{into_affine(g_a_opened), into_affine(g_c_opened), into_affine(g2_b_opened)};
    last_prove_times().finish_ms = ms_since(t_fin0);
    return proof;
  }

  // create_proof_with_assignment as the Rust drivers run it behind the reference's rayon_join5 (groth16.rs:227-294): five host threads,
  // each calling the driver's msm_public_points_hs with HOST scalars (one synchronous csh_msm per query: scalars up on the thread's
  // own lane stream, so one call's upload, sort and tail overlap another's accumulation); h is consumed from the host buffer the
  // witness map returned. No shared digit sort, no device-resident operands: nothing the unchanged reference could not do.
  struct MsmGroups {
    Proj<Fq> r_g1, s_g1, l_acc, h_acc;
    Proj<Fq2> s_g2;
  };
  // groth16.rs:227-294 on the trait path: the five closures of rayon_join5, no network involved
  static MsmGroups msm_groups_trait_path(int id, const ProvingKey<P>& pkey, const Share& r, const Share& s, const Half* h, size_t h_len,
                                         const std::vector<Fr>& input_assignment, const Half* aux, size_t n_aux) {
    const Proj<Fq> delta_g1 = into_group(pkey.delta_g1);
    const Proj<Fq2> delta_g2 = into_group(pkey.delta_g2);
    std::vector<Fr> inputs(input_assignment.begin() + 1, input_assignment.end());  // &input_assignment[1..]
    const size_t pub_len = inputs.size();
    MsmGroups g;
    const auto t_msm0 = std::chrono::steady_clock::now();
    int cur_dev = 0;
    (void)csh_current_device(&cur_dev);
    auto bind = [cur_dev] { check(csh_init(cur_dev), "csh_init"); };
    {
      Span sp_msm("5 msm groups, host scalars (trait path)");
      Joined t1([&] { bind(); g.r_g1 = finish_coeff<Fq>(id, T::template scalar_mul_public_point_hs<Fq>(delta_g1, T::to_half_share(r)), pkey.a_query, pkey.alpha_g1, inputs,
                                                       msm_device<Fq>(BasesView{pkey.a_query.dev, 1 + pub_len, pkey.a_query.size() - 1 - pub_len}, aux, n_aux)); });
      Joined t2([&] { bind(); g.s_g1 = finish_coeff<Fq>(id, T::template scalar_mul_public_point_hs<Fq>(delta_g1, T::to_half_share(s)), pkey.b_g1_query, pkey.beta_g1, inputs,
                                                       msm_device<Fq>(BasesView{pkey.b_g1_query.dev, 1 + pub_len, pkey.b_g1_query.size() - 1 - pub_len}, aux, n_aux)); });
      Joined t3([&] { bind(); g.s_g2 = finish_coeff<Fq2>(id, T::template scalar_mul_public_point_hs<Fq2>(delta_g2, T::to_half_share(s)), pkey.b_g2_query, pkey.beta_g2, inputs,
                                                        msm_device<Fq2>(BasesView{pkey.b_g2_query.dev, 1 + pub_len, pkey.b_g2_query.size() - 1 - pub_len}, aux, n_aux)); });
      Joined t4([&] { bind(); g.l_acc = msm_device<Fq>(BasesView{pkey.l_query.dev, pkey.l_query.lead, pkey.l_query.size()}, aux, n_aux); });
      Joined t5([&] { bind(); g.h_acc = msm_device<Fq>(BasesView{pkey.h_query.dev, 0, pkey.h_query.size()}, h, h_len); });
      t1.join(); t2.join(); t3.join(); t4.join(); t5.join();  // the first failure is rethrown; ~Joined reaps the rest
    }
    last_prove_times().msm_ms = ms_since(t_msm0);
    return g;
  }
  static Proof<P> create_proof_trait_path(const Net* net0, const Net* net1, State& state0, State& state1, const ProvingKey<P>& pkey, const Share& r,
                                          const Share& s, const Half* h, size_t h_len, const std::vector<Fr>& input_assignment, const Half* aux,
                                          size_t n_aux) {
    const MsmGroups g = msm_groups_trait_path(state0.id, pkey, r, s, h, h_len, input_assignment, aux, n_aux);
    return finish_proof(net0, net1, state0, state1, pkey, r, s, g.r_g1, g.s_g1, g.s_g2, g.l_acc, g.h_acc);
  }

  // the same with h and aux_assignment already resident on the device
  static Proof<P> create_proof_device(const Net* net0, const Net* net1, State& state0, State& state1, const ProvingKey<P>& pkey, const Share& r,
                                      const Share& s, const DeviceScalars& h_dev, const std::vector<Fr>& input_assignment,
                                      const DeviceScalars& aux_dev) {
    const Proj<Fq> delta_g1 = into_group(pkey.delta_g1);
    const Proj<Fq2> delta_g2 = into_group(pkey.delta_g2);
    const int id = state0.id;
    std::vector<Fr> inputs(input_assignment.begin() + 1, input_assignment.end());  // &input_assignment[1..]
    Proj<Fq> r_g1, s_g1, l_acc, h_acc;
    Proj<Fq2> s_g2;
    // rayon_join5 (:227-294): five independent MSM groups. The four that consume aux_assignment (A, B/G1, B/G2, L) share
    // one digit decomposition + bucket sort on the device (csh_msm_multi_dev); h_query runs from a second host thread.
    Span* sp_msm = new Span("5 msm groups (compute A, B/G1, B/G2, msm l_query, msm h_query)");
    const auto t_msm0 = std::chrono::steady_clock::now();
    int cur_dev = 0;
    (void)csh_current_device(&cur_dev);
    // workers carry their exception back to join() (a plain std::thread would std::terminate on a HIP OOM in an MSM)
    const size_t pub_len = inputs.size();
    const size_t n_aux = aux_dev.n;
    const bool same_len = pkey.a_query.size() == 1 + pub_len + n_aux && pkey.b_g1_query.size() == 1 + pub_len + n_aux &&
                          pkey.b_g2_query.size() == 1 + pub_len + n_aux && pkey.l_query.size() == n_aux && n_aux > 0 &&
                          !getenv("COG16_SEPARATE_MSMS");
    // (a placement only applies to the shared-sort path: the fallback below runs every MSM on this GPU)
    const bool h_at_home = !same_len || pkey.slot_has(0, ProvingKey<P>::Q_H);
    h_acc = Proj<Fq>::inf();
    Joined t5([&] {
      if (!h_at_home) return;
      check(csh_init(cur_dev), "csh_init");  // a new host thread is not bound to the parent's GPU
      size_t lo = 0, hi = pkey.h_query.size();
      if (same_len) pkey.slot_range(0, pkey.h_query.size() < h_dev.n ? pkey.h_query.size() : h_dev.n, &lo, &hi);
      if (lo == 0 && hi == pkey.h_query.size()) {
        h_acc = T::template msm_public_points_hs<Fq>(BasesView{pkey.h_query.dev, 0, pkey.h_query.size()}, h_dev);
      } else if (hi > lo) {  // BY_RANGE: this GPU's part of h, read in place (every driver's msm_public_points_hs is this MSM)
        h_acc = msm_device_resident_ptr<Fq>(BasesView{pkey.h_query.dev, lo, hi - lo}, static_cast<const char*>(h_dev.dev) + lo * sizeof(Half), hi - lo);
      }
    });
    // calculate_coeff's `initial` points (delta r, delta s in G1, delta_2 s in G2: groth16.rs:230-262) depend on r, s and the key only: a host
    // thread computes them while the device runs the MSMs (they were computed after the MSMs, on the thread that had just waited for them)
    Proj<Fq> init_r = Proj<Fq>::inf(), init_s = Proj<Fq>::inf();
    Proj<Fq2> init_s2 = Proj<Fq2>::inf();
    std::unique_ptr<Joined> t_init;
    if (same_len) {
      t_init.reset(new Joined([&] {
        init_r = T::template scalar_mul_public_point_hs<Fq>(delta_g1, T::to_half_share(r));
        init_s = T::template scalar_mul_public_point_hs<Fq>(delta_g1, T::to_half_share(s));
        init_s2 = T::template scalar_mul_public_point_hs<Fq2>(delta_g2, T::to_half_share(s));
      }));
    }
    if (same_len) {
      using PK = ProvingKey<P>;
      // One slot = the work placed on one GPU (ProvingKey::place; without a placement everything is slot 0 = this GPU): whole
      // queries (BY_QUERY) or the slot's contiguous range of every query (BY_RANGE). The aux queries of a slot share one digit
      // decomposition + bucket sort (csh_msm_multi_dev), G2 first: its host fold (Horner over Fp2 windows, ~3x a G1 fold) then runs
      // under the G1 bucket stages that follow. Results are per (slot, query) and summed on the host.
      const size_t nslots = pkey.slots();
      struct SlotOut {
        csh::Jac<Fq> a, b1, l;
        csh::Jac<Fq2> b2;
        Proj<Fq> h;
      };
      std::vector<SlotOut> outs_by_slot(nslots);
      for (auto& o : outs_by_slot) {
        o.a = o.b1 = o.l = csh::Jac<Fq>::inf();
        o.b2 = csh::Jac<Fq2>::inf();
        o.h = Proj<Fq>::inf();
      }
      auto run_aux_queries = [&](size_t slot, const void* aux_scalars_of_range) {
        SlotOut& o = outs_by_slot[slot];
        const int qs[4] = {PK::Q_B2, PK::Q_A, PK::Q_B1, PK::Q_L};
        void* const res[4] = {&o.b2, &o.a, &o.b1, &o.l};
        size_t lo = 0, hi = 0;
        pkey.slot_range(slot, n_aux, &lo, &hi);
        csh_bases_t hs[4];
        size_t offs[4];
        void* outs[4];
        size_t k = 0;
        for (int i = 0; i < 4; ++i) {
          if (!pkey.slot_has(slot, qs[i])) continue;
          hs[k] = pkey.handle_for(slot, qs[i]);
          offs[k] = pkey.offset_in(slot, qs[i], (qs[i] == PK::Q_L ? 0 : 1 + pub_len) + lo, hi - lo);
          outs[k] = res[i];
          ++k;
        }
        if (k && hi > lo) check(csh_msm_multi_dev(hs, offs, k, hi - lo, reinterpret_cast<const uint64_t*>(aux_scalars_of_range), 1, outs, nullptr), "csh_msm_multi_dev");
      };
      // the slots on other GPUs: one host thread each, bound to its GPU; its part of the scalars arrives by peer copy (32 bytes per
      // entry over xGMI), the results are 3 curve points per query on the host
      std::vector<std::unique_ptr<Joined>> remote;
      for (size_t sl = 1; sl < nslots; ++sl) {
        bool any_aux = false;
        for (int q : {PK::Q_B2, PK::Q_A, PK::Q_B1, PK::Q_L}) any_aux |= pkey.slot_has(sl, q);
        const bool has_h = pkey.slot_has(sl, PK::Q_H);
        if (!any_aux && !has_h) continue;
        const int dev = pkey.placement.devices[sl];
        remote.emplace_back(new Joined([&, sl, dev, any_aux, has_h] {
          check(csh_init(dev), "csh_init");
          Span sp("msm group on another GPU (peer copy of the scalars + its queries)");
          size_t lo = 0, hi = 0;
          if (any_aux) {
            pkey.slot_range(sl, n_aux, &lo, &hi);
            if (hi > lo) {
              const DeviceScalars aux_l(hi - lo);
              check(csh_memcpy_peer(aux_l.dev, dev, static_cast<const char*>(aux_dev.dev) + lo * sizeof(Half), aux_dev.device, (hi - lo) * sizeof(Half), nullptr),
                    "csh_memcpy_peer");
              run_aux_queries(sl, aux_l.dev);
            }
          }
          if (has_h) {
            pkey.slot_range(sl, pkey.h_query.size() < h_dev.n ? pkey.h_query.size() : h_dev.n, &lo, &hi);
            if (hi > lo) {
              const DeviceScalars h_l(hi - lo);
              check(csh_memcpy_peer(h_l.dev, dev, static_cast<const char*>(h_dev.dev) + lo * sizeof(Half), h_dev.device, (hi - lo) * sizeof(Half), nullptr),
                    "csh_memcpy_peer");
              outs_by_slot[sl].h = T::template msm_public_points_hs<Fq>(BasesView{pkey.handle_for(sl, PK::Q_H), pkey.offset_in(sl, PK::Q_H, lo, hi - lo), hi - lo}, h_l);
            }
          }
        }));
      }
      try {
        size_t lo = 0, hi = 0;
        pkey.slot_range(0, n_aux, &lo, &hi);
        run_aux_queries(0, static_cast<const char*>(aux_dev.dev) + lo * sizeof(Half));
      } catch (...) {
        t5.join_quiet();
        t_init->join_quiet();
        for (auto& r : remote) r->join_quiet();
        throw;
      }
      for (auto& r : remote) r->join();
      t5.join();
      t_init->join();
      auto to_proj = [](const auto& j) {
        using F = typename std::decay<decltype(j.x)>::type;
        return j.is_inf() ? Proj<F>::inf() : Proj<F>::from_affine(AffineT<F>{j.x, j.y});
      };
      Proj<Fq> ja = Proj<Fq>::inf(), jb1 = Proj<Fq>::inf(), jl = Proj<Fq>::inf();
      Proj<Fq2> jb2 = Proj<Fq2>::inf();
      for (size_t sl = 0; sl < nslots; ++sl) {  // BY_QUERY: one slot contributes per query; BY_RANGE: N partial sums each
        ja = point_add(ja, to_proj(outs_by_slot[sl].a));
        jb1 = point_add(jb1, to_proj(outs_by_slot[sl].b1));
        jl = point_add(jl, to_proj(outs_by_slot[sl].l));
        jb2 = point_add(jb2, to_proj(outs_by_slot[sl].b2));
        if (sl) h_acc = point_add(h_acc, outs_by_slot[sl].h);
      }
      r_g1 = finish_coeff<Fq>(id, init_r, pkey.a_query, pkey.alpha_g1, inputs, ja);
      s_g1 = finish_coeff<Fq>(id, init_s, pkey.b_g1_query, pkey.beta_g1, inputs, jb1);
      s_g2 = finish_coeff<Fq2>(id, init_s2, pkey.b_g2_query, pkey.beta_g2, inputs, jb2);
      l_acc = jl;
    } else {
      // (fallback: separate MSMs from separate host threads, each bound to the parent's GPU)
      auto bind = [cur_dev] { check(csh_init(cur_dev), "csh_init"); };
      Joined t1([&] { bind(); r_g1 = calculate_coeff<Fq>(id, T::template scalar_mul_public_point_hs<Fq>(delta_g1, T::to_half_share(r)), pkey.a_query, pkey.alpha_g1, inputs, aux_dev); });
      Joined t2([&] { bind(); s_g1 = calculate_coeff<Fq>(id, T::template scalar_mul_public_point_hs<Fq>(delta_g1, T::to_half_share(s)), pkey.b_g1_query, pkey.beta_g1, inputs, aux_dev); });
      Joined t3([&] { bind(); s_g2 = calculate_coeff<Fq2>(id, T::template scalar_mul_public_point_hs<Fq2>(delta_g2, T::to_half_share(s)), pkey.b_g2_query, pkey.beta_g2, inputs, aux_dev); });
      Joined t4([&] { bind(); l_acc = T::template msm_public_points_hs<Fq>(BasesView{pkey.l_query.dev, pkey.l_query.lead, pkey.l_query.size()}, aux_dev); });
      t1.join(); t2.join(); t3.join(); t4.join(); t5.join();  // the first failure is rethrown; ~Joined reaps the rest
    }
    delete sp_msm;
    last_prove_times().msm_ms = ms_since(t_msm0);
    return finish_proof(net0, net1, state0, state1, pkey, r, s, r_g1, s_g1, s_g2, l_acc, h_acc);
  }

  // prove_inner for a witness that is already a device vector of shares (e.g. ingested straight from a wtns image,
  // zkey.hpp parse_wtns_to_device): same sequence as the device-resident branch of prove_inner below
  template <class R>
  static Proof<P> prove_inner_device_witness(const Net* net0, const Net* net1, State& state0, State& state1, const ProvingKey<P>& pkey,
                                             const ConstraintMatrices<P>& matrices, const std::vector<Fr>& public_inputs,
                                             const DeviceScalars& wit_dev, const Share* r_in, const Share* s_in, std::vector<Half>* h_out = nullptr) {
    static_assert(std::is_same<Share, Half>::value, "device-ingested witnesses are single-component share vectors");
    if (public_inputs.size() != matrices.num_instance_variables)
      throw Error("amount of public inputs does not match with provided constraint system! Expected " +
                  std::to_string(matrices.num_instance_variables) + ", but got " + std::to_string(public_inputs.size()));
    if (wit_dev.n != matrices.num_witness_variables)
      throw Error("amount of private witness variables does not match with provided constraint system! Expected " +
                  std::to_string(matrices.num_witness_variables) + ", but got " + std::to_string(wit_dev.n));
    if (!R::template device_map_available<P>(matrices)) throw Error("constraint matrices are not on the device");
    const DeviceScalars h_dev = R::template witness_map_device<P, T>(state0, matrices, public_inputs, wit_dev);
    Share r = T::rand(net0, state0), s = T::rand(net0, state0);
    if (r_in) r = *r_in;
    if (s_in) s = *s_in;
    if (h_out) {
      h_out->resize(h_dev.n);
      check(csh_memcpy_d2h(h_out->data(), h_dev.dev, h_dev.n * sizeof(Half)), "csh_memcpy_d2h");
    }
    return create_proof_device(net0, net1, state0, state1, pkey, r, s, h_dev, public_inputs, wit_dev);
  }

  // groth16.rs:125-177 with r, s optionally supplied (the reference always draws them with T::rand)
  template <class R>
  static Proof<P> prove_inner(const Net* net0, const Net* net1, State& state0, State& state1, const ProvingKey<P>& pkey,
                              const ConstraintMatrices<P>& matrices, const SharedWitness<P, Share>& w, const Share* r_in, const Share* s_in,
                              std::vector<Half>* h_out = nullptr) {
    if (w.public_inputs.size() != matrices.num_instance_variables)
      throw Error("amount of public inputs does not match with provided constraint system! Expected " +
                  std::to_string(matrices.num_instance_variables) + ", but got " + std::to_string(w.public_inputs.size()));
    if (w.witness.size() != matrices.num_witness_variables)
      throw Error("amount of private witness variables does not match with provided constraint system! Expected " +
                  std::to_string(matrices.num_witness_variables) + ", but got " + std::to_string(w.witness.size()));
    if constexpr (R::HAS_DEVICE_MAP) {
      if (trait_path_flag().load(std::memory_order_relaxed) && R::template device_map_available<P>(matrices)) {
        // the Rust shim's sequence behind the unchanged reference (groth16.rs:125-177): host slices at every seam
        const auto t_wit0 = std::chrono::steady_clock::now();
        UninitBuf<Half> h = R::template witness_map_trait_path<P, T>(state0, matrices, w.public_inputs, w.witness);
        last_prove_times().witness_ms = ms_since(t_wit0);
        Share r = T::rand(net0, state0), s = T::rand(net0, state0);
        if (r_in) r = *r_in;
        if (s_in) s = *s_in;
        if (h_out) h_out->assign(h.data(), h.data() + h.size());
        // private_witness.into_iter().map(T::to_half_share).collect() (groth16.rs:159-163): Rust collects in place over the witness
        // allocation; the mirror's witness is const, so single-component shares are read where they lie and Rep3 shares are
        // compacted into a buffer that is not zero-filled first
        if constexpr (std::is_same<Share, Half>::value) {
          return create_proof_trait_path(net0, net1, state0, state1, pkey, r, s, h.data(), h.size(), w.public_inputs, w.witness.data(), w.witness.size());
        } else {
          const auto t_half0 = std::chrono::steady_clock::now();
          UninitBuf<Half> half(w.witness.size());
          parallel_for(half.size(), 1 << 16, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) half.data()[i] = T::to_half_share(w.witness[i]);
          });
          last_prove_times().half_ms = ms_since(t_half0);
          return create_proof_trait_path(net0, net1, state0, state1, pkey, r, s, h.data(), h.size(), w.public_inputs, half.data(), half.size());
        }
      }
      if (R::template device_map_available<P>(matrices)) {
        // device-resident proof: the witness shares cross PCIe once, h never leaves the device unless asked for
        constexpr size_t COMPS = sizeof(Share) / sizeof(Fr);
        const auto t_wit0 = std::chrono::steady_clock::now();
        Span* sp_up = new Span("upload witness shares");
        const DeviceScalars wit_dev(w.witness.data(), w.witness.size(), COMPS);
        delete sp_up;
        const DeviceScalars h_dev = R::template witness_map_device<P, T>(state0, matrices, w.public_inputs, wit_dev);
        last_prove_times().witness_ms = ms_since(t_wit0);
        Share r = T::rand(net0, state0), s = T::rand(net0, state0);
        if (r_in) r = *r_in;
        if (s_in) s = *s_in;
        if (h_out) {
          h_out->resize(h_dev.n);
          check(csh_memcpy_d2h(h_out->data(), h_dev.dev, h_dev.n * sizeof(Half)), "csh_memcpy_d2h");
        }
        if constexpr (std::is_same<Share, Half>::value) {
          return create_proof_device(net0, net1, state0, state1, pkey, r, s, h_dev, w.public_inputs, wit_dev);
        } else {
          // to_half_share over the witness (groth16.rs:159-163) on the device: a strided copy of the share component
          // the driver's to_half_share keeps (Rep3: `.a`), instead of a host loop + a second 32 MB upload
          static_assert(sizeof(Share) == COMPS * sizeof(Half), "share = COMPS field elements");
          Share probe{};
          reinterpret_cast<Fr*>(&probe)[0] = Fr::one();
          const uint32_t keep = T::to_half_share(probe) == Fr::one() ? 0u : 1u;
          const DeviceScalars aux_dev(w.witness.size());
          check(csh_extract_component_dev((const uint64_t*)wit_dev.dev, (uint32_t)COMPS, keep, w.witness.size(), (uint64_t*)aux_dev.dev, nullptr),
                "csh_extract_component_dev");
          check(csh_sync(nullptr), "csh_sync");
          return create_proof_device(net0, net1, state0, state1, pkey, r, s, h_dev, w.public_inputs, aux_dev);
        }
      }
    }
    std::vector<Half> h = R::template witness_map_from_matrices<P, T>(state0, matrices, w.public_inputs, w.witness);
    Share r = T::rand(net0, state0), s = T::rand(net0, state0);
    if (r_in) r = *r_in;
    if (s_in) s = *s_in;
    std::vector<Half> half(w.witness.size());
    for (size_t i = 0; i < half.size(); ++i) half[i] = T::to_half_share(w.witness[i]);
    if (h_out) *h_out = h;
    return create_proof_with_assignment(net0, net1, state0, state1, pkey, r, s, h, w.public_inputs, half);
  }
};

}  // namespace cosnarks
