// Call sites of the MSM / NTT / share-vector kernels in the reference's other two provers, mirrored
// above the C ABI with the reference's method names and argument meaning:
//   CircomPlonkProver::{local_mul_vec, fft, ifft, msm_public_points_g1}      co-circom/co-plonk/src/mpc.rs:56-166,
//       impls co-plonk/src/mpc/{plain.rs:60-185, rep3.rs:60-175, shamir.rs}, Domains (co-plonk/src/types.rs:70-109)
//   NoirUltraHonkProver::{local_mul_vec, msm_public_points, fft, ifft}        co-noir/co-noir-common/src/mpc/mod.rs:236-379,
//       impls mpc/{plain.rs:271, rep3.rs:258-288, shamir.rs:255}, HonkCurve::fast_msm (honk_curve.rs:35, 81-83, 175-177)
// Only these data-parallel methods are mirrored: the PLONK rounds / sumcheck / relations above them are control logic and
// stay in the Rust host (SURVEY.md 8 "out of scope").
#pragma once
#include "groth16.hpp"

namespace cosnarks {

// ark_poly::Radix2EvaluationDomain<F> as the provers use it: size = a power of two, group_gen either arkworks' default
// or overwritten with the snarkjs root (co-plonk/src/types.rs:92-99). fft / ifft are natural -> natural, the input is
// zero-padded to the domain size (EvaluationDomain::fft semantics), ifft scales by 1/n.
template <class P>
struct EvaluationDomain {
  csh_domain_t dom = nullptr;
  size_t size = 0;
  uint32_t log_size = 0;
  EvaluationDomain() = default;
  EvaluationDomain(const EvaluationDomain&) = delete;
  EvaluationDomain& operator=(const EvaluationDomain&) = delete;
  EvaluationDomain(EvaluationDomain&& o) noexcept : dom(o.dom), size(o.size), log_size(o.log_size) { o.dom = nullptr; }
  EvaluationDomain& operator=(EvaluationDomain&& o) noexcept {
    if (this != &o) {
      if (dom) csh_domain_free(dom);
      dom = o.dom;
      size = o.size;
      log_size = o.log_size;
      o.dom = nullptr;
    }
    return *this;
  }
  ~EvaluationDomain() {
    if (dom) csh_domain_free(dom);
  }
  // Radix2EvaluationDomain::new(n): smallest power of two >= n, arkworks root
  static EvaluationDomain arkworks(size_t n) { return make(n, nullptr); }
  // ... with group_gen := snarkjs roots_of_unity[log2 size] (types.rs:92-99)
  static EvaluationDomain snarkjs(size_t n) {
    size_t s = 1;
    uint32_t lg = 0;
    while (s < n) {
      s <<= 1;
      ++lg;
    }
    typename P::Fr gen, shift;
    groth16_roots_of_unity<typename P::Fr>(lg, gen, shift);
    return make(n, (const uint64_t*)&gen);
  }

 private:
  static EvaluationDomain make(size_t n, const uint64_t* gen) {
    EvaluationDomain d;
    d.size = 1;
    while (d.size < n) {
      d.size <<= 1;
      ++d.log_size;
    }
    const int rc = csh_domain_create(P::ID, d.log_size, gen, &d.dom);
    if (rc == CSH_ERR_DOMAIN) throw Error("PolynomialDegreeTooLarge");
    check(rc, "csh_domain_create");
    return d;
  }
};

// co-plonk/src/types.rs:36-109: the two domains of a PLONK proof (n and 4n, snarkjs roots)
template <class P>
struct PlonkDomains {
  EvaluationDomain<P> domain, extended_domain;
  explicit PlonkDomains(size_t domain_size) {
    if (domain_size == 0 || (domain_size & (domain_size - 1))) throw Error("InvalidDomainSize");
    domain = EvaluationDomain<P>::snarkjs(domain_size);
    extended_domain = EvaluationDomain<P>::snarkjs(domain_size * 4);
  }
};

namespace detail {
template <class P, class Share>
inline std::vector<Share> transform(const std::vector<Share>& data, const EvaluationDomain<P>& d, bool inverse) {
  if (data.size() > d.size) throw Error("fft: input longer than the domain");
  std::vector<Share> v(d.size);  // value-initialised: zero padding
  std::copy(data.begin(), data.end(), v.begin());
  constexpr uint32_t ncomp = sizeof(Share) / sizeof(typename P::Fr);
  check(inverse ? csh_ifft(d.dom, (uint64_t*)v.data(), ncomp) : csh_fft(d.dom, (uint64_t*)v.data(), ncomp), inverse ? "csh_ifft" : "csh_fft");
  return v;
}
// taceo_ark_algebra::msm::msm_unchecked on host slices: the shorter of the two lengths (honk_curve.rs:33-35)
template <class F>
inline Proj<F> msm_unchecked(csh_curve_t curve, csh_group_t group, const std::vector<AffineT<F>>& points, const void* scalars_mont, size_t n) {
  const size_t cnt = n < points.size() ? n : points.size();
  if (cnt == 0) return Proj<F>::inf();
  csh_bases_t h = nullptr;
  check(csh_bases_upload(curve, group, points.data(), cnt, 0, &h), "csh_bases_upload");
  csh::Jac<F> out;
  const int rc = csh_msm(h, 0, cnt, reinterpret_cast<const uint64_t*>(scalars_mont), 1, &out);
  csh_bases_free(h);
  check(rc, "csh_msm");
  if (out.is_inf()) return Proj<F>::inf();
  return Proj<F>::from_affine(AffineT<F>{out.x, out.y});
}
}  // namespace detail

template <class F>
struct Rep3PointShare {  // mpc-core/src/protocols/rep3/pointshare/types.rs:5-11
  Proj<F> a, b;
};

// A curve as the MSM entry points see it: BN254 / BLS12-381 G1 for PLONK and UltraHonk commitments, Grumpkin for the
// HonkCurve impl over the cycle curve.
template <class P>
struct G1Of {
  using Fr = typename P::Fr;
  using Fq = typename P::Fq;
  static constexpr csh_curve_t CURVE = P::ID;
  static constexpr csh_curve_t FIELD_OF = P::ID;
};
struct GrumpkinCurve {  // scalars live in BN254 Fq, coordinates in BN254 Fr
  using Fr = csh::Bn254Fq;
  using Fq = csh::Bn254Fr;
  static constexpr csh_curve_t CURVE = CSH_GRUMPKIN;
};

// HonkCurve::fast_msm (honk_curve.rs:35): msm_unchecked(bases, scalars)
template <class C>
inline Proj<typename C::Fq> fast_msm(const std::vector<AffineT<typename C::Fq>>& bases, const std::vector<typename C::Fr>& scalars) {
  return detail::msm_unchecked<typename C::Fq>(C::CURVE, CSH_G1, bases, scalars.data(), scalars.size());
}

// ---- plain drivers (co-plonk/src/mpc/plain.rs, co-noir-common/src/mpc/plain.rs) -----------------------------------
template <class P>
struct PlainPlonkDriver {
  using Fr = typename P::Fr;
  using Fq = typename P::Fq;
  using ArithmeticShare = Fr;
  using PointShareG1 = Proj<Fq>;
  using State = UnitState;
  static std::vector<Fr> local_mul_vec(const std::vector<Fr>& a, const std::vector<Fr>& b, State& st) {
    return PlainGroth16Driver<P>::local_mul_vec(a, b, st);
  }
  static std::vector<Fr> fft(const std::vector<Fr>& data, const EvaluationDomain<P>& d) { return detail::transform<P, Fr>(data, d, false); }
  static std::vector<Fr> ifft(const std::vector<Fr>& data, const EvaluationDomain<P>& d) { return detail::transform<P, Fr>(data, d, true); }
  static PointShareG1 msm_public_points_g1(const std::vector<AffineT<Fq>>& points, const std::vector<Fr>& scalars) {
    return detail::msm_unchecked<Fq>(P::ID, CSH_G1, points, scalars.data(), scalars.size());  // plain.rs:183
  }
  static PointShareG1 msm_public_points(const std::vector<AffineT<Fq>>& points, const std::vector<Fr>& scalars) {  // co-noir plain.rs:271
    return msm_public_points_g1(points, scalars);
  }
};

// ---- Rep3 drivers (co-plonk/src/mpc/rep3.rs, co-noir-common/src/mpc/rep3.rs) -------------------------------------------
template <class P>
struct Rep3PlonkDriver {
  using Fr = typename P::Fr;
  using Fq = typename P::Fq;
  using ArithmeticShare = Rep3PrimeFieldShare<Fr>;
  using PointShareG1 = Rep3PointShare<Fq>;
  using State = Rep3State;
  // arithmetic::local_mul_vec (rep3/arithmetic.rs:132-146): additive shares out, masked; io_round_mul_vec reshares them
  static std::vector<Fr> local_mul_vec(const std::vector<ArithmeticShare>& a, const std::vector<ArithmeticShare>& b, State& st) {
    return Rep3Groth16Driver<P>::local_mul_vec(a, b, st);
  }
  // DomainCoeff on a share: both components through the same transform (rep3.rs:140-152)
  static std::vector<ArithmeticShare> fft(const std::vector<ArithmeticShare>& data, const EvaluationDomain<P>& d) {
    return detail::transform<P, ArithmeticShare>(data, d, false);
  }
  static std::vector<ArithmeticShare> ifft(const std::vector<ArithmeticShare>& data, const EvaluationDomain<P>& d) {
    return detail::transform<P, ArithmeticShare>(data, d, true);
  }
  // pointshare::msm_public_points (rep3/pointshare.rs:201-222) / the fast_msm pair of co-noir rep3.rs:258-266: split the
  // shares into their a and b vectors, one MSM each
  static PointShareG1 msm_public_points_g1(const std::vector<AffineT<Fq>>& points, const std::vector<ArithmeticShare>& scalars) {
    const size_t cnt = std::min(scalars.size(), points.size());
    PointShareG1 out{Proj<Fq>::inf(), Proj<Fq>::inf()};
    if (cnt == 0) return out;
    csh_bases_t h = nullptr;
    check(csh_bases_upload(P::ID, CSH_G1, points.data(), cnt, 0, &h), "csh_bases_upload");  // uploaded once for both MSMs
    // the shares go up as they lie in memory ({a, b} pairs); the library cuts the two component vectors out on the device
    // (csh_msm_shares) instead of the host unzip + two uploads of the reference's call sites
    csh::Jac<Fq> ja, jb;
    void* outs[2] = {&ja, &jb};
    const int rc = csh_msm_shares(h, 0, cnt, reinterpret_cast<const uint64_t*>(scalars.data()), 2, 1, outs);
    csh_bases_free(h);
    check(rc, "csh_msm_shares");
    out.a = ja.is_inf() ? Proj<Fq>::inf() : Proj<Fq>::from_affine(AffineT<Fq>{ja.x, ja.y});
    out.b = jb.is_inf() ? Proj<Fq>::inf() : Proj<Fq>::from_affine(AffineT<Fq>{jb.x, jb.y});
    return out;
  }
  static PointShareG1 msm_public_points(const std::vector<AffineT<Fq>>& points, const std::vector<ArithmeticShare>& scalars) {
    return msm_public_points_g1(points, scalars);
  }
};

// ---- Shamir drivers (co-plonk/src/mpc/shamir.rs, co-noir-common/src/mpc/shamir.rs) ----------------------------------
template <class P>
struct ShamirPlonkDriver {
  using Fr = typename P::Fr;
  using Fq = typename P::Fq;
  using ArithmeticShare = Fr;  // ShamirPrimeFieldShare is repr(transparent)
  using PointShareG1 = Proj<Fq>;
  template <class State>
  static std::vector<Fr> local_mul_vec(const std::vector<Fr>& a, const std::vector<Fr>& b, State&) {  // shamir/arithmetic.rs:73-79
    std::vector<Fr> out(a.size());
    check(csh_vec_mul(P::ID, (const uint64_t*)a.data(), (const uint64_t*)b.data(), (uint64_t*)out.data(), a.size()), "csh_vec_mul");
    return out;
  }
  static std::vector<Fr> fft(const std::vector<Fr>& data, const EvaluationDomain<P>& d) { return detail::transform<P, Fr>(data, d, false); }
  static std::vector<Fr> ifft(const std::vector<Fr>& data, const EvaluationDomain<P>& d) { return detail::transform<P, Fr>(data, d, true); }
  static PointShareG1 msm_public_points_g1(const std::vector<AffineT<Fq>>& points, const std::vector<Fr>& scalars) {  // shamir/pointshare.rs:207-225
    return detail::msm_unchecked<Fq>(P::ID, CSH_G1, points, scalars.data(), scalars.size());
  }
  static PointShareG1 msm_public_points(const std::vector<AffineT<Fq>>& points, const std::vector<Fr>& scalars) {  // co-noir shamir.rs:255
    return msm_public_points_g1(points, scalars);
  }
};

}  // namespace cosnarks
