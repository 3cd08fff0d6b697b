// C entry points of the host mirror (for the Python tests / bench and as a template for the Rust shim):
//   cog16_prove_plain  == Groth16::<P>::plain_prove::<CircomReduction>            (groth16.rs:484-490)
//   cog16_prove_rep3   == three parties in one process, each running Rep3CoGroth16::prove::<_, CircomReduction>
//                         (groth16.rs:360-379) over LocalNetwork pairs -- the layout of the reference's own
//                         e2e test (tests/tests/circom/e2e_tests/rep3.rs:57-69): asserts the three proofs agree.
// curve: 0 BN254, 1 BLS12-381. r/s: canonical little-endian 4 x u64, or NULL to draw them with T::rand.
#include <atomic>
#include <random>

#include "groth16.hpp"
#include "zkey.hpp"

using namespace cosnarks;

namespace {

thread_local std::string g_err;

template <class P>
typename P::Fr fr_from_canonical(const uint64_t v[4]) {
  typename P::Fr f;
  memcpy(&f, v, 32);
  return f.to_mont();
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
  std::vector<Fr> w = parse_wtns<P>(wtns, wlen);
  SharedWitness<P, Fr> sw;
  sw.public_inputs.assign(w.begin(), w.begin() + m.num_instance_variables);
  sw.witness.assign(w.begin() + m.num_instance_variables, w.end());
  UnitState st0, st1;
  Fr rr, ss;
  if (r) rr = fr_from_canonical<P>(r);
  if (s) ss = fr_from_canonical<P>(s);
  std::vector<Fr> h;
  Proof<P> pr = CoGroth16<P, T>::template prove_inner<CircomReduction>(nullptr, nullptr, st0, st1, pk, m, sw, r ? &rr : nullptr, s ? &ss : nullptr, &h);
  if (h_out) {
    if (h.size() > h_cap) throw Error("h_out too small");
    memcpy(h_out, h.data(), h.size() * 32);
  }
  return write_out(proof_to_json(pr), out, cap);
}

template <class P>
int prove_rep3_t(const uint8_t* zkey, size_t zlen, const uint8_t* wtns, size_t wlen, uint64_t seed, const uint64_t* r, const uint64_t* s,
                 char* out, size_t cap, uint64_t* h_shares_out, size_t h_cap) {
  using T = Rep3Groth16Driver<P>;
  using Fr = typename P::Fr;
  using Share = Rep3PrimeFieldShare<Fr>;
  ProvingKey<P> pk;
  ConstraintMatrices<P> m;
  parse_zkey<P>(zkey, zlen, pk, m);
  std::vector<Fr> w = parse_wtns<P>(wtns, wlen);
  // share_field_elements (mpc-core/src/protocols/rep3.rs:281-292, 375-389) with a seeded RNG
  std::mt19937_64 gen(seed);
  auto rnd = [&] {
    uint8_t b[32];
    for (int i = 0; i < 4; ++i) {
      uint64_t v = gen();
      memcpy(b + 8 * i, &v, 8);
    }
    return from_be_bytes_mod_order<Fr>(b);
  };
  auto share = [&](const Fr& val, Share out3[3]) {
    Fr a = rnd(), b = rnd();
    Fr c = Fr::sub(Fr::sub(val, a), b);
    out3[0] = {a, c};
    out3[1] = {b, a};
    out3[2] = {c, b};
  };
  SharedWitness<P, Share> sw[3];
  const size_t npub = m.num_instance_variables;
  for (int p = 0; p < 3; ++p) sw[p].public_inputs.assign(w.begin(), w.begin() + npub);
  for (size_t i = npub; i < w.size(); ++i) {
    Share t[3];
    share(w[i], t);
    for (int p = 0; p < 3; ++p) sw[p].witness.push_back(t[p]);
  }
  Share r3[3], s3[3];
  if (r) share(fr_from_canonical<P>(r), r3);
  if (s) share(fr_from_canonical<P>(s), s3);
  auto nets0 = LocalNetwork::new_parties(3), nets1 = LocalNetwork::new_parties(3);
  Proof<P> proofs[3];
  std::vector<Fr> hs[3];
  std::string errs[3];
  std::vector<std::thread> th;
  int ndev = 1;
  csh_device_count(&ndev);
  for (int p = 0; p < 3; ++p) {
    th.emplace_back([&, p] {
      try {
        check(csh_init(ndev > 0 ? p % ndev : 0), "csh_init");  // one GPU per party when the node has them (BASELINE config 4)
        uint8_t my_seed[32];
        std::mt19937_64 g2(seed * 1000003ull + 17 * p + 1);
        for (int i = 0; i < 4; ++i) {
          uint64_t v = g2();
          memcpy(my_seed + 8 * i, &v, 8);
        }
        Rep3State state0 = Rep3State::create(nets0[p], my_seed);  // groth16.rs:371
        Rep3State state1 = state0.fork(0);                         // :372
        proofs[p] = CoGroth16<P, T>::template prove_inner<CircomReduction>(&nets0[p], &nets1[p], state0, state1, pk, m, sw[p],
                                                                            r ? &r3[p] : nullptr, s ? &s3[p] : nullptr, &hs[p]);
      } catch (const std::exception& e) {
        errs[p] = e.what();
      }
    });
  }
  for (auto& t : th) t.join();
  for (int p = 0; p < 3; ++p)
    if (!errs[p].empty()) throw Error("party " + std::to_string(p) + ": " + errs[p]);
  std::string j0 = proof_to_json(proofs[0]);
  if (j0 != proof_to_json(proofs[1]) || j0 != proof_to_json(proofs[2])) throw Error("the three parties disagree on the proof");
  if (h_shares_out) {
    const size_t n = hs[0].size();
    if (3 * n > h_cap) throw Error("h_shares_out too small");
    for (int p = 0; p < 3; ++p) memcpy(h_shares_out + 4 * n * p, hs[p].data(), 32 * n);
  }
  return write_out(j0, out, cap);
}

}  // namespace

extern "C" {

const char* cog16_last_error(void) { return g_err.c_str(); }

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
