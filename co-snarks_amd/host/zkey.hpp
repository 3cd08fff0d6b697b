// snarkjs binfile ingest (zkey / wtns) -> ProvingKey + ConstraintMatrices + witness.
// The reference delegates this to the external crate taceo-circom-types 0.3.1 (Cargo.lock:4799; call site
// co-circom/co-circom/src/bin/co-circom.rs:1005-1006); layout = the published iden3 format:
// sections (type u32, len u64); 2 header, 3 IC, 4 coeffs, 5 A, 6 B1, 7 B2, 8 L, 9 H. Point coordinates are
// already Montgomery little-endian (copied verbatim); coefficient values are doubly Montgomery-encoded.
#pragma once
#include <map>

#include "types.hpp"

namespace cosnarks {

struct Sections {
  std::map<uint32_t, std::pair<size_t, size_t>> pos;  // type -> (offset, len)
  Sections(const uint8_t* d, size_t n, const char* magic) {
    if (n < 12 || memcmp(d, magic, 4) != 0) throw Error(std::string("bad magic, expected ") + magic);
    uint32_t nsec;
    memcpy(&nsec, d + 8, 4);
    size_t off = 12;
    for (uint32_t i = 0; i < nsec; ++i) {
      if (n - off < 12) throw Error("truncated section table");
      uint32_t typ;
      uint64_t len;
      memcpy(&typ, d + off, 4);
      memcpy(&len, d + off + 4, 8);
      off += 12;
      if (len > n - off) throw Error("truncated section");  // overflow-safe: len is an attacker-controlled u64
      if (!pos.count(typ)) pos[typ] = {off, (size_t)len};
      off += len;
    }
  }
  std::pair<size_t, size_t> at(uint32_t t) const {
    auto it = pos.find(t);
    if (it == pos.end()) throw Error("missing section " + std::to_string(t));
    return it->second;
  }
};

// num_instance_variables (public inputs + the constant 1) from the zkey header alone
template <class P>
size_t zkey_num_instance_variables(const uint8_t* d, size_t n) {
  Sections s(d, n, "zkey");
  auto [h, hl] = s.at(2);
  const size_t need = 4 + sizeof(typename P::Fq) + 4 + sizeof(typename P::Fr) + 12;
  if (hl < need) throw Error("truncated zkey header");
  uint32_t n_public;
  memcpy(&n_public, d + h + need - 8, 4);
  return (size_t)n_public + 1;
}

template <class P>
void parse_zkey(const uint8_t* d, size_t n, ProvingKey<P>& pk, ConstraintMatrices<P>& m, bool upload = true) {
  using Fq = typename P::Fq;
  using Fr = typename P::Fr;
  Sections s(d, n, "zkey");
  {
    auto [off, len] = s.at(1);
    if (len < 4) throw Error("truncated zkey protocol section");
    uint32_t proto;
    memcpy(&proto, d + off, 4);
    if (proto != 1) throw Error("not a groth16 zkey");
  }
  auto [h, hl] = s.at(2);
  // header = n8q, q, n8r, r, nVars, nPublic, domainSize, alpha1, beta1, beta2, gamma2, delta1, delta2
  const size_t header_need = 4 + sizeof(Fq) + 4 + sizeof(Fr) + 12 + 3 * 2 * sizeof(Fq) + 3 * 4 * sizeof(Fq);
  if (hl < 4) throw Error("truncated zkey header");
  uint32_t n8q, n8r, n_vars, n_public, domain;
  memcpy(&n8q, d + h, 4);
  if (n8q != sizeof(Fq)) throw Error("zkey base field size does not match the selected curve");  // before the length check: a key of the other curve has another header size
  if (hl < header_need) throw Error("truncated zkey header");
  if (memcmp(d + h + 4, Fq::Params::MOD, n8q) != 0) throw Error("zkey base field modulus does not match the selected curve");
  size_t off = h + 4 + n8q;
  memcpy(&n8r, d + off, 4);
  if (n8r != sizeof(Fr)) throw Error("zkey scalar field size mismatch");
  if (memcmp(d + off + 4, Fr::Params::MOD, n8r) != 0) throw Error("zkey scalar field modulus mismatch");
  off += 4 + n8r;
  memcpy(&n_vars, d + off, 4);
  memcpy(&n_public, d + off + 4, 4);
  memcpy(&domain, d + off + 8, 4);
  off += 12;
  if ((uint64_t)n_public + 1 > n_vars) throw Error("zkey header: more public inputs than variables");
  if (domain == 0 || (domain & (domain - 1)) != 0) throw Error("zkey header: domain size is not a power of two");
  auto g1 = [&](size_t o) {
    AffineT<Fq> p;
    memcpy(&p, d + o, 2 * n8q);
    return p;
  };
  auto g2 = [&](size_t o) {
    AffineT<typename P::Fq2> p;
    memcpy(&p, d + o, 4 * n8q);
    return p;
  };
  pk.alpha_g1 = g1(off); off += 2 * n8q;
  pk.beta_g1 = g1(off);  off += 2 * n8q;
  pk.beta_g2 = g2(off);  off += 4 * n8q;
  pk.gamma_g2 = g2(off); off += 4 * n8q;
  pk.delta_g1 = g1(off); off += 2 * n8q;
  pk.delta_g2 = g2(off);
  // IC stays on the host; the five queries are uploaded from the file image itself ("zkey points are already
  // Montgomery LE -- they can be DMA'd without conversion"). calculate_coeff reads query[0 ..= n_public] on the host.
  // every point section must hold exactly the number of points the header implies (a short section would make the MSMs
  // read past the uploaded bases, a ragged one would shift every point after it)
  auto expect = [&](uint32_t sec, size_t l, size_t point_bytes, size_t count) {
    if (l != point_bytes * count)
      throw Error("zkey section " + std::to_string(sec) + ": " + std::to_string(l) + " bytes, expected " + std::to_string(count) + " points of " +
                  std::to_string(point_bytes) + " bytes");
  };
  const size_t keep = (size_t)n_public + 1;
  {
    auto [o, l] = s.at(3);
    expect(3, l, 2 * n8q, keep);
    pk.ic.resize(keep);
    memcpy(pk.ic.data(), d + o, l);
  }
  auto q1 = [&](uint32_t sec, Query<Fq>& q, size_t keep_host, size_t count, size_t lead_pad = 0) {
    auto [o, l] = s.at(sec);
    expect(sec, l, 2 * n8q, count);
    q.upload_from(P::ID, CSH_G1, d + o, count, keep_host, upload, lead_pad);
  };
  q1(5, pk.a_query, keep, n_vars);
  q1(6, pk.b_g1_query, keep, n_vars);
  q1(8, pk.l_query, 0, (size_t)n_vars - keep, keep);  // (padded to the a / b queries' length: Query::lead)
  q1(9, pk.h_query, 0, domain);
  {
    auto [o, l] = s.at(7);
    expect(7, l, 4 * n8q, n_vars);
    pk.b_g2_query.upload_from(P::ID, CSH_G2, d + o, n_vars, keep, upload);
  }
  if (upload) {
    pk.build_tables();
    pk.place_default();  // one prover's queries over several GPUs when cog16_set_prover_devices asked for it
  }
  // coefficients -> ConstraintMatrices (public-input rows, constraint index >= num_constraints, are dropped:
  // the reference overwrites exactly those evaluation slots, groth16/reduction.rs:111-113)
  auto [c, cl] = s.at(4);
  if (cl < 4) throw Error("truncated zkey coefficient section");
  uint32_t ncoef;
  memcpy(&ncoef, d + c, 4);
  if ((cl - 4) / (12 + (size_t)n8r) < ncoef) throw Error("zkey coefficient section shorter than its coefficient count");
  size_t o = c + 4;
  uint32_t max_constraint = 0;
  struct Coef { uint32_t m, row, sig; Fr v; };
  std::vector<Coef> coefs(ncoef);
  for (uint32_t i = 0; i < ncoef; ++i) {
    memcpy(&coefs[i].m, d + o, 4);
    memcpy(&coefs[i].row, d + o + 4, 4);
    memcpy(&coefs[i].sig, d + o + 8, 4);
    Fr raw;
    memcpy(&raw, d + o + 12, n8r);
    coefs[i].v = raw.from_mont();  // v R^2 -> v R
    o += 12 + n8r;
    if (coefs[i].m > 1) throw Error("zkey coefficient: matrix index must be 0 (A) or 1 (B)");
    if (coefs[i].sig >= n_vars) throw Error("zkey coefficient: signal index " + std::to_string(coefs[i].sig) + " >= nVars");  // an out-of-range column would be an out-of-bounds device read in k_eval_rows
    if (coefs[i].row >= domain) throw Error("zkey coefficient: constraint index beyond the domain");
    if (coefs[i].row > max_constraint) max_constraint = coefs[i].row;
  }
  if (max_constraint < n_public) throw Error("zkey coefficients: fewer rows than public inputs");
  m.num_instance_variables = n_public + 1;
  m.num_witness_variables = n_vars - n_public - 1;
  m.num_constraints = max_constraint - n_public;
  m.a.assign(m.num_constraints, {});
  m.b.assign(m.num_constraints, {});
  for (auto& cf : coefs)
    if (cf.row < m.num_constraints) (cf.m == 0 ? m.a : m.b)[cf.row].push_back({cf.v, cf.sig});
  (void)domain;
  if (upload) m.upload();
}

// witness values (canonical little-endian) -> Montgomery Fr
template <class P>
std::vector<typename P::Fr> parse_wtns(const uint8_t* d, size_t n) {
  using Fr = typename P::Fr;
  Sections s(d, n, "wtns");
  auto [h, hl] = s.at(1);
  if (hl < 4 + sizeof(Fr) + 4) throw Error("truncated wtns header");
  uint32_t n8, cnt;
  memcpy(&n8, d + h, 4);
  if (n8 != sizeof(Fr)) throw Error("wtns field size mismatch");
  if (memcmp(d + h + 4, Fr::Params::MOD, n8) != 0) throw Error("wtns field modulus does not match the selected curve");
  memcpy(&cnt, d + h + 4 + n8, 4);
  auto [o, l] = s.at(2);
  if (l < (size_t)cnt * n8) throw Error("truncated wtns");
  std::vector<Fr> out(cnt);
  for (uint32_t i = 0; i < cnt; ++i) {
    Fr raw;
    memcpy(&raw, d + o + (size_t)i * n8, n8);
    out[i] = raw.to_mont();
  }
  return out;
}

// the first `count` witness values only (the public inputs), Montgomery
template <class P>
std::vector<typename P::Fr> parse_wtns_prefix(const uint8_t* d, size_t n, size_t count) {
  using Fr = typename P::Fr;
  Sections s(d, n, "wtns");
  auto [h, hl] = s.at(1);
  if (hl < 4 + sizeof(Fr) + 4) throw Error("truncated wtns header");
  uint32_t n8, cnt;
  memcpy(&n8, d + h, 4);
  if (n8 != sizeof(Fr)) throw Error("wtns field size mismatch");
  if (memcmp(d + h + 4, Fr::Params::MOD, n8) != 0) throw Error("wtns field modulus does not match the selected curve");
  memcpy(&cnt, d + h + 4 + n8, 4);
  auto [o, l] = s.at(2);
  if (count > cnt || l < (size_t)cnt * n8) throw Error("truncated wtns");
  std::vector<Fr> out(count);
  for (size_t i = 0; i < count; ++i) {
    Fr raw;
    memcpy(&raw, d + o + i * n8, n8);
    out[i] = raw.to_mont();
  }
  return out;
}

// The same witness values as a device vector: the canonical little-endian bytes are uploaded as they lie in the file
// and converted to Montgomery form on the device (one multiplication by R^2 per element, csh_lincomb_dev with k = 1).
template <class P>
DeviceScalars parse_wtns_to_device(const uint8_t* d, size_t n, size_t skip_first = 0) {
  using Fr = typename P::Fr;
  Sections s(d, n, "wtns");
  auto [h, hl] = s.at(1);
  if (hl < 4 + sizeof(Fr) + 4) throw Error("truncated wtns header");
  uint32_t n8, cnt;
  memcpy(&n8, d + h, 4);
  if (n8 != sizeof(Fr)) throw Error("wtns field size mismatch");
  if (memcmp(d + h + 4, Fr::Params::MOD, n8) != 0) throw Error("wtns field modulus does not match the selected curve");
  memcpy(&cnt, d + h + 4 + n8, 4);
  auto [o, l] = s.at(2);
  if (l < (size_t)cnt * n8) throw Error("truncated wtns");
  if (skip_first > cnt) throw Error("wtns shorter than the public inputs");
  const size_t m = cnt - skip_first;
  DeviceScalars out(d + o + skip_first * n8, m);  // raw canonical limbs
  const Fr r2 = Fr::r2();
  const uint64_t* ptrs[1] = {reinterpret_cast<const uint64_t*>(out.dev)};
  check(csh_lincomb_dev(P::ID, ptrs, reinterpret_cast<const uint64_t*>(&r2), 1, reinterpret_cast<uint64_t*>(out.dev), m, nullptr), "csh_lincomb_dev");
  check(csh_sync(nullptr), "csh_sync");
  return out;
}

}  // namespace cosnarks
