// ark-serialize (uncompressed) wire formats either side of the hot path, restated from the published
// ark-serialize 0.5/0.6 semantics as the reference uses them:
//   * field element: little-endian canonical bytes (32 for the scalar fields)           CanonicalSerialize for Fp
//   * Vec<T>: u64 little-endian length, then the items                                  rep3/network.rs:103-109, 152-156
//   * Matrix<F> = Vec<Vec<(F, usize)>>, usize as u64                                     co-groth16/src/lib.rs:257-262
//   * short-Weierstrass affine point, uncompressed: x, then y with SWFlags in the two top bits of its last byte
//     (bit 7: y is the lexicographically larger root, bit 6: point at infinity, coordinates zero)
// The conversions to / from the Montgomery limbs the C ABI takes are done here on the host for small objects; bulk
// vectors go through `fr_vec_from_canonical`, one multiplication by R^2 per element on the device (csh_vec_mul_table
// with a constant table would cost the same traffic; the witness upload already crosses PCIe once).
#pragma once
#include "types.hpp"

namespace cosnarks {
namespace ark {

struct Reader {
  const uint8_t* p;
  size_t n, off = 0;
  Reader(const uint8_t* data, size_t len) : p(data), n(len) {}
  void need(size_t k) const {
    if (off + k > n) throw Error("ark-serialize: unexpected end of input");
  }
  uint64_t u64() {
    need(8);
    uint64_t v;
    memcpy(&v, p + off, 8);
    off += 8;
    return v;
  }
  // canonical little-endian field element -> Montgomery
  template <class Fr>
  Fr field() {
    need(sizeof(Fr));
    Fr raw;
    memcpy(&raw, p + off, sizeof(Fr));
    off += sizeof(Fr);
    if (csh::limbs_geq<Fr::N>(raw.l, Fr::Params::MOD)) throw Error("ark-serialize: field element not canonical");
    return raw.to_mont();
  }
  bool done() const { return off == n; }
};

template <class Fr>
inline std::vector<Fr> read_vec(Reader& r) {  // Vec<F>
  const uint64_t len = r.u64();
  if (len > (r.n - r.off) / sizeof(Fr)) throw Error("ark-serialize: Vec length exceeds the input");
  std::vector<Fr> v(len);
  for (auto& x : v) x = r.template field<Fr>();
  return v;
}

template <class Fr>
inline std::vector<std::vector<std::pair<Fr, size_t>>> read_matrix(Reader& r) {  // Matrix<F>
  const uint64_t rows = r.u64();
  if (rows > (r.n - r.off) / 8) throw Error("ark-serialize: Matrix length exceeds the input");
  std::vector<std::vector<std::pair<Fr, size_t>>> m(rows);
  for (auto& row : m) {
    const uint64_t len = r.u64();
    if (len > (r.n - r.off) / (sizeof(Fr) + 8)) throw Error("ark-serialize: row length exceeds the input");
    row.reserve(len);
    for (uint64_t i = 0; i < len; ++i) {
      Fr c = r.template field<Fr>();
      row.push_back({c, (size_t)r.u64()});
    }
  }
  return m;
}

template <class Fr>
inline void write_field(std::vector<uint8_t>& out, const Fr& mont) {
  const Fr c = mont.from_mont();
  const uint8_t* b = reinterpret_cast<const uint8_t*>(&c);
  out.insert(out.end(), b, b + sizeof(Fr));
}
inline void write_u64(std::vector<uint8_t>& out, uint64_t v) {
  const uint8_t* b = reinterpret_cast<const uint8_t*>(&v);
  out.insert(out.end(), b, b + 8);
}
template <class Fr>
inline void write_vec(std::vector<uint8_t>& out, const std::vector<Fr>& v) {
  write_u64(out, v.size());
  for (auto& x : v) write_field(out, x);
}

// y > -y as canonical integers (SWFlags::from_y_coordinate)
template <class Fq>
inline bool y_is_negative(const Fq& y_mont) {
  const Fq y = y_mont.from_mont(), ny = Fq::neg(y_mont).from_mont();
  for (int i = Fq::N - 1; i >= 0; --i) {
    if (y.l[i] != ny.l[i]) return y.l[i] > ny.l[i];
  }
  return false;
}
// G1 affine, uncompressed. The C ABI's convention for infinity is x = y = 0.
template <class Fq>
inline void write_g1(std::vector<uint8_t>& out, const AffineT<Fq>& pt) {
  const size_t at = out.size();
  if (pt.is_inf()) {
    out.resize(at + 2 * sizeof(Fq), 0);
    out.back() |= 0x40;
    return;
  }
  write_field(out, pt.x);
  write_field(out, pt.y);
  if (y_is_negative(pt.y)) out.back() |= 0x80;
}
template <class Fq>
inline AffineT<Fq> read_g1(Reader& r, bool check_sign_flag = true) {
  r.need(2 * sizeof(Fq));
  Fq x, y;
  memcpy(&x, r.p + r.off, sizeof(Fq));
  memcpy(&y, r.p + r.off + sizeof(Fq), sizeof(Fq));
  r.off += 2 * sizeof(Fq);
  const uint32_t flags = y.l[Fq::N - 1] >> 30;
  y.l[Fq::N - 1] &= 0x3fffffffu;
  if (flags & 1) return AffineT<Fq>::inf();  // bit 6 of the last byte
  if (csh::limbs_geq<Fq::N>(x.l, Fq::Params::MOD) || csh::limbs_geq<Fq::N>(y.l, Fq::Params::MOD)) throw Error("ark-serialize: coordinate not canonical");
  AffineT<Fq> pt{x.to_mont(), y.to_mont()};
  if (check_sign_flag && ((flags >> 1) & 1) != (y_is_negative(pt.y) ? 1u : 0u)) throw Error("ark-serialize: y-sign flag does not match y");
  return pt;
}

// G2 affine over Fq2 = (c0, c1), uncompressed: x.c0, x.c1, y.c0, y.c1, flags in the last byte; `y > -y` orders quadratic-extension
// elements by c1 first, then c0 (ark-ff's Ord for QuadExtField)
template <class Fq2>
inline bool y2_is_negative(const Fq2& y_mont) {
  using Fq = decltype(y_mont.c0);
  const Fq2 n = Fq2::neg(y_mont);
  const Fq y1 = y_mont.c1.from_mont(), n1 = n.c1.from_mont(), y0 = y_mont.c0.from_mont(), n0 = n.c0.from_mont();
  for (int i = Fq::N - 1; i >= 0; --i)
    if (y1.l[i] != n1.l[i]) return y1.l[i] > n1.l[i];
  for (int i = Fq::N - 1; i >= 0; --i)
    if (y0.l[i] != n0.l[i]) return y0.l[i] > n0.l[i];
  return false;
}
template <class Fq2>
inline void write_g2(std::vector<uint8_t>& out, const AffineT<Fq2>& pt) {
  const size_t at = out.size();
  if (pt.is_inf()) {
    out.resize(at + 2 * sizeof(Fq2), 0);
    out.back() |= 0x40;
    return;
  }
  write_field(out, pt.x.c0);
  write_field(out, pt.x.c1);
  write_field(out, pt.y.c0);
  write_field(out, pt.y.c1);
  if (y2_is_negative(pt.y)) out.back() |= 0x80;
}
template <class Fq2>
inline AffineT<Fq2> read_g2(Reader& r, bool check_sign_flag = true) {
  using Fq = decltype(Fq2().c0);
  r.need(2 * sizeof(Fq2));
  Fq c[4];
  memcpy(c, r.p + r.off, sizeof c);
  r.off += sizeof c;
  const uint32_t flags = c[3].l[Fq::N - 1] >> 30;
  c[3].l[Fq::N - 1] &= 0x3fffffffu;
  if (flags & 1) return AffineT<Fq2>::inf();
  for (auto& v : c)
    if (csh::limbs_geq<Fq::N>(v.l, Fq::Params::MOD)) throw Error("ark-serialize: coordinate not canonical");
  AffineT<Fq2> pt{{c[0].to_mont(), c[1].to_mont()}, {c[2].to_mont(), c[3].to_mont()}};
  if (check_sign_flag && ((flags >> 1) & 1) != (y2_is_negative(pt.y) ? 1u : 0u)) throw Error("ark-serialize: y-sign flag does not match y");
  return pt;
}

// ark-groth16 `ProvingKey<E>` as `deserialize_uncompressed_unchecked` reads it (co-groth16/src/lib.rs:257, Validate::No: no curve or
// subgroup check, the y-sign bit of an uncompressed point is not consulted): vk {alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1},
// beta_g1, delta_g1, a_query, b_g1_query, b_g2_query, h_query, l_query -- CanonicalSerialize's derive order; the leading vk is the
// layout of the reference's committed circuit.vk. The queries stay on the host here; the caller uploads them.
template <class P>
inline void read_proving_key(Reader& r, ProvingKey<P>& pk) {
  using Fq = typename P::Fq;
  using Fq2 = typename P::Fq2;
  auto g1 = [&] { return read_g1<Fq>(r, false); };
  auto g2 = [&] { return read_g2<Fq2>(r, false); };
  auto vec1 = [&](std::vector<AffineT<Fq>>& v) {
    const uint64_t n = r.u64();
    if (n > (r.n - r.off) / (2 * sizeof(Fq))) throw Error("ark-serialize: Vec length exceeds the input");
    v.resize(n);
    for (auto& pt : v) pt = g1();
  };
  pk.alpha_g1 = g1();
  pk.beta_g2 = g2();
  pk.gamma_g2 = g2();
  pk.delta_g2 = g2();
  vec1(pk.ic);
  pk.beta_g1 = g1();
  pk.delta_g1 = g1();
  vec1(pk.a_query.host);
  vec1(pk.b_g1_query.host);
  {
    const uint64_t n = r.u64();
    if (n > (r.n - r.off) / (2 * sizeof(Fq2))) throw Error("ark-serialize: Vec length exceeds the input");
    pk.b_g2_query.host.resize(n);
    for (auto& pt : pk.b_g2_query.host) pt = g2();
  }
  vec1(pk.h_query.host);
  vec1(pk.l_query.host);
}
// ark-groth16 `Proof<E>` {a, b, c}, Compress::No
template <class P>
inline std::vector<uint8_t> write_proof(const Proof<P>& pr) {
  std::vector<uint8_t> out;
  write_g1(out, pr.a);
  write_g2(out, pr.b);
  write_g1(out, pr.c);
  return out;
}

// ---- Rep3NetworkExt::{send_many, recv_many} payloads (mpc-core/src/protocols/rep3/network.rs:103-109, 152-156) ----------------------------
// What a GPU party has to put on / take off the wire to face two reference CPU parties: ONE `Network::send(to, Bytes)` per call whose
// payload is `data.serialize_uncompressed()` of the slice -- ark-serialize's impl for `[T]`: the count as u64 little-endian, then every
// item uncompressed -- and `recv_many` is `Vec::<F>::deserialize_uncompressed_unchecked(&data[..])` of one received message (Validate::No:
// no curve / subgroup check; a field element >= p still fails, Fp::from_bigint; an uncompressed point's y-sign bit is not consulted).
// `send_to` / `recv_from` / `send_next` / `broadcast` ... are the same with a one-item slice (:96-99, 140-149). The transports below
// (TCP / TLS / QUIC length-delimited frames, mpc-net) carry these payloads opaquely and stay in the Rust host.
template <class Fr>
inline std::vector<uint8_t> send_many_fields(const std::vector<Fr>& data) {
  std::vector<uint8_t> out;
  out.reserve(8 + sizeof(Fr) * data.size());  // Vec::with_capacity(data.serialized_size(Compress::No))
  write_vec(out, data);
  return out;
}
template <class Fr>
inline std::vector<Fr> recv_many_fields(const uint8_t* msg, size_t len) {
  Reader r(msg, len);
  std::vector<Fr> v = read_vec<Fr>(r);
  // trailing bytes are not an error for ark's reader (it deserializes a prefix); a short message is
  return v;
}
template <class Fq>
inline std::vector<uint8_t> send_many_g1(const std::vector<AffineT<Fq>>& data) {  // curve points travel in affine form, uncompressed
  std::vector<uint8_t> out;
  out.reserve(8 + 2 * sizeof(Fq) * data.size());
  write_u64(out, data.size());
  for (auto& pt : data) write_g1(out, pt);
  return out;
}
template <class Fq>
inline std::vector<AffineT<Fq>> recv_many_g1(const uint8_t* msg, size_t len) {
  Reader r(msg, len);
  const uint64_t n = r.u64();
  if (n > (r.n - r.off) / (2 * sizeof(Fq))) throw Error("ark-serialize: Vec length exceeds the input");
  std::vector<AffineT<Fq>> v(n);
  for (auto& pt : v) pt = read_g1<Fq>(r, /*check_sign_flag=*/false);
  return v;
}

// snarkjs wtns container as the reference's Witness::from_reader consumes it; tolerant of zeroed section headers (the
// Penumbra fixtures): positional fields -> Montgomery values
template <class Fr>
inline std::vector<Fr> read_wtns_positional(const uint8_t* d, size_t n) {
  if (n < 24 + 4 || memcmp(d, "wtns", 4) != 0) throw Error("bad wtns magic");
  size_t off = 24;
  uint32_t n8, cnt;
  memcpy(&n8, d + off, 4);
  off += 4;
  if (n8 != sizeof(Fr) || off + n8 + 4 + 12 > n) throw Error("wtns field size mismatch");
  if (memcmp(d + off, Fr::Params::MOD, n8) != 0) throw Error("wtns prime does not match the selected field");
  off += n8;
  memcpy(&cnt, d + off, 4);
  off += 4 + 12;
  if (off + (size_t)cnt * n8 != n) throw Error("unexpected wtns length");
  std::vector<Fr> out(cnt);
  for (uint32_t i = 0; i < cnt; ++i) {
    Fr raw;
    memcpy(&raw, d + off + (size_t)i * n8, n8);
    out[i] = raw.to_mont();
  }
  return out;
}

}  // namespace ark
}  // namespace cosnarks
