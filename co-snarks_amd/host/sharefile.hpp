// Witness-share files either side of the prover: what `co-circom split-witness` writes and
// `co-circom generate-proof` reads (co-circom/co-circom/src/bin/co-circom.rs:700-735, 1014-1035), restated from the
// published bincode 1.3 default configuration (fixed-width little-endian integers) and the reference's serde glue:
//   * `ark_se` (mpc-core/src/serde_compat.rs:7-15) turns a CanonicalSerialize value into ONE byte string (compressed
//     mode; for field elements that is the same as uncompressed) and hands it to `serialize_bytes`
//       -> bincode: u64 byte length, then the bytes; inside: ark Vec<T> = u64 item count, then the items
//   * a serde enum is a u32 variant index followed by the variant's fields (Rep3ShareVecType, rep3.rs:135-150)
//   * Rep3PrimeFieldShare derives CanonicalSerialize: a, then b (rep3/arithmetic/types.rs:9-28)
//   * ShamirPrimeFieldShare is a one-field struct: the element itself
// Files:
//   Rep3:   CompressedRep3SharedWitness { public_inputs: bytes(Vec<F>), witness: Rep3ShareVecType }
//           (co-circom-types/src/lib.rs:163-173); variants Replicated = 0 and Additive = 2 carry share data; the two
//           seeded variants (1, 3) carry RNG seeds and are rejected here with a clear error.
//   Shamir: SharedWitness { public_inputs: bytes(Vec<F>), witness: bytes(Vec<ShamirShare>) } (lib.rs:204-218)
// The reference commits no share-file fixture; the byte layout is pinned by an independent restatement
// in the test suite agreeing byte for byte (tests/test_share_files_cpu.py), and by proofs made from the files equalling the plain proof.
#pragma once
#include "arkwire.hpp"
#include "mpc.hpp"

namespace cosnarks {
namespace sharefile {

enum Rep3Variant : uint32_t { REPLICATED = 0, SEEDED_REPLICATED = 1, ADDITIVE = 2, SEEDED_ADDITIVE = 3 };

template <class P>
struct CompressedRep3SharedWitness {
  using Fr = typename P::Fr;
  std::vector<Fr> public_inputs;  // includes the constant 1
  Rep3Variant kind = REPLICATED;
  std::vector<Rep3PrimeFieldShare<Fr>> replicated;  // kind == REPLICATED
  std::vector<Fr> additive;                         // kind == ADDITIVE
};

inline void write_u32(std::vector<uint8_t>& out, uint32_t v) {
  const uint8_t* b = reinterpret_cast<const uint8_t*>(&v);
  out.insert(out.end(), b, b + 4);
}
// serialize_bytes(ark-serialize(Vec<T>)) with `items` field elements per T
template <class Fr>
inline void write_blob(std::vector<uint8_t>& out, const Fr* elems, size_t count, size_t items) {
  ark::write_u64(out, 8 + count * items * sizeof(Fr));
  ark::write_u64(out, count);
  for (size_t i = 0; i < count * items; ++i) ark::write_field(out, elems[i]);
}
template <class Fr>
inline std::vector<Fr> read_blob(ark::Reader& r, size_t items) {
  const uint64_t bytes = r.u64();
  if (bytes > r.n - r.off || bytes < 8) throw Error("share file: byte-string length exceeds the input");
  const size_t end = r.off + bytes;
  const uint64_t count = r.u64();
  if (count > (bytes - 8) / (items * sizeof(Fr)) || 8 + count * items * sizeof(Fr) != bytes)
    throw Error("share file: Vec length does not match its byte string");
  std::vector<Fr> v(count * items);
  for (auto& x : v) x = r.template field<Fr>();
  if (r.off != end) throw Error("share file: trailing bytes inside a byte string");
  return v;
}

template <class P>
inline std::vector<uint8_t> write_rep3(const CompressedRep3SharedWitness<P>& w) {
  using Fr = typename P::Fr;
  std::vector<uint8_t> out;
  write_blob<Fr>(out, w.public_inputs.data(), w.public_inputs.size(), 1);
  write_u32(out, w.kind);
  if (w.kind == REPLICATED)
    write_blob<Fr>(out, reinterpret_cast<const Fr*>(w.replicated.data()), w.replicated.size(), 2);
  else if (w.kind == ADDITIVE)
    write_blob<Fr>(out, w.additive.data(), w.additive.size(), 1);
  else
    throw Error("share file: seeded share variants are not supported");
  return out;
}

template <class P>
inline CompressedRep3SharedWitness<P> read_rep3(const uint8_t* d, size_t n) {
  using Fr = typename P::Fr;
  ark::Reader r(d, n);
  CompressedRep3SharedWitness<P> w;
  w.public_inputs = read_blob<Fr>(r, 1);
  r.need(4);
  uint32_t kind;
  memcpy(&kind, r.p + r.off, 4);
  r.off += 4;
  if (kind == REPLICATED) {
    std::vector<Fr> flat = read_blob<Fr>(r, 2);
    w.replicated.resize(flat.size() / 2);
    memcpy((void*)w.replicated.data(), flat.data(), flat.size() * sizeof(Fr));
  } else if (kind == ADDITIVE) {
    w.additive = read_blob<Fr>(r, 1);
  } else if (kind == SEEDED_REPLICATED || kind == SEEDED_ADDITIVE) {
    throw Error("share file: seeded share variants are not supported (split the witness with compression none or half-shares)");
  } else {
    throw Error("share file: unknown Rep3ShareVecType variant");
  }
  w.kind = (Rep3Variant)kind;
  if (!r.done()) throw Error("share file: trailing bytes");
  return w;
}

template <class P>
inline std::vector<uint8_t> write_shamir(const SharedWitness<P, typename P::Fr>& w) {
  using Fr = typename P::Fr;
  std::vector<uint8_t> out;
  write_blob<Fr>(out, w.public_inputs.data(), w.public_inputs.size(), 1);
  write_blob<Fr>(out, w.witness.data(), w.witness.size(), 1);
  return out;
}
template <class P>
inline SharedWitness<P, typename P::Fr> read_shamir(const uint8_t* d, size_t n) {
  using Fr = typename P::Fr;
  ark::Reader r(d, n);
  SharedWitness<P, Fr> w;
  w.public_inputs = read_blob<Fr>(r, 1);
  w.witness = read_blob<Fr>(r, 1);
  if (!r.done()) throw Error("share file: trailing bytes");
  return w;
}

// co_circom::uncompress_shared_witness (co-circom/co-circom/src/lib.rs:58-79): replicated shares pass through, additive
// half shares are completed with one reshare_vec round (rep3/arithmetic.rs:149-160: send mine to the next party, the
// previous party's becomes my `b`).
template <class P>
inline SharedWitness<P, Rep3PrimeFieldShare<typename P::Fr>> uncompress(CompressedRep3SharedWitness<P>&& c, const LocalNetwork& net) {
  using Fr = typename P::Fr;
  SharedWitness<P, Rep3PrimeFieldShare<Fr>> out;
  out.public_inputs = std::move(c.public_inputs);
  if (c.kind == REPLICATED) {
    out.witness = std::move(c.replicated);
    return out;
  }
  Bytes b(c.additive.size() * sizeof(Fr));
  if (!b.empty()) memcpy(b.data(), c.additive.data(), b.size());
  net.send((net.id() + 1) % 3, std::move(b));
  Bytes prev = net.recv((net.id() + 2) % 3);
  if (prev.size() != c.additive.size() * sizeof(Fr)) throw Error("reshare_vec: invalid number of elements received");
  out.witness.resize(c.additive.size());
  for (size_t i = 0; i < c.additive.size(); ++i) {
    out.witness[i].a = c.additive[i];
    memcpy(&out.witness[i].b, prev.data() + i * sizeof(Fr), sizeof(Fr));
  }
  return out;
}

}  // namespace sharefile
}  // namespace cosnarks
