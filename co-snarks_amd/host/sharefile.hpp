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
//           (co-circom-types/src/lib.rs:163-173); variants Replicated = 0 and Additive = 2 carry share data,
//           SeededReplicated = 1 { a, b: SeededType } and SeededAdditive = 3 (SeededType) -- what the `split-witness`
//           command writes (Compression::SeededHalfShares, co-circom.rs:693-697) -- carry either share data or a seed:
//           SeededType (rep3.rs:152-165) = enum { Shares(bytes(Vec<F>)) = 0, Seed([u8; 32], usize, PhantomData) = 1 };
//           a [u8; 32] is a serde tuple (32 raw bytes), usize a u64.
//   Seed expansion (rep3.rs:185-196): ChaCha12Rng::from_seed(seed) (SeedRng, rep3.rs:40; rand_chacha 0.3.1), `len`
//           draws of F::rand. ark-ff 0.6.0 (Cargo.lock:390; not vendored -- restated from the published source): four
//           next_u64 words = 32 keystream bytes as little-endian limbs, the top limb masked to MODULUS_BIT_SIZE bits,
//           redrawn while >= p; the limbs ARE the element's Montgomery representation.
//   Shamir: SharedWitness { public_inputs: bytes(Vec<F>), witness: bytes(Vec<ShamirShare>) } (lib.rs:204-218)
// The reference commits no share-file fixture; the byte layout is pinned by an independent restatement
// in the test suite agreeing byte for byte (tests/test_share_files_cpu.py), and by proofs made from the files equalling
// the plain proof. The seed expansion has no reference vector either: PARITY UNPINNED for the two seeded variants (a
// mismatch with the reference's sampler would show as shares that do not reconstruct, i.e. a proof that fails to verify).
#pragma once
#include "arkwire.hpp"
#include "mpc.hpp"

namespace cosnarks {
namespace sharefile {

enum Rep3Variant : uint32_t { REPLICATED = 0, SEEDED_REPLICATED = 1, ADDITIVE = 2, SEEDED_ADDITIVE = 3 };

// SeededType<Vec<F>, ChaCha12Rng> (rep3.rs:152-165)
template <class Fr>
struct SeededShare {
  bool is_seed = false;
  std::vector<Fr> shares;  // !is_seed
  uint8_t seed[32] = {0};  // is_seed
  uint64_t len = 0;
  size_t length() const { return is_seed ? (size_t)len : shares.size(); }
  // expand_vec (rep3.rs:185-196): F::rand over ChaCha12Rng::from_seed(seed)
  std::vector<Fr> expand() const {
    if (!is_seed) return shares;
    static_assert(sizeof(Fr) == 32, "32-byte scalar fields only");
    ChaCha12 rng(seed);
    const uint32_t top_mask = 0xffffffffu >> (256 - Fr::Params::BITS);
    std::vector<Fr> out((size_t)len);
    for (auto& x : out) {
      Fr raw;
      do {
        rng.fill_bytes(reinterpret_cast<uint8_t*>(&raw), 32);
        raw.l[Fr::N - 1] &= top_mask;
      } while (csh::limbs_geq<Fr::N>(raw.l, Fr::Params::MOD));
      x = raw;  // the sampled limbs are the Montgomery representation
    }
    return out;
  }
};

template <class P>
struct CompressedRep3SharedWitness {
  using Fr = typename P::Fr;
  std::vector<Fr> public_inputs;  // includes the constant 1
  Rep3Variant kind = REPLICATED;
  std::vector<Rep3PrimeFieldShare<Fr>> replicated;  // kind == REPLICATED
  std::vector<Fr> additive;                         // kind == ADDITIVE
  SeededShare<Fr> sa, sb;                           // kind == SEEDED_ADDITIVE (sa) / SEEDED_REPLICATED (sa, sb)
  size_t length() const {
    switch (kind) {
      case REPLICATED: return replicated.size();
      case ADDITIVE: return additive.size();
      case SEEDED_ADDITIVE: return sa.length();
      default: return sa.length();
    }
  }
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

template <class Fr>
inline void write_seeded(std::vector<uint8_t>& out, const SeededShare<Fr>& s) {
  write_u32(out, s.is_seed ? 1u : 0u);
  if (s.is_seed) {
    out.insert(out.end(), s.seed, s.seed + 32);
    ark::write_u64(out, s.len);
  } else {
    write_blob<Fr>(out, s.shares.data(), s.shares.size(), 1);
  }
}
template <class Fr>
inline SeededShare<Fr> read_seeded(ark::Reader& r) {
  SeededShare<Fr> s;
  r.need(4);
  uint32_t v;
  memcpy(&v, r.p + r.off, 4);
  r.off += 4;
  if (v == 0) {
    s.shares = read_blob<Fr>(r, 1);
  } else if (v == 1) {
    r.need(40);
    memcpy(s.seed, r.p + r.off, 32);
    r.off += 32;
    s.len = r.u64();
    if (s.len > (uint64_t(1) << 32)) throw Error("share file: seeded share length is implausible");
    s.is_seed = true;
  } else {
    throw Error("share file: unknown SeededType variant");
  }
  return s;
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
  else if (w.kind == SEEDED_ADDITIVE)
    write_seeded<Fr>(out, w.sa);
  else if (w.kind == SEEDED_REPLICATED) {
    write_seeded<Fr>(out, w.sa);
    write_seeded<Fr>(out, w.sb);
  } else
    throw Error("share file: unknown Rep3ShareVecType variant");
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
  } else if (kind == SEEDED_ADDITIVE) {
    w.sa = read_seeded<Fr>(r);
  } else if (kind == SEEDED_REPLICATED) {
    w.sa = read_seeded<Fr>(r);
    w.sb = read_seeded<Fr>(r);
    if (w.sa.length() != w.sb.length()) throw Error("share file: lengths of the two seeded shares do not match");  // rep3.rs:260-262
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
  if (c.kind == SEEDED_REPLICATED) {  // ReplicatedSeedType::expand_vec (rep3.rs:257-267)
    const std::vector<Fr> a = c.sa.expand(), b2 = c.sb.expand();
    if (a.size() != b2.size()) throw Error("share file: lengths of the two seeded shares do not match");
    out.witness.resize(a.size());
    for (size_t i = 0; i < a.size(); ++i) out.witness[i] = {a[i], b2[i]};
    return out;
  }
  if (c.kind == SEEDED_ADDITIVE) c.additive = c.sa.expand();  // then reshare_vec like the plain additive variant (lib.rs:70-72)
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
