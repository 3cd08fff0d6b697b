// Fused CircomReduction tail on the device: h = (A*B - C) evaluated on the coset, from the constraint
// evaluations a, b. Mirrors co-circom/co-groth16/src/groth16/reduction.rs:135-192 step by step
// (6 NTTs, 2 local_mul_vec, 3 coset-table multiplications, 1 subtraction) on one HIP stream, so a proof
// needs one upload of a, b (+ masks) and one download of h instead of 12 host<->device round trips.
#include "common.hpp"

using namespace csh;

extern "C" {

int csh_groth16_h_dev(csh_domain_t dom, const uint64_t shift[4], int protocol, uint64_t* a, uint64_t* b, const uint64_t* mask_c,
                      const uint64_t* mask_ab, uint64_t* h_out, void* stream) {
  CSH_REQUIRE(dom && shift && a && b && h_out, "NULL argument");
  CSH_REQUIRE(protocol == 0 || protocol == 1, "protocol must be 0 (plain/Shamir) or 1 (Rep3)");
  CSH_TRY(ensure_device());
  const Domain* d = reinterpret_cast<const Domain*>(dom);
  const size_t n = domain_size_of(d);
  const csh_curve_t f = domain_curve_of(d);
  const uint32_t ncomp = protocol == 1 ? 2 : 1;
  hipStream_t st = resolve_stream(stream);
  Arena& ar = arena_for((hipStream_t)((uintptr_t)st ^ 0x4));
  CSH_TRY(ar.reserve(2 * Arena::padded(32 * n)));
  uint64_t* table = ar.take<uint64_t>(4 * n);
  uint64_t* ab2 = ar.take<uint64_t>(4 * n);
  uint64_t* ab = h_out;

  CSH_TRY(ntt_coset_table(d, shift, table, st));                                             // reduction.rs:100 (45-60)
  if (protocol == 1)
    CSH_TRY(csh_rep3_local_mul_vec_dev(f, a, b, mask_c, ab, n, st));                         // :160
  else
    CSH_TRY(csh_vec_mul_dev(f, a, b, ab, n, st));
  for (uint64_t* v : {a, b}) {                                                               // :139-155
    CSH_TRY(ntt_run(d, v, ncomp, true, st));
    CSH_TRY(csh_vec_mul_table_dev(f, v, table, n, ncomp, st));
    CSH_TRY(ntt_run(d, v, ncomp, false, st));
  }
  CSH_TRY(ntt_run(d, ab, 1, true, st));                                                      // :163
  CSH_TRY(csh_vec_mul_table_dev(f, ab, table, n, 1, st));                                    // :165-171
  CSH_TRY(ntt_run(d, ab, 1, false, st));                                                     // :174
  if (protocol == 1)
    CSH_TRY(csh_rep3_local_mul_vec_dev(f, a, b, mask_ab, ab2, n, st));                       // :182
  else
    CSH_TRY(csh_vec_mul_dev(f, a, b, ab2, n, st));
  CSH_TRY(csh_vec_sub_dev(f, ab2, ab, h_out, n, 1, st));                                     // :185-190
  return CSH_OK;
}

int csh_groth16_h(csh_domain_t dom, const uint64_t shift[4], int protocol, uint64_t* a, uint64_t* b, const uint64_t* mask_c,
                  const uint64_t* mask_ab, uint64_t* h_out) {
  CSH_REQUIRE(dom && shift && a && b && h_out, "NULL argument");
  CSH_REQUIRE(protocol == 0 || protocol == 1, "protocol must be 0 (plain/Shamir) or 1 (Rep3)");
  const Domain* d = reinterpret_cast<const Domain*>(dom);
  const size_t n = domain_size_of(d);
  const size_t sb = 32 * n * (protocol == 1 ? 2 : 1), eb = 32 * n;
  HostStage h;
  CSH_TRY(h.begin(2 * Arena::padded(sb) + 3 * Arena::padded(eb)));
  uint64_t *da, *db, *dmc = nullptr, *dmab = nullptr, *dh;
  CSH_TRY(h.up(da, a, sb));
  CSH_TRY(h.up(db, b, sb));
  if (mask_c) CSH_TRY(h.up(dmc, mask_c, eb));
  if (mask_ab) CSH_TRY(h.up(dmab, mask_ab, eb));
  CSH_TRY(h.up(dh, nullptr, eb));
  CSH_TRY(csh_groth16_h_dev(dom, shift, protocol, da, db, dmc, dmab, dh, h.st));
  return h.down(h_out, dh, eb);
}

int csh_groth16_h_rep3_seeded(csh_domain_t dom, const uint64_t shift[4], uint64_t* a, uint64_t* b, const uint8_t seed1[32], uint64_t off1,
                              const uint8_t seed2[32], uint64_t off2, uint64_t* h_out) {
  CSH_REQUIRE(dom && shift && a && b && h_out && seed1 && seed2, "NULL argument");
  const Domain* d = reinterpret_cast<const Domain*>(dom);
  const size_t n = domain_size_of(d);
  const csh_curve_t f = domain_curve_of(d);
  const size_t sb = 64 * n, eb = 32 * n;
  HostStage h;
  CSH_TRY(h.begin(2 * Arena::padded(sb) + 3 * Arena::padded(eb)));
  uint64_t *da, *db, *dmc, *dmab, *dh;
  CSH_TRY(h.up(da, a, sb));
  CSH_TRY(h.up(db, b, sb));
  CSH_TRY(h.up(dmc, nullptr, eb));
  CSH_TRY(h.up(dmab, nullptr, eb));
  CSH_TRY(h.up(dh, nullptr, eb));
  CSH_TRY(csh_rep3_masks_dev(f, seed1, off1, seed2, off2, dmc, n, h.st));
  CSH_TRY(csh_rep3_masks_dev(f, seed1, off1 + n, seed2, off2 + n, dmab, n, h.st));
  CSH_TRY(csh_groth16_h_dev(dom, shift, 1, da, db, dmc, dmab, dh, h.st));
  return h.down(h_out, dh, eb);
}

}  // extern "C"
