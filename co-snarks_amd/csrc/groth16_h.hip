// CircomReduction tail on the device: h = (A*B - C) evaluated on the coset, from the constraint evaluations a, b:
// co-circom/co-groth16/src/groth16/reduction.rs:135-192 (6 NTTs, 2 local_mul_vec, 3 coset-table multiplications, 1 subtraction)
// on one HIP stream, so a proof needs one upload of a, b (+ masks) and one download of h instead of 12 host<->device round
// trips. Kernel-level fusion: the three coset-table multiplications ride on the last pass of the inverse transforms (the table
// carries their 1/n), and the final product and subtraction are one kernel: 8 launches / HBM sweeps of n elements less.
#include <string.h>

#include "common.hpp"
#include "field.hpp"

using namespace csh;

namespace {
template <class F>
int libsnark_consts(const uint64_t gen_words[4], size_t n, uint64_t v[4], uint64_t neg_v[4], uint64_t g_inv[4]) {
  F g;
  memcpy(&g, gen_words, sizeof g);
  F gn = F::pow_u64(g, (uint64_t)n);
  F d = F::sub(gn, F::one());
  if (d.is_zero()) return CSH_ERR_INVALID;
  F vi = F::inv(d), nv = F::neg(vi), gi = F::inv(g);
  memcpy(v, &vi, 32);
  memcpy(neg_v, &nv, 32);
  memcpy(g_inv, &gi, 32);
  return CSH_OK;
}
}  // namespace

extern "C" {

int csh_groth16_h_dev(csh_domain_t dom, const uint64_t shift[4], int protocol, uint64_t* a, uint64_t* b, const uint64_t* mask_c,
                      const uint64_t* mask_ab, uint64_t* h_out, void* stream) {
  CSH_REQUIRE(dom && shift && a && b && h_out, "NULL argument");
  CSH_REQUIRE(protocol == 0 || protocol == 1, "protocol must be 0 (plain/Shamir) or 1 (Rep3)");
  if (protocol == 1) CSH_TRY(require_rep3_masks(mask_c && mask_ab, "groth16_h"));
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

  // Fused form (default): the coset table carries the inverse transform's 1/n and is applied by the LAST PASS of each
  // ifft_in_to_out where that pass multiplies by 1/n anyway (no table sweep, no extra multiplication: three sweeps and 3 n
  // modmuls less), and h = a b - c is one kernel. The unfused sequence below it is the reference's order step by step and runs
  // when the lazy NTT passes are switched off (CSH_NTT_LAZY=0) or tune "h_unfused" asks for it (A/B, tests).
  if (ntt_scale_table_supported(d) && !tune().h_unfused.load(std::memory_order_relaxed)) {
    const uint64_t* ctab = nullptr;  // (kept with the domain after the first witness map: Domain::cs_table)
    CSH_TRY(ntt_coset_table_scaled_cached(d, shift, table, st, &ctab));                      // reduction.rs:100 (45-60), times 1/n
    if (protocol == 1)
      CSH_TRY(csh_rep3_local_mul_vec_dev(f, a, b, mask_c, ab, n, st));                       // :160
    else
      CSH_TRY(csh_vec_mul_dev(f, a, b, ab, n, st));
    // round 6: ifft_in_to_out + distribute_powers + fft_out_to_in per vector with the two passes over the contiguous tiles in ONE launch
    // (ntt_run_pair_table: the tile stays in LDS across the hand-over; tune "ntt_pair" = 0: two launches as in rounds 3-5)
    for (uint64_t* v : {a, b}) CSH_TRY(ntt_run_pair_table(d, v, ncomp, ctab, st));           // :139-155
    CSH_TRY(ntt_run_pair_table(d, ab, 1, ctab, st));                                         // :163-174
    if (protocol == 1)
      CSH_TRY(rep3_local_mul_sub_dev(f, a, b, mask_ab, ab, h_out, n, st));                   // :182-190 (ab aliases h_out: element-wise)
    else
      CSH_TRY(vec_mul_sub_dev(f, a, b, ab, h_out, n, st));
    return CSH_OK;
  }
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

// ---- LibSnarkReduction tail (reduction.rs:255-342) --------------------------------------------------------------
// a, b: constraint evaluations (ncomp per entry), c: half-share evaluations of the C matrix (1 component), all natural
// order over the arkworks domain. h = coefficients (natural order) of (A*B - C) / Z: three coset evaluations, one
// local_mul_vec, (ab - c) * (g^n - 1)^-1, interpolation over the coset, un-shift. The reference bit-reverses before it
// multiplies by g^-i sequentially (:327-340); multiplying the bit-reversed vector by the bit-reversed power table and
// permuting afterwards gives the same field elements.

int csh_groth16_h_libsnark_dev(csh_domain_t dom, const uint64_t generator[4], int protocol, uint64_t* a, uint64_t* b, uint64_t* c,
                               const uint64_t* mask, uint64_t* h_out, void* stream) {
  CSH_REQUIRE(dom && generator && a && b && c && h_out, "NULL argument");
  CSH_REQUIRE(protocol == 0 || protocol == 1, "protocol must be 0 (plain/Shamir) or 1 (Rep3)");
  if (protocol == 1) CSH_TRY(require_rep3_masks(mask != nullptr, "groth16_h_libsnark"));
  CSH_TRY(ensure_device());
  const Domain* d = reinterpret_cast<const Domain*>(dom);
  const size_t n = domain_size_of(d);
  const csh_curve_t f = domain_curve_of(d);
  const uint32_t ncomp = protocol == 1 ? 2 : 1;
  uint32_t log_n = 0;
  while ((size_t(1) << log_n) < n) ++log_n;
  uint64_t v[4], neg_v[4], g_inv[4];
  const int crc = f == CSH_BN254       ? libsnark_consts<Bn254Fr>(generator, n, v, neg_v, g_inv)
                  : f == CSH_BLS12_377 ? libsnark_consts<Bls377Fr>(generator, n, v, neg_v, g_inv)
                                       : libsnark_consts<Bls381Fr>(generator, n, v, neg_v, g_inv);
  CSH_REQUIRE(crc == CSH_OK, "libsnark reduction: the coset generator lies in the domain (g^n == 1)");
  hipStream_t st = resolve_stream(stream);
  Arena& ar = arena_for((hipStream_t)((uintptr_t)st ^ 0x4));
  CSH_TRY(ar.reserve(Arena::padded(32 * n)));
  uint64_t* table = ar.take<uint64_t>(4 * n);

  CSH_TRY(ntt_coset_table(d, generator, table, st));                                         // reduction.rs:255
  for (uint64_t* x : {a, b}) {                                                               // :270-272, :283-285
    CSH_TRY(ntt_run(d, x, ncomp, true, st));
    CSH_TRY(csh_vec_mul_table_dev(f, x, table, n, ncomp, st));
    CSH_TRY(ntt_run(d, x, ncomp, false, st));
  }
  if (protocol == 1)
    CSH_TRY(csh_rep3_local_mul_vec_dev(f, a, b, mask, h_out, n, st));                        // :289
  else
    CSH_TRY(csh_vec_mul_dev(f, a, b, h_out, n, st));
  CSH_TRY(ntt_run(d, c, 1, true, st));                                                       // :299
  CSH_TRY(csh_vec_mul_table_dev(f, c, table, n, 1, st));                                     // :300-305
  CSH_TRY(ntt_run(d, c, 1, false, st));                                                      // :306
  {                                                                                          // :311-322: (ab - c) * (g^n - 1)^-1
    const uint64_t* ptrs[2] = {h_out, c};
    uint64_t coeffs[8];
    memcpy(coeffs, v, 32);
    memcpy(coeffs + 4, neg_v, 32);
    CSH_TRY(csh_lincomb_dev(f, ptrs, coeffs, 2, h_out, n, st));
  }
  CSH_TRY(ntt_run(d, h_out, 1, true, st));                                                   // :327
  CSH_TRY(ntt_coset_table(d, g_inv, table, st));                                             // :329-340 (see header note)
  CSH_TRY(csh_vec_mul_table_dev(f, h_out, table, n, 1, st));
  CSH_TRY(ntt_bit_reverse(f, h_out, log_n, 1, st));                                          // :328
  return CSH_OK;
}

int csh_groth16_h_libsnark(csh_domain_t dom, const uint64_t generator[4], int protocol, uint64_t* a, uint64_t* b, uint64_t* c,
                           const uint64_t* mask, uint64_t* h_out) {
  CSH_REQUIRE(dom && generator && a && b && c && h_out, "NULL argument");
  CSH_REQUIRE(protocol == 0 || protocol == 1, "protocol must be 0 (plain/Shamir) or 1 (Rep3)");
  const Domain* d = reinterpret_cast<const Domain*>(dom);
  const size_t n = domain_size_of(d);
  const size_t sb = 32 * n * (protocol == 1 ? 2 : 1), eb = 32 * n;
  HostStage h;
  CSH_TRY(h.begin(2 * Arena::padded(sb) + 3 * Arena::padded(eb)));
  uint64_t *da, *db, *dc, *dm = nullptr, *dh;
  CSH_TRY(h.up(da, a, sb));
  CSH_TRY(h.up(db, b, sb));
  CSH_TRY(h.up(dc, c, eb));
  if (mask) CSH_TRY(h.up(dm, mask, eb));
  CSH_TRY(h.up(dh, nullptr, eb));
  CSH_TRY(csh_groth16_h_libsnark_dev(dom, generator, protocol, da, db, dc, dm, dh, h.st));
  return h.down(h_out, dh, eb);
}

}  // extern "C"
