/* liboracle -- plain-C CPU restatement of the co-snarks proof-generation hot path.
 *
 * ORACLE / TEST INFRASTRUCTURE ONLY: used by tests/ (large-size checker), __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg ("port"). Never linked into or called by the product (co-snarks_amd/).
 * Pinned against the pure-Python oracle (oracle/ Python modules), which is itself pinned on the reference's
 * test_vectors/Groth16 fixtures and the BN254 Fr product known-answer test (tests/tests/mpc/rep3.rs:286-345).
 *
 * Restated reference behaviour (paths relative to /root/reference):
 *   msm_unchecked / msm_bigint      external taceo-ark-algebra 0.1.0; call sites co-circom/co-groth16/src/groth16.rs:193-194,
 *                                   mpc/plain.rs:66-74, mpc/rep3.rs:124-132  -> ec_impl.h EC(msm)
 *   Domain::{ifft_in_to_out, fft_out_to_in}, bit_reverse   groth16/reduction.rs:38-60, 141-174 -> oc_ntt
 *   Rep3 local_mul_vec              mpc-core/src/protocols/rep3/arithmetic.rs:132-146, arithmetic/ops.rs:69-76
 *   Shamir/plain local_mul_vec      shamir/arithmetic.rs:73-79, mpc/plain.rs:83-89
 *   distribute_powers_and_mul_by_const   mpc/rep3.rs:95-106
 *   translate_primefield_repshare_vec    bridges/rep3_to_shamir.rs:43-62
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "constants.h"

/* ---- fields ------------------------------------------------------------------------------------------ */
#define FP_N 4
#define FP(x) bnq_##x
#define FP_P BN254_FQ_P
#define FP_R BN254_FQ_R
#define FP_R2 BN254_FQ_R2
#define FP_INV BN254_FQ_INV
#include "fp_impl.h"
#undef FP
#undef FP_P
#undef FP_R
#undef FP_R2
#undef FP_INV

#define FP(x) bnr_##x
#define FP_P BN254_FR_P
#define FP_R BN254_FR_R
#define FP_R2 BN254_FR_R2
#define FP_INV BN254_FR_INV
#include "fp_impl.h"
#undef FP
#undef FP_P
#undef FP_R
#undef FP_R2
#undef FP_INV

#define FP(x) blr_##x
#define FP_P BLS381_FR_P
#define FP_R BLS381_FR_R
#define FP_R2 BLS381_FR_R2
#define FP_INV BLS381_FR_INV
#include "fp_impl.h"
#undef FP
#undef FP_P
#undef FP_R
#undef FP_R2
#undef FP_INV

#define FP(x) b7r_##x
#define FP_P BLS377_FR_P
#define FP_R BLS377_FR_R
#define FP_R2 BLS377_FR_R2
#define FP_INV BLS377_FR_INV
#include "fp_impl.h"
#undef FP
#undef FP_P
#undef FP_R
#undef FP_R2
#undef FP_INV
#undef FP_N

#define FP_N 6
#define FP(x) blq_##x
#define FP_P BLS381_FQ_P
#define FP_R BLS381_FQ_R
#define FP_R2 BLS381_FQ_R2
#define FP_INV BLS381_FQ_INV
#include "fp_impl.h"
#undef FP
#undef FP_P
#undef FP_R
#undef FP_R2
#undef FP_INV

#define FP(x) b7q_##x
#define FP_P BLS377_FQ_P
#define FP_R BLS377_FQ_R
#define FP_R2 BLS377_FQ_R2
#define FP_INV BLS377_FQ_INV
#include "fp_impl.h"
#undef FP
#undef FP_P
#undef FP_R
#undef FP_R2
#undef FP_INV
#undef FP_N

#define F2(x) bnq2_##x
#define BF(x) bnq_##x
#include "fp2_impl.h"
#undef F2
#undef BF
#define F2(x) blq2_##x
#define BF(x) blq_##x
#include "fp2_impl.h"
#undef F2
#undef BF
#undef F2_NR
#define F2_NR 5   /* BLS12-377: Fq2 = Fq[u]/(u^2 + 5) */
#define F2(x) b7q2_##x
#define BF(x) b7q_##x
#include "fp2_impl.h"
#undef F2
#undef BF
#undef F2_NR

/* ---- groups -------------------------------------------------------------------------------------------- */
#define SF_N 4
#define EC(x) bn_g1_##x
#define FE(x) bnq_##x
#define SF(x) bnr_##x
#define SF_BITS 254
#include "ec_impl.h"
#undef EC
#undef FE
#define EC(x) bn_g2_##x
#define FE(x) bnq2_##x
#include "ec_impl.h"
#undef EC
#undef FE
#undef SF
#undef SF_BITS
#define EC(x) bl_g1_##x
#define FE(x) blq_##x
#define SF(x) blr_##x
#define SF_BITS 255
#include "ec_impl.h"
#undef EC
#undef FE
#define EC(x) bl_g2_##x
#define FE(x) blq2_##x
#include "ec_impl.h"
#undef EC
#undef FE
#undef SF
#undef SF_BITS
#define EC(x) b7_g1_##x
#define FE(x) b7q_##x
#define SF(x) b7r_##x
#define SF_BITS 253
#include "ec_impl.h"
#undef EC
#undef FE
#define EC(x) b7_g2_##x
#define FE(x) b7q2_##x
#include "ec_impl.h"
#undef EC
#undef FE
#undef SF
#undef SF_BITS

static int threads_or_default(int t) {
#ifdef _OPENMP
  return t > 0 ? t : omp_get_max_threads();
#else
  (void)t;
  return 1;
#endif
}

int oc_num_threads(void) { return threads_or_default(0); }

/* curve: 0 BN254, 1 BLS12-381, 3 BLS12-377 (the C ABI's csh_curve_t values); group: 0 G1, 1 G2. points: packed affine (all-zero = infinity);
 * out: packed affine. */
int oc_msm(int curve, int group, const uint64_t* points, const uint64_t* scalars, size_t n, int mont, int nthreads, uint64_t* out) {
  nthreads = threads_or_default(nthreads);
  if (curve == 0 && group == 0) { bn_g1_msm((bn_g1_aff*)out, (const bn_g1_aff*)points, scalars, n, mont, nthreads); return 0; }
  if (curve == 0 && group == 1) { bn_g2_msm((bn_g2_aff*)out, (const bn_g2_aff*)points, scalars, n, mont, nthreads); return 0; }
  if (curve == 1 && group == 0) { bl_g1_msm((bl_g1_aff*)out, (const bl_g1_aff*)points, scalars, n, mont, nthreads); return 0; }
  if (curve == 1 && group == 1) { bl_g2_msm((bl_g2_aff*)out, (const bl_g2_aff*)points, scalars, n, mont, nthreads); return 0; }
  if (curve == 3 && group == 0) { b7_g1_msm((b7_g1_aff*)out, (const b7_g1_aff*)points, scalars, n, mont, nthreads); return 0; }
  if (curve == 3 && group == 1) { b7_g2_msm((b7_g2_aff*)out, (const b7_g2_aff*)points, scalars, n, mont, nthreads); return 0; }
  return -1;
}

/* The tuned restatement (Booth digits, XYZZ buckets, thread-private bucket arrays): cpu_baseline "port" + full-size checker.
 * force_c = 0: cost model. stage_s (nullable, 6 doubles): [scalar conversion s, bucket tasks s, fold s, c, W, chunks]. */
int oc_msm_fast(int curve, int group, const uint64_t* points, const uint64_t* scalars, size_t n, int mont, int nthreads, int force_c, double* stage_s, uint64_t* out) {
  nthreads = threads_or_default(nthreads);
  if (curve == 0 && group == 0) { bn_g1_msm_fast((bn_g1_aff*)out, (const bn_g1_aff*)points, scalars, n, mont, nthreads, force_c, stage_s); return 0; }
  if (curve == 0 && group == 1) { bn_g2_msm_fast((bn_g2_aff*)out, (const bn_g2_aff*)points, scalars, n, mont, nthreads, force_c, stage_s); return 0; }
  if (curve == 1 && group == 0) { bl_g1_msm_fast((bl_g1_aff*)out, (const bl_g1_aff*)points, scalars, n, mont, nthreads, force_c, stage_s); return 0; }
  if (curve == 1 && group == 1) { bl_g2_msm_fast((bl_g2_aff*)out, (const bl_g2_aff*)points, scalars, n, mont, nthreads, force_c, stage_s); return 0; }
  if (curve == 3 && group == 0) { b7_g1_msm_fast((b7_g1_aff*)out, (const b7_g1_aff*)points, scalars, n, mont, nthreads, force_c, stage_s); return 0; }
  if (curve == 3 && group == 1) { b7_g2_msm_fast((b7_g2_aff*)out, (const b7_g2_aff*)points, scalars, n, mont, nthreads, force_c, stage_s); return 0; }
  return -1;
}

int oc_generate_bases(int curve, int group, uint64_t seed, size_t n, int nthreads, uint64_t* out) {
  nthreads = threads_or_default(nthreads);
  if (curve == 0 && group == 0) { bn_g1_gen_bases((bn_g1_aff*)out, (const bn_g1_aff*)BN254_G1_GEN, seed, n, nthreads); return 0; }
  if (curve == 0 && group == 1) { bn_g2_gen_bases((bn_g2_aff*)out, (const bn_g2_aff*)BN254_G2_GEN, seed, n, nthreads); return 0; }
  if (curve == 1 && group == 0) { bl_g1_gen_bases((bl_g1_aff*)out, (const bl_g1_aff*)BLS381_G1_GEN, seed, n, nthreads); return 0; }
  if (curve == 1 && group == 1) { bl_g2_gen_bases((bl_g2_aff*)out, (const bl_g2_aff*)BLS381_G2_GEN, seed, n, nthreads); return 0; }
  if (curve == 3 && group == 0) { b7_g1_gen_bases((b7_g1_aff*)out, (const b7_g1_aff*)BLS377_G1_GEN, seed, n, nthreads); return 0; }
  if (curve == 3 && group == 1) { b7_g2_gen_bases((b7_g2_aff*)out, (const b7_g2_aff*)BLS377_G2_GEN, seed, n, nthreads); return 0; }
  return -1;
}

/* out[i] = scalars[i] * G (canonical 4-limb scalars), G = the group's generator */
int oc_fixed_base_mul(int curve, int group, const uint64_t* scalars, size_t n, int nthreads, uint64_t* out) {
  nthreads = threads_or_default(nthreads);
  if (curve == 0 && group == 0) { bn_g1_mul_batch((bn_g1_aff*)out, (const bn_g1_aff*)BN254_G1_GEN, scalars, n, nthreads); return 0; }
  if (curve == 0 && group == 1) { bn_g2_mul_batch((bn_g2_aff*)out, (const bn_g2_aff*)BN254_G2_GEN, scalars, n, nthreads); return 0; }
  if (curve == 1 && group == 0) { bl_g1_mul_batch((bl_g1_aff*)out, (const bl_g1_aff*)BLS381_G1_GEN, scalars, n, nthreads); return 0; }
  if (curve == 1 && group == 1) { bl_g2_mul_batch((bl_g2_aff*)out, (const bl_g2_aff*)BLS381_G2_GEN, scalars, n, nthreads); return 0; }
  if (curve == 3 && group == 0) { b7_g1_mul_batch((b7_g1_aff*)out, (const b7_g1_aff*)BLS377_G1_GEN, scalars, n, nthreads); return 0; }
  if (curve == 3 && group == 1) { b7_g2_mul_batch((b7_g2_aff*)out, (const b7_g2_aff*)BLS377_G2_GEN, scalars, n, nthreads); return 0; }
  return -1;
}

int oc_generate_bases_wide(int curve, int group, uint64_t seed, size_t n, int nthreads, uint64_t* out) {
  nthreads = threads_or_default(nthreads);
  if (curve == 0 && group == 0) { bn_g1_gen_bases_wide((bn_g1_aff*)out, (const bn_g1_aff*)BN254_G1_GEN, seed, n, nthreads); return 0; }
  if (curve == 0 && group == 1) { bn_g2_gen_bases_wide((bn_g2_aff*)out, (const bn_g2_aff*)BN254_G2_GEN, seed, n, nthreads); return 0; }
  if (curve == 1 && group == 0) { bl_g1_gen_bases_wide((bl_g1_aff*)out, (const bl_g1_aff*)BLS381_G1_GEN, seed, n, nthreads); return 0; }
  if (curve == 1 && group == 1) { bl_g2_gen_bases_wide((bl_g2_aff*)out, (const bl_g2_aff*)BLS381_G2_GEN, seed, n, nthreads); return 0; }
  if (curve == 3 && group == 0) { b7_g1_gen_bases_wide((b7_g1_aff*)out, (const b7_g1_aff*)BLS377_G1_GEN, seed, n, nthreads); return 0; }
  if (curve == 3 && group == 1) { b7_g2_gen_bases_wide((b7_g2_aff*)out, (const b7_g2_aff*)BLS377_G2_GEN, seed, n, nthreads); return 0; }
  return -1;
}

int oc_generate_bases_progression(int curve, int group, uint64_t seed, size_t n, int nthreads, uint64_t* out) {
  nthreads = threads_or_default(nthreads);
  if (curve == 0 && group == 0) { bn_g1_gen_bases_progression((bn_g1_aff*)out, (const bn_g1_aff*)BN254_G1_GEN, seed, n, nthreads); return 0; }
  if (curve == 0 && group == 1) { bn_g2_gen_bases_progression((bn_g2_aff*)out, (const bn_g2_aff*)BN254_G2_GEN, seed, n, nthreads); return 0; }
  if (curve == 1 && group == 0) { bl_g1_gen_bases_progression((bl_g1_aff*)out, (const bl_g1_aff*)BLS381_G1_GEN, seed, n, nthreads); return 0; }
  if (curve == 1 && group == 1) { bl_g2_gen_bases_progression((bl_g2_aff*)out, (const bl_g2_aff*)BLS381_G2_GEN, seed, n, nthreads); return 0; }
  if (curve == 3 && group == 0) { b7_g1_gen_bases_progression((b7_g1_aff*)out, (const b7_g1_aff*)BLS377_G1_GEN, seed, n, nthreads); return 0; }
  if (curve == 3 && group == 1) { b7_g2_gen_bases_progression((b7_g2_aff*)out, (const b7_g2_aff*)BLS377_G2_GEN, seed, n, nthreads); return 0; }
  return -1;
}

/* BN254 G1 "hashed" points (family i): x = splitmix-derived field element, incremented until x^3 + 3 is
 * a square; y = rhs^((q+1)/4) (q = 3 mod 4), sign from a PRNG bit. Cofactor 1: every curve point is in the group. No
 * relation to the generator whatsoever. */
int oc_hash_points_bn254_g1(uint64_t seed, size_t n, int nthreads, uint64_t* out) {
  nthreads = threads_or_default(nthreads);
  uint64_t e[4];
  { /* (q + 1) / 4 */
    unsigned __int128 c = 1;
    uint64_t t[4];
    for (int i = 0; i < 4; i++) { c += BN254_FQ_P[i]; t[i] = (uint64_t)c; c >>= 64; }
    for (int i = 0; i < 4; i++) e[i] = (t[i] >> 2) | (i < 3 ? t[i + 1] << 62 : (uint64_t)c << 62);
  }
  bnq_t three; bnq_from_u64(&three, 3);
  bn_g1_aff* o = (bn_g1_aff*)out;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
  for (size_t i = 0; i < n; i++) {
    uint64_t k[4];
    for (int l = 0; l < 4; l++) {
      uint64_t x = seed + 4 * i + l + 0x9E3779B97F4A7C15ull;
      x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
      x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
      k[l] = x ^ (x >> 31);
    }
    const int sign = (int)(k[3] >> 63);
    k[3] &= (1ull << 61) - 1;                       /* < 2^253 < q: a canonical value; to_mont makes it a field element */
    bnq_t x, rhs, y, chk, one; memcpy(&x, k, 32); bnq_to_mont(&x, &x); bnq_set_one(&one);
    for (;;) {
      bnq_sqr(&rhs, &x); bnq_mul(&rhs, &rhs, &x); bnq_add(&rhs, &rhs, &three);
      bnq_pow(&y, &rhs, e, 4);
      bnq_sqr(&chk, &y);
      if (bnq_eq(&chk, &rhs)) break;
      bnq_add(&x, &x, &one);
    }
    if (sign) bnq_neg(&y, &y);
    o[i].x = x; o[i].y = y;
  }
  return 0;
}

/* ---- NTT / vector ops over Fr (both scalar fields are 4 limbs) ------------------------------------------- */
typedef struct { uint64_t l[4]; } fr_t;
typedef void (*fr_bin)(fr_t*, const fr_t*, const fr_t*);
static void bnr_mul_w(fr_t* r, const fr_t* a, const fr_t* b) { bnr_mul((bnr_t*)r, (const bnr_t*)a, (const bnr_t*)b); }
static void bnr_add_w(fr_t* r, const fr_t* a, const fr_t* b) { bnr_add((bnr_t*)r, (const bnr_t*)a, (const bnr_t*)b); }
static void bnr_sub_w(fr_t* r, const fr_t* a, const fr_t* b) { bnr_sub((bnr_t*)r, (const bnr_t*)a, (const bnr_t*)b); }
static void blr_mul_w(fr_t* r, const fr_t* a, const fr_t* b) { blr_mul((blr_t*)r, (const blr_t*)a, (const blr_t*)b); }
static void blr_add_w(fr_t* r, const fr_t* a, const fr_t* b) { blr_add((blr_t*)r, (const blr_t*)a, (const blr_t*)b); }
static void blr_sub_w(fr_t* r, const fr_t* a, const fr_t* b) { blr_sub((blr_t*)r, (const blr_t*)a, (const blr_t*)b); }

typedef struct { fr_bin mul, add, sub; fr_t one; } fr_ops;
static fr_ops ops_for(int curve) {
  fr_ops o;
  if (curve == 0) { o.mul = bnr_mul_w; o.add = bnr_add_w; o.sub = bnr_sub_w; memcpy(&o.one, BN254_FR_R, 32); }
  else { o.mul = blr_mul_w; o.add = blr_add_w; o.sub = blr_sub_w; memcpy(&o.one, BLS381_FR_R, 32); }
  return o;
}
static void fr_inv(int curve, fr_t* r, const fr_t* a) {
  if (curve == 0) bnr_inv((bnr_t*)r, (const bnr_t*)a); else blr_inv((blr_t*)r, (const blr_t*)a);
}
static void fr_from_u64(int curve, fr_t* r, uint64_t v) {
  if (curve == 0) bnr_from_u64((bnr_t*)r, v); else blr_from_u64((blr_t*)r, v);
}

static size_t bitrev(size_t i, int logn) {
  size_t r = 0;
  for (int k = 0; k < logn; k++) { r = (r << 1) | (i & 1); i >>= 1; }
  return r;
}

int oc_bit_reverse(uint64_t* data, int logn, int ncomp) {
  size_t n = (size_t)1 << logn;
  fr_t* v = (fr_t*)data;
  for (size_t i = 0; i < n; i++) {
    size_t j = bitrev(i, logn);
    if (i < j) for (int c = 0; c < ncomp; c++) { fr_t t = v[i * ncomp + c]; v[i * ncomp + c] = v[j * ncomp + c]; v[j * ncomp + c] = t; }
  }
  return 0;
}

/* dif = 0: fft_out_to_in (bit-reversed in -> natural out, DIT).  dif = 1: ifft_in_to_out (natural in ->
 * bit-reversed out, scaled by 1/n; decimation in frequency with the inverse root). gen: Montgomery. */
int oc_ntt(int curve, uint64_t* data, int logn, const uint64_t* gen, int ncomp, int dif, int nthreads) {
  nthreads = threads_or_default(nthreads);
  fr_ops o = ops_for(curve);
  size_t n = (size_t)1 << logn;
  fr_t* v = (fr_t*)data;
  if (logn == 0) return 0;
  fr_t w; memcpy(&w, gen, 32);
  if (dif) fr_inv(curve, &w, &w);
  size_t half = n / 2;
  fr_t* tw = (fr_t*)malloc(sizeof(fr_t) * half);
  tw[0] = o.one;
  for (size_t i = 1; i < half; i++) o.mul(&tw[i], &tw[i - 1], &w);
  for (int st = 0; st < logn; st++) {
    int s = dif ? logn - 1 - st : st;
    size_t m = (size_t)1 << s;
    size_t step = half >> s;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (size_t b = 0; b < half; b++) {
      size_t j = b & (m - 1);
      size_t i0 = ((b >> s) << (s + 1)) | j;
      size_t i1 = i0 + m;
      const fr_t* t = &tw[j * step];
      for (int c = 0; c < ncomp; c++) {
        fr_t* u = &v[i0 * ncomp + c];
        fr_t* x = &v[i1 * ncomp + c];
        if (dif) {
          fr_t s2, d;
          o.add(&s2, u, x); o.sub(&d, u, x); o.mul(&d, &d, t);
          *u = s2; *x = d;
        } else {
          fr_t y, s2, d;
          o.mul(&y, x, t); o.add(&s2, u, &y); o.sub(&d, u, &y);
          *u = s2; *x = d;
        }
      }
    }
  }
  if (dif) {
    fr_t ninv; fr_from_u64(curve, &ninv, (uint64_t)n); fr_inv(curve, &ninv, &ninv);
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (size_t i = 0; i < n * (size_t)ncomp; i++) o.mul(&v[i], &v[i], &ninv);
  }
  free(tw);
  return 0;
}

int oc_vec_mul(int curve, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, int nthreads) {
  nthreads = threads_or_default(nthreads);
  fr_ops o = ops_for(curve);
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (size_t i = 0; i < n; i++) o.mul((fr_t*)out + i, (const fr_t*)a + i, (const fr_t*)b + i);
  return 0;
}

int oc_rep3_local_mul_vec(int curve, const uint64_t* lhs, const uint64_t* rhs, const uint64_t* mask, uint64_t* out, size_t n, int nthreads) {
  nthreads = threads_or_default(nthreads);
  fr_ops o = ops_for(curve);
  const fr_t *l = (const fr_t*)lhs, *r = (const fr_t*)rhs, *m = (const fr_t*)mask;
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (size_t i = 0; i < n; i++) {
    fr_t t0, t1, t2;                         /* a*a' + a*b' + b*a' (ops.rs:69-76), three products as written */
    o.mul(&t0, &l[2 * i], &r[2 * i]);
    o.mul(&t1, &l[2 * i], &r[2 * i + 1]);
    o.mul(&t2, &l[2 * i + 1], &r[2 * i]);
    o.add(&t0, &t0, &t1); o.add(&t0, &t0, &t2);
    if (m) o.add(&t0, &t0, &m[i]);
    ((fr_t*)out)[i] = t0;
  }
  return 0;
}

int oc_vec_mul_table(int curve, uint64_t* v, const uint64_t* table, size_t n, int ncomp, int nthreads) {
  nthreads = threads_or_default(nthreads);
  fr_ops o = ops_for(curve);
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (size_t i = 0; i < n; i++)
    for (int c = 0; c < ncomp; c++) o.mul((fr_t*)v + i * ncomp + c, (fr_t*)v + i * ncomp + c, (const fr_t*)table + i);
  return 0;
}

int oc_vec_sub(int curve, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n_elems, int nthreads) {
  nthreads = threads_or_default(nthreads);
  fr_ops o = ops_for(curve);
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (size_t i = 0; i < n_elems; i++) o.sub((fr_t*)out + i, (const fr_t*)a + i, (const fr_t*)b + i);
  return 0;
}

/* Horner: out = sum_i coeffs[i * stride] * x^i (i < n), no NTT code involved (spot checks of full-size transforms) */
int oc_eval_poly(int curve, const uint64_t* coeffs, size_t n, size_t stride, const uint64_t* x, uint64_t* out) {
  fr_ops o = ops_for(curve);
  fr_t acc; memset(&acc, 0, sizeof acc);
  for (size_t i = n; i-- > 0;) { o.mul(&acc, &acc, (const fr_t*)x); o.add(&acc, &acc, (const fr_t*)coeffs + i * stride); }
  *(fr_t*)out = acc;
  return 0;
}

int oc_vec_add(int curve, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n_elems, int nthreads) {
  nthreads = threads_or_default(nthreads);
  fr_ops o = ops_for(curve);
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (size_t i = 0; i < n_elems; i++) o.add((fr_t*)out + i, (const fr_t*)a + i, (const fr_t*)b + i);
  return 0;
}

/* out[i] = sum_k coeffs[k] * shares[k][i] (Shamir reconstruct / open_vec, shamir.rs:483-491; Rep3 combine with coeffs = 1) */
int oc_lincomb(int curve, const uint64_t* const* shares, const uint64_t* coeffs, size_t k, uint64_t* out, size_t n, int nthreads) {
  nthreads = threads_or_default(nthreads);
  fr_ops o = ops_for(curve);
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (size_t i = 0; i < n; i++) {
    fr_t acc, t; memset(&acc, 0, sizeof acc);
    for (size_t j = 0; j < k; j++) { o.mul(&t, (const fr_t*)shares[j] + i, (const fr_t*)coeffs + j); o.add(&acc, &acc, &t); }
    ((fr_t*)out)[i] = acc;
  }
  return 0;
}

int oc_rep3_to_shamir_vec(int curve, const uint64_t* in, const uint64_t* x, const uint64_t* y, uint64_t* out, size_t n, int nthreads) {
  nthreads = threads_or_default(nthreads);
  fr_ops o = ops_for(curve);
  const fr_t* s = (const fr_t*)in;
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (size_t i = 0; i < n; i++) {
    fr_t t0, t1;
    o.mul(&t0, &s[2 * i], (const fr_t*)x);
    o.mul(&t1, &s[2 * i + 1], (const fr_t*)y);
    o.add((fr_t*)out + i, &t0, &t1);
  }
  return 0;
}

/* out[bitrev(i)] = shift^i (reduction.rs:45-60) */
int oc_coset_table(int curve, const uint64_t* shift, int logn, uint64_t* out) {
  fr_ops o = ops_for(curve);
  size_t n = (size_t)1 << logn;
  fr_t cur = o.one;
  for (size_t i = 0; i < n; i++) {
    ((fr_t*)out)[bitrev(i, logn)] = cur;
    o.mul(&cur, &cur, (const fr_t*)shift);
  }
  return 0;
}
