// Host-executed self-test hooks: the SAME templates the gfx950 kernels instantiate (field.hpp, curve.hpp,
// field29.hpp, msm_digits.hpp) run on the CPU so that `pytest -m "not gpu"` can check them against the
// oracle without a device. Test infrastructure; not declared in include/cosnarks_hip.h.
#define CSH_CHECK_BOUNDS 1  // host-side limb-bound contract checks in field29.hpp
#include <string.h>

#include <vector>

#include "common.hpp"
#include "curve.hpp"
#include "curve_lazy.hpp"
#include "chacha.hpp"
#include "field29.hpp"
#include "msm_digits.hpp"

using namespace csh;

namespace {

template <class F, bool IS_FP = true>
int field_op(int op, const void* a, const void* b, void* out) {
  F x, y, r;
  memcpy(&x, a, sizeof(F));
  if (b) memcpy(&y, b, sizeof(F)); else y = F::zero();
  switch (op) {
    case 0: r = F::add(x, y); break;
    case 1: r = F::sub(x, y); break;
    case 2: r = F::mul(x, y); break;
    case 3: r = F::inv(x); break;
    case 4: if constexpr (IS_FP) r = x.from_mont(); else return CSH_ERR_INVALID; break;
    case 5: if constexpr (IS_FP) r = x.to_mont(); else return CSH_ERR_INVALID; break;
    case 6: r = F::neg(x); break;
    case 7: r = F::sqr(x); break;
    default: return CSH_ERR_INVALID;
  }
  memcpy(out, &r, sizeof(F));
  return CSH_OK;
}

template <class Fq>
int curve_op(int op, const void* in1, const void* in2, uint32_t k, void* out) {
  XYZZ<Fq> acc, other;
  Affine<Fq> p;
  switch (op) {
    case 0:  // XYZZ += affine
      memcpy(&acc, in1, sizeof acc);
      memcpy(&p, in2, sizeof p);
      xyzz_madd(acc, p);
      memcpy(out, &acc, sizeof acc);
      return CSH_OK;
    case 1:  // XYZZ += XYZZ
      memcpy(&acc, in1, sizeof acc);
      memcpy(&other, in2, sizeof other);
      xyzz_add(acc, other);
      memcpy(out, &acc, sizeof acc);
      return CSH_OK;
    case 2:
      memcpy(&acc, in1, sizeof acc);
      acc = xyzz_dbl(acc);
      memcpy(out, &acc, sizeof acc);
      return CSH_OK;
    case 3:
      memcpy(&acc, in1, sizeof acc);
      acc = xyzz_mul_small(acc, k);
      memcpy(out, &acc, sizeof acc);
      return CSH_OK;
    case 4: {  // XYZZ -> affine
      memcpy(&acc, in1, sizeof acc);
      Affine<Fq> a = xyzz_to_affine(acc);
      memcpy(out, &a, sizeof a);
      return CSH_OK;
    }
    case 5: {  // affine -> XYZZ
      memcpy(&p, in1, sizeof p);
      acc = XYZZ<Fq>::from_affine(p);
      memcpy(out, &acc, sizeof acc);
      return CSH_OK;
    }
    default: return CSH_ERR_INVALID;
  }
}

// Lazy bucket accumulation on the host for any group: acc (empty) += sequence of `npts` affine points (negated
// where neg[i] != 0); out = XYZZ in arkworks words. Exercises exactly what k_msm_accum's lazy path does: storage
// repack, unpack, lazy_madd, export.
template <class L, class Fq>
int lazy_accumulate_t(const void* affine_pts, const uint8_t* neg, size_t npts, void* out_xyzz) {
  XYZZLazy<L> acc = XYZZLazy<L>::inf();
  const Affine<Fq>* pts = reinterpret_cast<const Affine<Fq>*>(affine_pts);
  for (size_t i = 0; i < npts; ++i) {
    Affine<Fq> p;
    memcpy(&p, pts + i, sizeof p);
    if (p.is_inf()) continue;
    Fq sx = L::repack_for_storage(p.x), sy = L::repack_for_storage(p.y);  // what Bases stores
    L x = L::unpack(sx), y = L::unpack(sy);
    if (neg && neg[i]) y = y.neg_unpacked();
    lazy_madd(acc, x, y);
  }
  XYZZ<Fq> r = lazy_to_xyzz<L, Fq>(acc);
  memcpy(out_xyzz, &r, sizeof r);
  return CSH_OK;
}

// Host run of the post-accumulation arithmetic (k_msm_merge / k_msm_reduce / k_msm_fold): groups of `group_len`
// points summed with lazy_madd, the group sums folded pairwise with lazy_add, the result times `weight`.
template <class L, class Fq>
int lazy_tree_t(const void* affine_pts, const uint8_t* neg, size_t npts, size_t group_len, uint32_t weight, void* out_xyzz) {
  const Affine<Fq>* pts = reinterpret_cast<const Affine<Fq>*>(affine_pts);
  std::vector<XYZZLazy<L>> parts;
  for (size_t g0 = 0; g0 < npts; g0 += group_len) {
    XYZZLazy<L> acc = XYZZLazy<L>::inf();
    for (size_t i = g0; i < npts && i < g0 + group_len; ++i) {
      Affine<Fq> p;
      memcpy(&p, pts + i, sizeof p);
      if (p.is_inf()) continue;
      L x = L::unpack(L::repack_for_storage(p.x)), y = L::unpack(L::repack_for_storage(p.y));
      if (neg && neg[i]) y = y.neg_unpacked();
      lazy_madd(acc, x, y);
    }
    parts.push_back(acc);
  }
  if (parts.empty()) parts.push_back(XYZZLazy<L>::inf());
  while (parts.size() > 1) {
    std::vector<XYZZLazy<L>> next;
    for (size_t i = 0; i + 1 < parts.size(); i += 2) {
      XYZZLazy<L> a = parts[i];
      if (i & 2) lazy_add_inl<L>(a, parts[i + 1]);  // both entry points
      else lazy_add_p<L>(&a, &parts[i + 1]);
      next.push_back(a);
    }
    if (parts.size() & 1) next.push_back(parts.back());
    parts.swap(next);
  }
  XYZZ<Fq> r = lazy_to_xyzz<L, Fq>(lazy_mul_small<L>(parts[0], weight));
  memcpy(out_xyzz, &r, sizeof r);
  return CSH_OK;
}

// Device-vs-host determinism check of the lazy bucket arithmetic: thread t accumulates the cyclic chain
// pts[(t + i) % n], i < len (sign from bit i of a hash) on the GPU; the host recomputes a sample of threads with the
// very same template code and compares the exported XYZZ words bit for bit.
template <class L, class Fq>
__host__ __device__ inline XYZZ<Fq> lazy_chain(const Affine<Fq>* pts, size_t n, size_t t, size_t len) {
  XYZZLazy<L> acc = XYZZLazy<L>::inf();
  len = 1 + (size_t)((t * 0x9E3779B1u) >> 8) % len;  // ragged chain lengths: lanes of a wave diverge, as in k_msm_accum
  for (size_t i = 0; i < len; ++i) {
    Affine<Fq> p = pts[(t + i) % n];
    if (p.is_inf()) continue;
    L x = L::unpack(p.x), y = L::unpack(p.y);
    if (((t * 2654435761u + i * 40503u) >> 7) & 1) y = y.neg_unpacked();
    lazy_madd(acc, x, y);
  }
  return lazy_to_xyzz<L, Fq>(acc);
}
template <class L, class Fq>
__global__ void k_lazy_chain(const Affine<Fq>* pts, size_t n, size_t len, size_t nthreads, XYZZ<Fq>* out) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t < nthreads) out[t] = lazy_chain<L, Fq>(pts, n, t, len);
}
template <class L, class Fq>
int lazy_chain_check_t(const void* affine_pts, size_t n, size_t len, size_t nthreads, size_t host_samples, int* mismatches) {
  std::vector<Affine<Fq>> st(n);
  const Affine<Fq>* in = reinterpret_cast<const Affine<Fq>*>(affine_pts);
  for (size_t i = 0; i < n; ++i) {
    st[i] = in[i];
    if (!st[i].is_inf()) st[i] = {L::repack_for_storage(in[i].x), L::repack_for_storage(in[i].y)};
  }
  Affine<Fq>* dpts;
  XYZZ<Fq>* dout;
  CSH_HIP(hipMalloc((void**)&dpts, n * sizeof(Affine<Fq>)));
  CSH_HIP(hipMalloc((void**)&dout, nthreads * sizeof(XYZZ<Fq>)));
  CSH_HIP(hipMemcpy(dpts, st.data(), n * sizeof(Affine<Fq>), hipMemcpyHostToDevice));
  hipLaunchKernelGGL((k_lazy_chain<L, Fq>), dim3((unsigned)((nthreads + 127) / 128)), dim3(128), 0, 0, dpts, n, len, nthreads, dout);
  std::vector<XYZZ<Fq>> got(nthreads);
  CSH_HIP(hipMemcpy(got.data(), dout, nthreads * sizeof(XYZZ<Fq>), hipMemcpyDeviceToHost));
  (void)hipFree(dpts);
  (void)hipFree(dout);
  int bad = 0;
  const size_t step = nthreads / host_samples ? nthreads / host_samples : 1;
  for (size_t t = 0; t < nthreads; t += step) {
    XYZZ<Fq> want = lazy_chain<L, Fq>(st.data(), n, t, len);
    if (memcmp(&want, &got[t], sizeof want) != 0) ++bad;
  }
  *mismatches = bad;
  return CSH_OK;
}

}  // namespace

// Host run of the lazy NTT butterfly arithmetic (ntt.hip k_ntt_pass_lazy): a, b, w arkworks-Montgomery elements of the
// scalar field; k times  acc = (acc +/- b*w).normalized()  with the twiddle in the R' domain, then canonical_wide().pack().
// Expected: a +/- k*b*w (still arkworks-Montgomery). Exercises the value drift a pass accumulates and its reduction.
template <class LZ, class F>
static int lazy_fr_chain_t(const uint64_t a[4], const uint64_t b[4], const uint64_t w[4], int k, int negative, uint64_t out[4]) {
  F fa, fb, fw;
  memcpy(&fa, a, 32);
  memcpy(&fb, b, 32);
  memcpy(&fw, w, 32);
  LZ acc = LZ::unpack(fa);
  const LZ lb = LZ::unpack(fb);
  const LZ lw = LZ::unpack(LZ::repack_for_storage(fw));
  for (int i = 0; i < k; ++i) {  // as the decimation-in-time stages do: the carry step only every second stage
    const LZ x = LZ::mul(lb, lw);
    acc = negative ? LZ::sub(acc, x) : LZ::add(acc, x);
    if (i & 1) acc = acc.normalized();
    const LZ as_operand = LZ::mul(acc, lw);  // the other role of a stage output: `v` of the next product (limb-bound asserted)
    (void)as_operand;
  }
  {  // sums feeding sums (decimation in frequency): doubling with fold_top must stay exact and in range
    LZ d = acc;
    for (int i = 0; i < 12; ++i) d = LZ::add(d, d).fold_top();
    LZ two12 = LZ::unpack(LZ::repack_for_storage(F::from_u64(4096)));
    const F ra = LZ::mul(acc, two12).canonical_wide().pack(), rb = d.canonical_wide().pack();
    if (memcmp(&ra, &rb, 32) != 0) return 2;
  }
  // one more product with a drifted operand (what the next stage does with it), divided out again by w^-1 is not
  // available here: multiply by the R'-domain one instead (value unchanged, reduced)
  const LZ one = LZ::one();
  const LZ red = LZ::mul(acc, one);
  const F r1 = acc.canonical_wide().pack(), r2 = red.canonical_wide().pack();
  if (memcmp(&r1, &r2, 32) != 0) return 1;
  memcpy(out, &r1, 32);
  return CSH_OK;
}

// Host run of the share-vector kernels' lazy products (vec_ops.hip): op 0: a*b; op 1: a*(c+d) + b*c + m (Rep3 local
// multiplication with la = a, lb = b, ra = c, rb = d, mask m). All arkworks-Montgomery in and out.
template <class LZ, class F>
static int lazy_vec_t(int op, const uint64_t* a, const uint64_t* b, const uint64_t* c, const uint64_t* d, const uint64_t* m, uint64_t* out) {
  F fa, fb, fc, fd, fm;
  memcpy(&fa, a, 32);
  memcpy(&fb, b, 32);
  memcpy(&fc, c, 32);
  memcpy(&fd, d, 32);
  memcpy(&fm, m, 32);
  F r;
  if (op == 0) {
    r = LZ::mul(LZ::unpack(fa), LZ::unpack(fb).times32()).canonical_wide().pack();
  } else {
    const LZ xa = LZ::unpack(fa), xb = LZ::unpack(fb), ya = LZ::unpack(fc), yb = LZ::unpack(fd);
    LZ t = LZ::reduce(LZ::mul_add_wide(xa, LZ::add(ya, yb).times32(), xb, ya.times32()));
    t = LZ::add(t, LZ::unpack(fm));
    r = t.canonical_wide().pack();
  }
  memcpy(out, &r, 32);
  return CSH_OK;
}

// Fp2 products of the signed lazy field on RAW limbs (no conversion in front): the caller chooses every limb, so the column bound the
// routines were derived for can be driven to its edge (all limbs at +/-(2^B + 8)) -- the values are arbitrary integers, only defined mod p.
// limbs: 4 elements x 2 components x NL int32; op 0: a b, 1: a^2, 2: a^2 - b, 3: a b - c d. out: the result as arkworks Montgomery limbs.
template <class L2, class F2>
static int fp2s_raw_t(int op, const int32_t* limbs, uint64_t* out) {
  using LF = decltype(L2().c0);
  constexpr int NL = LF::NL;
  L2 v[4];
  for (int e = 0; e < 4; ++e)
    for (int i = 0; i < NL; ++i) {
      v[e].c0.l[i] = limbs[(2 * e) * NL + i];
      v[e].c1.l[i] = limbs[(2 * e + 1) * NL + i];
    }
  const L2 r = op == 0 ? L2::mul(v[0], v[1]) : op == 1 ? L2::sqr(v[0]) : op == 2 ? L2::sqr_sub(v[0], v[1]) : L2::mul_sub(v[0], v[1], v[2], v[3]);
  const F2 f = r.to_fp();
  memcpy(out, &f, sizeof f);
  return CSH_OK;
}
extern "C" {

int csh_selftest_lazy_chain_dev(int curve, int group, const void* affine_pts, size_t n, size_t len, size_t nthreads, size_t host_samples,
                                int* mismatches) {
  CSH_TRY(ensure_device());
  if (curve == CSH_BN254 && group == CSH_G1) return lazy_chain_check_t<Fq29s, Bn254Fq>(affine_pts, n, len, nthreads, host_samples, mismatches);
  if (curve == CSH_BN254 && group == CSH_G2) return lazy_chain_check_t<Fq29s2, Bn254Fq2>(affine_pts, n, len, nthreads, host_samples, mismatches);
  if (curve == CSH_BLS12_381 && group == CSH_G1) return lazy_chain_check_t<Fq28s, Bls381Fq>(affine_pts, n, len, nthreads, host_samples, mismatches);
  if (curve == CSH_BLS12_381 && group == CSH_G2) return lazy_chain_check_t<Fq28s2, Bls381Fq2>(affine_pts, n, len, nthreads, host_samples, mismatches);
  if (curve == CSH_GRUMPKIN && group == CSH_G1) return lazy_chain_check_t<Fr29s, Bn254Fr>(affine_pts, n, len, nthreads, host_samples, mismatches);
  if (curve == CSH_BLS12_377 && group == CSH_G1) return lazy_chain_check_t<Fq28s377, Bls377Fq>(affine_pts, n, len, nthreads, host_samples, mismatches);
  if (curve == CSH_BLS12_377 && group == CSH_G2) return lazy_chain_check_t<Fq28s377x2, Bls377Fq2>(affine_pts, n, len, nthreads, host_samples, mismatches);
  return CSH_ERR_INVALID;
}


// field: 0 BN254 Fq, 1 BN254 Fr, 2 BLS12-381 Fq, 3 BLS12-381 Fr, 4 BLS12-377 Fq, 5 BLS12-377 Fr
int csh_selftest_field_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
  switch (field) {
    case 0: return field_op<Bn254Fq>(op, a, b, out);
    case 1: return field_op<Bn254Fr>(op, a, b, out);
    case 2: return field_op<Bls381Fq>(op, a, b, out);
    case 3: return field_op<Bls381Fr>(op, a, b, out);
    case 4: return field_op<Bls377Fq>(op, a, b, out);
    case 5: return field_op<Bls377Fr>(op, a, b, out);
    default: return CSH_ERR_INVALID;
  }
}

int csh_selftest_fp2_op(int curve, int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
  if (curve == CSH_BN254) return field_op<Bn254Fq2, false>(op, a, b, out);
  if (curve == CSH_BLS12_381) return field_op<Bls381Fq2, false>(op, a, b, out);
  if (curve == CSH_BLS12_377) return field_op<Bls377Fq2, false>(op, a, b, out);
  return CSH_ERR_INVALID;
}

int csh_selftest_curve_op(int curve, int group, int op, const void* in1, const void* in2, uint32_t k, void* out) {
  if (curve == CSH_BN254 && group == CSH_G1) return curve_op<Bn254Fq>(op, in1, in2, k, out);
  if (curve == CSH_BN254 && group == CSH_G2) return curve_op<Bn254Fq2>(op, in1, in2, k, out);
  if (curve == CSH_BLS12_381 && group == CSH_G1) return curve_op<Bls381Fq>(op, in1, in2, k, out);
  if (curve == CSH_BLS12_381 && group == CSH_G2) return curve_op<Bls381Fq2>(op, in1, in2, k, out);
  if (curve == CSH_GRUMPKIN && group == CSH_G1) return curve_op<Bn254Fr>(op, in1, in2, k, out);
  if (curve == CSH_BLS12_377 && group == CSH_G1) return curve_op<Bls377Fq>(op, in1, in2, k, out);
  if (curve == CSH_BLS12_377 && group == CSH_G2) return curve_op<Bls377Fq2>(op, in1, in2, k, out);
  return CSH_ERR_INVALID;
}

int csh_selftest_lazy_accumulate(int curve, int group, const void* affine_pts, const uint8_t* neg, size_t npts, void* out_xyzz) {
  if (curve == CSH_BN254 && group == CSH_G1) return lazy_accumulate_t<Fq29s, Bn254Fq>(affine_pts, neg, npts, out_xyzz);
  if (curve == CSH_BN254 && group == CSH_G2) return lazy_accumulate_t<Fq29s2, Bn254Fq2>(affine_pts, neg, npts, out_xyzz);
  if (curve == CSH_BLS12_381 && group == CSH_G1) return lazy_accumulate_t<Fq28s, Bls381Fq>(affine_pts, neg, npts, out_xyzz);
  if (curve == CSH_BLS12_381 && group == CSH_G2) return lazy_accumulate_t<Fq28s2, Bls381Fq2>(affine_pts, neg, npts, out_xyzz);
  if (curve == CSH_GRUMPKIN && group == CSH_G1) return lazy_accumulate_t<Fr29s, Bn254Fr>(affine_pts, neg, npts, out_xyzz);
  if (curve == CSH_BLS12_377 && group == CSH_G1) return lazy_accumulate_t<Fq28s377, Bls377Fq>(affine_pts, neg, npts, out_xyzz);
  if (curve == CSH_BLS12_377 && group == CSH_G2) return lazy_accumulate_t<Fq28s377x2, Bls377Fq2>(affine_pts, neg, npts, out_xyzz);
  return CSH_ERR_INVALID;
}

int csh_selftest_lazy_tree(int curve, int group, const void* affine_pts, const uint8_t* neg, size_t npts, size_t group_len, uint32_t weight,
                           void* out_xyzz) {
  if (group_len == 0) return CSH_ERR_INVALID;
  if (curve == CSH_BN254 && group == CSH_G1) return lazy_tree_t<Fq29s, Bn254Fq>(affine_pts, neg, npts, group_len, weight, out_xyzz);
  if (curve == CSH_BN254 && group == CSH_G2) return lazy_tree_t<Fq29s2, Bn254Fq2>(affine_pts, neg, npts, group_len, weight, out_xyzz);
  if (curve == CSH_BLS12_381 && group == CSH_G1) return lazy_tree_t<Fq28s, Bls381Fq>(affine_pts, neg, npts, group_len, weight, out_xyzz);
  if (curve == CSH_BLS12_381 && group == CSH_G2) return lazy_tree_t<Fq28s2, Bls381Fq2>(affine_pts, neg, npts, group_len, weight, out_xyzz);
  if (curve == CSH_GRUMPKIN && group == CSH_G1) return lazy_tree_t<Fr29s, Bn254Fr>(affine_pts, neg, npts, group_len, weight, out_xyzz);
  if (curve == CSH_BLS12_377 && group == CSH_G1) return lazy_tree_t<Fq28s377, Bls377Fq>(affine_pts, neg, npts, group_len, weight, out_xyzz);
  if (curve == CSH_BLS12_377 && group == CSH_G2) return lazy_tree_t<Fq28s377x2, Bls377Fq2>(affine_pts, neg, npts, group_len, weight, out_xyzz);
  return CSH_ERR_INVALID;
}

int csh_selftest_lazy_fr_chain(int field_of, const uint64_t a[4], const uint64_t b[4], const uint64_t w[4], int k, int negative, uint64_t out[4]) {
  if (field_of == CSH_BN254) return lazy_fr_chain_t<Fr29s, Bn254Fr>(a, b, w, k, negative, out);
  if (field_of == CSH_BLS12_381) return lazy_fr_chain_t<Bls381Fr29s, Bls381Fr>(a, b, w, k, negative, out);
  if (field_of == CSH_BLS12_377) return lazy_fr_chain_t<Bls377Fr29s, Bls377Fr>(a, b, w, k, negative, out);
  return CSH_ERR_INVALID;
}

int csh_selftest_lazy_vec(int field_of, int op, const uint64_t* a, const uint64_t* b, const uint64_t* c, const uint64_t* d, const uint64_t* m,
                          uint64_t* out) {
  if (field_of == CSH_BN254) return lazy_vec_t<Fr29s, Bn254Fr>(op, a, b, c, d, m, out);
  if (field_of == CSH_BLS12_381) return lazy_vec_t<Bls381Fr29s, Bls381Fr>(op, a, b, c, d, m, out);
  if (field_of == CSH_BLS12_377) return lazy_vec_t<Bls377Fr29s, Bls377Fr>(op, a, b, c, d, m, out);
  return CSH_ERR_INVALID;
}

// out = to_fp(mul(from_fp(a) (+/-) from_fp(b), from_fp(c))) for the signed lazy field: op 0: (a+b)*c, 1: (a-b)*c
int csh_selftest_lazys_op(int op, const uint64_t a[4], const uint64_t b[4], const uint64_t c[4], uint64_t out[4]) {
  using L = Fq29s;
  Bn254Fq fa, fb, fc;
  memcpy(&fa, a, 32);
  memcpy(&fb, b, 32);
  memcpy(&fc, c, 32);
  L la = L::from_fp(fa), lb = L::from_fp(fb), lc = L::from_fp(fc);
  L s = op == 0 ? L::add(la, lb) : L::sub(la, lb);
  Bn254Fq r = L::mul(s, lc).to_fp();
  memcpy(out, &r, 32);
  return s.is_zero() ? 1 : 0;   // also reports the zero test of (a +/- b)
}

int csh_selftest_fp2s_raw(int curve, int op, const int32_t* limbs, uint64_t* out) {
  if (curve == CSH_BN254) return fp2s_raw_t<Fq29s2, Bn254Fq2>(op, limbs, out);
  if (curve == CSH_BLS12_381) return fp2s_raw_t<Fq28s2, Bls381Fq2>(op, limbs, out);
  if (curve == CSH_BLS12_377) return fp2s_raw_t<Fq28s377x2, Bls377Fq2>(op, limbs, out);
  return CSH_ERR_INVALID;
}

// host execution of the on-device Rep3 mask generator (same template code as k_rep3_masks)
int csh_selftest_rep3_masks_host(int curve, const uint8_t seed1[32], uint64_t e1, const uint8_t seed2[32], uint64_t e2, uint64_t* out, size_t n) {
  uint32_t k1[8], k2[8];
  memcpy(k1, seed1, 32);
  memcpy(k2, seed2, 32);
  for (size_t i = 0; i < n; ++i) {
    if (curve == CSH_BN254) {
      Bn254Fr v = rep3_mask_element<Bn254Fr>(k1, k2, e1 + i, e2 + i);
      memcpy(out + 4 * i, &v, 32);
    } else {
      Bls381Fr v = rep3_mask_element<Bls381Fr>(k1, k2, e1 + i, e2 + i);
      memcpy(out + 4 * i, &v, 32);
    }
  }
  return CSH_OK;
}

// canonical scalar limbs -> signed digits (digits_out[w], w < *W_out)
int csh_selftest_digits(int curve, const uint64_t scalar[4], int c, int32_t* digits_out, int* W_out) {
  const int bits = curve == CSH_BN254 ? Bn254FrParams::BITS : curve == CSH_BLS12_377 ? Bls377FrParams::BITS : Bls381FrParams::BITS;
  const int W = windows_for(bits, c);
  uint32_t s[8];
  memcpy(s, scalar, 32);
  for (int w = 0; w < W; ++w) digits_out[w] = 0;
  for_each_digit<8>(s, c, W, [&](int w, uint32_t b, uint32_t neg) { digits_out[w] = neg ? -(int32_t)b : (int32_t)b; });
  *W_out = W;
  return CSH_OK;
}

}  // extern "C"
