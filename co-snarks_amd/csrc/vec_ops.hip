// Element-wise secret-shared field arithmetic (Rep3 / Shamir / plain) on gfx950.
// HBM-bound streaming kernels: one 32-byte field element per lane per step, 2 x 16-byte accesses,
// grid-stride over <= 65536 workgroups of 256 threads. See DESIGN.md section "share-vector kernels".
// Products run in the signed lazy 9 x 29-bit field (field29.hpp): operands are re-sliced as they are, one of them is
// scaled by 2^5 = R'/2^256 so the lazy Montgomery product is the arkworks one; 162 multiply-adds instead of a 32-bit CIOS
// product with its carry chains, which keeps these kernels on the HBM side of the roofline.
#include "common.hpp"
#include "field.hpp"
#include "field29.hpp"
#include "chacha.hpp"
#include <stdlib.h>
#include <string.h>

namespace csh {

constexpr int VB = 256;
static int vec_grid(size_t n) {
  int mb = tune().vec_max_blocks.load(std::memory_order_relaxed);  // default 65536: up to one element per lane at 2^24, measured 8-10 % faster than 4096 blocks + grid stride
  if (mb <= 0) mb = 65536;
  return grid_for(n, VB, mb);
}

template <class F>
__global__ __launch_bounds__(VB) void k_vec_mul(const F* __restrict__ a, const F* __restrict__ b, F* out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) {
    using LZ = typename LazyOf<F>::type;
    out[i] = LZ::mul(LZ::unpack(a[i]), LZ::unpack(b[i]).times32()).canonical_wide().pack();
  }
}
// out = a * b - c in one sweep (h = a b - c, reduction.rs:176-190): the difference is limb-wise on the reduced product
template <class F>
__global__ __launch_bounds__(VB) void k_vec_mul_sub(const F* __restrict__ a, const F* __restrict__ b, const F* c /* may alias out */, F* out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) {
    using LZ = typename LazyOf<F>::type;
    out[i] = LZ::sub(LZ::mul(LZ::unpack(a[i]), LZ::unpack(b[i]).times32()), LZ::unpack(c[i])).canonical_wide().pack();
  }
}

template <class F, bool SUB>
__global__ __launch_bounds__(VB) void k_vec_addsub(const F* __restrict__ a, const F* __restrict__ b, F* out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) {
    out[i] = SUB ? F::sub(a[i], b[i]) : F::add(a[i], b[i]);
  }
}

// v[i*ncomp + c] *= table[i]
template <class F>
__global__ __launch_bounds__(VB) void k_vec_mul_table(F* v, const F* __restrict__ table, size_t n_elems, uint32_t ncomp) {
  for (size_t e = blockIdx.x * (size_t)VB + threadIdx.x; e < n_elems; e += (size_t)gridDim.x * VB) {
    size_t i = ncomp == 1 ? e : e / ncomp;
    using LZ = typename LazyOf<F>::type;
    v[e] = LZ::mul(LZ::unpack(v[e]), LZ::unpack(table[i]).times32()).canonical_wide().pack();
  }
}

// Rep3 local multiplication (mpc-core rep3/arithmetic/ops.rs:69-76) + mask
template <class F>
__global__ __launch_bounds__(VB) void k_rep3_local_mul(const F* __restrict__ lhs, const F* __restrict__ rhs,
                                                       const F* __restrict__ mask, const F* sub /* may alias out */, F* out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) {
    F la = lhs[2 * i], lb = lhs[2 * i + 1];
    F ra = rhs[2 * i], rb = rhs[2 * i + 1];
    // a*a' + a*b' + b*a' = la*(ra+rb) + lb*ra  (2 multiplications instead of 3; same field element)
    //   both products accumulate double-width before ONE reduction
    using LZ = typename LazyOf<F>::type;
    const LZ xa = LZ::unpack(la), xb = LZ::unpack(lb), ya = LZ::unpack(ra), yb = LZ::unpack(rb);
    LZ r = LZ::reduce(LZ::mul_add_wide(xa, LZ::add(ya, yb).times32(), xb, ya.times32()));
    if (mask) r = LZ::add(r, LZ::unpack(mask[i]));
    if (sub) r = LZ::sub(r, LZ::unpack(sub[i]));
    out[i] = r.canonical_wide().pack();
  }
}

template <class F>
__global__ __launch_bounds__(VB) void k_rep3_to_shamir(const F* __restrict__ in, F x, F y, F* out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) {
    using LZ = typename LazyOf<F>::type;
    const LZ lx = LZ::unpack(x).times32(), ly = LZ::unpack(y).times32();
    out[i] = LZ::reduce(LZ::mul_add_wide(LZ::unpack(in[2 * i]), lx, LZ::unpack(in[2 * i + 1]), ly)).canonical_wide().pack();
  }
}

// Rep3 correlated masks on the device: out[i] = from_be(stream1 chunk e1+i) - from_be(stream2 chunk e2+i)
struct ChaChaKeys {
  uint32_t k1[8], k2[8];
};
template <class F>
__global__ __launch_bounds__(VB) void k_rep3_masks(ChaChaKeys keys, uint64_t e1, uint64_t e2, F* out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) {
    out[i] = rep3_mask_element<F>(keys.k1, keys.k2, e1 + i, e2 + i);
  }
}

constexpr int MAX_LINCOMB = 16;
template <class F>
struct LincombArgs {
  const F* shares[MAX_LINCOMB];
  F coeffs[MAX_LINCOMB];
  int k;
  int unit;  // all coefficients are 1 (Rep3 combine/open): plain sum
};

template <class F>
__global__ __launch_bounds__(VB) void k_lincomb(LincombArgs<F> args, F* out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)VB + threadIdx.x; i < n; i += (size_t)gridDim.x * VB) {
    F acc = F::zero();
    for (int k = 0; k < args.k; ++k) {
      F s = args.shares[k][i];
      acc = F::add(acc, args.unit ? s : F::mul(s, args.coeffs[k]));
    }
    out[i] = acc;
  }
}

// ---- typed launchers ---------------------------------------------------------------------------
template <class F>
static int vec_mul_t(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, hipStream_t st) {
  if (n == 0) return CSH_OK;
  hipLaunchKernelGGL(k_vec_mul<F>, dim3(vec_grid(n)), dim3(VB), 0, st, (const F*)a, (const F*)b, (F*)out, n);
  CSH_HIP(hipGetLastError());
  return CSH_OK;
}
template <class F>
static int vec_mul_sub_t(const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t* out, size_t n, hipStream_t st) {
  if (n == 0) return CSH_OK;
  hipLaunchKernelGGL(k_vec_mul_sub<F>, dim3(vec_grid(n)), dim3(VB), 0, st, (const F*)a, (const F*)b, (const F*)c, (F*)out, n);
  CSH_HIP(hipGetLastError());
  return CSH_OK;
}
template <class F>
static int vec_addsub_t(bool sub, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n_elems, hipStream_t st) {
  if (n_elems == 0) return CSH_OK;
  if (sub)
    hipLaunchKernelGGL((k_vec_addsub<F, true>), dim3(vec_grid(n_elems)), dim3(VB), 0, st, (const F*)a, (const F*)b, (F*)out, n_elems);
  else
    hipLaunchKernelGGL((k_vec_addsub<F, false>), dim3(vec_grid(n_elems)), dim3(VB), 0, st, (const F*)a, (const F*)b, (F*)out, n_elems);
  CSH_HIP(hipGetLastError());
  return CSH_OK;
}
template <class F>
static int vec_mul_table_t(uint64_t* v, const uint64_t* table, size_t n, uint32_t ncomp, hipStream_t st) {
  if (n == 0) return CSH_OK;
  hipLaunchKernelGGL(k_vec_mul_table<F>, dim3(vec_grid(n * ncomp)), dim3(VB), 0, st, (F*)v, (const F*)table, n * ncomp, ncomp);
  CSH_HIP(hipGetLastError());
  return CSH_OK;
}
template <class F>
static int rep3_local_mul_t(const uint64_t* l, const uint64_t* r, const uint64_t* m, uint64_t* out, size_t n, hipStream_t st, const uint64_t* sub = nullptr) {
  if (n == 0) return CSH_OK;
  hipLaunchKernelGGL(k_rep3_local_mul<F>, dim3(vec_grid(n)), dim3(VB), 0, st, (const F*)l, (const F*)r, (const F*)m, (const F*)sub, (F*)out, n);
  CSH_HIP(hipGetLastError());
  return CSH_OK;
}
template <class F>
static int rep3_to_shamir_t(const uint64_t* in, const uint64_t* x, const uint64_t* y, uint64_t* out, size_t n, hipStream_t st) {
  if (n == 0) return CSH_OK;
  F fx, fy;
  memcpy(&fx, x, sizeof(F));
  memcpy(&fy, y, sizeof(F));
  hipLaunchKernelGGL(k_rep3_to_shamir<F>, dim3(vec_grid(n)), dim3(VB), 0, st, (const F*)in, fx, fy, (F*)out, n);
  CSH_HIP(hipGetLastError());
  return CSH_OK;
}
template <class F>
static int rep3_masks_t(const uint8_t* seed1, uint64_t e1, const uint8_t* seed2, uint64_t e2, uint64_t* out, size_t n, hipStream_t st) {
  if (n == 0) return CSH_OK;
  ChaChaKeys keys;
  memcpy(keys.k1, seed1, 32);
  memcpy(keys.k2, seed2, 32);
  hipLaunchKernelGGL(k_rep3_masks<F>, dim3(vec_grid(n)), dim3(VB), 0, st, keys, e1, e2, (F*)out, n);
  CSH_HIP(hipGetLastError());
  return CSH_OK;
}
template <class F>
static int lincomb_t(const uint64_t* const* shares, const uint64_t* coeffs, size_t k, uint64_t* out, size_t n, hipStream_t st) {
  if (n == 0) return CSH_OK;
  LincombArgs<F> args;
  args.k = (int)k;
  F one = F::one();
  int unit = 1;
  for (size_t j = 0; j < k; ++j) {
    args.shares[j] = (const F*)shares[j];
    memcpy(&args.coeffs[j], coeffs + 4 * j, sizeof(F));
    if (!(args.coeffs[j] == one)) unit = 0;
  }
  args.unit = unit;
  hipLaunchKernelGGL(k_lincomb<F>, dim3(vec_grid(n)), dim3(VB), 0, st, args, (F*)out, n);
  CSH_HIP(hipGetLastError());
  return CSH_OK;
}

}  // namespace csh

using namespace csh;

#define FR_DISPATCH(field_of, CALL)                                  \
  switch (field_of) {                                                \
    case CSH_BN254: { using F = Bn254Fr; return CALL; }              \
    case CSH_BLS12_381: { using F = Bls381Fr; return CALL; }         \
    case CSH_BLS12_377: { using F = Bls377Fr; return CALL; }         \
    default: set_error("unknown curve %d", (int)(field_of)); return CSH_ERR_INVALID; \
  }

namespace csh {
int vec_mul_sub_dev(csh_curve_t f, const uint64_t* a, const uint64_t* b, const uint64_t* c, uint64_t* out, size_t n, hipStream_t st) {
  FR_DISPATCH(f, vec_mul_sub_t<F>(a, b, c, out, n, st));
}
int rep3_local_mul_sub_dev(csh_curve_t f, const uint64_t* a, const uint64_t* b, const uint64_t* mask, const uint64_t* c, uint64_t* out, size_t n, hipStream_t st) {
  FR_DISPATCH(f, rep3_local_mul_t<F>(a, b, mask, out, n, st, c));
}
}  // namespace csh

extern "C" {

int csh_vec_mul_dev(csh_curve_t f, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream) {
  CSH_TRY(ensure_device());
  hipStream_t st = resolve_stream(stream);
  FR_DISPATCH(f, vec_mul_t<F>(a, b, out, n, st));
}
int csh_vec_add_dev(csh_curve_t f, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, uint32_t ncomp, void* stream) {
  CSH_TRY(ensure_device());
  hipStream_t st = resolve_stream(stream);
  FR_DISPATCH(f, vec_addsub_t<F>(false, a, b, out, n * ncomp, st));
}
int csh_vec_sub_dev(csh_curve_t f, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, uint32_t ncomp, void* stream) {
  CSH_TRY(ensure_device());
  hipStream_t st = resolve_stream(stream);
  FR_DISPATCH(f, vec_addsub_t<F>(true, a, b, out, n * ncomp, st));
}
int csh_vec_mul_table_dev(csh_curve_t f, uint64_t* v, const uint64_t* table, size_t n, uint32_t ncomp, void* stream) {
  CSH_REQUIRE(ncomp >= 1 && ncomp <= 2, "ncomp must be 1 or 2");
  CSH_TRY(ensure_device());
  hipStream_t st = resolve_stream(stream);
  FR_DISPATCH(f, vec_mul_table_t<F>(v, table, n, ncomp, st));
}
int csh_rep3_local_mul_vec_dev(csh_curve_t f, const uint64_t* l, const uint64_t* r, const uint64_t* m, uint64_t* out, size_t n, void* stream) {
  CSH_TRY(require_rep3_masks(m != nullptr || n == 0, "rep3_local_mul_vec"));
  CSH_TRY(ensure_device());
  hipStream_t st = resolve_stream(stream);
  FR_DISPATCH(f, rep3_local_mul_t<F>(l, r, m, out, n, st));
}
int csh_rep3_to_shamir_vec_dev(csh_curve_t f, const uint64_t* in, const uint64_t x[4], const uint64_t y[4], uint64_t* out, size_t n, void* stream) {
  CSH_REQUIRE(x && y, "translation points are NULL");
  CSH_TRY(ensure_device());
  hipStream_t st = resolve_stream(stream);
  FR_DISPATCH(f, rep3_to_shamir_t<F>(in, x, y, out, n, st));
}
int csh_rep3_masks_dev(csh_curve_t f, const uint8_t seed1[32], uint64_t elem_offset1, const uint8_t seed2[32], uint64_t elem_offset2,
                       uint64_t* out, size_t n, void* stream) {
  CSH_REQUIRE(seed1 && seed2 && (out || n == 0), "rep3_masks: NULL argument");
  CSH_TRY(ensure_device());
  hipStream_t st = resolve_stream(stream);
  FR_DISPATCH(f, rep3_masks_t<F>(seed1, elem_offset1, seed2, elem_offset2, out, n, st));
}
int csh_lincomb_dev(csh_curve_t f, const uint64_t* const* shares, const uint64_t* coeffs, size_t k, uint64_t* out, size_t n, void* stream) {
  CSH_REQUIRE(k >= 1 && k <= (size_t)MAX_LINCOMB, "lincomb: 1 <= k <= 16");
  CSH_REQUIRE(shares && coeffs, "lincomb: NULL argument");
  CSH_TRY(ensure_device());
  hipStream_t st = resolve_stream(stream);
  FR_DISPATCH(f, lincomb_t<F>(shares, coeffs, k, out, n, st));
}

// ---- host-pointer convenience wrappers: H2D, compute, D2H on the thread's stream (HostStage) -------
int csh_vec_mul(csh_curve_t f, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  HostStage h;
  size_t eb = 32 * n;
  CSH_TRY(h.begin(3 * Arena::padded(eb)));
  uint64_t *da, *db, *dout;
  CSH_TRY(h.up(da, a, eb));
  CSH_TRY(h.up(db, b, eb));
  CSH_TRY(h.up(dout, nullptr, eb));
  CSH_TRY(csh_vec_mul_dev(f, da, db, dout, n, h.st));
  return h.down(out, dout, eb);
}
static int addsub_host(bool sub, csh_curve_t f, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, uint32_t ncomp) {
  HostStage h;
  size_t eb = 32 * n * ncomp;
  CSH_TRY(h.begin(3 * Arena::padded(eb)));
  uint64_t *da, *db, *dout;
  CSH_TRY(h.up(da, a, eb));
  CSH_TRY(h.up(db, b, eb));
  CSH_TRY(h.up(dout, nullptr, eb));
  CSH_TRY(sub ? csh_vec_sub_dev(f, da, db, dout, n, ncomp, h.st) : csh_vec_add_dev(f, da, db, dout, n, ncomp, h.st));
  return h.down(out, dout, eb);
}
int csh_vec_add(csh_curve_t f, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, uint32_t ncomp) {
  return addsub_host(false, f, a, b, out, n, ncomp);
}
int csh_vec_sub(csh_curve_t f, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, uint32_t ncomp) {
  return addsub_host(true, f, a, b, out, n, ncomp);
}
int csh_vec_mul_table(csh_curve_t f, uint64_t* v, const uint64_t* table, size_t n, uint32_t ncomp) {
  HostStage h;
  size_t vb = 32 * n * ncomp, tb = 32 * n;
  CSH_TRY(h.begin(Arena::padded(vb) + Arena::padded(tb)));
  uint64_t *dv, *dt;
  CSH_TRY(h.up(dv, v, vb));
  CSH_TRY(h.up(dt, table, tb));
  CSH_TRY(csh_vec_mul_table_dev(f, dv, dt, n, ncomp, h.st));
  return h.down(v, dv, vb);
}
int csh_rep3_local_mul_vec(csh_curve_t f, const uint64_t* l, const uint64_t* r, const uint64_t* m, uint64_t* out, size_t n) {
  HostStage h;
  size_t sb = 64 * n, eb = 32 * n;
  CSH_TRY(h.begin(2 * Arena::padded(sb) + 2 * Arena::padded(eb)));
  uint64_t *dl, *dr, *dm = nullptr, *dout;
  CSH_TRY(h.up(dl, l, sb));
  CSH_TRY(h.up(dr, r, sb));
  if (m) CSH_TRY(h.up(dm, m, eb));
  CSH_TRY(h.up(dout, nullptr, eb));
  CSH_TRY(csh_rep3_local_mul_vec_dev(f, dl, dr, dm, dout, n, h.st));
  return h.down(out, dout, eb);
}
int csh_rep3_to_shamir_vec(csh_curve_t f, const uint64_t* in, const uint64_t x[4], const uint64_t y[4], uint64_t* out, size_t n) {
  HostStage h;
  size_t sb = 64 * n, eb = 32 * n;
  CSH_TRY(h.begin(Arena::padded(sb) + Arena::padded(eb)));
  uint64_t *din, *dout;
  CSH_TRY(h.up(din, in, sb));
  CSH_TRY(h.up(dout, nullptr, eb));
  CSH_TRY(csh_rep3_to_shamir_vec_dev(f, din, x, y, dout, n, h.st));
  return h.down(out, dout, eb);
}
int csh_rep3_masks(csh_curve_t f, const uint8_t seed1[32], uint64_t elem_offset1, const uint8_t seed2[32], uint64_t elem_offset2,
                   uint64_t* out, size_t n) {
  HostStage h;
  CSH_TRY(h.begin(Arena::padded(32 * n)));
  uint64_t* dout;
  CSH_TRY(h.up(dout, nullptr, 32 * n));
  CSH_TRY(csh_rep3_masks_dev(f, seed1, elem_offset1, seed2, elem_offset2, dout, n, h.st));
  return h.down(out, dout, 32 * n);
}
int csh_lincomb(csh_curve_t f, const uint64_t* const* shares, const uint64_t* coeffs, size_t k, uint64_t* out, size_t n) {
  CSH_REQUIRE(k >= 1 && k <= (size_t)MAX_LINCOMB, "lincomb: 1 <= k <= 16");
  CSH_REQUIRE(shares && coeffs, "lincomb: NULL argument");
  HostStage h;
  size_t eb = 32 * n;
  CSH_TRY(h.begin((k + 1) * Arena::padded(eb)));
  const uint64_t* dsh[MAX_LINCOMB];
  for (size_t j = 0; j < k; ++j) {
    uint64_t* d;
    CSH_TRY(h.up(d, shares[j], eb));
    dsh[j] = d;
  }
  uint64_t* dout;
  CSH_TRY(h.up(dout, nullptr, eb));
  CSH_TRY(csh_lincomb_dev(f, dsh, coeffs, k, dout, n, h.st));
  return h.down(out, dout, eb);
}

}  // extern "C"
