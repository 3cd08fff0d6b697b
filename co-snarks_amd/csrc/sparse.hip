// Sparse constraint evaluation (CSR rows x (public || witness)) on gfx950 and the fully device-resident
// CircomReduction::witness_map_from_matrices (co-circom/co-groth16/src/groth16/reduction.rs:77-193).
// One lane per constraint row (circom rows hold a handful of terms); gathers are 32/64-byte reads through L2.
#include <string.h>

#include <chrono>

#include "common.hpp"
#include "field.hpp"

namespace csh {

struct Matrix {
  csh_curve_t curve;
  int device;        // handles are per device (checked on use, like bases and domains)
  uint32_t max_col;  // largest column index (validated against n_public + n_witness where the caller states them)
  size_t n_rows, nnz;
  uint64_t* row_ptr;  // device, n_rows + 1
  uint32_t* col_idx;  // device, nnz
  void* coeffs;       // device, nnz Montgomery elements
};

// protocol 0: out[row] = sum c * v ; protocol 1 (Rep3): out[row] = {sum c * v.a (+ public on party 0), sum c * v.b (+ public on party 1)}
template <class F, int PROTOCOL>
__global__ __launch_bounds__(256) void k_eval_rows(const uint64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col_idx,
                                                   const F* __restrict__ coeffs, size_t n_rows, const F* __restrict__ pub, size_t n_public,
                                                   const F* __restrict__ wit, int party, F* out, size_t n_out) {
  for (size_t row = blockIdx.x * (size_t)256 + threadIdx.x; row < n_out; row += (size_t)gridDim.x * 256) {
    F acc_a = F::zero(), acc_b = F::zero();
    if (row < n_rows) {
      const uint64_t lo = row_ptr[row], hi = row_ptr[row + 1];
      for (uint64_t e = lo; e < hi; ++e) {
        const uint32_t col = col_idx[e];
        const F c = coeffs[e];
        if (col < n_public) {
          const F m = F::mul(pub[col], c);
          if (PROTOCOL == 0 || party == 0) acc_a = F::add(acc_a, m);
          else if (party == 1) acc_b = F::add(acc_b, m);
        } else {
          const size_t w = col - n_public;
          if (PROTOCOL == 0) {
            acc_a = F::add(acc_a, F::mul(wit[w], c));
          } else {
            acc_a = F::add(acc_a, F::mul(wit[2 * w], c));
            acc_b = F::add(acc_b, F::mul(wit[2 * w + 1], c));
          }
        }
      }
    }
    if (PROTOCOL == 0) {
      out[row] = acc_a;
    } else {
      out[2 * row] = acc_a;
      out[2 * row + 1] = acc_b;
    }
  }
}

// a[num_constraints + i] = promote_to_trivial_share(public[i]) (reduction.rs:111-113; rep3 types.rs:69-82)
template <class F, int PROTOCOL>
__global__ void k_promote_publics(F* a, size_t num_constraints, const F* __restrict__ pub, size_t n_public, int party) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n_public) return;
  if (PROTOCOL == 0) {
    a[num_constraints + i] = pub[i];
  } else {
    a[2 * (num_constraints + i)] = party == 0 ? pub[i] : F::zero();
    a[2 * (num_constraints + i) + 1] = party == 1 ? pub[i] : F::zero();
  }
}

template <class F>
static int eval_t(const Matrix* m, int protocol, int party, const uint64_t* pub, size_t n_public, const uint64_t* wit, uint64_t* out, size_t n_out,
                  hipStream_t st) {
  if (n_out == 0) return CSH_OK;
  const dim3 g(grid_for(n_out, 256)), b(256);
  if (protocol == 0)
    hipLaunchKernelGGL((k_eval_rows<F, 0>), g, b, 0, st, m->row_ptr, m->col_idx, (const F*)m->coeffs, m->n_rows, (const F*)pub, n_public,
                       (const F*)wit, party, (F*)out, n_out);
  else
    hipLaunchKernelGGL((k_eval_rows<F, 1>), g, b, 0, st, m->row_ptr, m->col_idx, (const F*)m->coeffs, m->n_rows, (const F*)pub, n_public,
                       (const F*)wit, party, (F*)out, n_out);
  CSH_HIP(hipGetLastError());
  return CSH_OK;
}

template <class F>
static int promote_t(int protocol, uint64_t* a, size_t num_constraints, const uint64_t* pub, size_t n_public, int party, hipStream_t st) {
  if (n_public == 0) return CSH_OK;
  const dim3 g((unsigned)((n_public + 63) / 64)), b(64);
  if (protocol == 0)
    hipLaunchKernelGGL((k_promote_publics<F, 0>), g, b, 0, st, (F*)a, num_constraints, (const F*)pub, n_public, party);
  else
    hipLaunchKernelGGL((k_promote_publics<F, 1>), g, b, 0, st, (F*)a, num_constraints, (const F*)pub, n_public, party);
  CSH_HIP(hipGetLastError());
  return CSH_OK;
}

}  // namespace csh

using namespace csh;

extern "C" {

int csh_matrix_upload(csh_curve_t field_of, const uint64_t* row_ptr, const uint32_t* col_idx, const uint64_t* coeffs, size_t n_rows, size_t nnz,
                      csh_matrix_t* out) {
  CSH_REQUIRE(out && row_ptr && (nnz == 0 || (col_idx && coeffs)), "matrix_upload: NULL argument");
  CSH_REQUIRE(field_of == CSH_BN254 || field_of == CSH_BLS12_381 || field_of == CSH_BLS12_377, "unknown curve");
  CSH_REQUIRE(row_ptr[n_rows] == nnz, "matrix_upload: row_ptr[n_rows] != nnz");
  CSH_REQUIRE(row_ptr[0] == 0, "matrix_upload: row_ptr[0] != 0");
  for (size_t i = 0; i < n_rows; ++i) CSH_REQUIRE(row_ptr[i] <= row_ptr[i + 1], "matrix_upload: row_ptr is not monotone");
  uint32_t max_col = 0;
  for (size_t i = 0; i < nnz; ++i) max_col = col_idx[i] > max_col ? col_idx[i] : max_col;
  CSH_TRY(ensure_device());
  Matrix* m = new Matrix();
  m->curve = field_of;
  m->max_col = max_col;
  if (hipGetDevice(&m->device) != hipSuccess) m->device = 0;
  m->n_rows = n_rows;
  m->nnz = nnz;
  m->row_ptr = nullptr;
  m->col_idx = nullptr;
  m->coeffs = nullptr;
  hipError_t e1 = hipMalloc((void**)&m->row_ptr, (n_rows + 1) * 8);
  hipError_t e2 = hipMalloc((void**)&m->col_idx, (nnz ? nnz : 1) * 4);
  hipError_t e3 = hipMalloc(&m->coeffs, (nnz ? nnz : 1) * 32);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
    csh_matrix_free(reinterpret_cast<csh_matrix_t>(m));
    set_error("matrix_upload: hipMalloc failed");
    return CSH_ERR_OOM;
  }
  {  // the caller's CSR arrays are typically temporaries it frees right after this call: uploads follow the "host_h2d" policy (staged by default)
    hipStream_t st = resolve_stream(nullptr);
    CSH_TRY(upload_h2d(m->row_ptr, row_ptr, (n_rows + 1) * 8, st, 0));
    if (nnz) {
      CSH_TRY(upload_h2d(m->col_idx, col_idx, nnz * 4, st, 1));
      CSH_TRY(upload_h2d(m->coeffs, coeffs, nnz * 32, st, 2));
    }
    CSH_HIP(hipStreamSynchronize(st));
  }
  *out = reinterpret_cast<csh_matrix_t>(m);
  return CSH_OK;
}

int csh_matrix_free(csh_matrix_t mm) {
  if (!mm) return CSH_OK;
  Matrix* m = reinterpret_cast<Matrix*>(mm);
  if (m->row_ptr) (void)hipFree(m->row_ptr);
  if (m->col_idx) (void)hipFree(m->col_idx);
  if (m->coeffs) (void)hipFree(m->coeffs);
  delete m;
  return CSH_OK;
}

// a matrix handle used from a thread bound to another GPU would hand the kernels pointers of the wrong device
static int check_matrix_device(const Matrix* m) {
  int cur = -1;
  if (hipGetDevice(&cur) == hipSuccess && cur != m->device) {
    set_error("constraint matrix was uploaded on device %d but the calling thread is bound to device %d (csh_init): upload a copy per device", m->device, cur);
    return CSH_ERR_INVALID;
  }
  return CSH_OK;
}

int csh_matrix_info(csh_matrix_t mm, size_t* n_rows, size_t* nnz, uint32_t* max_column, int* device) {
  CSH_REQUIRE(mm, "matrix is NULL");
  const Matrix* m = reinterpret_cast<const Matrix*>(mm);
  if (n_rows) *n_rows = m->n_rows;
  if (nnz) *nnz = m->nnz;
  if (max_column) *max_column = m->max_col;
  if (device) *device = m->device;
  return CSH_OK;
}

int csh_evaluate_constraints_dev(csh_matrix_t mm, int protocol, int party_id, const uint64_t* public_dev, size_t n_public,
                                 const uint64_t* witness_dev, size_t n_witness, uint64_t* out_dev, size_t n_out, void* stream) {
  CSH_REQUIRE(mm && out_dev, "evaluate_constraints: NULL argument");
  CSH_REQUIRE(protocol == 0 || protocol == 1, "protocol must be 0 (plain/Shamir) or 1 (Rep3)");
  CSH_TRY(ensure_device());
  Matrix* m = reinterpret_cast<Matrix*>(mm);
  CSH_TRY(check_matrix_device(m));
  // a matrix from an untrusted circuit must not index past the vectors the caller handed over (the kernel gathers by column)
  CSH_REQUIRE(m->nnz == 0 || (size_t)m->max_col < n_public + n_witness, "evaluate_constraints: a matrix column index exceeds n_public + n_witness");
  hipStream_t st = resolve_stream(stream);
  if (m->curve == CSH_BN254) return eval_t<Bn254Fr>(m, protocol, party_id, public_dev, n_public, witness_dev, out_dev, n_out, st);
  if (m->curve == CSH_BLS12_377) return eval_t<Bls377Fr>(m, protocol, party_id, public_dev, n_public, witness_dev, out_dev, n_out, st);
  return eval_t<Bls381Fr>(m, protocol, party_id, public_dev, n_public, witness_dev, out_dev, n_out, st);
}

}  // extern "C"

namespace {
// Where the two Rep3 mask vectors of a witness map come from (reduction.rs:160 then :182): the party's ChaCha12 keys (generated on the
// device) or two vectors the caller drew itself through the reference's public surface (`masking_field_elements_vec`, rngs.rs:137).
struct MaskSource {
  const uint8_t* seed1 = nullptr;
  uint64_t off1 = 0;
  const uint8_t* seed2 = nullptr;
  uint64_t off2 = 0;
  const uint64_t* mask_c_dev = nullptr;
  const uint64_t* mask_ab_dev = nullptr;
  bool have() const { return (seed1 && seed2) || (mask_c_dev && mask_ab_dev); }
};

// Device-resident core: witness already on the device, h stays on the device (scratch from the stream's arena)
int witness_map_core(csh_domain_t dom, const uint64_t shift[4], int protocol, int party_id, csh_matrix_t ma, csh_matrix_t mb, size_t num_constraints,
                     const uint64_t* public_inputs, size_t n_public, const uint64_t* witness_dev, size_t n_witness, const MaskSource& ms,
                     uint64_t* h_out_dev, void* stream) {
  CSH_REQUIRE(dom && shift && ma && mb && h_out_dev && (public_inputs || n_public == 0) && witness_dev, "witness_map_dev: NULL argument");
  CSH_REQUIRE(protocol == 0 || protocol == 1, "protocol must be 0 (plain/Shamir) or 1 (Rep3)");
  if (protocol == 1) CSH_TRY(require_rep3_masks(ms.have(), "groth16_witness_map"));
  CSH_TRY(ensure_device());
  const Domain* d = reinterpret_cast<const Domain*>(dom);
  const size_t n = domain_size_of(d);
  const csh_curve_t f = domain_curve_of(d);
  CSH_REQUIRE(num_constraints + n_public <= n, "Polynomial Degree too large");
  const size_t comp = protocol == 1 ? 2 : 1;
  const size_t sb = 32 * n * comp, eb = 32 * n;
  hipStream_t st = resolve_stream(stream);
  Arena& ar = arena_for((hipStream_t)((uintptr_t)st ^ 0x1));  // distinct arena from kernel-internal scratch
  CSH_TRY(ar.reserve(2 * Arena::padded(sb) + 2 * Arena::padded(eb) + Arena::padded(32 * n_public + 32)));
  uint64_t* da = reinterpret_cast<uint64_t*>(ar.take<char>(sb));
  uint64_t* db = reinterpret_cast<uint64_t*>(ar.take<char>(sb));
  uint64_t* dpub = reinterpret_cast<uint64_t*>(ar.take<char>(32 * n_public + 32));
  const uint64_t *dmc = ms.mask_c_dev, *dmab = ms.mask_ab_dev;
  if (n_public) CSH_HIP(hipMemcpyAsync(dpub, public_inputs, 32 * n_public, hipMemcpyHostToDevice, st));
  CSH_TRY(csh_evaluate_constraints_dev(ma, protocol, party_id, dpub, n_public, witness_dev, n_witness, da, n, st));   // reduction.rs:102-110
  CSH_TRY(csh_evaluate_constraints_dev(mb, protocol, party_id, dpub, n_public, witness_dev, n_witness, db, n, st));   // :118-127
  if (f == CSH_BN254) CSH_TRY(promote_t<Bn254Fr>(protocol, da, num_constraints, dpub, n_public, party_id, st));  // :111-113
  else if (f == CSH_BLS12_377) CSH_TRY(promote_t<Bls377Fr>(protocol, da, num_constraints, dpub, n_public, party_id, st));
  else CSH_TRY(promote_t<Bls381Fr>(protocol, da, num_constraints, dpub, n_public, party_id, st));
  if (protocol == 1 && !(dmc && dmab) && ms.seed1 && ms.seed2) {
    uint64_t* gc = reinterpret_cast<uint64_t*>(ar.take<char>(eb));
    uint64_t* gab = reinterpret_cast<uint64_t*>(ar.take<char>(eb));
    CSH_TRY(csh_rep3_masks_dev(f, ms.seed1, ms.off1, ms.seed2, ms.off2, gc, n, st));
    CSH_TRY(csh_rep3_masks_dev(f, ms.seed1, ms.off1 + n, ms.seed2, ms.off2 + n, gab, n, st));
    dmc = gc, dmab = gab;
  }
  CSH_TRY(csh_groth16_h_dev(dom, shift, protocol, da, db, dmc, dmab, h_out_dev, st));
  if (n_public) CSH_HIP(hipStreamSynchronize(st));  // public_inputs (host) must outlive the async copy
  return CSH_OK;
}

// Host-facing shell: witness shares (and, for the masks variant, the two mask vectors) up, h down; the pages of h_out are populated
// from host threads while the device works (HostXfer, common.hpp), so a freshly allocated result vector costs no first-touch in the copy.
int witness_map_host(csh_domain_t dom, const uint64_t shift[4], int protocol, int party_id, csh_matrix_t ma, csh_matrix_t mb, size_t num_constraints,
                     const uint64_t* public_inputs, size_t n_public, const uint64_t* witness, size_t n_witness, const uint8_t* seed1, uint64_t off1,
                     const uint8_t* seed2, uint64_t off2, const uint64_t* mask_c, const uint64_t* mask_ab, uint64_t* h_out) {
  CSH_REQUIRE(dom && ma && mb && h_out && (witness || n_witness == 0), "witness_map: NULL argument");
  CSH_REQUIRE(protocol == 0 || protocol == 1, "protocol must be 0 (plain/Shamir) or 1 (Rep3)");
  {  // a column index beyond public || witness would be an out-of-bounds device read in the row kernel
    const Matrix* A = reinterpret_cast<const Matrix*>(ma);
    const Matrix* B = reinterpret_cast<const Matrix*>(mb);
    const uint32_t mc = A->max_col > B->max_col ? A->max_col : B->max_col;
    CSH_REQUIRE((A->nnz == 0 && B->nnz == 0) || (size_t)mc < n_public + n_witness, "witness_map: a matrix column index exceeds n_public + n_witness");
  }
  const size_t n = domain_size_of(reinterpret_cast<const Domain*>(dom));
  const size_t comp = protocol == 1 ? 2 : 1;
  const bool host_masks = protocol == 1 && mask_c && mask_ab;
  CSH_TRY(ensure_device());
  hipStream_t st = resolve_stream(nullptr);
  Arena& ar = arena_for((hipStream_t)((uintptr_t)st ^ 0x2));  // staging arena (the core uses ^0x1 and the stream's own)
  CSH_TRY(ar.reserve(Arena::padded(32 * comp * n_witness + 32) + (host_masks ? 3 : 1) * Arena::padded(32 * n)));
  uint64_t* dwit = reinterpret_cast<uint64_t*>(ar.take<char>(32 * comp * n_witness + 32));
  uint64_t* dh = reinterpret_cast<uint64_t*>(ar.take<char>(32 * n));
  HostXfer pins;
  pins.expect_d2h(h_out, 32 * n);
  MaskSource ms;
  ms.seed1 = seed1, ms.off1 = off1, ms.seed2 = seed2, ms.off2 = off2;
  const bool timing = tune().host_timing.load(std::memory_order_relaxed) != 0;  // diagnostics only: the extra synchronisations cost ~20 us each
  auto t_phase = std::chrono::steady_clock::now();
  auto phase_done = [&](std::atomic<int>& counter) {
    if (!timing) return;
    (void)hipStreamSynchronize(st);
    const auto now = std::chrono::steady_clock::now();
    counter.fetch_add((int)std::chrono::duration_cast<std::chrono::microseconds>(now - t_phase).count(), std::memory_order_relaxed);
    t_phase = now;
  };
  if (n_witness) CSH_TRY(pins.h2d(dwit, witness, 32 * comp * n_witness, st));
  if (host_masks) {
    uint64_t* dmc = reinterpret_cast<uint64_t*>(ar.take<char>(32 * n));
    uint64_t* dmab = reinterpret_cast<uint64_t*>(ar.take<char>(32 * n));
    CSH_TRY(pins.h2d(dmc, mask_c, 32 * n, st));
    CSH_TRY(pins.h2d(dmab, mask_ab, 32 * n, st));
    ms.mask_c_dev = dmc, ms.mask_ab_dev = dmab;
  }
  phase_done(tune().stat_wm_h2d_us);
  CSH_TRY(witness_map_core(dom, shift, protocol, party_id, ma, mb, num_constraints, public_inputs, n_public, dwit, n_witness, ms, dh, st));
  phase_done(tune().stat_wm_dev_us);
  CSH_TRY(pins.d2h(h_out, dh, 32 * n, st));
  const int rc = pins.finish(st);
  phase_done(tune().stat_wm_d2h_us);
  return rc;
}
}  // namespace

extern "C" {

int csh_groth16_witness_map_dev(csh_domain_t dom, const uint64_t shift[4], int protocol, int party_id, csh_matrix_t ma, csh_matrix_t mb,
                                size_t num_constraints, const uint64_t* public_inputs, size_t n_public, const uint64_t* witness_dev, size_t n_witness,
                                const uint8_t seed1[32], uint64_t off1, const uint8_t seed2[32], uint64_t off2, uint64_t* h_out_dev, void* stream) {
  MaskSource ms;
  ms.seed1 = seed1, ms.off1 = off1, ms.seed2 = seed2, ms.off2 = off2;
  return witness_map_core(dom, shift, protocol, party_id, ma, mb, num_constraints, public_inputs, n_public, witness_dev, n_witness, ms, h_out_dev, stream);
}

int csh_groth16_witness_map_masks_dev(csh_domain_t dom, const uint64_t shift[4], int protocol, int party_id, csh_matrix_t ma, csh_matrix_t mb,
                                      size_t num_constraints, const uint64_t* public_inputs, size_t n_public, const uint64_t* witness_dev, size_t n_witness,
                                      const uint64_t* mask_c_dev, const uint64_t* mask_ab_dev, uint64_t* h_out_dev, void* stream) {
  MaskSource ms;
  ms.mask_c_dev = mask_c_dev, ms.mask_ab_dev = mask_ab_dev;
  return witness_map_core(dom, shift, protocol, party_id, ma, mb, num_constraints, public_inputs, n_public, witness_dev, n_witness, ms, h_out_dev, stream);
}

int csh_groth16_witness_map(csh_domain_t dom, const uint64_t shift[4], int protocol, int party_id, csh_matrix_t ma, csh_matrix_t mb,
                            size_t num_constraints, const uint64_t* public_inputs, size_t n_public, const uint64_t* witness, size_t n_witness,
                            const uint8_t seed1[32], uint64_t off1, const uint8_t seed2[32], uint64_t off2, uint64_t* h_out) {
  return witness_map_host(dom, shift, protocol, party_id, ma, mb, num_constraints, public_inputs, n_public, witness, n_witness, seed1, off1, seed2, off2,
                          nullptr, nullptr, h_out);
}

// The same with the two mask vectors handed over by the caller (host), in the order the reference draws them: mask_c for "c:
// local_mul_vec" (reduction.rs:160), mask_ab for the last product (:182). This is the form an UNCHANGED reference can drive: Rep3Rand's
// generators are private (rngs.rs:83-86), `masking_field_elements_vec` (rngs.rs:137) is public, and for a generic driver
// T::local_mul_vec on two zero vectors returns exactly the mask. NULL masks for protocol 0.
int csh_groth16_witness_map_masks(csh_domain_t dom, const uint64_t shift[4], int protocol, int party_id, csh_matrix_t ma, csh_matrix_t mb,
                                  size_t num_constraints, const uint64_t* public_inputs, size_t n_public, const uint64_t* witness, size_t n_witness,
                                  const uint64_t* mask_c, const uint64_t* mask_ab, uint64_t* h_out) {
  if (protocol == 1) CSH_TRY(require_rep3_masks(mask_c && mask_ab, "groth16_witness_map_masks"));
  return witness_map_host(dom, shift, protocol, party_id, ma, mb, num_constraints, public_inputs, n_public, witness, n_witness, nullptr, 0, nullptr, 0,
                          mask_c, mask_ab, h_out);
}

// LibSnarkReduction::witness_map_from_matrices (reduction.rs:241-342) on the device: a, b as above; c through
// evaluate_constraint_half_share (mpc/rep3.rs:51-74: the `a` component of the full-share row kernel, public terms on
// party 0 only; plain / Shamir: the row value itself), then csh_groth16_h_libsnark_dev. One mask vector (one local_mul_vec).
static int witness_map_libsnark_host(csh_domain_t dom, const uint64_t generator[4], int protocol, int party_id, csh_matrix_t ma, csh_matrix_t mb,
                                     csh_matrix_t mc, size_t num_constraints, const uint64_t* public_inputs, size_t n_public, const uint64_t* witness,
                                     size_t n_witness, const uint8_t* seed1, uint64_t off1, const uint8_t* seed2, uint64_t off2, const uint64_t* mask,
                                     uint64_t* h_out) {
  CSH_REQUIRE(dom && generator && ma && mb && mc && h_out && (public_inputs || n_public == 0) && (witness || n_witness == 0),
              "witness_map_libsnark: NULL argument");
  CSH_REQUIRE(protocol == 0 || protocol == 1, "protocol must be 0 (plain/Shamir) or 1 (Rep3)");
  if (protocol == 1) CSH_TRY(require_rep3_masks((seed1 && seed2) || mask, "groth16_witness_map_libsnark"));
  for (csh_matrix_t mm : {ma, mb, mc}) {
    const Matrix* M = reinterpret_cast<const Matrix*>(mm);
    CSH_REQUIRE(M->nnz == 0 || (size_t)M->max_col < n_public + n_witness, "witness_map_libsnark: a matrix column index exceeds n_public + n_witness");
  }
  const Domain* d = reinterpret_cast<const Domain*>(dom);
  const size_t n = domain_size_of(d);
  const csh_curve_t f = domain_curve_of(d);
  CSH_REQUIRE(num_constraints + n_public <= n, "Polynomial Degree too large");
  const size_t comp = protocol == 1 ? 2 : 1;
  const size_t sb = 32 * n * comp, eb = 32 * n;
  HostStage h;
  CSH_TRY(h.begin(3 * Arena::padded(sb) + 3 * Arena::padded(eb) + Arena::padded(32 * n_public + 32) + Arena::padded(32 * comp * n_witness + 32)));
  HostXfer x;
  x.expect_d2h(h_out, eb);
  uint64_t *da, *db, *dcf, *dc, *dm = nullptr, *dh, *dpub, *dwit;
  CSH_TRY(h.up(da, nullptr, sb));
  CSH_TRY(h.up(db, nullptr, sb));
  CSH_TRY(h.up(dcf, nullptr, sb));
  CSH_TRY(h.up(dh, nullptr, eb));
  CSH_TRY(h.up(dpub, public_inputs, 32 * n_public));
  CSH_TRY(h.up(dwit, witness, 32 * comp * n_witness));
  CSH_TRY(csh_evaluate_constraints_dev(ma, protocol, party_id, dpub, n_public, dwit, n_witness, da, n, h.st));    // reduction.rs:260-266
  if (f == CSH_BN254) CSH_TRY(promote_t<Bn254Fr>(protocol, da, num_constraints, dpub, n_public, party_id, h.st));  // :267-269
  else if (f == CSH_BLS12_377) CSH_TRY(promote_t<Bls377Fr>(protocol, da, num_constraints, dpub, n_public, party_id, h.st));
  else CSH_TRY(promote_t<Bls381Fr>(protocol, da, num_constraints, dpub, n_public, party_id, h.st));
  CSH_TRY(csh_evaluate_constraints_dev(mb, protocol, party_id, dpub, n_public, dwit, n_witness, db, n, h.st));    // :276-282
  CSH_TRY(csh_evaluate_constraints_dev(mc, protocol, party_id, dpub, n_public, dwit, n_witness, dcf, n, h.st));   // :292-298
  if (protocol == 1) {  // half share = component a
    CSH_TRY(h.up(dc, nullptr, eb));
    CSH_HIP(hipMemcpy2DAsync(dc, 32, dcf, 64, 32, n, hipMemcpyDeviceToDevice, h.st));
    if (mask) {
      CSH_TRY(h.up(dm, mask, eb));
    } else if (seed1 && seed2) {
      CSH_TRY(h.up(dm, nullptr, eb));
      CSH_TRY(csh_rep3_masks_dev(f, seed1, off1, seed2, off2, dm, n, h.st));
    }
  } else {
    dc = dcf;
  }
  CSH_TRY(csh_groth16_h_libsnark_dev(dom, generator, protocol, da, db, dc, dm, dh, h.st));
  CSH_TRY(x.d2h(h_out, dh, eb, h.st));
  return x.finish(h.st);
}

int csh_groth16_witness_map_libsnark(csh_domain_t dom, const uint64_t generator[4], int protocol, int party_id, csh_matrix_t ma, csh_matrix_t mb,
                                     csh_matrix_t mc, size_t num_constraints, const uint64_t* public_inputs, size_t n_public,
                                     const uint64_t* witness, size_t n_witness, const uint8_t seed1[32], uint64_t off1, const uint8_t seed2[32],
                                     uint64_t off2, uint64_t* h_out) {
  return witness_map_libsnark_host(dom, generator, protocol, party_id, ma, mb, mc, num_constraints, public_inputs, n_public, witness, n_witness, seed1, off1,
                                   seed2, off2, nullptr, h_out);
}

// The same with the one mask vector of its local_mul_vec (reduction.rs:289) handed over by the caller: see csh_groth16_witness_map_masks.
int csh_groth16_witness_map_libsnark_masks(csh_domain_t dom, const uint64_t generator[4], int protocol, int party_id, csh_matrix_t ma, csh_matrix_t mb,
                                           csh_matrix_t mc, size_t num_constraints, const uint64_t* public_inputs, size_t n_public,
                                           const uint64_t* witness, size_t n_witness, const uint64_t* mask, uint64_t* h_out) {
  if (protocol == 1) CSH_TRY(require_rep3_masks(mask != nullptr, "groth16_witness_map_libsnark_masks"));
  return witness_map_libsnark_host(dom, generator, protocol, party_id, ma, mb, mc, num_constraints, public_inputs, n_public, witness, n_witness, nullptr, 0,
                                   nullptr, 0, mask, h_out);
}

}  // extern "C"
