// Integer-pipe micro-benchmarks for gfx950 ("measure, don't guess"): issue rates of the instructions a
// Montgomery multiplication is made of, and modmul throughput of candidate limb representations.
// Results drive the choice of field representation in the MSM kernels (DESIGN.md "integer roofline").
// Not part of include/cosnarks_hip.h; exported for bench/profiling harnesses only.
// (the signed-lazy multiplier is probed in the form the accumulate and NTT kernels run it: multiply-add order pinned)
#define CSH_PIN_MADS 3
#include "common.hpp"
#include "field.hpp"
#include "field29.hpp"
#include <string.h>

namespace csh {

constexpr int UB_BLK = 256;
constexpr int UB_CHAINS = 8;

// kind: 0 v_mad_u64_u32, 1 v_lshl_add_u64, 2 v_add_co_u32 + v_addc_co_u32 (2 instructions), 3 v_mul_lo_u32, 4 v_add_u32,
// 5 v_mul_hi_u32, 6 v_ashrrev_i64, 7 v_alignbit_b32 + v_ashrrev_i32 (2 instructions), 8 v_mad_i64_i32, 9 v_lshl_add_u64
template <int KIND>
__global__ __launch_bounds__(UB_BLK) void k_ub(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + seed * 3u + 1u;
  uint64_t acc[UB_CHAINS];
#pragma unroll
  for (int k = 0; k < UB_CHAINS; ++k) acc[k] = ((uint64_t)a << 20) + k * 977u + b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int k = 0; k < UB_CHAINS; ++k) {
        // every instruction is forced with inline asm: left to itself the compiler collapses unrolled chains of constant
        // adds / multiplies / shifts into fewer instructions and the "rate" comes out several times too high
        if (KIND == 0) {
          asm volatile("v_mad_u64_u32 %0, s[4:5], %1, %2, %0" : "+v"(acc[k]) : "v"((uint32_t)acc[(k + 1) % UB_CHAINS]), "s"(b) : "s4", "s5");
        } else if (KIND == 1) {
          asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[k]) : "v"(acc[(k + 1) % UB_CHAINS]));
        } else if (KIND == 2) {  // 64-bit add as a carry pair (counted as 2 instructions)
          uint32_t lo = (uint32_t)acc[k], hi = (uint32_t)(acc[k] >> 32);
          asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
          acc[k] = ((uint64_t)hi << 32) | lo;
        } else if (KIND == 3) {
          uint32_t lo = (uint32_t)acc[k];
          asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(b | 1u));
          acc[k] = lo;
        } else if (KIND == 4) {
          uint32_t lo = (uint32_t)acc[k];
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(b));
          acc[k] = lo;
        } else if (KIND == 5) {
          uint32_t lo = (uint32_t)acc[k];
          asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(b | 0x80000001u));
          acc[k] = lo;
        } else if (KIND == 6) {  // v_ashrrev_i64 (forced: the compiler would merge a chain of constant shifts)
          asm volatile("v_ashrrev_i64 %0, 3, %0" : "+v"(acc[k]));
        } else if (KIND == 7) {  // the same shift from 32-bit halves: v_alignbit_b32 + v_ashrrev_i32
          uint32_t lo = (uint32_t)acc[k], hi = (uint32_t)(acc[k] >> 32);
          asm volatile("v_alignbit_b32 %0, %1, %0, 3\n\tv_ashrrev_i32 %1, 3, %1" : "+v"(lo), "+v"(hi));
          acc[k] = ((uint64_t)hi << 32) | lo;
        } else if (KIND == 8) {  // 64-bit += sign-extended 32-bit value through v_mad_i64_i32 (multiplier in an SGPR)
          asm volatile("v_mad_i64_i32 %0, s[4:5], %1, %2, %0" : "+v"(acc[k]) : "v"((uint32_t)acc[(k + 1) % UB_CHAINS]), "s"(b) : "s4", "s5");
        } else {  // 9: v_lshl_add_u64 (forced)
          asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[k]) : "v"(acc[(k + 1) % UB_CHAINS]));
        }
      }
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < UB_CHAINS; ++k) s ^= acc[k];
  out[blockIdx.x * UB_BLK + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

// Dependent-chain probe: CH independent accumulators per lane, every v_mad_i64_i32 depends on the previous one of its own chain
// only (the multiplicands are loop-invariant) -- the dependency structure of the pinned product-scanning multiplication, where a
// lane's multiply-adds form ONE chain. Launched with a chosen number of waves per SIMD, it tells how much instruction-level or
// wave-level parallelism the 64-bit multiply-add pipe needs before it issues at its peak rate.
template <int CH>
__global__ __launch_bounds__(UB_BLK) void k_ub_chain(uint32_t* out, int iters, uint32_t seed) {
  const uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + seed * 3u + 1u;
  uint64_t acc[CH];
#pragma unroll
  for (int k = 0; k < CH; ++k) acc[k] = ((uint64_t)a << 20) + k * 977u + b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int k = 0; k < CH; ++k) asm volatile("v_mad_i64_i32 %0, s[4:5], %1, %2, %0" : "+v"(acc[k]) : "v"(a + k), "s"(b) : "s4", "s5");
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < CH; ++k) s ^= acc[k];
  out[blockIdx.x * UB_BLK + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

// v_fma_f64 issue rate (kind 13): the building block of the double-precision "split product" multiplier (two FMAs give the
// exact high and low halves of a 52 x 52-bit limb product). Measured to price that alternative against the 29-bit integer
// scheme: each limb product there costs 2 FMA + 1 subtraction + 2 64-bit integer additions.
__global__ __launch_bounds__(UB_BLK) void k_ub_fma64(double* out, int iters, double seed) {
  double acc[UB_CHAINS];
  const double b = 1.0 + seed * 1e-9, c = 1e-3 * (threadIdx.x + 1);
#pragma unroll
  for (int k = 0; k < UB_CHAINS; ++k) acc[k] = seed + k * 0.5 + blockIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int k = 0; k < UB_CHAINS; ++k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[k]) : "v"(b), "v"(c));
    }
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < UB_CHAINS; ++k) s += acc[k];
  out[blockIdx.x * UB_BLK + threadIdx.x] = s;
}

// modmul chains: kind 10 = Fp<Bn254Fq>::mul (8 x 32-bit CIOS), 11 = Fp29 (9 x 29-bit lazy), 12 = Bls381Fq
template <class F>
__global__ __launch_bounds__(UB_BLK) void k_modmul(const F* in, F* out, int iters) {
  const int i = blockIdx.x * UB_BLK + threadIdx.x;
  F x = in[i], y = in[i ^ 1];
  for (int it = 0; it < iters; ++it) {
    x = F::mul(x, y);
    y = F::mul(y, x);
  }
  out[i] = F::add(x, y);
}

__global__ __launch_bounds__(UB_BLK) void k_modmul29(const Bn254Fq* in, Bn254Fq* out, int iters) {
  const int i = blockIdx.x * UB_BLK + threadIdx.x;
  Fq29 x = Fq29::from_fp(in[i]), y = Fq29::from_fp(in[i ^ 1]);
  for (int it = 0; it < iters; ++it) {
    x = Fq29::mul(x, y);
    y = Fq29::mul(y, x);
  }
  out[i] = Fq29::add(x, y).to_fp();
}

// the multiplier of the bucket and NTT kernels: signed lazy 9 x 29-bit limbs, product scanning (field29.hpp FpS::mul)
__global__ __launch_bounds__(UB_BLK) void k_modmul29s(const Bn254Fq* in, Bn254Fq* out, int iters) {
  const int i = blockIdx.x * UB_BLK + threadIdx.x;
  Fq29s x = Fq29s::from_fp(in[i]), y = Fq29s::from_fp(in[i ^ 1]);
  for (int it = 0; it < iters; ++it) {
    x = Fq29s::mul(x, y);
    y = Fq29s::mul(y, x);
  }
  out[i] = Fq29s::add(x, y).normalized().to_fp();
}

// FETCH_SIZE / WRITE_SIZE calibration (tools/gpu_calib.py under rocprofv3 --pmc): every lane reads ONE record of REC bytes
// from a table far larger than the 256 MiB Infinity Cache -- at a hashed index (the access pattern of k_msm_accum's base
// gather: REC = 64 / 96 / 128 / 192 for the four groups) or at its own index (SEQ: the coalesced streaming pattern the
// guide's x2 correction was calibrated on) -- and writes 4 bytes. Known traffic: n * REC read, n * 4 written.
template <int REC, bool SEQ>
__global__ __launch_bounds__(256) void k_gather_calib(const uint4* __restrict__ table, uint32_t rec_mask, uint32_t* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  uint64_t x = (uint64_t)i + 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  const uint32_t r = SEQ ? (i & rec_mask) : ((uint32_t)(x ^ (x >> 31)) & rec_mask);
  const uint4* p = table + (size_t)r * (REC / 16);
  uint4 acc = p[0];
#pragma unroll
  for (int k = 1; k < REC / 16; ++k) {
    const uint4 v = p[k];
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  out[i] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

template <class K, class... Args>
static int time_kernel(K kern, dim3 grid, dim3 blk, float* ms, Args... args) {
  hipStream_t st = resolve_stream(nullptr);
  hipEvent_t e0, e1;
  CSH_HIP(hipEventCreate(&e0));
  CSH_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, grid, blk, 0, st, args...);  // warm-up
  CSH_HIP(hipEventRecord(e0, st));
  hipLaunchKernelGGL(kern, grid, blk, 0, st, args...);
  CSH_HIP(hipEventRecord(e1, st));
  CSH_HIP(hipEventSynchronize(e1));
  CSH_HIP(hipEventElapsedTime(ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return CSH_OK;
}

}  // namespace csh

using namespace csh;

extern "C" {

// Returns lane-operations per second of the selected instruction / modmul variant.
int csh_microbench(int kind, int iters, double* ops_per_s) {
  CSH_REQUIRE(ops_per_s && iters > 0, "bad argument");
  CSH_TRY(ensure_device());
  const int blocks = 256 * 8;
  const size_t threads = (size_t)blocks * UB_BLK;
  float ms = 0;
  if (kind == 13) {
    double* out;
    CSH_HIP(hipMalloc((void**)&out, threads * 8));
    const int rc = time_kernel(k_ub_fma64, dim3(blocks), dim3(UB_BLK), &ms, out, iters, 1.5);
    (void)hipFree(out);
    if (rc != CSH_OK) return rc;
    *ops_per_s = (double)iters * 4 * UB_CHAINS * threads / (ms * 1e-3);
    return CSH_OK;
  }
  if (kind < 10) {
    uint32_t* out;
    CSH_HIP(hipMalloc((void**)&out, threads * 4));
    int rc = CSH_OK;
    switch (kind) {
      case 0: rc = time_kernel(k_ub<0>, dim3(blocks), dim3(UB_BLK), &ms, out, iters, 7u); break;
      case 1: rc = time_kernel(k_ub<1>, dim3(blocks), dim3(UB_BLK), &ms, out, iters, 7u); break;
      case 2: rc = time_kernel(k_ub<2>, dim3(blocks), dim3(UB_BLK), &ms, out, iters, 7u); break;
      case 3: rc = time_kernel(k_ub<3>, dim3(blocks), dim3(UB_BLK), &ms, out, iters, 7u); break;
      case 4: rc = time_kernel(k_ub<4>, dim3(blocks), dim3(UB_BLK), &ms, out, iters, 7u); break;
      case 5: rc = time_kernel(k_ub<5>, dim3(blocks), dim3(UB_BLK), &ms, out, iters, 7u); break;
      case 6: rc = time_kernel(k_ub<6>, dim3(blocks), dim3(UB_BLK), &ms, out, iters, 7u); break;
      case 7: rc = time_kernel(k_ub<7>, dim3(blocks), dim3(UB_BLK), &ms, out, iters, 7u); break;
      case 8: rc = time_kernel(k_ub<8>, dim3(blocks), dim3(UB_BLK), &ms, out, iters, 7u); break;
      default: rc = time_kernel(k_ub<9>, dim3(blocks), dim3(UB_BLK), &ms, out, iters, 7u); break;
    }
    (void)hipFree(out);
    if (rc != CSH_OK) return rc;
    double per_lane = (double)iters * 4 * UB_CHAINS * ((kind == 2 || kind == 7) ? 2 : 1);  // instructions per lane
    *ops_per_s = per_lane * threads / (ms * 1e-3);
    return CSH_OK;
  }
  // modmul variants: inputs = small Montgomery-form values
  const size_t esz = kind == 12 ? sizeof(Bls381Fq) : sizeof(Bn254Fq);
  void *in, *out;
  CSH_HIP(hipMalloc(&in, threads * esz));
  CSH_HIP(hipMalloc(&out, threads * esz));
  CSH_HIP(hipMemset(in, 0x11, threads * esz));  // 0x1111... < p for all fields here
  int rc;
  if (kind == 10)
    rc = time_kernel(k_modmul<Bn254Fq>, dim3(blocks), dim3(UB_BLK), &ms, (const Bn254Fq*)in, (Bn254Fq*)out, iters);
  else if (kind == 11)
    rc = time_kernel(k_modmul29, dim3(blocks), dim3(UB_BLK), &ms, (const Bn254Fq*)in, (Bn254Fq*)out, iters);
  else if (kind == 14)
    rc = time_kernel(k_modmul29s, dim3(blocks), dim3(UB_BLK), &ms, (const Bn254Fq*)in, (Bn254Fq*)out, iters);
  else
    rc = time_kernel(k_modmul<Bls381Fq>, dim3(blocks), dim3(UB_BLK), &ms, (const Bls381Fq*)in, (Bls381Fq*)out, iters);
  (void)hipFree(in);
  (void)hipFree(out);
  if (rc != CSH_OK) return rc;
  *ops_per_s = 2.0 * iters * threads / (ms * 1e-3);
  return CSH_OK;
}

// chains in {1, 2, 4}, waves_per_simd in 1..8: lane-ops/s of dependent v_mad_i64_i32 chains at that occupancy (workgroups of four
// waves = one per SIMD of a CU; waves_per_simd workgroups per CU)
int csh_microbench_chain(int chains, int waves_per_simd, int iters, double* ops_per_s) {
  CSH_REQUIRE(ops_per_s && iters > 0 && waves_per_simd >= 1 && waves_per_simd <= 8 && (chains == 1 || chains == 2 || chains == 4), "bad argument");
  CSH_TRY(ensure_device());
  const int blocks = (device_simds() / 4) * waves_per_simd;
  const size_t threads = (size_t)blocks * UB_BLK;
  uint32_t* out;
  CSH_HIP(hipMalloc((void**)&out, threads * 4));
  float ms = 0;
  int rc;
  if (chains == 1) rc = time_kernel(k_ub_chain<1>, dim3(blocks), dim3(UB_BLK), &ms, out, iters, 7u);
  else if (chains == 2) rc = time_kernel(k_ub_chain<2>, dim3(blocks), dim3(UB_BLK), &ms, out, iters, 7u);
  else rc = time_kernel(k_ub_chain<4>, dim3(blocks), dim3(UB_BLK), &ms, out, iters, 7u);
  (void)hipFree(out);
  if (rc != CSH_OK) return rc;
  *ops_per_s = (double)iters * 16 * chains * threads / (ms * 1e-3);
  return CSH_OK;
}

// One calibration launch (+ a warm-up): n = 2^log_n lanes, each reading one rec_bytes record out of 2^log_records.
// ms: duration of the timed launch. bytes_read / bytes_written: the known traffic of that launch.
int csh_microbench_gather(int rec_bytes, int log_records, int log_n, int sequential, float* ms, double* bytes_read, double* bytes_written) {
  CSH_REQUIRE(ms && bytes_read && bytes_written, "NULL argument");
  CSH_REQUIRE(rec_bytes == 64 || rec_bytes == 96 || rec_bytes == 128 || rec_bytes == 192, "record size must be 64, 96, 128 or 192");
  CSH_REQUIRE(log_records >= 10 && log_records <= 27 && log_n >= 10 && log_n <= 27, "bad sizes");
  CSH_TRY(ensure_device());
  const size_t recs = size_t(1) << log_records, n = size_t(1) << log_n;
  void* table;
  uint32_t* out;
  CSH_HIP(hipMalloc(&table, recs * rec_bytes));
  CSH_HIP(hipMalloc((void**)&out, n * 4));
  CSH_HIP(hipMemset(table, 0x5a, recs * rec_bytes));
  const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
  const uint32_t mask = (uint32_t)(recs - 1);
  int rc = CSH_OK;
#define CSH_CALIB(REC)                                                                                                             \
  rc = sequential ? time_kernel(k_gather_calib<REC, true>, grid, blk, ms, (const uint4*)table, mask, out, (uint32_t)n)              \
                  : time_kernel(k_gather_calib<REC, false>, grid, blk, ms, (const uint4*)table, mask, out, (uint32_t)n)
  if (rec_bytes == 64) { CSH_CALIB(64); }
  else if (rec_bytes == 96) { CSH_CALIB(96); }
  else if (rec_bytes == 128) { CSH_CALIB(128); }
  else { CSH_CALIB(192); }
#undef CSH_CALIB
  (void)hipFree(table);
  (void)hipFree(out);
  *bytes_read = (double)n * rec_bytes;
  *bytes_written = (double)n * 4;
  return rc;
}

// Host-side self-check hook for the 29-bit representation: out = to_fp(mul29(from_fp(a), from_fp(b)))
// (values in the same Montgomery domain as Fp<Bn254Fq>: R = 2^256).
int csh_test_mul29_host(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
  Bn254Fq fa, fb;
  memcpy(&fa, a, 32);
  memcpy(&fb, b, 32);
  Bn254Fq r = Fq29::mul(Fq29::from_fp(fa), Fq29::from_fp(fb)).to_fp();
  memcpy(out, &r, 32);
  return CSH_OK;
}

}  // extern "C"
