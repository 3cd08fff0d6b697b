// Synthetic-input helpers (bench / large-size parity): known-discrete-log bases generated on the device.
#include <string.h>

#include "common.hpp"
#include "curve.hpp"

namespace csh {

__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// out[i] = k_i * G, k_i = splitmix64(seed + i) | 1  (affine, Montgomery)
template <class Fq>
__global__ __launch_bounds__(128) void k_gen_bases(Affine<Fq> gen, uint64_t seed, size_t n, Affine<Fq>* out) {
  const size_t i = blockIdx.x * (size_t)128 + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = splitmix64(seed + i) | 1ull;
  XYZZ<Fq> acc = XYZZ<Fq>::inf();
  for (int b = 63; b >= 0; --b) {
    acc = xyzz_dbl(acc);
    if ((k >> b) & 1) xyzz_madd(acc, gen);
  }
  out[i] = xyzz_to_affine(acc);
}

template <class Fq>
static int gen_bases_t(const uint32_t* gen_words, uint64_t seed, size_t n, void* out_dev, hipStream_t st) {
  Affine<Fq> g;
  memcpy(&g, gen_words, sizeof g);
  if (n == 0) return CSH_OK;
  hipLaunchKernelGGL(k_gen_bases<Fq>, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, g, seed, n, (Affine<Fq>*)out_dev);
  CSH_HIP(hipGetLastError());
  return CSH_OK;
}

}  // namespace csh

using namespace csh;

extern "C" {

uint64_t csh_util_splitmix64(uint64_t x) { return splitmix64(x); }

int csh_util_generate_bases_dev(csh_curve_t curve, csh_group_t group, uint64_t seed, size_t n, void* out_dev, void* stream) {
  CSH_REQUIRE(out_dev || n == 0, "out_dev is NULL");
  CSH_TRY(ensure_device());
  hipStream_t st = resolve_stream(stream);
  if (curve == CSH_BN254 && group == CSH_G1) return gen_bases_t<Bn254Fq>(Bn254G1Gen, seed, n, out_dev, st);
  if (curve == CSH_BN254 && group == CSH_G2) return gen_bases_t<Bn254Fq2>(Bn254G2Gen, seed, n, out_dev, st);
  if (curve == CSH_BLS12_381 && group == CSH_G1) return gen_bases_t<Bls381Fq>(Bls381G1Gen, seed, n, out_dev, st);
  if (curve == CSH_BLS12_381 && group == CSH_G2) return gen_bases_t<Bls381Fq2>(Bls381G2Gen, seed, n, out_dev, st);
  if (curve == CSH_GRUMPKIN && group == CSH_G1) return gen_bases_t<Bn254Fr>(GrumpkinG1Gen, seed, n, out_dev, st);
  if (curve == CSH_BLS12_377 && group == CSH_G1) return gen_bases_t<Bls377Fq>(Bls377G1Gen, seed, n, out_dev, st);
  if (curve == CSH_BLS12_377 && group == CSH_G2) return gen_bases_t<Bls377Fq2>(Bls377G2Gen, seed, n, out_dev, st);
  set_error("unknown curve/group");
  return CSH_ERR_INVALID;
}

}  // extern "C"
