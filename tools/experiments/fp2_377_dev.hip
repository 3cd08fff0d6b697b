// Device vs host: Fp2T<Bls377Fq, 5> products / squares / inverses and XYZZ steps on the 32-bit path (debug aid).
#include <stdio.h>
#include <string.h>
#include "../../co-snarks_amd/csrc/curve.hpp"
using namespace csh;
using F2 = Bls377Fq2;
struct Out { F2 m, s, i, a3; XYZZ<F2> d, md; Affine<F2> aff; };
__host__ __device__ void work(const Affine<F2>& g, Out* o) {
  o->m = F2::mul(g.x, g.y);
  o->s = F2::sqr(g.x);
  o->i = F2::inv(g.y);
  o->a3 = F2::mul3(g.x);
  XYZZ<F2> acc = XYZZ<F2>::inf();
  xyzz_madd(acc, g);
  acc = xyzz_dbl(acc);
  o->d = acc;
  xyzz_madd(acc, g);
  o->md = acc;
  o->aff = xyzz_to_affine(acc);
}
__global__ void k(Affine<F2> g, Out* o) { work(g, o); }
__host__ __device__ inline uint64_t sm64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
template <class Fq>
__host__ __device__ Affine<Fq> genone(const Affine<Fq>& gen, uint64_t k, int nb) {
  XYZZ<Fq> acc = XYZZ<Fq>::inf();
  for (int b = nb - 1; b >= 0; --b) {
    acc = xyzz_dbl(acc);
    if ((k >> b) & 1) xyzz_madd(acc, gen);
  }
  return xyzz_to_affine(acc);
}
template <class Fq>
__global__ __launch_bounds__(128) void kg(Affine<Fq> gen, uint64_t seed, size_t n, int nb, Affine<Fq>* out) {
  const size_t i = blockIdx.x * (size_t)128 + threadIdx.x;
  if (i >= n) return;
  out[i] = genone(gen, sm64(seed + i) | 1ull, nb);
}
template <class Fq>
__global__ __launch_bounds__(128) void k_gen_bases(Affine<Fq> gen, uint64_t seed, size_t n, Affine<Fq>* out) {
  const size_t i = blockIdx.x * (size_t)128 + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = sm64(seed + i) | 1ull;
  XYZZ<Fq> acc = XYZZ<Fq>::inf();
  for (int b = 63; b >= 0; --b) {
    acc = xyzz_dbl(acc);
    if ((k >> b) & 1) xyzz_madd(acc, gen);
  }
  out[i] = xyzz_to_affine(acc);
}
extern "C" int csh_util_generate_bases_dev(int, int, uint64_t, size_t, void*, void*);
int main() {
  {
    Affine<F2> g, *dd, hd[4], first[4];
    memcpy(&g, Bls377G2Gen, sizeof g);
    hipMalloc(&dd, sizeof(hd));
    csh_util_generate_bases_dev(3, 1, 99, 4, dd, nullptr);   // first kernel of the process
    hipDeviceSynchronize();
    hipMemcpy(first, dd, sizeof(first), hipMemcpyDeviceToHost);
    { int ok = 0; for (int i = 0; i < 4; ++i) { Affine<F2> w = genone(g, sm64(99 + i) | 1ull, 64); ok += !memcmp(&w, &first[i], sizeof w); }
      printf("library call as the FIRST kernel of the process: %d/4 equal host\n", ok); }
    hipLaunchKernelGGL(k_gen_bases<F2>, dim3(1), dim3(128), 0, 0, g, (uint64_t)99, (size_t)4, dd);
    hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
    int ok = 0;
    for (int i = 0; i < 4; ++i) { Affine<F2> w = genone(g, sm64(99 + i) | 1ull, 64); ok += !memcmp(&w, &hd[i], sizeof w); }
    printf("const-64 loop: %d/4 equal host\n", ok);
    Affine<F2> ld[4];
    int rc = csh_util_generate_bases_dev(3, 1, 99, 4, dd, nullptr);
    hipDeviceSynchronize();
    hipMemcpy(ld, dd, sizeof(ld), hipMemcpyDeviceToHost);
    ok = 0;
    for (int i = 0; i < 4; ++i) ok += !memcmp(&ld[i], &hd[i], sizeof(ld[i]));
    printf("library rc %d: %d/4 equal the standalone kernel\n", rc, ok);
  }
  {
    Affine<F2> g, *dd, hd[4];
    memcpy(&g, Bls377G2Gen, sizeof g);
    hipMalloc(&dd, sizeof(hd));
    for (int nb : {2, 3, 4, 8, 16, 32, 64}) {
      hipLaunchKernelGGL(kg<F2>, dim3(1), dim3(128), 0, 0, g, (uint64_t)99, (size_t)4, nb, dd);
      hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
      int ok = 0;
      for (int i = 0; i < 4; ++i) { Affine<F2> w = genone(g, sm64(99 + i) | 1ull, nb); ok += !memcmp(&w, &hd[i], sizeof w); }
      printf("nb %d: %d/4 equal host\n", nb, ok);
      if (nb == 64) { const uint32_t* w = (const uint32_t*)&hd[0]; printf("P0"); for (int j = 0; j < 48; ++j) printf(" %08x", w[j]); printf("\n"); }
    }
  }
  Affine<F2> g;
  memcpy(&g, Bls377G2Gen, sizeof g);
  Out h, d, *dd;
  work(g, &h);
  hipMalloc(&dd, sizeof(Out));
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, g, dd);
  hipMemcpy(&d, dd, sizeof(Out), hipMemcpyDeviceToHost);
  printf("mul %d sqr %d inv %d mul3 %d dbl %d madd %d affine %d\n", !memcmp(&h.m, &d.m, sizeof h.m), !memcmp(&h.s, &d.s, sizeof h.s), !memcmp(&h.i, &d.i, sizeof h.i),
         !memcmp(&h.a3, &d.a3, sizeof h.a3), !memcmp(&h.d, &d.d, sizeof h.d), !memcmp(&h.md, &d.md, sizeof h.md), !memcmp(&h.aff, &d.aff, sizeof h.aff));
  return 0;
}
