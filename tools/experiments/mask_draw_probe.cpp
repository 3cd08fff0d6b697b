// Host-only probe: how does the mirror's Rep3 mask draw (host/mpc.hpp masking_field_elements_vec: two ChaCha12 keystreams + two
// from_be_bytes_mod_order per element) scale over host threads on this box? Build: hipcc -O2 -std=c++17 -x hip --offload-arch=gfx950 ...
#include <chrono>
#include <cstdio>
#include "../../co-snarks_amd/host/mpc.hpp"
using namespace cosnarks;
int main() {
  uint8_t s1[32] = {1}, s2[32] = {2};
  using Fr = csh::Bn254Fr;
  const size_t len = size_t(1) << 20;
  printf("host_threads() = %zu\n", host_threads());
  for (int nt : {1, 2, 4, 8, 16, 32, 64}) {
    ChaCha12 r1(s1), r2(s2);
    double best = 1e30;
    for (int it = 0; it < 3; ++it) {
      UninitBuf<Fr> out(len);
      Fr* o = out.data();
      auto t0 = std::chrono::steady_clock::now();
      std::vector<std::thread> th;
      const size_t per = (len + nt - 1) / nt;
      for (int t = 0; t < nt; ++t) {
        const size_t lo = t * per, hi = std::min(len, lo + per);
        th.emplace_back([&, lo, hi] {
          ChaCha12 a = r1, b = r2;
          a.seek(32 * (uint64_t)lo);
          b.seek(32 * (uint64_t)lo);
          uint8_t x[32], y[32];
          for (size_t i = lo; i < hi; ++i) {
            a.fill_bytes(x, 32);
            b.fill_bytes(y, 32);
            o[i] = mask_element_from_be_bytes<Fr>(x, y);
          }
        });
      }
      for (auto& t : th) t.join();
      best = std::min(best, ms_since(t0));
    }
    printf("threads %2d: %.1f ms per 2^20-element draw\n", nt, best);
  }
  // pieces, one thread
  ChaCha12 c(s1);
  std::vector<uint8_t> buf(32 << 20);
  auto t0 = std::chrono::steady_clock::now();
  c.fill_bytes(buf.data(), buf.size());
  printf("chacha12 32 MB, 1 thread: %.1f ms\n", ms_since(t0));
  t0 = std::chrono::steady_clock::now();
  Fr acc = Fr::zero();
  for (size_t i = 0; i + 1 < len; ++i) acc = Fr::add(acc, mask_element_from_be_bytes<Fr>(&buf[32 * i], &buf[32 * i + 32]));
  printf("2 conversions + sub, 2^20 elements, 1 thread: %.1f ms (%08x)\n", ms_since(t0), acc.l[0]);
  return 0;
}
