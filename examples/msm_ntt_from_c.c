/* Plain-C caller of the drop-in boundary (include/cosnarks_hip.h): what a cgo / JNI / N-API binding does, without the
 * binding.  2 * G + 3 * (2G) over BN254 G1 through csh_msm (the msm_unchecked replacement), then an NTT round trip
 * (Domain::ifft_in_to_out followed by fft_out_to_in returns the input).
 *
 *   gcc -std=c11 -Iinclude examples/msm_ntt_from_c.c -Lco-snarks_amd/lib -lcosnarks_hip \
 *       -Wl,-rpath,$PWD/co-snarks_amd/lib -o /tmp/msm_ntt_from_c && /tmp/msm_ntt_from_c
 *
 * Exit status: 0 = both checks passed on the GPU; 3 = no usable GPU (the library has no CPU fallback and says so);
 * 1 = wrong result.  All values are arkworks' in-memory encoding: little-endian Montgomery limbs. */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "cosnarks_hip.h"

/* BN254 Fq Montgomery constants: 1*R, 2*R mod q (generator G = (1, 2)); 2G = (x2, y2) in Montgomery form */
static const uint64_t ONE_R[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL};
static const uint64_t TWO_R[4] = {0xa6ba871b8b1e1b3aULL, 0x14f1d651eb8e167bULL, 0xccdd46def0f28c58ULL, 0x1c14ef83340fbe5eULL};
/* BN254 Fr Montgomery: 2*R mod r */
static const uint64_t FR_TWO[4] = {0x592c68389ffffff6ULL, 0x6df8ed2b3ec19a53ULL, 0xccdd46def0f28c5cULL, 0x1c14ef83340fbe5eULL};

static int fail(const char* what, int rc) {
  fprintf(stderr, "%s failed (%d): %s\n", what, rc, csh_last_error());
  return rc == CSH_ERR_NO_DEVICE ? 3 : 2;
}

int main(void) {
  int rc = csh_init(0);
  if (rc != CSH_OK) return fail("csh_init", rc);

  /* ---- NTT round trip over the BN254 scalar field, domain 2^10, arkworks' default root */
  enum { LOGN = 10, N = 1 << LOGN };
  static uint64_t data[N][4], orig[N][4];
  for (int i = 0; i < N; ++i) {
    /* any canonical Montgomery values will do: alternate two known elements below r */
    memcpy(data[i], (i & 1) ? FR_TWO : ONE_R, 32);
    memcpy(orig[i], data[i], 32);
  }
  csh_domain_t dom;
  rc = csh_domain_create(CSH_BN254, LOGN, NULL, &dom);
  if (rc != CSH_OK) return fail("csh_domain_create", rc);
  rc = csh_ifft_in_to_out(dom, &data[0][0], 1);
  if (rc != CSH_OK) return fail("csh_ifft_in_to_out", rc);
  if (memcmp(data, orig, sizeof data) == 0) {
    fprintf(stderr, "the inverse transform left the data unchanged\n");
    return 1;
  }
  rc = csh_fft_out_to_in(dom, &data[0][0], 1);
  if (rc != CSH_OK) return fail("csh_fft_out_to_in", rc);
  csh_domain_free(dom);
  if (memcmp(data, orig, sizeof data) != 0) {
    fprintf(stderr, "NTT round trip mismatch\n");
    return 1;
  }

  /* ---- MSM: bases {G, G}, scalars {2, 2} (Montgomery) -> 4G; compare X/Z^2 against the x of 2*(2G) computed by a
   * second MSM with bases {G, G, G, G} and scalars {1, 1, 1, 1} */
  uint64_t bases[4][8];
  for (int i = 0; i < 4; ++i) {
    memcpy(bases[i], ONE_R, 32);
    memcpy(bases[i] + 4, TWO_R, 32);
  }
  csh_bases_t h;
  rc = csh_bases_upload(CSH_BN254, CSH_G1, bases, 4, 0, &h);
  if (rc != CSH_OK) return fail("csh_bases_upload", rc);
  uint64_t s2[2][4], s1[4][4], fr_one[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
  memcpy(s2[0], FR_TWO, 32);
  memcpy(s2[1], FR_TWO, 32);
  for (int i = 0; i < 4; ++i) memcpy(s1[i], fr_one, 32);
  uint64_t p[12], q[12];
  rc = csh_msm(h, 0, 2, &s2[0][0], 1, p);
  if (rc != CSH_OK) return fail("csh_msm", rc);
  rc = csh_msm(h, 0, 4, &s1[0][0], 1, q);
  if (rc != CSH_OK) return fail("csh_msm", rc);
  csh_bases_free(h);
  /* both results are normalised Jacobian points (x, y, 1) of the same group element */
  if (memcmp(p, q, sizeof p) != 0) {
    fprintf(stderr, "2G + 2G != G + G + G + G\n");
    return 1;
  }
  printf("ok: NTT round trip (2^%d) and MSM agree\n", LOGN);
  csh_shutdown();
  return 0;
}
