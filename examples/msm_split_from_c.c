/* Plain-C caller of the split-MSM entry points (include/cosnarks_hip.h): ONE multi-scalar multiplication cut into contiguous
 * point ranges over every GPU of the node, driven from a single host thread -- the shape a Rust / Go / Java host would use.
 *
 *   gcc -std=c11 -Iinclude examples/msm_split_from_c.c -Lco-snarks_amd/lib -lcosnarks_hip \
 *       -Wl,-rpath,$PWD/co-snarks_amd/lib -o /tmp/msm_split_from_c && /tmp/msm_split_from_c [log2 points, default 18]
 *
 * Bases: k_i G with k_i = csh_util_splitmix64(seed + i) | 1 generated on each device for its own range
 * (csh_util_generate_bases_dev); scalars: all equal to 2 (Montgomery), so the expected result 2 * sum_i k_i G is computed a
 * second way -- as the one-range MSM on device 0 over the same family -- and compared limb for limb. Every exchange
 * (hipMemcpyPeer to the first device, host copies, grouped RCCL all-gather) must give the same Jacobian (X, Y, 1).
 * Exit status: 0 = all exchanges agree with the unsplit MSM; 3 = no usable GPU; 1 = mismatch. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cosnarks_hip.h"

/* BN254 Fr Montgomery: 2 * R mod r */
static const uint64_t FR_TWO[4] = {0x592c68389ffffff6ULL, 0x6df8ed2b3ec19a53ULL, 0xccdd46def0f28c5cULL, 0x1c14ef83340fbe5eULL};
#define MAX_DEV 16

static int fail(const char* what, int rc) {
  fprintf(stderr, "%s failed (%d): %s\n", what, rc, csh_last_error());
  return rc == CSH_ERR_NO_DEVICE ? 3 : 2;
}

int main(int argc, char** argv) {
  const int logn = argc > 1 ? atoi(argv[1]) : 18;
  const size_t n = (size_t)1 << logn;
  const uint64_t seed = 0x5EED;
  int ndev = 0, rc = csh_device_count(&ndev);
  if (rc != CSH_OK || ndev < 1) return fail("csh_device_count", rc ? rc : CSH_ERR_NO_DEVICE);
  if (ndev > MAX_DEV) ndev = MAX_DEV;
  /* on a one-GPU box the ranges share device 0: same code path, same exchange */
  const int parts = ndev > 1 ? ndev : 3;
  csh_bases_t bases[MAX_DEV];
  const uint64_t* scal[MAX_DEV];
  size_t off[MAX_DEV], cnt[MAX_DEV];
  uint64_t* host_sc = malloc(32 * n);
  if (!host_sc) return 2;
  for (size_t i = 0; i < n; ++i) memcpy(host_sc + 4 * i, FR_TWO, 32);
  size_t start = 0;
  for (int p = 0; p < parts; ++p) {
    const size_t len = p + 1 < parts ? n / parts : n - start;
    if ((rc = csh_init(ndev > 1 ? p : 0)) != CSH_OK) return fail("csh_init", rc);
    void *pts = NULL, *sc = NULL;
    if ((rc = csh_malloc(&pts, len * 64)) != CSH_OK) return fail("csh_malloc", rc);
    if ((rc = csh_util_generate_bases_dev(CSH_BN254, CSH_G1, seed + start, len, pts, NULL)) != CSH_OK) return fail("generate_bases", rc);
    if ((rc = csh_sync(NULL)) != CSH_OK) return fail("csh_sync", rc);
    if ((rc = csh_bases_upload_dev(CSH_BN254, CSH_G1, pts, len, 0, NULL, &bases[p])) != CSH_OK) return fail("csh_bases_upload_dev", rc);
    csh_free(pts);
    if ((rc = csh_malloc(&sc, len * 32)) != CSH_OK) return fail("csh_malloc", rc);
    if ((rc = csh_memcpy_h2d(sc, host_sc + 4 * start, len * 32)) != CSH_OK) return fail("csh_memcpy_h2d", rc);
    scal[p] = sc;
    off[p] = 0;
    cnt[p] = len;
    start += len;
  }
  /* reference: the same family as ONE range on device 0 */
  uint64_t want[12], got[12];
  {
    if ((rc = csh_init(0)) != CSH_OK) return fail("csh_init", rc);
    void* pts = NULL;
    csh_bases_t all;
    if ((rc = csh_malloc(&pts, n * 64)) != CSH_OK) return fail("csh_malloc", rc);
    if ((rc = csh_util_generate_bases_dev(CSH_BN254, CSH_G1, seed, n, pts, NULL)) != CSH_OK) return fail("generate_bases", rc);
    if ((rc = csh_sync(NULL)) != CSH_OK) return fail("csh_sync", rc);
    if ((rc = csh_bases_upload_dev(CSH_BN254, CSH_G1, pts, n, 0, NULL, &all)) != CSH_OK) return fail("csh_bases_upload_dev", rc);
    csh_free(pts);
    if ((rc = csh_msm(all, 0, n, host_sc, 1, want)) != CSH_OK) return fail("csh_msm", rc);
    csh_bases_free(all);
  }
  int bad = 0;
  const char* names[3] = {"hipMemcpyPeer", "host copies", "RCCL all-gather"};
  for (int mode = CSH_SPLIT_PEER; mode <= CSH_SPLIT_RCCL; ++mode) {
    csh_comm_t comms[MAX_DEV];
    const csh_comm_t* cp = NULL;
    if (mode == CSH_SPLIT_RCCL) {
      if (ndev < 2) continue; /* RCCL needs distinct devices per rank */
      int devs[MAX_DEV];
      for (int p = 0; p < parts; ++p) devs[p] = p;
      if ((rc = csh_comm_init_all(devs, parts, comms)) != CSH_OK) return fail("csh_comm_init_all", rc);
      cp = comms;
    }
    memset(got, 0, sizeof got);
    rc = csh_msm_split(bases, off, cnt, scal, (size_t)parts, 1, mode, cp, got);
    if (rc != CSH_OK) return fail("csh_msm_split", rc);
    const int ok = memcmp(got, want, sizeof want) == 0;
    printf("%-16s %d ranges over %d GPU(s), 2^%d points: %s\n", names[mode], parts, ndev, logn, ok ? "ok" : "MISMATCH");
    bad |= !ok;
    if (cp)
      for (int p = 0; p < parts; ++p) csh_comm_destroy(comms[p]);
  }
  for (int p = 0; p < parts; ++p) {
    csh_init(ndev > 1 ? p : 0);
    csh_free((void*)scal[p]);
    csh_bases_free(bases[p]);
  }
  free(host_sc);
  return bad ? 1 : 0;
}
