/*
 * cosnarks_hip.h -- C ABI of the MI355X (gfx950) proof-generation hot path for co-snarks.
 *
 * This is the drop-in boundary: the reference (TaceoLabs/co-snarks, 100 % Rust) has no FFI today; its
 * hot path sits behind Rust traits/free functions whose arithmetic lives in the un-vendored crates
 * taceo-ark-algebra 0.1.0 / ark-poly 0.6.0.  Each entry point below names the reference interface it
 * replaces (paths relative to the reference root).  INTEGRATION.md shows the Rust `extern "C"` block and
 * the trait implementors (`HipCircomReduction: R1CSToQAP`, `Hip*Groth16Driver: CircomGroth16Prover`)
 * that bind it.
 *
 * Data conventions (identical to arkworks' in-memory layout, so Rust slices can be passed as-is):
 *  - field element  = N64 little-endian u64 limbs of x*R mod p (Montgomery, R = 2^(64*N64));
 *                     Fr: 4 limbs (32 B) on both curves; Fq: 4 limbs (BN254) / 6 limbs (BLS12-381).
 *  - affine point   = x || y (G2: x.c0 || x.c1 || y.c0 || y.c1), Montgomery; infinity = all-zero bytes
 *                     (the zkey convention).  `stride_bytes` lets the caller pass arkworks `Affine`
 *                     (x, y, infinity flag, padding) without repacking -- the flag byte is ignored.
 *  - result point   = arkworks `Projective` = Jacobian (X, Y, Z), Montgomery; infinity = (1, 1, 0).
 *                     The library always returns Z in {0, 1} (normalised), which is a valid Jacobian
 *                     representative; the reference compares/serialises affine forms only
 *                     (co-circom/co-groth16/src/groth16.rs:333-337).
 *  - Rep3 share     = {a, b} = 2 consecutive Fr elements (64 B)
 *                     (mpc-core/src/protocols/rep3/arithmetic/types.rs:21-28).
 *  - Shamir share   = 1 Fr element (repr(transparent), shamir/arithmetic/types.rs:9-13).
 *
 * Pointers named `*_dev` / functions suffixed `_dev` take DEVICE pointers and a HIP stream (NULL = the
 * library's per-thread stream) and are asynchronous unless they return data to the host; all other
 * pointers are HOST pointers and the call is synchronous.  All functions return 0 on success or a
 * negative csh_status; csh_last_error() returns a thread-local message.  Every entry point is
 * re-entrant and thread-safe (the reference calls MSMs/NTTs concurrently from rayon workers:
 * groth16.rs:227-294, reduction.rs:135-178).  There is no CPU fallback: without a HIP device every
 * compute entry point fails with CSH_ERR_NO_DEVICE.
 *
 * Host threads the library itself starts (all joined before the call returns unless said otherwise; none touches caller memory after
 * the return):
 *  - entry points that move >= 4 MiB between caller memory and the device through HOST pointers -- csh_groth16_witness_map / _masks /
 *    _libsnark*, csh_groth16_h*, the host-pointer transforms and share-vector calls (no _dev suffix), csh_msm / csh_msm_shares with host
 *    scalars -- start ONE short-lived thread per result that populates the destination's pages while the device works ("host_populate"),
 *    and hand the chunks of a staged transfer ("host_d2h" / "host_h2d" = 1, the default) to a process-wide pool of up to eight parked
 *    copier threads, created on first use and kept for the life of the process; they touch caller memory only while the call that
 *    enlisted them is running. csh_tune_set("host_populate", 0) + ("host_d2h", 0) + ("host_h2d", 0) turns all of it off (the call then
 *    never leaves the calling thread, and the runtime pins the caller's pages itself);
 *  - csh_msm with host scalars ("msm_share_uploads" = 2, the default): a call that finds another call of the same process uploading the
 *    SAME scalar slice (pointer, length, device, curve, encoding) hands that call its request and blocks; the uploading call's thread
 *    runs both as one csh_msm_multi_dev and writes the result into the waiting call's `out` before either returns;
 *  - csh_comm_init_rank with nranks > 1 runs the collective ncclCommInitRank on a helper thread against "comm_timeout_ms"; a helper
 *    whose peers never arrive is ABANDONED (still blocked inside RCCL after the call returned its error): leave such a process
 *    through _exit.
 *  Everything else (csh_msm_dev / _multi_dev / _partial_dev, csh_msm_split*, every other *_dev entry point) runs on the calling thread
 *  only. Page-locked staging memory is bounded: at most 32 MiB per (thread lane, stream, slot) -- larger transfers cycle through a ring
 *  of 2 MiB chunks (round 6).
 */
#ifndef COSNARKS_HIP_H
#define COSNARKS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CSH_GRUMPKIN (MSM entry points only, group CSH_G1): the BN254 cycle curve y^2 = x^3 - 17 over BN254 Fr with scalar field
 * BN254 Fq -- HonkCurve::fast_msm for short_weierstrass::Projective<GrumpkinConfig> (co-noir-common/src/honk_curve.rs:163-177). */
/* CSH_BLS12_377: the curve of the reference's LibSnarkReduction fixtures (Groth16::<Bls12_377>::plain_prove::<LibSnarkReduction>,
 * co-groth16/src/lib.rs:231-300): `field_of` of the NTT / share-vector / sparse-matrix / reduction entry points (its scalar field Fr)
 * and, since round 6, both MSM groups (G1: y^2 = x^3 + 1 over the 377-bit Fq, 96-byte affine points; G2 over Fq[u]/(u^2 + 5),
 * 192 bytes) through every csh_bases_* / csh_msm* entry point, tables and split ranges included. */
typedef enum { CSH_BN254 = 0, CSH_BLS12_381 = 1, CSH_GRUMPKIN = 2, CSH_BLS12_377 = 3 } csh_curve_t;
typedef enum { CSH_G1 = 0, CSH_G2 = 1 } csh_group_t;

typedef enum {
  CSH_OK = 0,
  CSH_ERR_INVALID = -1,    /* bad argument */
  CSH_ERR_NO_DEVICE = -2,  /* no HIP device / runtime failure at init */
  CSH_ERR_HIP = -3,        /* a HIP runtime call failed (see csh_last_error) */
  CSH_ERR_OOM = -4,
  CSH_ERR_DOMAIN = -5      /* "Polynomial Degree too large" (reduction.rs:87-94) */
} csh_status;

typedef struct csh_bases_s* csh_bases_t;   /* proving-key query resident on the device */
typedef struct csh_domain_s* csh_domain_t; /* radix-2 evaluation domain (twiddles on device) */

/* ---- context ------------------------------------------------------------------------------- */
int csh_init(int device);            /* select device for the calling thread, create context lazily */
int csh_shutdown(void);              /* free all cached workspaces/streams of every device */
const char* csh_last_error(void);
const char* csh_version(void);
int csh_device_count(int* count);
/* Process-wide tuning knobs for A/B runs and tests (initial values come from the environment once, at load: CSH_MSM_C ...;
 * no entry point reads the environment afterwards). Keys: "msm_c" (forced window width, 0 = cost model), "msm_l" (entries per
 * accumulate lane, 0 = round-count cost model), "msm_seg_buckets" (buckets per window-reduction segment, 0 = as many segments
 * as fit one round of waves), "msm_timing" (record csh_msm_last_timing), "msm_no_table", "msm_multi_overlap", "acc_blk",
 * "sort_two_level" (-1 auto), "vec_max_blocks", "ntt_lazy", "ntt_threads", "ntt_variant" (A/B forms with equal results; its timing-experiment
 * bits 12-13 / 16-19 and "allow_unmasked_rep3" return WRONG / unmasked results and exist only in builds with -DCSH_EXPERIMENTS: the product
 * library refuses them with CSH_ERR_INVALID and never takes them from the environment), "msm_wide_lb" / "msm_wide_chunks" (wide sort
 * stage of the fixed-base MSM: log2 buckets per second-level partition 8 .. 11, first-level blocks; 0 = defaults), "msm_variant" (bit mask of non-default kernel forms, same results: bit 0 = window reduction
 * lane-serial on G1 / four-lane on G2, bit 1 = the other form of the G2 accumulate kernel (BN254 G2: two lanes per point instead of whole points; BLS12-381 G2: whole
 * points instead of two lanes per point), bit 2 = lane-serial window
 * reduction on G2, bit 3 = 8-byte sort records at every size, bit 4 = merge fused into the window reduction, bit 5 = level 2 of the
 * two-level sort with one block per partition instead of one per tile-sized slice), "h_unfused", "h_table_cache" (1 = default: the witness map's scaled coset table stays with the
 * domain after the first call, 32 bytes per point up to 2^25 points; 0 = rebuilt per call), "comm_timeout_ms" (csh_comm_init_rank),
 * "host_populate" (results of >= 4 MiB handed back in pageable memory: low byte = host threads that populate the destination's
 * pages while the device works, 0 = none; bit 8 = transparent-huge-page hint on the range; default 0x101), "host_d2h" (the copy of such
 * a result: 0 = one DMA into the caller's pages, 1 = default since round 5: staged through the lane's page-locked buffer and moved on by
 * host threads, 2 = direct and timed; two stalled copies in a row on one lane -- the stream + scratch leased to one host thread at a time, so
 * concurrent callers do not race on the detector -- switch every large transfer of the process, both directions, to the staged paths for
 * the next 4096 transfers: the stall is a process-wide condition of the runtime having pinned caller memory that was unmapped since, and
 * it only goes away while the runtime sees no caller memory at all; counter "stat_stage_all_switches"),
 * "host_h2d" (uploads of >= 4 MiB from caller memory: 0 = one copy from the caller's pages, 1 = staged through the lane's page-locked
 * buffers in 2 MiB chunks copied by host threads -- the driver never pins caller memory; the default, 2 = direct and timed, feeding the same
 * process-wide switch as "host_d2h"; counters "stat_h2d_slow", "stat_h2d_staged". Staging is the default in both directions because a
 * process whose runtime has pinned caller memory that was unmapped afterwards can stall 10-30 ms per later call for the rest of its life
 * (DESIGN.md 3.4), and that state cannot be left once entered),
 * "msm_share_uploads" (concurrent csh_msm calls handed the same host scalar slice: 1 = they share one upload; 2 = default since round 6:
 * the calls that arrive while the first one is uploading are also RUN by it, as one csh_msm_multi_dev -- one digit sort for the handles of
 * equal length and offset -- and return when it has written their results; 0 = every call uploads and runs for itself; counter
 * "stat_uploads_shared"), "host_timing" (diagnostics: phase times of the host-facing witness map in "stat_wm_h2d_us" / "_dev_us" / "_d2h_us"),
 * "msm_balanced" (1 = default: the MSM's windows share the scalar bits evenly, widths c and c - 1; 0 = uniform c-bit windows),
 * "msm_w" (balanced windows: forced number of windows, 0 = the tuned count). Read-only counters
 * (csh_tune_get): "stat_arena_grows", "stat_lanes", "stat_populate_us", "stat_join_wait_us", "stat_finish_us", "stat_d2h_slow",
 * "stat_d2h_staged". */
int csh_tune_set(const char* key, int value);
int csh_tune_get(const char* key, int* value);

/* plain device-memory plumbing for harnesses without their own HIP binding (tests, the Rust shim) */
/* the device the calling thread is bound to (csh_init, default 0): handles (bases, domains, matrices) are per device */
int csh_current_device(int* device);
int csh_malloc(void** dev_ptr, size_t bytes);
int csh_free(void* dev_ptr);
int csh_memcpy_h2d(void* dev_dst, const void* host_src, size_t bytes);
/* waits first for whatever the CALLING thread queued with stream = NULL (its lane stream): a result some csh_*_dev call of this thread
 * is still producing is complete when the copy reads it */
int csh_memcpy_d2h(void* host_dst, const void* dev_src, size_t bytes);
/* device-to-device between GPUs (scalars of one prover to the GPU that runs one of its MSMs). stream NULL = synchronous. */
int csh_memcpy_peer(void* dst, int dst_device, const void* src, int src_device, size_t bytes, void* stream);
int csh_sync(void* stream);
/* out[i] = component `comp` of the i-th share (ncomp 32-byte field elements per share), device to device: the
 * `to_half_share` map over the witness (groth16.rs:159-163; Rep3 = take `.a`, mpc/rep3.rs:120-122) without a host pass. */
int csh_extract_component_dev(const uint64_t* shares_dev, uint32_t ncomp, uint32_t comp, size_t n, uint64_t* out_dev, void* stream);

/* ---- MSM ------------------------------------------------------------------------------------
 * Replaces taceo_ark_algebra::msm::msm_unchecked(&[Affine<C>], &[F]) -> Projective<C> and
 * msm::msm_bigint(&[Affine<C>], &[F::BigInt]); call sites groth16.rs:193-194, mpc/plain.rs:66-74,
 * mpc/rep3.rs:124-132, mpc/shamir.rs:111-119, mpc-core/src/protocols/rep3/pointshare.rs:201-222,
 * shamir/pointshare.rs:207-225, co-noir/co-noir-common/src/honk_curve.rs:81-83.
 * Bases (ProvingKey queries a_query/b_g1_query/b_g2_query/l_query/h_query, groth16.rs:219-290) are
 * uploaded once and reused across proofs. */
int csh_bases_upload(csh_curve_t curve, csh_group_t group, const void* affine_points, size_t n,
                     size_t stride_bytes /* 0 = packed */, csh_bases_t* out);
int csh_bases_upload_dev(csh_curve_t curve, csh_group_t group, const void* affine_points_dev, size_t n,
                         size_t stride_bytes, void* stream, csh_bases_t* out);
int csh_bases_len(csh_bases_t bases, size_t* n);
/* Optional, for bases reused across many MSMs (proving-key queries: uploaded once per key, groth16.rs:219-225): builds
 * fixed-base window tables 2^(c w) P_i on the device (c = 0: automatic; c in 4 .. 22; W x the memory of the points). MSMs on the
 * handle then put all windows into ONE set of 2^(c-1) buckets: one bucket reduction instead of W, no Horner over windows, and
 * -- with c = 17 .. 22, round 6 -- FEWER mixed additions per point than any plain plan (W = ceil((bits + 1) / c): 15 at c = 17, 13
 * at c = 20, against 17 / 16 at the plain plan's c = 15 / 16). Results are the same group elements. Handles with fewer than 1024
 * points are left unchanged. Measured on MI355X against the plain handle (profiles/r06_e_*, r06_f_*): BN254 G1 2^16 0.355 against
 * 0.40 ms, 2^18 0.56 / 0.71, 2^20 +15 %, 2^24 (c = 20) +15 .. 17 % points/s; BN254 G2 +14 %, BLS12-381 G1 / G2 +15 / +10 % at 2^20. */
int csh_bases_precompute(csh_bases_t bases, int c);
/* The same with `groups` (2..128) table rows instead of one per window: row k holds 2^(c W' k) P_i, W' = ceil(windows / groups),
 * so that windows w and w + W' k share a bucket set: W' bucket reductions instead of W and a host Horner over W' windows, for
 * groups x the memory of the points (c <= 16). groups >= the window count is the full merge above (the only form for c > 16).
 * Same results; replaces any tables already on the handle. */
int csh_bases_precompute_grouped(csh_bases_t bases, int c, int groups);
/* The table policy of the library for the queries of ONE proving key whose largest query has `key_points` points (host-only,
 * no device needed): window width *c_out and row count *rows_out to pass to csh_bases_precompute_grouped for every query of
 * the key (equal (c, rows) on all of them lets csh_msm_multi_dev share one digit pass), or *rows_out = 0 when tables do not pay
 * (keys below 2^14 or above 2^26 points). Round 6: one row per window (rows_out >= the window count of either scalar field), c = 17
 * from 2^15 to 3 * 2^20 points, c = 20 above; 2^14 .. 2^15: c <= 16. The C++ host mirror
 * (ProvingKey::build_tables) and the Rust bases cache (rust/co-groth16-hip/src/bases.rs) both take the policy from here. */
int csh_bases_table_policy(size_t key_points, int* c_out, int* rows_out);
/* Free the fixed-base tables of a handle (the plain points stay): the fallback when a key's tables do not fit the device. */
int csh_bases_drop_tables(csh_bases_t bases);
/* A copy of a handle -- points and fixed-base tables, as stored -- on GPU `device` (device-to-device, no re-encoding): how one
 * prover places its five independent query MSMs (groth16.rs:227-294, rayon_join5) on several GPUs, and how a party gets its own
 * copy of a key. Free with csh_bases_free. */
int csh_bases_clone(csh_bases_t bases, int device, csh_bases_t* out);
/* The same for the points [offset, offset + n) only (and the matching columns of every table row): a GPU that works on the k-th range
 * of every query (placement by range) holds 1/N of the key instead of all of it. Point i of the clone is point offset + i of the source. */
int csh_bases_clone_range(csh_bases_t bases, size_t offset, size_t n, int device, csh_bases_t* out);
int csh_bases_free(csh_bases_t bases);

/* sum_{i<n} scalars[i] * bases[offset+i].  "unchecked": the caller passes the shorter length
 * (honk_curve.rs:33-34).  scalars_are_montgomery=1 <=> msm_unchecked (&[F]); 0 <=> msm_bigint.
 * out_jacobian: 3 base-field elements (X, Y, Z) on the host. */
int csh_msm(csh_bases_t bases, size_t offset, size_t n, const uint64_t* scalars,
            int scalars_are_montgomery, void* out_jacobian);
/* One MSM per share component over the same bases, from ONE host vector of shares: shares = n entries of ncomp (1 or 2) consecutive
 * field elements, outs[c] = Jacobian result of component c. ncomp = 2 is the Rep3PointShare {a, b} of pointshare::msm_public_points
 * (mpc-core/src/protocols/rep3/pointshare.rs:201-222; call sites CircomPlonkProver::msm_public_points_g1, co-plonk/src/mpc/rep3.rs:170-175,
 * and NoirUltraHonkProver::msm_public_points, co-noir-common/src/mpc/rep3.rs:259-266, which unzip the shares on the host and run two
 * MSMs): here the share vector crosses PCIe once and the components are cut out on the device. */
int csh_msm_shares(csh_bases_t bases, size_t offset, size_t n, const uint64_t* shares, uint32_t ncomp, int scalars_are_montgomery,
                   void* const* outs_jacobian);
int csh_msm_dev(csh_bases_t bases, size_t offset, size_t n, const uint64_t* scalars_dev,
                int scalars_are_montgomery, void* out_jacobian_host, void* stream);
/* Split-MSM building blocks (the entry points below compose them): the un-normalised partial result of one range -- a
 * 32-byte header {magic, c, W} + up to 128 window sums as XYZZ points (4 base-field elements each, arkworks encoding) --
 * left ON THE DEVICE (synchronous: the buffer is complete when the call returns), and the host-side fold of gathered
 * partials. Ranges may use different window widths; an empty range (n = 0) folds as the identity. */
int csh_msm_partial_dev(csh_bases_t bases, size_t offset, size_t n, const uint64_t* scalars_dev,
                        int scalars_are_montgomery, void* out_xyzz_dev /* csh_msm_partial_bytes */,
                        void* stream);
/* ---- one MSM split over the GPUs of a node (BASELINE config 5) ---------------------------------------------
 * Same seam as csh_msm_dev -- the MSM closures of groth16.rs:227-294 -- for a query too large or too urgent for one GPU:
 * the points are cut into contiguous ranges (one csh_bases_t per GPU, uploaded once), every GPU reduces its range to window
 * sums, ONE exchange moves those partial buffers (csh_msm_partial_bytes each, <= 48 KiB) and the host folds them. The
 * exchange is RCCL ncclAllGather over xGMI (bound with dlopen: no link-time dependency, no torch), hipMemcpyPeer to a root
 * device, or one device-to-host copy per GPU. RCCL has no elliptic-curve reduction operator, so this is an all-gather. */
#define CSH_COMM_ID_BYTES 128
typedef struct csh_comm_s* csh_comm_t;
/* One process (or thread) per GPU: rank 0 draws an id (ncclGetUniqueId), ships the 128 bytes to the other ranks by whatever
 * channel the host has, every rank calls csh_comm_init_rank on the thread bound (csh_init) to its GPU. nranks == 1 with
 * id == NULL makes a local communicator without loading RCCL. A communicator is used by one thread at a time.
 * ncclCommInitRank is collective and cannot be interrupted, so with more than one rank the construction runs on a helper thread and the
 * caller waits for it against a deadline -- csh_tune_set("comm_timeout_ms", ms), default 120000, 0 = construct on the calling thread: a
 * rank whose peers never arrive gets CSH_ERR_HIP with "did not come up within ..." instead of a hang (the abandoned helper takes the
 * communicator down itself if the bootstrap ever completes). */
int csh_comm_unique_id(uint8_t id[CSH_COMM_ID_BYTES]);
int csh_comm_init_rank(const uint8_t id[CSH_COMM_ID_BYTES], int nranks, int rank, csh_comm_t* out);
/* One process driving `ndev` distinct GPUs (ncclCommInitAll): out[i] = rank i on devices[i]. */
int csh_comm_init_all(const int* devices, int ndev, csh_comm_t* out);
int csh_comm_info(csh_comm_t comm, int* rank, int* nranks, int* device); /* any out pointer may be NULL */
int csh_comm_destroy(csh_comm_t comm);
/* This rank's share of one split MSM: sum_{i<n} scalars[i] * bases[offset+i] over its own range -> window sums -> all-gather
 * over the communicator -> fold. Every rank receives the full result in out_jacobian (host, as csh_msm). Synchronous;
 * collective: every rank of the communicator must call it (n may be 0 on a rank). A rank whose own range fails on the device still
 * takes part in the exchange with an empty record, so that its peers return an error ("bad partial header") instead of waiting for
 * it; argument errors (NULL pointers, wrong device, range outside the bases) are reported before the collective and must be avoided
 * by the caller on every rank alike. */
int csh_msm_split_rank_dev(csh_comm_t comm, csh_bases_t bases, size_t offset, size_t n, const uint64_t* scalars_dev,
                           int scalars_are_montgomery, void* out_jacobian, void* stream);
/* One thread driving all GPUs: part i = counts[i] points from offsets[i] of bases[i] (a handle on any device; several parts
 * may share a device) with device scalars scalars_dev[i] on that device. The ranges run concurrently on their devices'
 * streams; `mode` picks the exchange. CSH_SPLIT_RCCL needs comms[i] = rank i of csh_comm_init_all over the parts' (distinct)
 * devices; `comms` is ignored otherwise. The calling thread's device binding is restored on return. Synchronous. */
typedef enum { CSH_SPLIT_PEER = 0 /* hipMemcpyPeerAsync to the first part's device */, CSH_SPLIT_HOST = 1 /* one D2H per part */,
               CSH_SPLIT_RCCL = 2 /* grouped ncclAllGather */ } csh_split_mode_t;
int csh_msm_split(const csh_bases_t* bases, const size_t* offsets, const size_t* counts, const uint64_t* const* scalars_dev,
                  size_t k, int scalars_are_montgomery, int mode, const csh_comm_t* comms, void* out_jacobian);

/* k MSMs over ONE device scalar vector -- the four aux-assignment MSMs of a Groth16 proof (a_query, b_g1_query, b_g2_query,
 * l_query: groth16.rs:237-284, calculate_coeff :179-203): the signed-digit decomposition and the bucket sort depend on the
 * scalars only and are computed once. bases[i]: handles of one curve (G1 and G2 may be mixed), each MSM takes n points
 * from offsets[i]; outs_host[i]: Jacobian as csh_msm. Synchronous.
 * On handles with fixed-base tables the sorted list holds table indices (row * handle length + offset + i), so consecutive handles
 * share it only while their (length, offset) pair repeats; a handle with another pair gets its own scatter (~0.2 ms at 2^20). A
 * prover that wants ONE sort for all four uploads the l query behind 1 + n_public padding points (any valid points; never read),
 * which gives it the length and the offset of the a / b queries: the C++ mirror does (host/types.hpp, Query::lead). */
int csh_msm_multi_dev(const csh_bases_t* bases, const size_t* offsets, size_t k, size_t n, const uint64_t* scalars_dev,
                      int scalars_are_montgomery, void* const* outs_host, void* stream);
int csh_msm_partial_bytes(csh_curve_t curve, csh_group_t group, size_t* bytes);
int csh_msm_fold_partials(csh_curve_t curve, csh_group_t group, const void* partials_host, size_t nparts,
                          void* out_jacobian);

/* ---- NTT ------------------------------------------------------------------------------------
 * Replaces taceo_ark_algebra::fft::Domain<F>::{new, with_group_gen, size, ifft_in_to_out,
 * fft_out_to_in}, fft::bit_reverse (reduction.rs:10, 93, 141-174, 249-328) and
 * ark_poly::EvaluationDomain::{fft, ifft} (co-plonk/src/mpc/plain.rs:149-161).
 * `ncomp` = field elements per entry: 1 for F / Shamir shares, 2 for Rep3 shares (DomainCoeff<F>). */
int csh_domain_create(csh_curve_t field_of, uint32_t log_n, const uint64_t group_gen[4] /* Montgomery; NULL =
                      arkworks default 2-adic root (Domain::new) */, csh_domain_t* out);
int csh_domain_size(csh_domain_t dom, size_t* n);
int csh_domain_free(csh_domain_t dom);
/* natural-order evaluations -> coefficients in bit-reversed order (scaled by 1/n) */
int csh_ifft_in_to_out(csh_domain_t dom, uint64_t* data, uint32_t ncomp);
/* bit-reversed coefficients -> natural-order evaluations */
int csh_fft_out_to_in(csh_domain_t dom, uint64_t* data, uint32_t ncomp);
/* natural -> natural (EvaluationDomain::{fft, ifft}); data holds exactly domain-size entries */
int csh_fft(csh_domain_t dom, uint64_t* data, uint32_t ncomp);
int csh_ifft(csh_domain_t dom, uint64_t* data, uint32_t ncomp);
int csh_bit_reverse(csh_curve_t field_of, uint64_t* data, uint32_t log_n, uint32_t ncomp);
/* bit_reversed_coset_table(shift, size) (reduction.rs:45-60): out[bitrev(i)] = shift^i */
int csh_coset_table(csh_domain_t dom, const uint64_t shift[4], uint64_t* out);

int csh_ifft_in_to_out_dev(csh_domain_t dom, uint64_t* data_dev, uint32_t ncomp, void* stream);
int csh_fft_out_to_in_dev(csh_domain_t dom, uint64_t* data_dev, uint32_t ncomp, void* stream);
int csh_fft_dev(csh_domain_t dom, uint64_t* data_dev, uint32_t ncomp, void* stream);
int csh_ifft_dev(csh_domain_t dom, uint64_t* data_dev, uint32_t ncomp, void* stream);
int csh_bit_reverse_dev(csh_curve_t field_of, uint64_t* data_dev, uint32_t log_n, uint32_t ncomp, void* stream);
int csh_coset_table_dev(csh_domain_t dom, const uint64_t shift[4], uint64_t* out_dev, void* stream);

/* ---- element-wise share arithmetic -----------------------------------------------------------
 * `field_of` selects Fr of the curve.  n = number of entries. In-place allowed (out == an input). */
/* out[i] = a[i]*b[i]: plain/Shamir local_mul_vec (mpc/plain.rs:83-89, shamir/arithmetic.rs:73-79) */
int csh_vec_mul(csh_curve_t field_of, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
/* out[i] = a[i] +/- b[i], ncomp components per entry (reduction.rs:185-190, rep3/arithmetic.rs:61-82) */
int csh_vec_add(csh_curve_t field_of, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, uint32_t ncomp);
int csh_vec_sub(csh_curve_t field_of, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, uint32_t ncomp);
/* v[i] *= table[i] for every component: distribute_powers_and_mul_by_const (mpc.rs:90-94,
 * mpc/rep3.rs:95-106, mpc/shamir.rs:85-96, mpc/plain.rs:91-98; half-share form reduction.rs:166-171) */
int csh_vec_mul_table(csh_curve_t field_of, uint64_t* v, const uint64_t* table, size_t n, uint32_t ncomp);
/* out[i] = l.a*r.a + l.a*r.b + l.b*r.a + mask[i]: Rep3 local_mul_vec (rep3/arithmetic.rs:132-146,
 * arithmetic/ops.rs:69-76); mask = Rep3Rand::masking_field_elements_vec (rngs.rs:137-156). The reference always adds the
 * mask: an unmasked product leaks cross terms once opened, so mask == NULL (and NULL masks / seeds of every protocol-1 entry
 * point below) is refused with CSH_ERR_INVALID (the override "allow_unmasked_rep3" exists only in builds with -DCSH_EXPERIMENTS; pass an all-zero mask vector to test the arithmetic). */
int csh_rep3_local_mul_vec(csh_curve_t field_of, const uint64_t* lhs_ab, const uint64_t* rhs_ab,
                           const uint64_t* mask, uint64_t* out, size_t n);
/* out[i] = in[i].a*x + in[i].b*y: translate_primefield_repshare_vec (bridges/rep3_to_shamir.rs:43-62) */
int csh_rep3_to_shamir_vec(csh_curve_t field_of, const uint64_t* in_ab, const uint64_t x[4],
                           const uint64_t y[4], uint64_t* out, size_t n);
/* out[i] = sum_k coeffs[k] * shares[k][i]: Shamir reconstruct / open_vec (shamir.rs:483-491,
 * shamir/arithmetic.rs:191-213); with coeffs = 1 it is Rep3 combine/open (rep3.rs:583-605,
 * rep3/arithmetic.rs:249-271). `shares` = k host pointers. */
int csh_lincomb(csh_curve_t field_of, const uint64_t* const* shares, const uint64_t* coeffs /* k*4 */,
                size_t k, uint64_t* out, size_t n);

/* Rep3 correlated masks generated on the device ("next" row f2): out[i] = from_be_bytes_mod_order(a_i) -
 * from_be_bytes_mod_order(b_i), a_i / b_i = the 32-byte chunks number elem_offset{1,2} + i of the ChaCha12 keystreams
 * of seed1 (own key) and seed2 (previous party's key): byte-compatible with Rep3Rand::masking_field_elements_vec
 * (mpc-core/src/protocols/rep3/rngs.rs:137-156; RngType = rand_chacha::ChaCha12Rng, mpc-core/src/lib.rs:13), so the
 * three parties' masks still cancel. The caller advances its two generators by 32*n bytes. */
int csh_rep3_masks(csh_curve_t field_of, const uint8_t seed1[32], uint64_t elem_offset1, const uint8_t seed2[32],
                   uint64_t elem_offset2, uint64_t* out, size_t n);
int csh_rep3_masks_dev(csh_curve_t field_of, const uint8_t seed1[32], uint64_t elem_offset1, const uint8_t seed2[32],
                       uint64_t elem_offset2, uint64_t* out_dev, size_t n, void* stream);

int csh_vec_mul_dev(csh_curve_t field_of, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, void* stream);
int csh_vec_add_dev(csh_curve_t field_of, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, uint32_t ncomp, void* stream);
int csh_vec_sub_dev(csh_curve_t field_of, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, uint32_t ncomp, void* stream);
int csh_vec_mul_table_dev(csh_curve_t field_of, uint64_t* v, const uint64_t* table, size_t n, uint32_t ncomp, void* stream);
int csh_rep3_local_mul_vec_dev(csh_curve_t field_of, const uint64_t* lhs_ab, const uint64_t* rhs_ab,
                               const uint64_t* mask, uint64_t* out, size_t n, void* stream);
int csh_rep3_to_shamir_vec_dev(csh_curve_t field_of, const uint64_t* in_ab, const uint64_t x[4],
                               const uint64_t y[4], uint64_t* out, size_t n, void* stream);
int csh_lincomb_dev(csh_curve_t field_of, const uint64_t* const* shares_dev /* host array of k device ptrs */,
                    const uint64_t* coeffs, size_t k, uint64_t* out, size_t n, void* stream);

/* ---- fused CircomReduction tail (reduction.rs:135-192) -------------------------------------------
 * Given the constraint evaluations a, b (natural order, n = domain size, ncomp comps per entry) computes
 * h[i] = (A*B - C)(shift * w^i) entirely on the device: 6 NTTs + 2 local_mul_vec + 3 table muls + 1 sub.
 * protocol: 0 plain / Shamir (ncomp 1, out = a*b), 1 Rep3 (ncomp 2, local mul with masks).
 * mask_c / mask_ab: the two mask vectors drawn (in this order) by the two local_mul_vec calls
 * (reduction.rs:160 then :182); NULL for protocol 0.  a and b are clobbered. */
int csh_groth16_h(csh_domain_t dom, const uint64_t shift[4], int protocol, uint64_t* a, uint64_t* b,
                  const uint64_t* mask_c, const uint64_t* mask_ab, uint64_t* h_out);
int csh_groth16_h_dev(csh_domain_t dom, const uint64_t shift[4], int protocol, uint64_t* a_dev, uint64_t* b_dev,
                      const uint64_t* mask_c_dev, const uint64_t* mask_ab_dev, uint64_t* h_out_dev, void* stream);
/* Rep3 variant with the two mask vectors generated on the device from the party's ChaCha12 keys: mask_c = chunks
 * [off, off+n), mask_ab = chunks [off+n, off+2n) of each stream (the order of the two local_mul_vec calls,
 * reduction.rs:160 then :182). Host pointers for a, b, h. */
int csh_groth16_h_rep3_seeded(csh_domain_t dom, const uint64_t shift[4], uint64_t* a, uint64_t* b, const uint8_t seed1[32],
                              uint64_t elem_offset1, const uint8_t seed2[32], uint64_t elem_offset2, uint64_t* h_out);

/* ---- LibSnarkReduction tail (reduction.rs:255-342) -------------------------------------------------
 * a, b: constraint evaluations incl. the promoted public inputs (ncomp per entry), c: half-share evaluations of the C
 * matrix (evaluate_constraint_half_share, 1 component), natural order over Domain::new(n) (csh_domain_create with a
 * NULL generator). generator = F::GENERATOR (Montgomery). h_out[i] = i-th coefficient of (A*B - C)/Z (natural order):
 * 7 NTTs, 1 local_mul_vec (mask = its mask vector, NULL for protocol 0), 4 table multiplications. a, b, c are clobbered.
 * The reference's fixtures for this path (Penumbra BLS12-377 circuits) lack their proving keys: parity is against the
 * oracle's restatement, itself checked on that fixture data with the identity H(t) Z(t) = A(t) B(t) - C(t) (tests/). */
int csh_groth16_h_libsnark(csh_domain_t dom, const uint64_t generator[4], int protocol, uint64_t* a, uint64_t* b, uint64_t* c,
                           const uint64_t* mask, uint64_t* h_out);
int csh_groth16_h_libsnark_dev(csh_domain_t dom, const uint64_t generator[4], int protocol, uint64_t* a_dev, uint64_t* b_dev,
                               uint64_t* c_dev, const uint64_t* mask_dev, uint64_t* h_out_dev, void* stream);

/* ---- sparse constraint evaluation on the device ("next" row f3) ------------------------------------------------
 * Replaces evaluate_constraint over a ConstraintMatrices side (reduction.rs:196-210) + the driver row kernels
 * (mpc/plain.rs:28-43, mpc/rep3.rs:31-49, mpc/shamir.rs:29-49). The matrix (CSR: row_ptr[n_rows+1], col_idx[nnz],
 * Montgomery coeffs[nnz]) is uploaded once per circuit. values = public_inputs || private_witness by column index.
 * protocol 0: plain / Shamir (1 component per value; public terms are added to every share);
 * protocol 1: Rep3 (witness entries are {a, b}; a public term coeff*pub goes to component a on party 0, to component
 *             b on party 1, nowhere on party 2 -- rep3/arithmetic.rs:52-58).
 * out has n_out entries (ncomp components each); rows >= n_rows are zero (the resize to domain_size). */
typedef struct csh_matrix_s* csh_matrix_t;
int csh_matrix_upload(csh_curve_t field_of, const uint64_t* row_ptr, const uint32_t* col_idx, const uint64_t* coeffs,
                      size_t n_rows, size_t nnz, csh_matrix_t* out);
int csh_matrix_free(csh_matrix_t m);
/* rows, non-zeros, the largest column index (every entry point below checks it against the n_public + n_witness the caller
 * states, host or device pointers alike) and the device the handle lives on; any pointer may be NULL */
int csh_matrix_info(csh_matrix_t m, size_t* n_rows, size_t* nnz, uint32_t* max_column, int* device);
int csh_evaluate_constraints_dev(csh_matrix_t m, int protocol, int party_id, const uint64_t* public_dev, size_t n_public,
                                 const uint64_t* witness_dev, size_t n_witness, uint64_t* out_dev, size_t n_out, void* stream);
/* witness_map_from_matrices of CircomReduction (reduction.rs:77-193) entirely on the device: evaluate A and B rows,
 * overwrite the public-input slots a[num_constraints .. +n_public] with the promoted public inputs (:111-113), then the
 * fused pipeline of csh_groth16_h. Host pointers; witness = n_witness entries (1 or 2 components). For protocol 1 the
 * masks come from the ChaCha12 seeds as in csh_groth16_h_rep3_seeded (required for protocol 1; NULL for protocol 0). */
int csh_groth16_witness_map(csh_domain_t dom, const uint64_t shift[4], int protocol, int party_id, csh_matrix_t a, csh_matrix_t b,
                            size_t num_constraints, const uint64_t* public_inputs, size_t n_public, const uint64_t* witness,
                            size_t n_witness, const uint8_t seed1[32], uint64_t elem_offset1, const uint8_t seed2[32],
                            uint64_t elem_offset2, uint64_t* h_out);
/* Same with the witness shares already on the device and h left on the device (feeds csh_msm_dev of h_query directly,
 * groth16.rs:286-292): no PCIe traffic besides the n_public public inputs (host pointer). */
int csh_groth16_witness_map_dev(csh_domain_t dom, const uint64_t shift[4], int protocol, int party_id, csh_matrix_t a, csh_matrix_t b,
                                size_t num_constraints, const uint64_t* public_inputs, size_t n_public, const uint64_t* witness_dev,
                                size_t n_witness, const uint8_t seed1[32], uint64_t elem_offset1, const uint8_t seed2[32],
                                uint64_t elem_offset2, uint64_t* h_out_dev, void* stream);
/* The same map with the two Rep3 mask vectors handed over by the CALLER instead of ChaCha12 seeds: mask_c is the vector the
 * "c: local_mul_vec" draws (reduction.rs:160), mask_ab the one of the last product (:182), n = domain size elements each; NULL for
 * protocol 0. This is the form an unchanged reference can drive in ONE call per witness map: `Rep3Rand`'s generators are private
 * (mpc-core/src/protocols/rep3/rngs.rs:83-86) but `masking_field_elements_vec` (rngs.rs:137-156) is public, and for a generic
 * driver `T::local_mul_vec` of two zero vectors returns exactly the mask (rep3/arithmetic.rs:132-146) -- so the implementor of
 * R1CSToQAP (reduction.rs:27-36) draws both vectors through the public surface, in the reference's order, and passes them here.
 * Host pointers: witness shares (+ masks) up, h down; the pages of h_out are populated from host threads while the device works, so
 * h_out may be freshly allocated, uninitialised memory (Vec::with_capacity + set_len). The _dev variant takes device pointers for the
 * witness shares, the masks and h (public_inputs stays a host pointer) and is asynchronous on `stream` like csh_groth16_witness_map_dev. */
int csh_groth16_witness_map_masks(csh_domain_t dom, const uint64_t shift[4], int protocol, int party_id, csh_matrix_t a, csh_matrix_t b,
                                  size_t num_constraints, const uint64_t* public_inputs, size_t n_public, const uint64_t* witness,
                                  size_t n_witness, const uint64_t* mask_c, const uint64_t* mask_ab, uint64_t* h_out);
int csh_groth16_witness_map_masks_dev(csh_domain_t dom, const uint64_t shift[4], int protocol, int party_id, csh_matrix_t a, csh_matrix_t b,
                                      size_t num_constraints, const uint64_t* public_inputs, size_t n_public, const uint64_t* witness_dev,
                                      size_t n_witness, const uint64_t* mask_c_dev, const uint64_t* mask_ab_dev, uint64_t* h_out_dev,
                                      void* stream);
/* LibSnarkReduction::witness_map_from_matrices (reduction.rs:241-342) on the device: rows of a, b through
 * evaluate_constraint, rows of c through evaluate_constraint_half_share (mpc/rep3.rs:51-74, mpc/shamir.rs:51-68,
 * mpc/plain.rs:45-60), then csh_groth16_h_libsnark. dom = Domain::new (NULL generator at csh_domain_create),
 * generator = F::GENERATOR. Host pointers; one mask vector from the seeds (required for protocol 1). */
int csh_groth16_witness_map_libsnark(csh_domain_t dom, const uint64_t generator[4], int protocol, int party_id, csh_matrix_t a,
                                     csh_matrix_t b, csh_matrix_t c, size_t num_constraints, const uint64_t* public_inputs,
                                     size_t n_public, const uint64_t* witness, size_t n_witness, const uint8_t seed1[32],
                                     uint64_t elem_offset1, const uint8_t seed2[32], uint64_t elem_offset2, uint64_t* h_out);
/* ... and with its ONE mask vector (the local_mul_vec of reduction.rs:289) drawn by the caller, as for csh_groth16_witness_map_masks. */
int csh_groth16_witness_map_libsnark_masks(csh_domain_t dom, const uint64_t generator[4], int protocol, int party_id, csh_matrix_t a,
                                           csh_matrix_t b, csh_matrix_t c, size_t num_constraints, const uint64_t* public_inputs,
                                           size_t n_public, const uint64_t* witness, size_t n_witness, const uint64_t* mask, uint64_t* h_out);

/* ---- measurement hooks (bench.py / profiles) ----------------------------------------------------------
 * HIP-event timing on the stream the kernels are launched on. */
int csh_event_create(void** ev);
int csh_event_record(void* ev, void* stream);
int csh_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);
int csh_event_destroy(void* ev);
/* Per-MSM stage timings (ms) of the last csh_msm*_dev call on this thread: [digits+histogram, scan,
 * scatter, bucket accumulate, bucket reduce, total]; valid only while csh_tune_set("msm_timing", 1) is in effect. */
int csh_msm_last_timing(float out_ms[6]);
/* Pipeline parameters of the last csh_msm*_dev call on this thread: [window bits c, windows W, entries per lane L,
 * reduction segments S]. */
int csh_msm_last_params(uint32_t out[4]);
/* The plan an n-point MSM on `curve` would run with on the calling thread's device (no device: a 256-CU part is assumed), without
 * running it: [window bits c, windows W, entries per accumulate lane L, window-reduction segments S, accumulate waves, SIMDs].
 * The accumulate kernel takes ceil(waves / SIMDs) rounds of L mixed additions; L and S are chosen so that every SIMD runs a
 * whole number of equal rounds. This is the plan of the G2 groups and of plans shared by several groups
 * (csh_msm_multi_dev); a G1 MSM on its own whose launch would leave SIMDs with fewer accumulate waves than fit them together (below
 * ~2^18 points) runs with a shorter lane, down to 8 entries: csh_msm_last_params reports what ran. Host-only: for capacity planning
 * and for tests of the planner. */
int csh_msm_plan(csh_curve_t curve, size_t n, uint32_t out[6]);

/* ---- synthetic inputs (bench / full-size parity) -------------------------------------------------------
 * out[i] = k_i * G (affine, packed), k_i = csh_util_splitmix64(seed + i) | 1: known discrete logs, so an MSM
 * over them has the closed form (sum_i s_i k_i mod r) * G at any size (SURVEY 8d "known-dlog" family). */
uint64_t csh_util_splitmix64(uint64_t x);
int csh_util_generate_bases_dev(csh_curve_t curve, csh_group_t group, uint64_t seed, size_t n, void* out_dev,
                                void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COSNARKS_HIP_H */
