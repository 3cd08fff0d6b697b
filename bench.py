#!/usr/bin/env python3
"""bench.py -- BN254 G1 MSM points/s on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log-n 20]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N

A "step" = one pass of the hot path over one batch: one Pippenger MSM of 2^log_n BN254 G1 bases per rank
(BASELINE config 2: 2^20 uniform 254-bit scalars, the reference's `msm_unchecked` call, scalars in
Montgomery form) through the C ABI (`csh_msm_dev`), with bases and scalars already resident in HBM.
N > 1: ONE MSM of N*2^log_n points split by contiguous point ranges: every rank reduces its
range to per-window partial sums (`csh_msm_partial_dev`), the partials (a few KiB) are exchanged with an
RCCL all-gather over xGMI, and every rank folds them -- weak scaling, value = total points / time.

Extra objects on the JSON line: "roofline" (dominant kernel k_msm_accum, HIP-event timed on the launch
stream) and "cpu_baseline" (the oracle's C restatement timed on this box's host cores; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# k_msm_accum HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; see tools/pmc_summary.py).
# The kernel gathers every 64-byte base once per window (W = 17 at 2^20): 1.14 GB is inherent to Pippenger; the rest is
# the x2 FETCH_SIZE correction applied to 64-byte gathers (raw counter: 1.44 GB), the sorted-index reads and the
# partial-sum writes (0.14 GB).
PMC_TRAFFIC_BYTES = {20: 3026656721}
MADS_PER_MADD = 1467   # 10 products (6 mul 81 + 2 sqr 45 + fused 2x81) + 9 Montgomery reductions x 81, 9-limb 29-bit field
MAD_PEAK_T = 31.0      # measured v_mad_u64_u32 issue rate, T lane-ops/s


def secondary_metrics(hip, B, L, C, np, torch, dev, stream):
    """MSM at 2^24, NTT/iNTT at 2^22 (BASELINE config 3), Rep3 local_mul_vec, Groth16 prove on a synthetic 2^20 circuit."""
    out = {}
    torch.cuda.synchronize()
    # MSM 2^24
    n = 1 << 24
    pts = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    B._check(L.csh_util_generate_bases_dev(hip.BN254, hip.G1, C.c_uint64(77), C.c_size_t(n), C.c_void_p(pts.data_ptr()), C.c_void_p(stream)))
    h = C.c_void_p()
    B._check(L.csh_bases_upload_dev(hip.BN254, hip.G1, C.c_void_p(pts.data_ptr()), C.c_size_t(n), C.c_size_t(0), C.c_void_p(stream), C.byref(h)))
    del pts
    sc = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device=dev)
    sc[:, 3] >>= 1
    res = np.zeros(12, dtype=np.uint64)
    run = lambda: B._check(L.csh_msm_dev(h, C.c_size_t(0), C.c_size_t(n), C.c_void_p(sc.data_ptr()), 1, res.ctypes.data_as(C.c_void_p), C.c_void_p(stream)))
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    out["msm_bn254_g1_2p24"] = {"points_per_s": n / dt, "ms": dt * 1e3}
    L.csh_bases_free(h)
    del sc
    # NTT 2^22 (snarkjs root), data resident
    logn = 22
    r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    g = pow(pow(5, (r - 1) >> 28, r), 1 << (28 - logn), r) * ((1 << 256) % r) % r
    gen = np.array([(g >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
    dom = hip.Domain(hip.BN254, logn, gen)
    data = torch.randint(0, 1 << 62, (1 << logn, 4), dtype=torch.int64, device=dev)
    data[:, 3] >>= 1   # canonical (< r)
    # HIP events recorded on the stream the kernels run on (torch's default stream handle is 0 = the library's own
    # per-thread stream here, which torch.cuda.Event would not observe)
    e0, e1 = B.Event(), B.Event()
    dom.ifft_in_to_out_dev(data.data_ptr(), 1, stream)
    e0.record(stream)
    for _ in range(10):
        dom.ifft_in_to_out_dev(data.data_ptr(), 1, stream)
        dom.fft_out_to_in_dev(data.data_ptr(), 1, stream)
    e1.record(stream)
    ms = e0.elapsed_ms(e1) / 20
    modmuls = (1 << logn) // 2 * logn                      # one twiddle multiplication per butterfly
    out["ntt_bn254_2p22"] = {"elements_per_s": (1 << logn) / ms * 1e3, "ms": ms, "alg_GBps": 64.0 * (1 << logn) / ms / 1e6,
                             "hbm_peak_frac": 64.0 * (1 << logn) / ms / 1e6 / 8000.0,
                             # integer roofline: lazy-field modular multiplications/s against the measured 156 G/s of that
                             # multiplier; the NTT is ALU-bound, not HBM-bound
                             "modmul_per_s": modmuls / ms * 1e3, "modmul_peak_frac": modmuls / ms * 1e3 / 156e9}
    dom.free()
    del data
    # Rep3 local_mul_vec 2^24 (192 B/element)
    n = 1 << 24
    a = torch.randint(0, 1 << 62, (2 * n, 4), dtype=torch.int64, device=dev)
    b = torch.randint(0, 1 << 62, (2 * n, 4), dtype=torch.int64, device=dev)
    m = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device=dev)
    o = torch.empty((n, 4), dtype=torch.int64, device=dev)
    for x in (a, b, m):
        x[:, 3] >>= 1
    f = lambda: B._check(L.csh_rep3_local_mul_vec_dev(hip.BN254, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(m.data_ptr()),
                                                       C.c_void_p(o.data_ptr()), C.c_size_t(n), C.c_void_p(stream)))
    f()
    e0.record(stream)
    for _ in range(10):
        f()
    e1.record(stream)
    ms = e0.elapsed_ms(e1) / 10
    out["rep3_local_mul_vec_2p24"] = {"elements_per_s": n / ms * 1e3, "ms": ms, "alg_GBps": 192.0 * n / ms / 1e6, "hbm_peak_frac": 192.0 * n / ms / 1e6 / 8000.0}
    del a, b, m, o
    # Groth16 plain prove, synthetic 2^20-constraint circuit with a known-dlog key (closed-form checked), key resident
    from cosnarks_amd import groth16 as g16
    out["groth16_prove_synthetic_2p20"] = g16.bench_synthetic(hip.BN254, 20, 3, with_rep3=True)
    # BASELINE configs 1 and 4 on the reference's own circuits (tests/golden copies of test_vectors/Groth16/bn254): plain
    # prove of multiplier2 (domain 4) and poseidon (domain 256), and a three-party Rep3 prove of poseidon (in-process
    # parties sharing this GPU). Wall ms per proof incl. zkey parse + key upload; these sizes are launch-latency bound.
    gold = os.path.join(ROOT, "tests", "golden", "Groth16", "bn254")
    small = {}
    for circ in ("multiplier2", "poseidon"):
        zk = open(os.path.join(gold, circ, "circuit.zkey"), "rb").read()
        wt = open(os.path.join(gold, circ, "witness.wtns"), "rb").read()
        g16.prove_plain(hip.BN254, zk, wt, 123456789, 987654321)
        ts = []
        for _ in range(10):
            t0 = time.perf_counter()
            g16.prove_plain(hip.BN254, zk, wt, 123456789, 987654321)
            ts.append((time.perf_counter() - t0) * 1e3)
        small[f"plain_{circ}_ms"] = sorted(ts)[len(ts) // 2]
        if circ == "poseidon":
            g16.prove_rep3(hip.BN254, zk, wt, 7, 123456789, 987654321)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                g16.prove_rep3(hip.BN254, zk, wt, 7, 123456789, 987654321)
                ts.append((time.perf_counter() - t0) * 1e3)
            small["rep3_poseidon_3_parties_ms"] = sorted(ts)[len(ts) // 2]
    out["groth16_prove_reference_circuits"] = small
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary metrics (MSM 2^24, NTT 2^22, Groth16 prove ms)")
    args = ap.parse_args()

    import torch  # FIRST: torch bundles its own libamdhip64.so.7; our library must bind to the same runtime
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # One rank per GPU over RCCL is the real configuration. BENCH_DIST_BACKEND=gloo (with ranks folded onto the GPUs
    # that exist) is a debugging aid to exercise the N > 1 code path on a box with fewer GPUs than ranks.
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import ctypes as C

    import numpy as np

    import cosnarks_amd as hip
    from cosnarks_amd import bindings as B
    L = hip.lib()
    B._check(L.csh_init(dev_index))

    n = 1 << args.log_n
    stream = torch.cuda.current_stream().cuda_stream

    # ---- synthetic inputs, generated on the device (known-dlog bases, uniform 253-bit Montgomery scalars)
    seed = 0x00C0FFEE5EED + rank * (1 << 32)
    pts = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    B._check(L.csh_util_generate_bases_dev(hip.BN254, hip.G1, C.c_uint64(seed), C.c_size_t(n), C.c_void_p(pts.data_ptr()), C.c_void_p(stream)))
    torch.cuda.synchronize()
    bases_h = C.c_void_p()
    B._check(L.csh_bases_upload_dev(hip.BN254, hip.G1, C.c_void_p(pts.data_ptr()), C.c_size_t(n), C.c_size_t(0), C.c_void_p(stream), C.byref(bases_h)))
    del pts
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    sc = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device=dev, generator=g)
    sc = sc * 2 + torch.randint(0, 2, (n, 4), dtype=torch.int64, device=dev, generator=g)   # 63 random bits/limb
    sc[:, 3] >>= 2                                                                           # < 2^253 < r
    torch.cuda.synchronize()

    out = np.zeros(12, dtype=np.uint64)
    pbytes = hip.msm_partial_bytes(hip.BN254, hip.G1)
    part = torch.zeros(pbytes, dtype=torch.uint8, device=dev)
    from cosnarks_amd.distributed import allgather_and_fold

    def step():
        if world == 1:
            B._check(L.csh_msm_dev(bases_h, C.c_size_t(0), C.c_size_t(n), C.c_void_p(sc.data_ptr()), 1, out.ctypes.data_as(C.c_void_p), C.c_void_p(stream)))
            return out
        B._check(L.csh_msm_partial_dev(bases_h, C.c_size_t(0), C.c_size_t(n), C.c_void_p(sc.data_ptr()), 1, C.c_void_p(part.data_ptr()), C.c_void_p(stream)))
        return allgather_and_fold(part if backend == "nccl" else part.cpu(), hip.BN254, hip.G1, world, dist)   # RCCL all-gather over xGMI + host fold

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        B.sync()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * n * args.steps / dt

    # ---- per-kernel timing for the roofline object (separate untimed passes, HIP events on `stream`)
    roofline = None
    stage_ms = None
    def local_msm():  # the device part of a step on this rank (no collective): what the roofline object describes
        if world == 1:
            return step()
        B._check(L.csh_msm_partial_dev(bases_h, C.c_size_t(0), C.c_size_t(n), C.c_void_p(sc.data_ptr()), 1, C.c_void_p(part.data_ptr()), C.c_void_p(stream)))

    if True:  # every rank runs the same untimed passes (keeps the ranks in step); rank 0 reports
        os.environ["CSH_MSM_TIMING"] = "1"
        acc = []
        for _ in range(5):
            local_msm()
            acc.append(B.msm_last_timing())
        del os.environ["CSH_MSM_TIMING"]
        stage_ms = [float(np.mean([a[i] for a in acc])) for i in range(6)]
        c_bits, n_win, lane_len, n_seg = B.msm_last_params()
        t_acc = stage_ms[3] * 1e-3
        alg_bytes = n * 96.0                                 # 32 B scalar + 64 B affine base per point
        achieved = alg_bytes / t_acc / 1e9
        # HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/ (FETCH_SIZE x2
        # per the gfx950 correction + WRITE_SIZE, KiB -> bytes); measured for the default 2^20 workload only.
        traffic = PMC_TRAFFIC_BYTES.get(args.log_n)
        roofline = {"bound": "hbm", "kernel": "k_msm_accum<Bn254G1>", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                    "traffic_source": "profiles/r01_f_msm_bn254g1_2p20_pmc_hbm_bytes.csv" if traffic else None,
                    "note": "MSM is integer-ALU (v_mad_u64_u32 / v_mad_i64_i32) bound, not HBM bound; see DESIGN.md 3.1",
                    # integer roofline of the same kernel: mixed additions x 64-bit multiply-adds per addition (ISA count of
                    # the k_msm_accum<Bn254G1> loop body) over the measured issue peak of those instructions
                    # (tools/gpu_probe.py microbench, profiles/*probe*: 31e12 lane-ops/s)
                    "alu": {"unit": "Tmad/s", "achieved": round(n * n_win * MADS_PER_MADD / t_acc / 1e12, 2), "peak": MAD_PEAK_T,
                            "frac": round(n * n_win * MADS_PER_MADD / t_acc / 1e12 / MAD_PEAK_T, 3), "mads_per_madd": MADS_PER_MADD,
                            "madds_per_s": round(n * n_win / t_acc, 0)},
                    "msm_params": {"c": c_bits, "windows": n_win, "lane_len": lane_len, "segments": n_seg},
                    "stage_ms": {"hist": stage_ms[0], "scan": stage_ms[1], "scatter": stage_ms[2], "accum": stage_ms[3],
                                 "reduce": stage_ms[4], "total_device": stage_ms[5]}}

    # ---- correctness of the timed result: closed form (sum s_i k_i) * G via a second, tiny MSM
    check = None
    if not args.no_check and args.log_n <= 20:
        if world == 1:
            from tests.check_closed_form import closed_form_ok
            check = bool(closed_form_ok(hip, L, seed, n, sc.cpu().numpy(), res))
        else:
            # every rank contributes sum_i s_i k_i of its own range (32-byte integer), rank 0 checks the folded MSM
            from tests.check_closed_form import closed_form_ok_split, local_dlog_sum
            mine = local_dlog_sum(seed, n, sc.cpu().numpy())
            buf = torch.tensor(list(mine.to_bytes(32, "little")), dtype=torch.uint8, device=dev if backend == "nccl" else "cpu")
            allb = torch.empty(32 * world, dtype=torch.uint8, device=buf.device)
            dist.all_gather_into_tensor(allb, buf)
            if rank == 0:
                raw = bytes(allb.cpu().tolist())
                sums = [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(world)]
                check = bool(closed_form_ok_split(sums, res))

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            from oracle.cbridge import cpu_msm_baseline
            cpu_baseline = cpu_msm_baseline(target_seconds=12.0)
        except Exception as e:  # the baseline is a reported extra, never part of the measured path
            cpu_baseline = {"error": repr(e)}

    # ---- secondary metrics of BASELINE.json (untimed extras, N = 1 only): MSM 2^24, NTT 2^22, Groth16 prove ms
    extras = None
    if rank == 0 and world == 1 and not args.no_extras:
        extras = {}
        try:
            extras.update(secondary_metrics(hip, B, L, C, np, torch, dev, stream))
        except Exception as e:
            extras["error"] = repr(e)

    if rank == 0:
        line = {
            "metric": "BN254 G1 MSM points/sec", "value": value, "unit": "points/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "i32x9 29-bit-limb Montgomery, i64 accumulate (BN254 Fq 254-bit)", "data": "synthetic",
            "config": {"workload": f"BN254 G1 Pippenger MSM, 2^{args.log_n} uniform scalars/points per GPU (BASELINE config 2)",
                       "points_per_gpu": n, "split": "contiguous point ranges + RCCL all-gather of window partials" if world > 1 else "single GPU"},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "result_check": check, "secondary": extras,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
