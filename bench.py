#!/usr/bin/env python3
"""bench.py -- MSM points/s on MI355X (BASELINE.json metric), one process per GPU, everything through the C ABI.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload bn254_g1] [--log-n 20] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" = one pass of the hot path over one batch: one Pippenger MSM (the reference's `msm_unchecked` call: uniform scalars
in Montgomery form, bases and scalars already resident in HBM).
  N = 1   `csh_msm_dev` on 2^log_n points (default: BN254 G1, 2^20 = BASELINE config 2).
  N > 1   ONE MSM split by contiguous point ranges over the N ranks: every rank reduces its range to window sums,
          ONE RCCL all-gather over xGMI moves the partial buffers, every rank folds them -- all inside
          `csh_msm_split_rank_dev` (RCCL bound by the library itself; torch.distributed/gloo is only this harness's control
          plane: rendezvous, barriers, max-over-ranks timing).
          --scaling weak (default): 2^log_n points per rank;  --scaling strong: 2^log_n points in total (BASELINE config 5 is
          `--workload bls12_381_g1|bls12_381_g2 --log-n 24 --scaling strong`).
`value` = points of the whole job per second. The timed result is checked against the closed form (sum s_i k_i) G at every N.

Extra objects on the JSON line: "roofline" (dominant kernel k_msm_accum, HIP-event timed on the launch stream; constants
and PMC traffic read from profiles/roofline_inputs.json), "cpu_baseline" (oracle/c `oc_msm_fast` on this box's host cores on
the step's own inputs; rank 0, N = 1 only) and "secondary" (MSM 2^24, NTT 2^22, Rep3 local_mul_vec, Groth16 prove ms, and the
config-5 workloads at this N).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
try:
    AFFINITY0 = os.sched_getaffinity(0)             # before any OpenMP runtime pins this thread
    HOST_CPUS = len(AFFINITY0)
except Exception:
    AFFINITY0 = None
    HOST_CPUS = os.cpu_count() or 1
os.environ.setdefault("OMP_PROC_BIND", "spread")   # cpu_baseline leg: one pinned OpenMP thread per core
os.environ.setdefault("OMP_PLACES", "cores")

WORKLOADS = {
    # name: (curve id, group id, bytes per affine point, label, arithmetic type)
    "bn254_g1": (0, 0, 64, "BN254 G1", "i32x9 29-bit-limb lazy Montgomery, i64 accumulate (BN254 Fq 254-bit)"),
    "bn254_g2": (0, 1, 128, "BN254 G2", "Fp2 over i32x9 29-bit-limb lazy Montgomery, i64 accumulate (BN254 Fq2)"),
    "bls12_381_g1": (1, 0, 96, "BLS12-381 G1", "i32x14 28-bit-limb lazy Montgomery, i64 accumulate (BLS12-381 Fq 381-bit)"),
    "bls12_381_g2": (1, 1, 192, "BLS12-381 G2", "Fp2 over i32x14 28-bit-limb lazy Montgomery, i64 accumulate (BLS12-381 Fq2)"),
    "grumpkin_g1": (2, 0, 64, "Grumpkin G1", "i32x9 29-bit-limb lazy Montgomery, i64 accumulate (BN254 Fr 254-bit)"),
    # the curve of the reference's LibSnarkReduction fixtures (co-circom/co-groth16/src/lib.rs:231-300); Fq2 = Fq[u]/(u^2 + 5)
    "bls12_377_g1": (3, 0, 96, "BLS12-377 G1", "i32x14 28-bit-limb lazy Montgomery, i64 accumulate (BLS12-377 Fq 377-bit)"),
    "bls12_377_g2": (3, 1, 192, "BLS12-377 G2", "Fp2 (u^2 = -5) over i32x14 28-bit-limb lazy Montgomery, i64 accumulate (BLS12-377 Fq2)"),
}
CURVE_NAMES = {0: "bn254", 1: "bls12_381", 2: "grumpkin", 3: "bls12_377"}
SEED = 0x00C0FFEE5EED


# scalar-field moduli (BN254 Fr, BLS12-381 Fr, Grumpkin Fr = BN254 Fq)
SCALAR_MODULUS = {
    0: 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    1: 52435875175126190479447740508185965837690552500527637822603658699938581184513,
    2: 21888242871839275222246405745257275088696311157297823662689037894645226208583,
    3: 8444461749428370424248824938781546531375899335154063827935233455917409239041,
}


def ge_modulus(torch, v, modulus):
    """Row-wise (v >= modulus) for n x 4 little-endian u64 limbs held as int64 bit patterns. Unsigned limb order is mapped onto signed
    order on BOTH sides by subtracting 2^63: the tensor limb u becomes u ^ (1 << 63) = u - 2^63 as a signed value, the modulus limb m
    becomes m - 2^63 (the top limb of a < 2^255 modulus is below 2^63 and compares as it is)."""
    m = [(modulus >> (64 * i)) & (2**64 - 1) for i in range(4)]
    assert m[3] < (1 << 63)
    flip = -(1 << 63)
    ms = [x - (1 << 63) if i < 3 else x for i, x in enumerate(m)]
    ge = torch.ones(v.shape[0], dtype=torch.bool, device=v.device)     # equal so far => (v >= m) holds on the empty suffix
    for i in range(4):                                                 # from the least significant limb up
        x = v[:, i] ^ flip if i < 3 else v[:, i]
        ge = (x > ms[i]) | ((x == ms[i]) & ge)
    return ge


def uniform_scalars(torch, dev, n, modulus, seed):
    """n x 4 little-endian u64 limbs (as int64 bit patterns), uniform in [0, modulus) by rejection sampling (SURVEY 8d config 2:
    "random scalars"): every limb carries 64 random bits, the top limb as many as the modulus has, draws >= modulus are redrawn.
    Taken as Montgomery encodings they decode to uniform field elements."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    top_bits = modulus.bit_length() - 192

    def draw(k):
        lo = torch.randint(0, 1 << 32, (k, 4), dtype=torch.int64, device=dev, generator=g)
        hi = torch.randint(-(1 << 31), 1 << 31, (k, 4), dtype=torch.int64, device=dev, generator=g)
        v = (hi << 32) | lo
        v[:, 3] = (v[:, 3] >> (64 - top_bits)) & ((1 << top_bits) - 1)     # logical shift: top_bits random bits
        return v

    sc = draw(n)
    bad = ge_modulus(torch, sc, modulus)
    while bool(bad.any()):
        idx = bad.nonzero().flatten()
        sc[idx] = draw(idx.numel())
        bad = ge_modulus(torch, sc, modulus)
    return sc


def load_roofline_inputs():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "roofline_inputs.json")))
    except Exception:
        return {}


class Ctx:
    """Everything a workload needs: the library, the device, the stream, the harness's process group."""

    def __init__(self, args):
        import torch  # FIRST: torch bundles its own libamdhip64.so.7; our library must bind to the same runtime
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        assert self.world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={self.world}"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # BENCH_FOLD_RANKS=1 folds the ranks onto the GPUs that exist (debugging the N > 1 code path on a smaller box; the
        # exchange then falls back to the harness all-gather because RCCL cannot put two ranks on one device).
        self.folded = bool(os.environ.get("BENCH_FOLD_RANKS"))
        self.dev_index = local_rank % max(1, torch.cuda.device_count()) if self.folded else local_rank
        torch.cuda.set_device(self.dev_index)
        self.dev = torch.device("cuda", self.dev_index)
        if self.world > 1:
            if os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # one node: never depend on the container's hostname resolving
            dist.init_process_group("gloo", rank=self.rank, world_size=self.world)   # control plane only
        import ctypes as C

        import numpy as np

        import cosnarks_amd as hip
        from cosnarks_amd import bindings as B
        self.C, self.np, self.hip, self.B = C, np, hip, B
        self.L = hip.lib()
        # OMP_PROC_BIND (set above for the cpu_baseline leg) makes libgomp bind the INITIAL thread to one core the moment the library is
        # loaded (torch brings it in), and every host thread created afterwards inherits that one-core mask: the five MSM threads of a
        # prove, the library's page-population / copy helpers and the mirror's parallel loops all ran on ONE core in rounds 1-4's bench
        # process (round 5: a Rep3 mask draw took 67 ms in here against 6 ms in a process of its own, profiles/archive/r05_i_mask_draw_probe.log).
        # The main thread gets its full mask back before anything is measured; OpenMP's own workers keep their places.
        self.affinity_after_imports = len(os.sched_getaffinity(0)) if AFFINITY0 is not None else None
        if AFFINITY0 is not None:
            try:
                os.sched_setaffinity(0, AFFINITY0)
            except OSError:
                pass
        B._check(self.L.csh_init(self.dev_index))
        self.stream = torch.cuda.current_stream().cuda_stream
        self.comm = None
        self.hard_exit = False   # set when csh_comm_init_rank timed out: a detached helper thread is still blocked inside ncclCommInitRank
        self.exchange = "single GPU"
        self.exchange_mode = "none"       # "rccl" | "harness_gloo" | "none": what the timed steps of an N > 1 run actually used
        self.rccl_ranks_seen = 0          # ranks of the RCCL communicator the library built (0: no communicator)
        if self.world > 1:
            self._make_comm(args)

    def _make_comm(self, args):
        """RCCL communicator behind the C ABI: rank 0 draws the id, gloo ships the 128 bytes."""
        torch, dist, B = self.torch, self.dist, self.B
        ok, err = 1, ""
        want_rccl = args.exchange == "rccl" and (not self.folded or bool(os.environ.get("BENCH_FORCE_RCCL")))   # forced + folded: exercises the fallback
        if want_rccl:
            try:
                uid = torch.zeros(B.COMM_ID_BYTES, dtype=torch.uint8)
                if self.rank == 0:
                    uid = torch.tensor(list(B.comm_unique_id()), dtype=torch.uint8)
                dist.broadcast(uid, 0)
                # ncclCommInitRank is collective: the library makes the blocking call on a helper thread and waits for it against a deadline
                # (csh_comm_init_rank, tune comm_timeout_ms), so a wedged bootstrap comes back as an error on every rank and the ranks
                # agree below to fall back to the harness exchange. The helper thread of a timed-out call stays blocked inside RCCL: the
                # process then leaves through os._exit once its line is printed (self.hard_exit), never through interpreter teardown.
                B.tune_set("comm_timeout_ms", int(float(os.environ.get("BENCH_RCCL_INIT_TIMEOUT", "120")) * 1000))
                self.comm = B.Comm.init_rank(bytes(uid.tolist()), self.world, self.rank)
            except Exception as e:  # noqa: BLE001
                ok, err = 0, repr(e)
                if "did not come up within" in err:
                    self.hard_exit = True
        flag = torch.tensor([ok if want_rccl else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            self.exchange_mode = "rccl"
            self.rccl_ranks_seen = self.world
            self.exchange = "csh_msm_split_rank_dev: RCCL ncclAllGather of window partials over xGMI, behind the C ABI"
        else:
            if self.comm is not None:
                self.comm.destroy()
                self.comm = None
            self.exchange_mode = "harness_gloo"
            self.exchange = "csh_msm_partial_dev + harness all-gather (gloo) + csh_msm_fold_partials" + (f" [RCCL path unavailable: {err}]" if err else "")

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()
        self.B.sync()

    def max_over_ranks(self, x: float) -> float:
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


class MsmJob:
    """One rank's share of one MSM: known-dlog bases [start, start + n) of the global family k_i = splitmix64(SEED + i) | 1,
    uniform 253-bit scalars (valid Montgomery encodings of uniform field elements), both resident in HBM."""

    def __init__(self, cx: Ctx, workload: str, start: int, n: int, scalar_seed: int):
        C, torch, B, L = cx.C, cx.torch, cx.B, cx.L
        self.cx, self.workload, self.start, self.n = cx, workload, start, n
        self.curve, self.group, self.pbytes, self.label, self.dtype = WORKLOADS[workload]
        pts = torch.empty(n * self.pbytes, dtype=torch.uint8, device=cx.dev)
        B._check(L.csh_util_generate_bases_dev(self.curve, self.group, C.c_uint64((SEED + start) & (2**64 - 1)), C.c_size_t(n), C.c_void_p(pts.data_ptr()), C.c_void_p(cx.stream)))
        torch.cuda.synchronize()
        B.sync()
        self.h = C.c_void_p()
        B._check(L.csh_bases_upload_dev(self.curve, self.group, C.c_void_p(pts.data_ptr()), C.c_size_t(n), C.c_size_t(0), C.c_void_p(cx.stream), C.byref(self.h)))
        self.pts_host = None
        self._pts_dev = pts          # kept only until the caller asks for a host copy (cpu_baseline) or drops it
        self.sc = uniform_scalars(torch, cx.dev, n, SCALAR_MODULUS[self.curve], scalar_seed)
        torch.cuda.synchronize()
        self.out = cx.np.zeros(3 * self.pbytes // 16, dtype=cx.np.uint64)
        self.part = None

    def drop_point_copy(self):
        self._pts_dev = None

    def points_to_host(self):
        p = self._pts_dev.cpu().numpy().view(self.cx.np.uint64).reshape(self.n, -1)
        self._pts_dev = None
        return p

    def step(self):
        cx, C, B, L = self.cx, self.cx.C, self.cx.B, self.cx.L
        if cx.world == 1:
            B._check(L.csh_msm_dev(self.h, C.c_size_t(0), C.c_size_t(self.n), C.c_void_p(self.sc.data_ptr()), 1, self.out.ctypes.data_as(C.c_void_p), C.c_void_p(cx.stream)))
        elif cx.comm is not None:
            B._check(L.csh_msm_split_rank_dev(cx.comm.h, self.h, C.c_size_t(0), C.c_size_t(self.n), C.c_void_p(self.sc.data_ptr()), 1,
                                              self.out.ctypes.data_as(C.c_void_p), C.c_void_p(cx.stream)))
        else:
            from cosnarks_amd.distributed import allgather_and_fold
            if self.part is None:
                self.part = cx.torch.zeros(cx.hip.msm_partial_bytes(self.curve, self.group), dtype=cx.torch.uint8, device=cx.dev)
            self.local()
            self.out = allgather_and_fold(self.part.cpu(), self.curve, self.group, cx.world, cx.dist)
        return self.out

    def local(self):
        """The device part of a step on this rank, no exchange (what the roofline object describes)."""
        cx, C, B, L = self.cx, self.cx.C, self.cx.B, self.cx.L
        if cx.world == 1:
            return self.step()
        if self.part is None:
            self.part = cx.torch.zeros(cx.hip.msm_partial_bytes(self.curve, self.group), dtype=cx.torch.uint8, device=cx.dev)
        B._check(L.csh_msm_partial_dev(self.h, C.c_size_t(0), C.c_size_t(self.n), C.c_void_p(self.sc.data_ptr()), 1, C.c_void_p(self.part.data_ptr()), C.c_void_p(cx.stream)))

    def spin_up(self, min_s: float = 0.3, max_s: float = 3.0, batch: int = 10, tol: float = 0.02):
        """Untimed steps BEFORE the W warm-up steps, whatever flags were passed: an idle MI355X needs a few hundred ms of continuous
        work to reach its steady clocks (profiles/archive/r04_j_ntt_context.log), and a 20-step region of 1.6 ms steps sits entirely inside that
        ramp. Batches of `batch` steps run for at least `min_s` and until two consecutive batch medians agree within `tol` (cap `max_s`).
        -> (steps issued, wall ms, median ms of the last batch)."""
        cx = self.cx
        t_begin = time.perf_counter()
        prev, issued, last = None, 0, None
        while True:
            ts = []
            for _ in range(batch):
                t0 = time.perf_counter()
                self.step()
                ts.append(time.perf_counter() - t0)
            issued += batch
            last = sorted(ts)[len(ts) // 2]
            el = time.perf_counter() - t_begin
            done = el >= min_s and prev is not None and abs(last - prev) <= tol * prev
            if cx.world > 1:   # all ranks leave together (a rank that stopped early would stall a collective step)
                done = cx.max_over_ranks(0.0 if done or el >= max_s else 1.0) == 0.0
            elif el >= max_s:
                done = True
            if done:
                break
            prev = last
        return issued, (time.perf_counter() - t_begin) * 1e3, last * 1e3

    def timed(self, steps: int, warmup: int):
        cx = self.cx
        for _ in range(warmup):
            self.step()
        cx.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = self.step()
        cx.barrier()
        dt = cx.max_over_ranks(time.perf_counter() - t0)
        return dt, res

    def check(self, res) -> bool:
        """Closed form: the folded MSM must equal (sum over ranks of sum_i s_i k_i) * G. Checker only (uses the oracle)."""
        from oracle import curves as cv
        from tests import helpers as H
        from tests.check_closed_form import dlogs, weighted_sum
        cx, np = self.cx, self.cx.np
        name = CURVE_NAMES[self.curve]
        F = H.FR[name]
        mine = weighted_sum(self.sc.cpu().numpy().view(np.uint64), dlogs(SEED, self.n, self.start)) % F.p
        sums = [mine]
        if cx.world > 1:
            buf = cx.torch.tensor(list(mine.to_bytes(32, "little")), dtype=cx.torch.uint8)
            allb = [cx.torch.empty(32, dtype=cx.torch.uint8) for _ in range(cx.world)]
            cx.dist.all_gather(allb, buf)
            sums = [int.from_bytes(bytes(t.tolist()), "little") for t in allb]
        if cx.rank != 0:
            return True
        G = cv.CURVES[name][self.group]
        S = sum(sums) % F.p * F.Rinv % F.p
        return bool(G.eq(H.jac_to_affine(G, res), G.mul(G.gen, S)))

    def affine_words(self, res):
        """Jacobian (X, Y, Z in {0, 1}) -> packed affine words (all-zero = infinity), for the bit-exact CPU comparison."""
        np = self.cx.np
        w = res.size // 3
        return np.zeros(2 * w, dtype=np.uint64) if not res[2 * w:].any() else np.ascontiguousarray(res[:2 * w])

    def stage_timing(self, reps=5):
        cx, B, np = self.cx, self.cx.B, self.cx.np
        B.tune_set("msm_timing", 1)
        acc = []
        for _ in range(reps):
            self.local()
            acc.append(B.msm_last_timing())
        B.tune_set("msm_timing", 0)
        return [float(np.mean([a[i] for a in acc])) for i in range(6)], B.msm_last_params()

    def free(self):
        self.cx.L.csh_bases_free(self.h)
        self.sc = None
        self._pts_dev = None


_LIVE = {}


def live_probe(cx, kind: int, iters: int):
    """Instruction-rate / modmul-rate probe of the library (csh_microbench, the same chains tools/gpu_probe.py runs) on THIS box, once
    per process: the boxes of the pool differ by ~5 % in clocks, so a peak read from a file measured elsewhere would shift every
    `alu.frac` by that much. None if the probe is unavailable."""
    if kind not in _LIVE:
        try:
            v = cx.C.c_double(0)
            best = 0.0
            for _ in range(3):
                cx.B._check(cx.L.csh_microbench(kind, iters, cx.C.byref(v)))
                best = max(best, v.value)
            _LIVE[kind] = best
        except Exception:  # noqa: BLE001
            _LIVE[kind] = None
    return _LIVE[kind]


def msm_roofline(job: MsmJob, rin: dict, log_n_local: int):
    stage_ms, (c_bits, n_win, lane_len, n_seg) = job.stage_timing()
    t_acc = stage_ms[3] * 1e-3
    kname = {"bn254_g1": "Bn254G1", "bn254_g2": "Bn254G2", "bls12_381_g1": "Bls381G1", "bls12_381_g2": "Bls381G2", "grumpkin_g1": "GrumpkinG1",
             "bls12_377_g1": "Bls377G1", "bls12_377_g2": "Bls377G2"}[job.workload]
    alg_bytes = job.n * (32.0 + job.pbytes)                  # scalar + affine base per point
    achieved = alg_bytes / t_acc / 1e9
    hbm_peak = float(rin.get("hbm_peak_GBps", 8000.0))     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s
    kin = rin.get("kernels", {}).get(f"k_msm_accum<{kname}> 2^{log_n_local}", {})
    mads = rin.get("mads_per_madd", {}).get(kname)
    mad_peak, mad_src = rin.get("mad_peak_T"), rin.get("mad_peak_source")
    lv = [x for x in (live_probe(job.cx, 0, 2000), live_probe(job.cx, 8, 2000)) if x]
    if lv:   # forced v_mad_u64_u32 / v_mad_i64_i32 chains on this box, lane-ops/s
        mad_peak, mad_src = round(max(lv) / 1e12, 2), "csh_microbench on this box, this run (v_mad_u64_u32 / v_mad_i64_i32 chains; file value: %s T/s, %s)" % (rin.get("mad_peak_T"), rin.get("mad_peak_source"))
    # `bound`: what bounds the kernel (the driver keeps the scalar fields of this object only, so the integer-issue figures
    # are flattened into it below). achieved / peak / frac stay the HBM figures the contract asks for: algorithmic bytes per launch over
    # the kernel's event time against 8 TB/s -- ~1 % by construction, the kernel is bound by v_mad_i64_i32 issue (alu_frac).
    roof = {"bound": "valu_int_mad", "kernel": f"k_msm_accum<{kname}Cfg>", "achieved": round(achieved, 2), "peak": hbm_peak, "unit": "GB/s",
            "frac": round(achieved / hbm_peak, 5), "traffic": kin.get("traffic_bytes"), "traffic_source": kin.get("file"),
            "algorithmic_bytes_per_launch": alg_bytes, "hbm_frac": round(achieved / hbm_peak, 5),
            "accum_ms": round(stage_ms[3], 4), "c": c_bits, "windows": n_win, "lane_len": lane_len, "device_ms": round(stage_ms[5], 4),
            "sort_ms": round(stage_ms[0] + stage_ms[1] + stage_ms[2], 4), "tail_ms": round(stage_ms[4], 4),
            "note": "the MSM is integer-ALU bound (v_mad_i64_i32 issue), not HBM bound, by construction (W mixed additions ~ 160 modmuls per 96 B); "
                    "see `alu` and DESIGN.md 3.1",
            "msm_params": {"c": c_bits, "windows": n_win, "lane_len": lane_len, "segments": n_seg},
            "stage_ms": {"digits+hist": stage_ms[0], "scan": stage_ms[1], "scatter": stage_ms[2], "accum": stage_ms[3], "merge+reduce+fold": stage_ms[4],
                         "total_device": stage_ms[5]}}
    if mads and mad_peak:
        a = job.n * n_win * mads / t_acc / 1e12
        roof["alu"] = {"unit": "Tmad/s", "achieved": round(a, 2), "peak": mad_peak, "frac": round(a / mad_peak, 3), "mads_per_madd": mads,
                       "madds_per_s": round(job.n * n_win / t_acc), "peak_source": mad_src}
        # a denominator this repository did not measure itself: one VALU lane-operation per lane and clock, 256 CUs x 4 SIMDs x 16 lanes x
        # 2.4 GHz (/opt/skills/guides/MI355X_MICROARCH.md) = 39.3 T lane-ops/s; a 64-bit multiply-add is one such operation at best
        nominal = 256 * 4 * 16 * 2.4e9 / 1e12
        roof["alu"]["peak_nominal"] = round(nominal, 1)
        roof["alu"]["frac_vs_nominal"] = round(a / nominal, 3)
        roof.update({"alu_unit": "Tmad/s", "alu_achieved_Tmad": round(a, 2), "alu_peak_Tmad": mad_peak, "alu_frac": round(a / mad_peak, 3),
                     "alu_peak_nominal_Tmad": round(nominal, 1), "alu_frac_vs_nominal": round(a / nominal, 3), "mads_per_madd": mads})
        if rin.get("mad_peak_T"):   # the constant of profiles/roofline_inputs.json (tools/gpu_probe.py: one cold pass per process), for continuity with earlier rounds
            roof["alu"]["peak_file"] = rin["mad_peak_T"]
            roof["alu"]["frac_vs_file"] = round(a / rin["mad_peak_T"], 3)
    return roof


def ntt_transforms(cx: Ctx, logn: int, warm_transforms: int, timed_pairs: int):
    """One independent BN254 transform pipeline on this rank's GPU (snarkjs root, data resident): `timed_pairs` back-to-back
    ifft_in_to_out + fft_out_to_in pairs between two HIP events on the launch stream, after `warm_transforms` untimed transforms.
    -> ms per transform (warm), the same measured right after a single warm-up transform, and the number of untimed transforms."""
    hip, B, torch = cx.hip, cx.B, cx.torch
    r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    g = pow(pow(5, (r - 1) >> 28, r), 1 << (28 - logn), r) * ((1 << 256) % r) % r
    gen = cx.np.array([(g >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=cx.np.uint64)
    dom = hip.Domain(hip.BN254, logn, gen)
    data = torch.randint(0, 1 << 62, (1 << logn, 4), dtype=torch.int64, device=cx.dev)
    data[:, 3] >>= 1   # canonical (< r)
    torch.cuda.synchronize()
    e0, e1 = B.Event(), B.Event()

    def pairs(k):
        e0.record(cx.stream)
        for _ in range(k):
            dom.ifft_in_to_out_dev(data.data_ptr(), 1, cx.stream)
            dom.fft_out_to_in_dev(data.data_ptr(), 1, cx.stream)
        e1.record(cx.stream)
        return e0.elapsed_ms(e1) / (2 * k)

    dom.ifft_in_to_out_dev(data.data_ptr(), 1, cx.stream)
    first = pairs(timed_pairs)
    done = 1 + 2 * timed_pairs
    # spin-up, same rule as the MSM lines: untimed batches (10 pairs) for at least `warm_transforms` transforms AND 0.3 s, until two
    # consecutive batches agree within 2 % (cap 3 s)
    t_begin, prev = time.perf_counter(), None
    while True:
        cur = pairs(10)
        done += 20
        el = time.perf_counter() - t_begin
        if el >= 3.0 or (done >= warm_transforms and el >= 0.3 and prev is not None and abs(cur - prev) <= 0.02 * prev):
            break
        prev = cur
    ms = pairs(timed_pairs)
    dom.free()
    del data
    return {"ms": ms, "ms_first_batch": first, "warm_transforms": done, "spinup_ms": round((time.perf_counter() - t_begin) * 1e3, 1)}


def secondary_single_gpu(cx: Ctx, rin: dict, args):
    """N = 1 extras: MSM 2^24 (BN254 G1 + the config-5 groups, closed-form checked), NTT 2^22 (config 3), Rep3 local_mul_vec,
    Groth16 prove ms (configs 1, 4 and the synthetic 2^20 circuit)."""
    hip, B, L, C, np, torch, dev, stream = cx.hip, cx.B, cx.L, cx.C, cx.np, cx.torch, cx.dev, cx.stream
    out = {}
    torch.cuda.synchronize()
    cpu_inputs = None
    for wl, logn, steps in (("bn254_g1", args.split_log_n, 5), ("bls12_381_g1", args.split_log_n, 3), ("bls12_381_g2", args.split_log_n, 2)):
        job = MsmJob(cx, wl, 0, 1 << logn, 4321 + logn)
        keep = wl == "bn254_g1" and not args.no_cpu_baseline
        pts_host = job.points_to_host() if keep else None
        job.drop_point_copy()
        spin = job.spin_up(min_s=args.spinup_s, batch=3) if args.spinup_s > 0 else (0, 0.0, None)
        dt, res = job.timed(steps, 1)
        roof = msm_roofline(job, rin, logn)
        out[f"msm_{wl}_2p{logn}"] = {"points_per_s": job.n * steps / dt, "ms": dt / steps * 1e3, "result_check": job.check(res),
                                     "steps": steps, "spinup_steps": spin[0], "spinup_ms": round(spin[1], 1),
                                     "roofline": {k: roof[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "traffic", "alu", "stage_ms") if k in roof}}
        if keep:
            cpu_inputs = (pts_host, job.sc.cpu().numpy().view(np.uint64), job.affine_words(res))
        job.free()
        del job
    # BLS12-377 (the curve of the reference's LibSnarkReduction fixtures; round 6): both groups at 2^20, closed-form checked
    for wl, steps in (("bls12_377_g1", 5), ("bls12_377_g2", 3)):
        try:
            job = MsmJob(cx, wl, 0, 1 << 20, 377)
            job.drop_point_copy()
            if args.spinup_s > 0:
                job.spin_up(min_s=args.spinup_s, batch=3)
            dt, res = job.timed(steps, 1)
            out[f"msm_{wl}_2p20"] = {"points_per_s": job.n * steps / dt, "ms": dt / steps * 1e3, "result_check": job.check(res), "steps": steps}
            job.free()
            del job
        except Exception as e:  # noqa: BLE001
            out[f"msm_{wl}_2p20"] = {"error": repr(e)}
    # The headline workload and the 2^24 workload again with fixed-base tables on the handle, as a prover holds its key (round 6:
    # csh_bases_table_policy -> one table row per window, ONE bucket set, 17- / 20-bit windows: 15 / 13 mixed additions per point instead of
    # 17 / 16; msm_sort_wide.hip). The headline itself stays on the plain handle, as the reference's msm_unchecked is variable-base.
    for logn, steps, key in ((20, 20, "msm_bn254_g1_2p20_fixed_base_tables"), (args.split_log_n, 5, f"msm_bn254_g1_2p{args.split_log_n}_fixed_base_tables")):
        try:
            job = MsmJob(cx, "bn254_g1", 0, 1 << logn, 1234 if logn == 20 else 4321 + logn)
            job.drop_point_copy()
            tc, rows = C.c_int(0), C.c_int(0)
            B._check(L.csh_bases_table_policy(C.c_size_t(job.n), C.byref(tc), C.byref(rows)))
            t0 = time.perf_counter()
            B._check(L.csh_bases_precompute_grouped(job.h, tc.value, rows.value))
            B.sync()
            build_ms = (time.perf_counter() - t0) * 1e3
            if args.spinup_s > 0:
                job.spin_up(min_s=args.spinup_s, batch=10 if logn <= 20 else 3)
            dt, res = job.timed(steps, 5 if logn <= 20 else 1)
            stage_ms, (c_bits, n_win, lane_len, n_seg) = job.stage_timing(reps=3)
            out[key] = {"points_per_s": job.n * steps / dt, "ms": dt / steps * 1e3, "result_check": job.check(res), "steps": steps,
                        "window_bits": tc.value, "table_rows_asked": rows.value, "bucket_sets": n_win, "table_build_ms": round(build_ms, 1),
                        "table_bytes": int(job.n) * 64 * (-(-255 // tc.value)),
                        "stage_ms": {"sort_level1": stage_ms[0], "bucket_hist+scan": stage_ms[1], "sort_level2": stage_ms[2], "accum": stage_ms[3],
                                     "merge+reduce+fold": stage_ms[4], "total_device": stage_ms[5]}}
            job.free()
            del job
        except Exception as e:  # noqa: BLE001
            out[key] = {"error": repr(e)}
    # The headline workload issued the way the reference issues its MSMs: two host threads, each calling the synchronous entry point back to
    # back (rayon_join5 in groth16.rs:227-294 runs the five MSM closures of one proof concurrently). Every thread has its own stream in
    # the library (NULL stream = the calling thread's lane), so one call's sort, bucket tail, result copy and host fold overlap the other's
    # accumulation. Reported beside the headline, never instead of it: the headline step stays one synchronous call.
    try:
        import threading
        job = MsmJob(cx, "bn254_g1", 0, 1 << 20, 1234)
        job.drop_point_copy()
        per_thread, outs, errs = 12, [np.zeros_like(job.out) for _ in range(2)], []

        def caller(buf, k):
            try:
                for _ in range(k):
                    B._check(L.csh_msm_dev(job.h, C.c_size_t(0), C.c_size_t(job.n), C.c_void_p(job.sc.data_ptr()), 1, buf.ctypes.data_as(C.c_void_p), None))
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))

        def run(k):
            th = [threading.Thread(target=caller, args=(outs[i], k)) for i in range(2)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            return time.perf_counter() - t0

        run(3)
        torch.cuda.synchronize()
        dt = run(per_thread)
        out["msm_bn254_g1_2p20_two_callers"] = {"points_per_s": job.n * 2 * per_thread / dt, "ms_per_msm": dt / (2 * per_thread) * 1e3,
                                                "result_check": bool(job.check(outs[0]) and job.check(outs[1])) and not errs, "callers": 2,
                                                "msms": 2 * per_thread, "errors": errs or None}
        job.free()
        del job
    except Exception as e:  # noqa: BLE001
        out["msm_bn254_g1_2p20_two_callers"] = {"error": repr(e)}
    # The headline workload on a witness-like scalar vector instead of a uniform one (what the PLAIN prover's a / b / l queries see on a
    # circuit full of booleans; secret shares of a witness are uniform, so the MPC provers see the headline's distribution): a quarter of
    # the scalars 0, a quarter 1 (every one of them in bucket 1 of window 0), a quarter one repeated value (one bucket per window holds a
    # quarter of the window's entries), a quarter uniform. Exercises the tile-parallel level 2 of the sort and the sliced merge of
    # oversized buckets; closed-form checked like every other line.
    try:
        job = MsmJob(cx, "bn254_g1", 0, 1 << 20, 1234)
        job.drop_point_copy()
        gk = torch.Generator(device=dev)
        gk.manual_seed(4242)
        kind = torch.randint(0, 4, (job.n,), device=dev, generator=gk)
        r_mod = SCALAR_MODULUS[job.curve]
        mont_one = (1 << 256) % r_mod
        one = torch.from_numpy(np.array([(mont_one >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64).view(np.int64)).to(dev)
        rep = job.sc[0].clone()
        job.sc[kind == 0] = 0
        job.sc[kind == 1] = one
        job.sc[kind == 2] = rep
        torch.cuda.synchronize()
        if args.spinup_s > 0:
            job.spin_up(min_s=args.spinup_s)
        dt, res = job.timed(20, 5)
        out["msm_bn254_g1_2p20_witness_like"] = {"points_per_s": job.n * 20 / dt, "ms": dt / 20 * 1e3, "result_check": job.check(res),
                                                 "scalars": "1/4 zero, 1/4 one, 1/4 one repeated value, 1/4 uniform (Montgomery form)"}
        job.free()
        del job
    except Exception as e:  # noqa: BLE001
        out["msm_bn254_g1_2p20_witness_like"] = {"error": repr(e)}
    # NTT 2^22 (snarkjs root), data resident; HIP events on the launch stream
    logn = args.ntt_log_n
    nt = ntt_transforms(cx, logn, warm_transforms=120, timed_pairs=20)
    ms = nt["ms"]
    modmuls = (1 << logn) // 2 * logn                      # one twiddle multiplication per butterfly
    mm_peak, mm_src = rin.get("modmul_peak_G"), rin.get("modmul_peak_source")
    lv = [x for x in (live_probe(cx, 11, 200), live_probe(cx, 14, 200)) if x]
    if lv:   # dependent 9 x 29-bit Montgomery products (unsigned row-wise / signed product-scanning multiplier) on this box
        mm_peak, mm_src = round(max(lv) / 1e9, 2), "csh_microbench on this box, this run (k_modmul29 / k_modmul29s; file value: %s G/s)" % rin.get("modmul_peak_G")
    kin = rin.get("kernels", {}).get("k_ntt_pass_lazy 2^22", {}) if logn == 22 else {}
    out[f"ntt_bn254_2p{logn}"] = {"elements_per_s": (1 << logn) / ms * 1e3, "ms": ms,
                             # the same 20 pairs right after ONE warm-up transform (what rounds 1-3 reported): the clocks of an idle GPU take tens of
                             # ms of continuous work to come up (profiles/archive/r04_j_ntt_context.log: 0.566 -> 0.503 -> 0.480 -> 0.467 ms over four batches)
                             "ms_first_batch_after_idle": nt["ms_first_batch"], "warm_up_transforms": nt["warm_transforms"], "spinup_ms": nt["spinup_ms"],
                             "roofline": {"bound": "hbm", "kernel": "k_ntt_pass_r4 / k_ntt_pass_lazy (all passes of one transform)", "achieved": round(64.0 * (1 << logn) / ms / 1e6, 1),
                                          "peak": float(rin.get("hbm_peak_GBps", 8000.0)), "unit": "GB/s", "frac": round(64.0 * (1 << logn) / ms / 1e6 / float(rin.get("hbm_peak_GBps", 8000.0)), 4), "traffic": kin.get("traffic_bytes"),
                                          "traffic_source": kin.get("file"),
                                          "alu": {"unit": "G modmul/s", "achieved": round(modmuls / ms / 1e6, 1), "peak": mm_peak,
                                                  "frac": round(modmuls / ms / 1e6 / mm_peak, 3) if mm_peak else None, "peak_source": mm_src}}}
    e0, e1 = B.Event(), B.Event()
    # Rep3 local_mul_vec 2^24 (192 B/element)
    n = 1 << (18 if args.quick else 24)
    a = torch.randint(0, 1 << 62, (2 * n, 4), dtype=torch.int64, device=dev)
    b = torch.randint(0, 1 << 62, (2 * n, 4), dtype=torch.int64, device=dev)
    m = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device=dev)
    o = torch.empty((n, 4), dtype=torch.int64, device=dev)
    for x in (a, b, m):
        x[:, 3] >>= 1
    f = lambda: B._check(L.csh_rep3_local_mul_vec_dev(hip.BN254, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(m.data_ptr()),
                                                       C.c_void_p(o.data_ptr()), C.c_size_t(n), C.c_void_p(stream)))
    for _ in range(3):
        f()
    e0.record(stream)
    for _ in range(20):
        f()
    e1.record(stream)
    ms = e0.elapsed_ms(e1) / 20
    kin = rin.get("kernels", {}).get("k_rep3_local_mul 2^24", {})
    out["rep3_local_mul_vec_2p24"] = {"elements_per_s": n / ms * 1e3, "ms": ms,
                                      "roofline": {"bound": "hbm", "kernel": "k_rep3_local_mul<Bn254Fr>", "achieved": round(192.0 * n / ms / 1e6, 1), "peak": float(rin.get("hbm_peak_GBps", 8000.0)), "unit": "GB/s",
                                                   "frac": round(192.0 * n / ms / 1e6 / float(rin.get("hbm_peak_GBps", 8000.0)), 4), "traffic": kin.get("traffic_bytes"), "traffic_source": kin.get("file")}}
    del a, b, m, o
    # Groth16 plain prove, synthetic 2^20-constraint circuit with a known-dlog key (closed-form checked), key resident
    from cosnarks_amd import groth16 as g16
    xfer_keys = ("stat_populate_us", "stat_join_wait_us", "stat_finish_us", "stat_d2h_slow", "stat_d2h_staged", "stat_h2d_slow", "stat_h2d_staged",
                 "stat_stage_all_switches", "stat_uploads_shared")
    xfer0 = {k: cx.B.tune_get(k) for k in xfer_keys}
    out[f"groth16_prove_synthetic_2p{args.prove_log_n}"] = g16.bench_synthetic(hip.BN254, args.prove_log_n, 7 if args.quick else 21, with_rep3=True)
    # the host side of the trait path on THIS box: page population (worker time, the caller's wait for it), result copies that stalled /
    # were staged (HostXfer, csrc/capi.hip) -- summed over the witness maps of the entry above
    out[f"groth16_prove_synthetic_2p{args.prove_log_n}"]["host_result_copies"] = {k[5:]: cx.B.tune_get(k) - xfer0[k] for k in xfer_keys}
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
    return out, cpu_inputs


def secondary_multi_gpu(cx: Ctx, args):
    """N > 1 extras: the strong-scaling workloads of BASELINE config 5 (ONE MSM of 2^24 points split over the N ranks) on
    BLS12-381 G1 / G2 and BN254 G1, closed-form checked; then, on rank 0 alone, the single-thread `csh_msm_split` over all N
    devices with its three exchanges (hipMemcpyPeer / host copies / grouped RCCL) so both exchanges are measured."""
    out = {}
    logn = args.split_log_n
    total = 1 << logn
    per = total // cx.world
    for wl, steps in (("bn254_g1", 5), ("bls12_381_g1", 3), ("bls12_381_g2", 2)):
        start = cx.rank * per
        cnt = per if cx.rank < cx.world - 1 else total - start
        job = MsmJob(cx, wl, start, cnt, 777 + cx.rank)
        job.drop_point_copy()
        if args.spinup_s > 0:
            job.spin_up(min_s=args.spinup_s, batch=3)
        dt, res = job.timed(steps, 1)
        ok = job.check(res)
        out[f"msm_{wl}_2p{logn}_strong"] = {"points_per_s": total * steps / dt, "ms": dt / steps * 1e3, "result_check": ok, "ranks": cx.world,
                                            "points_per_rank": per}
        job.free()
        del job
    # independent NTT instances, one per rank (north star: "independent MSM/NTT instances ... shard across the GPUs": replicas, no
    # collective): every rank transforms its own 2^k vector; elements/s summed over the ranks on the slowest rank's clock
    try:
        nl = args.ntt_log_n
        cx.barrier()
        nt = ntt_transforms(cx, nl, warm_transforms=120, timed_pairs=20)
        slow = cx.max_over_ranks(nt["ms"])
        out[f"ntt_bn254_2p{nl}_replicas"] = {"elements_per_s": cx.world * (1 << nl) / slow * 1e3, "ms_per_transform_slowest_rank": slow,
                                             "ms_per_transform_this_rank": nt["ms"], "ranks": cx.world, "scaling": "weak (one independent transform per GPU)"}
    except Exception as e:  # noqa: BLE001
        out[f"ntt_bn254_2p{args.ntt_log_n}_replicas"] = {"error": repr(e)}
    cx.barrier()
    # BASELINE config 4 with one GPU per party: three in-process Rep3 parties on devices 0..2 (folded: on the GPUs that exist), each with
    # its own copy of the proving key; rank 0 drives them (the parties are host threads of one process, as in the reference's tests)
    if cx.rank == 0:
        try:
            from cosnarks_amd import groth16 as g16
            ndev = 1 if cx.folded else cx.world
            out["groth16_rep3_party_per_gpu"] = g16.bench_rep3_party_per_gpu(cx.hip.BN254, args.prove_log_n, [p % ndev for p in range(3)], iters=2)
        except Exception as e:  # noqa: BLE001
            out["groth16_rep3_party_per_gpu"] = {"error": repr(e)}
    cx.barrier()
    if cx.rank == 0:   # folded (BENCH_FOLD_RANKS): every range on device 0, so the code path runs in the one-GPU suite too
        try:
            out[f"single_process_split_bn254_g1_2p{logn}"] = single_process_split(cx, 0, 0, logn)
        except Exception as e:  # noqa: BLE001
            out[f"single_process_split_bn254_g1_2p{logn}"] = {"error": repr(e)}
    cx.barrier()
    # the metric's second half at this N: ONE plain prover with its five query MSMs placed on the N GPUs (rank 0, subprocess)
    if cx.rank == 0:
        try:
            devs = [0] * cx.world if cx.folded else list(range(cx.world))
            out[f"groth16_prove_synthetic_2p{args.prove_log_n}_placed"] = {"by_query": prove_over_devices(devs, logn=args.prove_log_n, mode=1),
                                                                            "by_range": prove_over_devices(devs, logn=args.prove_log_n, mode=2)}
        except Exception as e:  # noqa: BLE001
            out[f"groth16_prove_synthetic_2p{args.prove_log_n}_placed"] = {"error": repr(e)}
    cx.barrier()
    return out


def prove_over_devices(devices, logn=20, steps=5, warmup=2, timeout=300, mode=0):
    """Plain Groth16 prove of the synthetic 2^logn circuit with the five query MSMs placed on `devices` (one process, one host
    thread per GPU): tools/bench_prove_devices.py in a subprocess with a timeout, so nothing it does can stall the caller."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_prove_devices.py"), "--devices", ",".join(str(d) for d in devices),
           "--log-n", str(logn), "--steps", str(steps), "--warmup", str(warmup), "--mode", str(mode)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": (r.stderr or r.stdout)[-400:]}
    return json.loads(lines[-1])


def run_groth16_prove(cx: Ctx, args):
    """--workload groth16_prove: a step = one plain Groth16 prove (prove_inner: witness upload, device witness map, five MSM
    groups, finish) of the synthetic 2^log_n-constraint BN254 circuit, key resident. ONE prover: at N > 1 rank 0 drives all N
    GPUs (its five independent query MSMs -- rayon_join5, groth16.rs:227-294 -- placed on them, scalars by peer copy); the other
    ranks only keep the barriers. value = ms per proof (lower is better), scaling = strong."""
    from cosnarks_amd import groth16 as g16
    world, rank = cx.world, cx.rank
    if world == 1:
        devices = [cx.dev_index]
    elif cx.folded:                      # BENCH_FOLD_RANKS: logical slots on the GPUs that exist
        devices = [cx.dev_index] * world
    else:
        devices = [cx.dev_index] + [d for d in range(world) if d != cx.dev_index]
    circ, err = None, None
    if rank == 0:
        try:
            g16.set_prover_devices(devices if len(devices) > 1 else None, args.placement)
            circ = g16.SynthCircuit(cx.hip.BN254, args.log_n)
            for _ in range(args.warmup + 2):      # + 2 untimed setup proofs: the slot threads' streams, arenas and pooled buffers are
                circ.prove()                      # created on first use (the same role as --spinup in the MSM workloads)
        except Exception as e:  # noqa: BLE001
            err = repr(e)
    cx.barrier()
    t0 = time.perf_counter()
    phases = []
    if rank == 0 and err is None:
        try:
            for _ in range(args.steps):
                phases.append(circ.prove())
        except Exception as e:  # noqa: BLE001
            err = repr(e)
    cx.barrier()
    dt = cx.max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        ok = None
        if err is None and not args.no_check:
            ok = circ.check()
        ms = dt / args.steps * 1e3
        med = lambda k: sorted(p[k] for p in phases)[len(phases) // 2] if phases else None
        line = {"metric": "Groth16 prove ms", "value": ms if err is None else None, "unit": "ms", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": ms, "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
                "dtype": "BN254: i32x9 29-bit-limb lazy Montgomery (MSM, NTT butterflies), Fp2 for the G2 query", "data": "synthetic",
                "config": {"workload": f"plain Groth16 prove (CircomReduction), synthetic 2^{args.log_n}-constraint BN254 circuit with a known-dlog key, key resident; "
                                       "one prover, its five query MSMs placed on the GPUs", "devices": devices,
                           "placement": {0: "auto (whole queries per GPU up to 2 GPUs, ranges of every query from 3 on)", 1: "whole queries per GPU (LPT, G2 = 2.5 x G1)",
                                         2: "the k-th range of every query per GPU"}[args.placement] + "; csh_bases_clone + csh_memcpy_peer, one host thread per GPU"},
                "phases_ms_median": {k: med(k) for k in ("witness_upload_and_map", "msm_groups", "finish")}, "result_check": ok, "error": err,
                "roofline": None, "cpu_baseline": None}
        print(json.dumps(line))
        if circ is not None:
            circ.close()
        g16.set_prover_devices(None)
    sys.stdout.flush()
    if world > 1:
        cx.dist.destroy_process_group()


def single_process_split(cx: Ctx, curve, group, logn):
    """Rank 0 drives every GPU of the node from one thread (csh_msm_split: PEER / HOST / RCCL exchange). Runs in a subprocess with
    a timeout so that nothing it does can stall the ranks waiting at the barrier."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_single_process_split.py"), "--devices", str(cx.world), "--curve", str(curve),
           "--group", str(group), "--log-n", str(logn)] + (["--fold"] if cx.folded else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": (r.stderr or r.stdout)[-400:]}
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # 100 steps of 1.7 ms: the 20-step region of rounds 1-3 (34 ms) sat inside the clock ramp of an idle GPU
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + ["groth16_prove"], default="bn254_g1")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--exchange", choices=["rccl", "harness"], default="rccl", help="N > 1: RCCL behind the C ABI (default) or the gloo harness all-gather")
    ap.add_argument("--spinup", type=int, default=int(os.environ.get("BENCH_SPINUP", "0")),
                    help="untimed steps issued during setup, before the W warmup steps (brings the GPU out of its idle clocks)")
    ap.add_argument("--spinup-s", type=float, default=float(os.environ.get("BENCH_SPINUP_S", "0.3")),
                    help="minimum wall time of the automatic spin-up (0 disables it)")
    ap.add_argument("--placement", type=int, choices=[0, 1, 2], default=0, help="groth16_prove at N > 1: 0 auto, 1 whole queries per GPU, 2 ranges of every query per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary metrics")
    ap.add_argument("--quick", action="store_true", help="shrink the secondary workloads (schema tests of the N > 1 line on a small box): "
                                                         "split MSM 2^18, NTT 2^16, prove 2^14")
    args = ap.parse_args()
    args.split_log_n, args.ntt_log_n, args.prove_log_n = (18, 16, 14) if args.quick else (24, 22, 20)

    if args.workload == "groth16_prove":
        args.exchange = "harness"      # no split MSM in this workload: no RCCL communicator to build
    cx = Ctx(args)
    if args.workload == "groth16_prove":
        return run_groth16_prove(cx, args)
    rin = load_roofline_inputs()
    world, rank = cx.world, cx.rank
    if args.scaling == "weak":
        n_local = 1 << args.log_n
        start = rank * n_local
        total = world * n_local
    else:
        total = 1 << args.log_n
        per = total // world
        start = rank * per
        n_local = per if rank < world - 1 else total - start

    job = MsmJob(cx, args.workload, start, n_local, 1234 + rank)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "bn254_g1"
    pts_host = job.points_to_host() if want_cpu else None
    job.drop_point_copy()
    # the first 20 steps after setup (what rounds 1-4 recorded under the driver's `--steps 20 --warmup 5`: inside the clock ramp of an idle GPU),
    # kept beside `value`; then the spin-up (untimed, >= 0.3 s, until two consecutive 10-step medians agree within 2 %), then W warm-up
    # steps and EXACTLY K timed steps between barriers
    cold_dt, _ = job.timed(20, 3)
    value_first_20 = total * 20 / cold_dt
    for _ in range(args.spinup):
        job.step()
    spin_steps, spin_ms, spin_last_ms = job.spin_up(min_s=args.spinup_s) if args.spinup_s > 0 else (0, 0.0, None)
    dt, res = job.timed(args.steps, args.warmup)
    ms_per_step = dt / args.steps * 1e3
    value = total * args.steps / dt

    log_local = max(0, n_local.bit_length() - 1)
    roofline = msm_roofline(job, rin, log_local)          # every rank runs the same untimed passes (keeps the ranks in step)
    check = None if args.no_check else job.check(res)

    extras, cpu_inputs24 = None, None
    if not args.no_extras:
        try:
            if world == 1:
                extras, cpu_inputs24 = secondary_single_gpu(cx, rin, args)
            else:
                extras = secondary_multi_gpu(cx, args)
        except Exception as e:  # noqa: BLE001 -- extras never break the headline
            extras = {"error": repr(e)}

    cpu_baseline = None
    if want_cpu:
        try:
            from oracle.cbridge import cpu_baseline_suite
            sc_host = job.sc.cpu().numpy().view(cx.np.uint64)
            p24, s24, a24 = cpu_inputs24 if cpu_inputs24 else (None, None, None)
            # G2 bases of the same known-dlog family for the CPU prove composition (generated on the GPU, copied back; untimed)
            g2_host = None
            try:
                g2 = cx.torch.empty(job.n * 128, dtype=cx.torch.uint8, device=cx.dev)
                cx.B._check(cx.L.csh_util_generate_bases_dev(0, 1, cx.C.c_uint64(SEED), cx.C.c_size_t(job.n), cx.C.c_void_p(g2.data_ptr()), cx.C.c_void_p(cx.stream)))
                cx.torch.cuda.synchronize()
                cx.B.sync()
                g2_host = g2.cpu().numpy().view(cx.np.uint64).reshape(job.n, -1)
                del g2
            except Exception:  # noqa: BLE001
                g2_host = None
            cpu_baseline = cpu_baseline_suite(pts_host, sc_host, job.affine_words(res), p24, s24, a24, host_cpus=HOST_CPUS, pts20_g2=g2_host)
            cpu_baseline["gpu_over_cpu_2p20"] = round(value / cpu_baseline["value"], 1)
            try:   # a CPU figure beside the GPU's wall ms on the reference's own circuits
                from oracle.cbridge import groth16_reference_circuit_composition
                ref = groth16_reference_circuit_composition(os.path.join(ROOT, "tests", "golden", "Groth16", "bn254"), threads=1, reps=20)
                gpu_ref = (extras or {}).get("groth16_prove_reference_circuits") or {}
                if gpu_ref:
                    ref["gpu_wall_ms_beside_it"] = {k: gpu_ref.get(k) for k in ("plain_multiplier2_ms", "plain_poseidon_ms", "rep3_poseidon_3_parties_ms")}
                    m2c, m2g = ref.get("plain_multiplier2_ms"), gpu_ref.get("plain_multiplier2_ms")
                    if m2c and m2g:
                        ref["note"] = ("multiplier2 (domain 4): GPU %.2f ms >= CPU %.2f ms -- at this size the prove is launch latency on the GPU and the CPU wins, as "
                                       "SURVEY 8d config 1 expects; the GPU figures are whole proves (zkey parse + key upload + finish), the CPU figures the hot stages only"
                                       % (m2g, m2c))
                cpu_baseline["groth16_prove_reference_circuits"] = ref
            except Exception as e:  # noqa: BLE001
                cpu_baseline["groth16_prove_reference_circuits"] = {"error": repr(e)}
            if extras and "msm_2p24" in cpu_baseline and "msm_bn254_g1_2p24" in extras:
                cpu_baseline["gpu_over_cpu_2p24"] = round(extras["msm_bn254_g1_2p24"]["points_per_s"] / cpu_baseline["msm_2p24"]["points_per_s"], 1)
        except Exception as e:  # the baseline is a reported extra, never part of the measured path
            cpu_baseline = {"error": repr(e)}

    if rank == 0:
        label = WORKLOADS[args.workload][3]
        per_rank = f"2^{args.log_n} points per GPU" if args.scaling == "weak" else f"2^{args.log_n} points in total, {total // world} per GPU"
        line = {
            "metric": f"{label} MSM points/sec", "value": value, "unit": "points/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": WORKLOADS[args.workload][4], "data": "synthetic",
            "config": {"workload": f"{label} Pippenger MSM, uniform scalars / known-dlog points, {per_rank}"
                                   + (" (BASELINE config 2)" if args.workload == "bn254_g1" and args.log_n == 20 else "")
                                   + (" (BASELINE config 5)" if args.workload.startswith("bls12_381") and args.log_n == 24 and args.scaling == "strong" else ""),
                       "points_total": total, "points_per_gpu": n_local, "split": cx.exchange,
                       "exchange_mode": cx.exchange_mode, "rccl_ranks_seen": cx.rccl_ranks_seen,
                       "host_cpus": {"affinity_at_start": HOST_CPUS, "affinity_after_imports_before_restore": cx.affinity_after_imports,
                                     "affinity_measured_under": len(os.sched_getaffinity(0)) if AFFINITY0 is not None else None},
                       "spinup_steps": args.spinup + spin_steps,
                       "spinup_ms": round(spin_ms, 1), "spinup_last_batch_median_ms": spin_last_ms,
                       "spinup_rule": "untimed steps before the W warm-up steps: batches of 10 for >= %.1f s until two consecutive batch medians agree within 2 %% (cap 3 s)" % args.spinup_s},
            "value_first_20_steps": value_first_20, "ms_per_step_first_20_steps": cold_dt / 20 * 1e3,
            # the same workload on a handle that carries the library's fixed-base tables (round 6: what a prover gets at key load through
            # csh_bases_table_policy; `value` itself stays the plain handle) -- top-level scalars, copied from `secondary`
            "value_fixed_base_tables": ((extras or {}).get(f"msm_{args.workload}_2p{args.log_n}_fixed_base_tables") or {}).get("points_per_s"),
            "ms_per_step_fixed_base_tables": ((extras or {}).get(f"msm_{args.workload}_2p{args.log_n}_fixed_base_tables") or {}).get("ms"),
            "roofline": roofline, "cpu_baseline": cpu_baseline, "result_check": check, "secondary": extras,
            # the host side of every large host-pointer result of this process (HostXfer, csrc/capi.hip): page-population worker time, the
            # callers' wait for it, final stream waits, copies that stalled (> 3 x their PCIe time + 4 ms) and copies that went through the
            # staged path -- a box whose driver stalls on freshly populated caller pages shows up HERE, on the headline object
            "host_result_copies": {k[5:]: cx.B.tune_get(k) for k in ("stat_populate_us", "stat_join_wait_us", "stat_finish_us", "stat_d2h_slow", "stat_d2h_staged",
                                                                      "stat_h2d_slow", "stat_h2d_staged", "stat_stage_all_switches", "stat_uploads_shared")},
        }
        print(json.dumps(line))
    sys.stdout.flush()
    if cx.hard_exit:           # a wedged RCCL bootstrap left a helper thread inside ncclCommInitRank: normal teardown could hang on it
        if world > 1:
            cx.dist.barrier()  # every rank has printed / finished its part
        sys.stderr.flush()
        os._exit(0)
    if cx.comm is not None:
        cx.comm.destroy()
    if world > 1:
        cx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
