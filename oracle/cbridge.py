"""ctypes bridge to oracle/c/liboracle.so (the plain-C restatement). ORACLE / TEST INFRASTRUCTURE ONLY.

Used by tests/ as the large-size checker and by bench.py's ``cpu_baseline`` leg (kind "port": the reference's
Rust/arkworks path cannot be built here -- no Rust toolchain, un-vendored crates)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        p = os.path.join(_HERE, "c", "liboracle.so")
        if not os.path.exists(p):
            subprocess.run(["make", "-s", "-C", os.path.join(_HERE, "c")], check=True)
        _LIB = C.CDLL(p)
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def num_threads() -> int:
    return lib().oc_num_threads()


_DEFAULT_THREADS = None


def default_threads() -> int:
    """Threads for a call that did not ask for a count: the CPUs this process may actually use -- the affinity mask capped by the
    container's CPU-time quota (the GPU boxes show 256 CPUs and grant 16: 256 OpenMP threads there are 10-100x slower)."""
    global _DEFAULT_THREADS
    if _DEFAULT_THREADS is None:
        n = _available_cpus()
        q = _cgroup_cpu_quota()
        if q:
            n = max(1, min(n, int(q + 0.5)))
        _DEFAULT_THREADS = n
    return _DEFAULT_THREADS


def _t(threads: int) -> int:
    return int(threads) if threads and threads > 0 else default_threads()


def point_words(curve: int, group: int) -> int:
    return (4 if curve in (0, 2) else 6) * 2 * (2 if group else 1)   # 0 BN254, 1 BLS12-381, 3 BLS12-377 (48-byte Fq)


def msm(curve: int, group: int, points, scalars, montgomery=True, threads=0):
    pts = np.ascontiguousarray(points, dtype=np.uint64)
    sc = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = sc.size // 4
    out = np.zeros(point_words(curve, group), dtype=np.uint64)
    assert lib().oc_msm(curve, group, _p(pts), _p(sc), C.c_size_t(n), int(montgomery), _t(threads), _p(out)) == 0
    return out


def msm_fast(curve: int, group: int, points, scalars, montgomery=True, threads=0, c=0, stages=None):
    """The tuned restatement (Booth digits, XYZZ buckets, thread-private bucket arrays): -> packed affine result.
    stages (optional list): receives [convert s, bucket tasks s, fold s, c, W, chunks]."""
    pts = np.ascontiguousarray(points, dtype=np.uint64)
    sc = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = sc.size // 4
    out = np.zeros(point_words(curve, group), dtype=np.uint64)
    st = (C.c_double * 6)()
    assert lib().oc_msm_fast(curve, group, _p(pts), _p(sc), C.c_size_t(n), int(montgomery), _t(threads), int(c), st, _p(out)) == 0
    if stages is not None:
        stages[:] = list(st)
    return out


def generate_bases(curve: int, group: int, seed: int, n: int, threads=0):
    out = np.zeros((n, point_words(curve, group)), dtype=np.uint64)
    assert lib().oc_generate_bases(curve, group, C.c_uint64(seed), C.c_size_t(n), _t(threads), _p(out)) == 0
    return out


def fixed_base_mul(curve: int, group: int, scalars_canonical, threads=0):
    """out[i] = k_i * G for canonical scalars (n, 4) u64 -> packed affine points (n, point_words)."""
    sc = np.ascontiguousarray(scalars_canonical, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros((len(sc), point_words(curve, group)), dtype=np.uint64)
    assert lib().oc_fixed_base_mul(curve, group, _p(sc), C.c_size_t(len(sc)), _t(threads), _p(out)) == 0
    return out


def generate_bases_wide(curve: int, group: int, seed: int, n: int, threads=0):
    """bases[i] = k_i * G with 253-bit k_i (four splitmix64 outputs); ~1 in 4096 is the point at infinity."""
    out = np.zeros((n, point_words(curve, group)), dtype=np.uint64)
    assert lib().oc_generate_bases_wide(curve, group, C.c_uint64(seed), C.c_size_t(n), _t(threads), _p(out)) == 0
    return out


def generate_bases_progression(curve: int, group: int, seed: int, n: int, threads=0):
    """Full-range points in blocks of 256: S_b + j D_b with 253-bit discrete logs s_b, d_b per block (one addition per point: 2^20
    G2 points in seconds where k G per point takes minutes)."""
    out = np.zeros((n, point_words(curve, group)), dtype=np.uint64)
    assert lib().oc_generate_bases_progression(curve, group, C.c_uint64(seed), C.c_size_t(n), _t(threads), _p(out)) == 0
    return out


def hash_points_bn254_g1(seed: int, n: int, threads=0):
    """SURVEY 8d family (i): x hashed, incremented until x^3 + 3 is a square; y = sqrt, sign from a PRNG bit."""
    out = np.zeros((n, 8), dtype=np.uint64)
    assert lib().oc_hash_points_bn254_g1(C.c_uint64(seed), C.c_size_t(n), _t(threads), _p(out)) == 0
    return out


def eval_poly(curve, coeffs, x, stride=1, offset=0):
    """Horner evaluation of sum_i coeffs[offset + i * stride] x^i over Fr (Montgomery limbs in and out)."""
    c = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    xx = np.ascontiguousarray(x, dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    n = (c.shape[0] - offset + stride - 1) // stride
    lib().oc_eval_poly(curve, C.c_void_p(c.ctypes.data + 32 * offset), C.c_size_t(n), C.c_size_t(stride), _p(xx), _p(out))
    return out


def vec_add(curve, a, b, threads=0):
    a, b = np.ascontiguousarray(a, dtype=np.uint64), np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty_like(a)
    lib().oc_vec_add(curve, _p(a), _p(b), _p(out), C.c_size_t(a.size // 4), _t(threads))
    return out


def lincomb(curve, shares, coeffs, threads=0):
    sh = [np.ascontiguousarray(s, dtype=np.uint64) for s in shares]
    k, n = len(sh), sh[0].size // 4
    arr = (C.c_void_p * k)(*[s.ctypes.data for s in sh])
    co = np.ascontiguousarray(coeffs, dtype=np.uint64)
    out = np.empty(n * 4, dtype=np.uint64)
    lib().oc_lincomb(curve, arr, _p(co), C.c_size_t(k), _p(out), C.c_size_t(n), _t(threads))
    return out


def ntt(curve: int, data, logn: int, gen, ncomp=1, dif=False, threads=0):
    d = np.ascontiguousarray(data, dtype=np.uint64).copy()
    g = np.ascontiguousarray(gen, dtype=np.uint64)
    assert lib().oc_ntt(curve, _p(d), logn, _p(g), ncomp, int(dif), _t(threads)) == 0
    return d


def bit_reverse(data, logn: int, ncomp=1):
    d = np.ascontiguousarray(data, dtype=np.uint64).copy()
    lib().oc_bit_reverse(_p(d), logn, ncomp)
    return d


def coset_table(curve: int, shift, logn: int):
    out = np.zeros((1 << logn) * 4, dtype=np.uint64)
    s = np.ascontiguousarray(shift, dtype=np.uint64)
    lib().oc_coset_table(curve, _p(s), logn, _p(out))
    return out


def vec_mul(curve, a, b, threads=0):
    a, b = np.ascontiguousarray(a, dtype=np.uint64), np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty_like(a)
    lib().oc_vec_mul(curve, _p(a), _p(b), _p(out), C.c_size_t(a.size // 4), _t(threads))
    return out


def rep3_local_mul_vec(curve, l, r, mask=None, threads=0):
    l, r = np.ascontiguousarray(l, dtype=np.uint64), np.ascontiguousarray(r, dtype=np.uint64)
    m = np.ascontiguousarray(mask, dtype=np.uint64) if mask is not None else None
    n = l.size // 8
    out = np.empty(n * 4, dtype=np.uint64)
    lib().oc_rep3_local_mul_vec(curve, _p(l), _p(r), _p(m), _p(out), C.c_size_t(n), _t(threads))
    return out


def vec_mul_table(curve, v, table, ncomp=1, threads=0):
    v = np.ascontiguousarray(v, dtype=np.uint64).copy()
    t = np.ascontiguousarray(table, dtype=np.uint64)
    lib().oc_vec_mul_table(curve, _p(v), _p(t), C.c_size_t(t.size // 4), ncomp, _t(threads))
    return v


def vec_sub(curve, a, b, threads=0):
    a, b = np.ascontiguousarray(a, dtype=np.uint64), np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty_like(a)
    lib().oc_vec_sub(curve, _p(a), _p(b), _p(out), C.c_size_t(a.size // 4), _t(threads))
    return out


def rep3_to_shamir_vec(curve, in_ab, x, y, threads=0):
    a = np.ascontiguousarray(in_ab, dtype=np.uint64)
    n = a.size // 8
    out = np.empty(n * 4, dtype=np.uint64)
    lib().oc_rep3_to_shamir_vec(curve, _p(a), _p(np.ascontiguousarray(x, dtype=np.uint64)), _p(np.ascontiguousarray(y, dtype=np.uint64)),
                                _p(out), C.c_size_t(n), _t(threads))
    return out


def use_native_build():
    """Rebuild the restatement with -march=native on the box it is timed on (oracle/c/liboracle_native.so, git-ignored) and
    switch this bridge to it. Falls back silently to the portable x86-64-v3 build when gcc is absent. Returns the flags used."""
    global _LIB
    d = os.path.join(_HERE, "c")
    try:
        subprocess.run(["make", "-s", "-C", d, "native"], check=True, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _LIB = C.CDLL(os.path.join(d, "liboracle_native.so"))
        return "-O3 -march=native"
    except Exception:
        lib()
        return "-O3 -march=x86-64-v3 (portable prebuilt)"


def _available_cpus() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def _cgroup_cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown. A box can
    expose 256 CPUs in its affinity mask and still be throttled to a few CPUs of time: more threads than that only adds
    contention (measured on the GPU boxes: linear to 16 threads, flat total throughput beyond)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def _best_of(f, reps=2):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        r = f()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best, r


def groth16_prove_composition(pts_g1, sc, pts_g2, logn, threads):
    """CPU figure beside `groth16_prove_synthetic_2p20`: the stages of one plain Groth16 prove of a 2^logn-constraint circuit
    (groth16.rs:125-338 with CircomReduction, reduction.rs:77-193) composed from this restatement's own routines on inputs of the
    prove's shapes: witness map = c = a.b, 3 x (ifft_in_to_out, coset-table multiplication, fft_out_to_in), a.b - c over BN254 Fr
    at n = 2^logn; then four G1 MSMs and one G2 MSM of 2^logn points (the bench step's own G1 bases and scalars; G2 bases from
    the same known-dlog family generated on the GPU and copied back). Stages run back to back (the reference overlaps them with
    rayon; with every core busy in each stage the sum is the comparable figure)."""
    n = 1 << logn
    r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    g = pow(pow(5, (r - 1) >> 28, r), 1 << (28 - logn), r) * ((1 << 256) % r) % r
    gen = np.array([(g >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
    rs = np.random.RandomState(11)
    a, b = (np.ascontiguousarray(rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64)) for _ in range(2))
    table = np.ascontiguousarray(rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64))
    fn = lib().oc_ntt
    t0 = time.perf_counter()
    c = vec_mul(0, a, b, threads=threads)                                   # reduction.rs:155-156 (on the evaluations)
    vs = [a.copy(), b.copy(), np.ascontiguousarray(c)]
    for v in vs:                                                            # :141-174
        assert fn(0, _p(v), logn, _p(gen), 1, 1, int(threads)) == 0        # ifft_in_to_out (DIF)
        lib().oc_vec_mul_table(0, _p(v), _p(table), C.c_size_t(n), 1, _t(threads))   # distribute_powers_and_mul_by_const / *= table (in place)
        assert fn(0, _p(v), logn, _p(gen), 1, 0, int(threads)) == 0        # fft_out_to_in (DIT)
    ab = vec_mul(0, vs[0], vs[1], threads=threads)                          # :176-181
    vec_sub(0, ab, vs[2], threads=threads)                                  # :185-190
    t_map = time.perf_counter() - t0
    p1 = np.ascontiguousarray(pts_g1).reshape(-1, 8)[:n]
    s1 = np.ascontiguousarray(sc).reshape(-1, 4)[:n]
    t0 = time.perf_counter()
    for _ in range(4):                                                      # a_query, b_g1_query, l_query, h_query (groth16.rs:237-292)
        msm_fast(0, 0, p1, s1, True, threads=threads)
    t_g1 = time.perf_counter() - t0
    t_g2 = None
    if pts_g2 is not None:
        p2 = np.ascontiguousarray(pts_g2).reshape(-1, 16)[:n]
        t0 = time.perf_counter()
        msm_fast(0, 1, p2, s1, True, threads=threads)                       # b_g2_query (:262-272)
        t_g2 = time.perf_counter() - t0
    total = t_map + t_g1 + (t_g2 or 0.0)
    return {"prove_ms": round(total * 1e3, 1), "witness_map_ms": round(t_map * 1e3, 1), "four_g1_msms_ms": round(t_g1 * 1e3, 1),
            "g2_msm_ms": round(t_g2 * 1e3, 1) if t_g2 is not None else None, "threads": threads, "log_n": logn,
            "sample": "stage sum of oracle/c routines on prove-shaped inputs (see groth16_prove_composition); no zkey parsing, no openings"}


def groth16_reference_circuit_composition(gold_dir, threads=1, reps=20):
    """CPU figures beside `groth16_prove_reference_circuits`: the hot stages of a plain Groth16 prove of the
    reference's own BN254 circuits (tests/golden copies of test_vectors/Groth16/bn254/{multiplier2, poseidon}) from this restatement's
    C routines ON THE CIRCUIT'S OWN DATA -- witness map (3 x ifft_in_to_out / coset table / fft_out_to_in, a.b, a.b - c on the real
    constraint evaluations) and the five query MSMs (the zkey's points, the witness / h as scalars) -- and the same for one Rep3 party
    (two-component share vectors through the transforms, local_mul_vec, one MSM per share component; x 3 for three parties on the same
    cores). Excluded because they are Python in this harness: zkey parsing, the sparse row evaluation (a few hundred rows), the three
    finishing scalar multiplications. Median of `reps`. A restatement, not arkworks; at these sizes (domain 4 / 256) a CPU wins on latency."""
    import os
    import statistics
    from . import curves as cv
    from . import groth16 as og
    from . import ntt as ontt
    from . import zkey as oz
    out = {}

    def pack_fr(F, xs):
        a = np.zeros((len(xs), 4), dtype=np.uint64)
        for i, x in enumerate(xs):
            v = x * F.R % F.p
            for k in range(4):
                a[i, k] = (v >> (64 * k)) & (2**64 - 1)
        return a

    for circ in ("multiplier2", "poseidon"):
        zk = oz.parse_zkey(open(os.path.join(gold_dir, circ, "circuit.zkey"), "rb").read())
        w = oz.parse_wtns(open(os.path.join(gold_dir, circ, "witness.wtns"), "rb").read())
        F = zk.Fr
        npub = zk.num_inputs
        pub = [x % F.p for x in w[:npub]]
        wit = [x % F.p for x in w[npub:]]
        drv = og.PlainDriver(F)
        A, Bm = zk.matrices()
        n = 1
        while n < zk.num_constraints + npub:
            n *= 2
        logn = n.bit_length() - 1
        gen, shift = ontt.groth16_roots_of_unity(F, logn)
        ev = lambda M: [drv.eval_row(r, pub, wit) for r in M] + [0] * (n - len(M))
        a = ev(A)
        a[zk.num_constraints:zk.num_constraints + npub] = pub[:npub]
        b = ev(Bm)
        pa, pb = pack_fr(F, a), pack_fr(F, b)
        table = pack_fr(F, ontt.bit_reversed_coset_table(F, shift, n))
        pg = pack_fr(F, [gen])
        fn = lib().oc_ntt
        aux = pack_fr(F, wit)
        G1, G2 = zk.G1, zk.G2
        q = {"a": cv.pack_points(G1, zk.a_query[1 + npub - 1:][: len(wit)] if False else zk.a_query[npub:][: len(wit)]),
             "b1": cv.pack_points(G1, zk.b_g1_query[npub:][: len(wit)]), "l": cv.pack_points(G1, zk.l_query[: len(wit)]),
             "h": cv.pack_points(G1, zk.h_query[:n]), "b2": cv.pack_points(G2, zk.b_g2_query[npub:][: len(wit)])}
        n_l = len(zk.l_query)

        def witness_map(ncomp):
            rep = lambda x: np.repeat(x, ncomp, axis=0) if ncomp > 1 else x   # a two-component share vector has the same shape of work
            va, vb = rep(pa).copy(), rep(pb).copy()
            c = (rep3_local_mul_vec(0, va, vb, None, threads=threads) if ncomp > 1 else vec_mul(0, va, vb, threads=threads)).reshape(-1, 4)
            vs = [va, vb, np.ascontiguousarray(c)]
            for k, v in enumerate(vs):
                nc = ncomp if k < 2 else 1
                assert fn(0, _p(v), logn, _p(pg), nc, 1, _t(threads)) == 0
                lib().oc_vec_mul_table(0, _p(v), _p(table), C.c_size_t(n), nc, _t(threads))
                assert fn(0, _p(v), logn, _p(pg), nc, 0, _t(threads)) == 0
            ab = (rep3_local_mul_vec(0, vs[0], vs[1], None, threads=threads) if ncomp > 1 else vec_mul(0, vs[0], vs[1], threads=threads)).reshape(-1, 4)
            return vec_sub(0, ab, vs[2], threads=threads)

        def msms(ncomp, h):
            for _ in range(ncomp):
                msm_fast(0, 0, q["a"], aux, True, threads=threads)
                msm_fast(0, 0, q["b1"], aux, True, threads=threads)
                msm_fast(0, 1, q["b2"], aux, True, threads=threads)
                msm_fast(0, 0, q["l"][: n_l], aux[-n_l:] if n_l else aux[:0], True, threads=threads)
            msm_fast(0, 0, q["h"], h[: len(zk.h_query)], True, threads=threads)     # h is a half share: one component

        def once(ncomp):
            t0 = time.perf_counter()
            h = witness_map(ncomp)
            t1 = time.perf_counter()
            msms(ncomp, np.ascontiguousarray(h).reshape(-1, 4))
            return (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3

        for ncomp, key in ((1, f"plain_{circ}"),) + (((2, "rep3_poseidon_one_party"),) if circ == "poseidon" else ()):
            once(ncomp)
            runs = [once(ncomp) for _ in range(reps)]
            wm = statistics.median(r[0] for r in runs)
            ms = statistics.median(r[1] for r in runs)
            out[key + "_ms"] = round(wm + ms, 3)
            out[key + "_stages_ms"] = {"witness_map": round(wm, 3), "five_query_msms": round(ms, 3)}
        if circ == "poseidon":
            out["rep3_poseidon_3_parties_ms"] = round(3 * out["rep3_poseidon_one_party_ms"], 3)
    out["threads"] = threads
    out["sample"] = ("hot-stage sum (witness map + query MSMs) of oracle/c routines on the reference's own circuits, median of %d; excludes zkey parsing, "
                     "sparse row evaluation and the finishing scalar multiplications (Python in this harness); kind \"port\", not arkworks" % reps)
    return out


def cpu_baseline_suite(pts20, sc20, gpu_affine20=None, pts24=None, sc24=None, gpu_affine24=None, curve=0, group=0, budget_s=30.0, host_cpus=None,
                       pts20_g2=None):
    """bench.py's ``cpu_baseline`` object (kind "port"): the tuned restatement `oc_msm_fast` timed on THIS box's host cores on
    the SAME inputs as the GPU line (the arrays are the GPU's bases / scalars copied back): BN254 G1 MSM at 2^20 (the bench
    workload) and at 2^24 (the north star's >= 10x target size), a thread-scaling table, plus NTT 2^22 and Rep3 local_mul_vec
    2^20 (mpc-core/benches/local_mul_vec.rs sizes). gpu_affine*: the GPU's affine results for a bit-exact comparison."""
    flags = use_native_build()
    # host_cpus: the CPUs this process may use, counted BEFORE any OpenMP runtime pinned the main thread to its first place
    # (with OMP_PROC_BIND set, sched_getaffinity of the main thread shrinks to one core once libgomp has initialised)
    hw = host_cpus or _available_cpus()
    n20 = np.ascontiguousarray(sc20).size // 4
    spent = time.perf_counter()
    table = []
    quota = _cgroup_cpu_quota()
    cands = sorted({t for t in (2, 4, 8, 16, 32, 64, 128, 256, hw) if t <= hw})
    best_t, best_v, res20 = 1, 0.0, None
    for t in cands:
        st = []
        dt, r = _best_of(lambda: msm_fast(curve, group, pts20, sc20, True, threads=t, stages=st), reps=2)
        W = st[4]
        table.append({"threads": t, "points_per_s": round(n20 / dt), "ms": round(dt * 1e3, 2), "mixed_adds_per_s_per_thread": round(n20 * W / max(st[1], 1e-9) / t),
                      "c": int(st[3]), "windows": int(W)})
        if n20 / dt > best_v:
            best_t, best_v, res20 = t, n20 / dt, r
        if time.perf_counter() - spent > budget_s * 0.4:
            break
        if len(table) >= 2 and table[-1]["points_per_s"] < 0.9 * max(r["points_per_s"] for r in table[:-1]) and t >= 32:
            break   # past the plateau (CPU-time quota of the container / shared host): more threads only add contention
    # single-thread row on a 2^17 prefix (bounded): the per-thread mixed-addition rate without any sharing effects
    m = min(n20, 1 << 17)
    st = []
    dt1, _ = _best_of(lambda: msm_fast(curve, group, np.ascontiguousarray(pts20).reshape(n20, -1)[:m], np.ascontiguousarray(sc20).reshape(n20, 4)[:m], True, threads=1, stages=st), reps=1)
    table.insert(0, {"threads": 1, "points_per_s": round(m / dt1), "ms": round(dt1 * 1e3, 2), "mixed_adds_per_s_per_thread": round(m * st[4] / max(st[1], 1e-9)),
                     "c": int(st[3]), "windows": int(st[4]), "sample": f"first 2^{m.bit_length() - 1} points"})
    # `cores` = CPUs' worth of compute the measurement actually had (the thread count capped by the container's CPU-time quota);
    # `threads` = OpenMP threads of the best row; `cpus_granted` = what the cgroup grants this container (None = unlimited)
    eff = best_t if quota is None else max(1, min(best_t, int(round(quota))))
    out = {"value": best_v, "unit": "points/s", "cores": eff, "threads": best_t, "cpus_granted": quota, "kind": "port", "host_cpus_available": hw,
           "cgroup_cpu_quota": quota, "build": flags,
           "sample": f"BN254 G1 MSM, the bench step's own 2^{n20.bit_length() - 1} bases and scalars copied back from the GPU; oracle/c oc_msm_fast "
                     f"(Booth signed digits, XYZZ mixed additions, thread-private buckets, __int128 Montgomery; OpenMP x{best_t}), best of 2; "
                     "a restatement of the published Pippenger shape, NOT arkworks (no Rust toolchain / un-vendored crates here)",
           "thread_scaling": table}
    if gpu_affine20 is not None:
        out["bit_exact_vs_gpu_2p20"] = bool((np.asarray(res20) == np.asarray(gpu_affine20)).all())
    if pts24 is not None and time.perf_counter() - spent < budget_s:
        n24 = np.ascontiguousarray(sc24).size // 4
        # bounded: the full 2^24 only if the 2^20 rate predicts <= ~12 s, otherwise the largest power-of-two prefix that does
        # (the bit-exact comparison with the GPU needs the full size)
        m24 = n24
        while m24 > (1 << 20) and m24 / best_v > 12.0:
            m24 >>= 1
        p24 = np.ascontiguousarray(pts24).reshape(n24, -1)[:m24]
        s24 = np.ascontiguousarray(sc24).reshape(n24, 4)[:m24]
        st = []
        dt, r24 = _best_of(lambda: msm_fast(curve, group, p24, s24, True, threads=best_t, stages=st), reps=1)
        out["msm_2p24"] = {"points_per_s": round(m24 / dt), "ms": round(dt * 1e3, 1), "threads": best_t, "c": int(st[3]), "windows": int(st[4]),
                           "mixed_adds_per_s_per_thread": round(m24 * st[4] / max(st[1], 1e-9) / best_t),
                           "sample": "all 2^24 points" if m24 == n24 else f"first 2^{m24.bit_length() - 1} of the 2^24 points (bounded sample)"}
        if gpu_affine24 is not None and m24 == n24:
            out["msm_2p24"]["bit_exact_vs_gpu"] = bool((np.asarray(r24) == np.asarray(gpu_affine24)).all())
    # NTT 2^22 (BASELINE config 3) and Rep3 local_mul_vec 2^20, BN254 Fr, uniform canonical inputs
    rs = np.random.RandomState(5)
    try:
        logn = 22
        r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
        g = pow(pow(5, (r - 1) >> 28, r), 1 << (28 - logn), r) * ((1 << 256) % r) % r
        gen = np.array([(g >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
        v = rs.randint(0, 1 << 62, size=(1 << logn, 4), dtype=np.uint64)
        d = np.ascontiguousarray(v).copy()
        fn = lib().oc_ntt
        best = None
        for t in sorted({min(hw, x) for x in (32, 64, best_t)}):
            t0 = time.perf_counter()
            assert fn(0, _p(d), logn, _p(gen), 1, 1, int(t)) == 0
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, t)
        out["ntt_2p22"] = {"elements_per_s": round((1 << logn) / best[0]), "ms": round(best[0] * 1e3, 1), "threads": best[1],
                           "sample": "oc_ntt ifft_in_to_out, radix-2 in place, twiddle table, OpenMP per stage"}
        n = 1 << 20
        a, b, mk = (rs.randint(0, 1 << 62, size=(k * n, 4), dtype=np.uint64) for k in (2, 2, 1))
        best = None
        for t in sorted({min(hw, x) for x in (8, 32, best_t)}):
            dt, _ = _best_of(lambda: rep3_local_mul_vec(0, a, b, mk, threads=t), reps=2)
            if best is None or dt < best[0]:
                best = (dt, t)
        out["rep3_local_mul_vec_2p20"] = {"elements_per_s": round(n / best[0]), "ms": round(best[0] * 1e3, 2), "threads": best[1],
                                          "sample": "oc_rep3_local_mul_vec (three products as written, ops.rs:69-76), incl. the output allocation"}
    except Exception as e:  # noqa: BLE001 -- extras never break the baseline
        out["extras_error"] = repr(e)
    try:
        if n20 >= (1 << 20):
            out["groth16_prove_synthetic_2p20"] = groth16_prove_composition(pts20, sc20, pts20_g2, 20, best_t)
    except Exception as e:  # noqa: BLE001
        out["groth16_prove_error"] = repr(e)
    out["wall_s"] = round(time.perf_counter() - spent, 1)
    return out
