"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/cosnarks_hip.h declares,
and -- with no GPU in this container -- every compute entry point fails loudly (there is no CPU fallback)."""
import ctypes as C
import os

import numpy as np
import pytest


def test_library_exports_every_declared_symbol(hip):
    from cosnarks_amd import bindings
    L = hip.lib()
    names = bindings.declared_symbols()
    assert len(names) >= 60
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_header_cites_reference_interfaces():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "include", "cosnarks_hip.h")).read()
    for cite in ["groth16.rs:193-194", "reduction.rs", "rep3/arithmetic.rs:132-146", "rngs.rs:137-156", "bridges/rep3_to_shamir.rs:43-62",
                 "mpc/rep3.rs:124-132", "shamir.rs:483-491"]:
        assert cite in txt, cite


def test_host_mirror_library_loads(hip):
    from cosnarks_amd import groth16
    G = groth16.glib()
    for sym in ["cog16_prove_plain", "cog16_prove_rep3", "cog16_prove_shamir", "cog16_bench_synthetic"]:
        assert hasattr(G, sym)


def test_no_cpu_fallback_without_device(hip):
    """In the CPU container the product must refuse to compute (CSH_ERR_NO_DEVICE), never fall back to a CPU path."""
    if hip.have_device():
        pytest.skip("a HIP device is present")
    a = np.zeros(8, dtype=np.uint64)
    with pytest.raises(hip.CoSnarksHipError, match="no HIP device|no CPU fallback"):
        hip.vec_mul(hip.BN254, a, a)
    with pytest.raises(hip.CoSnarksHipError, match="no HIP device|no CPU fallback"):
        hip.Domain(hip.BN254, 4, None)
    with pytest.raises(hip.CoSnarksHipError, match="no HIP device|no CPU fallback"):
        hip.Bases(hip.BN254, hip.G1, np.zeros(16, dtype=np.uint64))
    from cosnarks_amd import groth16 as g
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "Groth16", "bn254", "multiplier2")
    with pytest.raises(hip.CoSnarksHipError, match="no HIP device|no CPU fallback"):
        g.prove_plain(0, open(os.path.join(gold, "circuit.zkey"), "rb").read(), open(os.path.join(gold, "witness.wtns"), "rb").read(), 1, 2)


def test_split_msm_entry_points_without_device(hip):
    """The split-MSM / communicator / tuning entry points on a machine without a GPU: the argument checks work, the host-side
    fold works (it is pure host code: identity for empty partials), everything that needs a device fails loudly."""
    import ctypes as C
    from cosnarks_amd import bindings as B
    L = hip.lib()
    # tuning knobs are host state
    B.tune_set("msm_c", 13)
    assert B.tune_get("msm_c") == 13
    B.tune_set("msm_c", 0)
    with pytest.raises(hip.CoSnarksHipError, match="unknown key"):
        B.tune_set("no_such_knob", 1)
    # fold of well-formed empty partials = the identity (1, 1, 0) in Montgomery form; a bad header is refused
    for curve, group in ((0, 0), (1, 1), (2, 0)):
        pb = hip.msm_partial_bytes(curve, group)
        assert pb == 32 + 128 * 2 * hip.point_bytes(curve, group)
        parts = np.zeros(3 * pb, dtype=np.uint8)
        for k in range(3):
            parts[k * pb:k * pb + 4] = np.frombuffer((0x4d534d50).to_bytes(4, "little"), dtype=np.uint8)
        out = hip.msm_fold_partials(curve, group, parts, 3)
        w = out.size // 3
        assert not out[2 * w:].any() and out[:w].any()                 # Z = 0, X = Montgomery one
        with pytest.raises(hip.CoSnarksHipError, match="bad partial header"):
            hip.msm_fold_partials(curve, group, np.zeros(pb, dtype=np.uint8), 1)
    if hip.have_device():
        return
    with pytest.raises(hip.CoSnarksHipError, match="no HIP device|no CPU fallback"):
        hip.Comm.init_rank(None, 1, 0)
    with pytest.raises(hip.CoSnarksHipError):
        hip.Comm.init_rank(None, 2, 0)                                  # nranks > 1 needs an id
    with pytest.raises(hip.CoSnarksHipError):
        hip.Comm.init_all([0, 0])


def test_msm_planner_invariants(hip):
    """csh_msm_plan (host-only): the launch geometry the MSM would use. Lane length within [16, 1024]; the plan's round-count
    cost is never worse than the former power-of-two rule under the same model; the window reduction's segments cover every
    bucket and fit one round of waves; where n W / 64 is a multiple of the SIMD count the plan lands on a whole number of
    rounds (2^22 -> 128, 2^24 -> 256 on a 256-CU part); planning does not disturb csh_msm_last_params."""
    import math
    B = hip.bindings
    before = B.msm_last_params()

    def cost(n, W, L, simds):
        lanes = -(-n // L)
        waves = W * -(-lanes // 128) * 2
        return (math.ceil(waves / simds) + 0.2) * L + 4.6e-5 * n * W / L

    for curve in (0, 1, 2):
        for n in (1, 700, 5000, 1 << 14, 70001, 1 << 17, 1 << 18, 300007, 1 << 19, 1 << 20, 1200000, 1 << 21, 3000000, 1 << 22, 1 << 24, 20000003):
            c, W, L, S, waves, simds = B.msm_plan(curve, n)
            assert 2 <= c <= 16 and W * c >= 254 and 16 <= L <= 1024 and simds >= 4, (curve, n, c, W, L)
            assert waves == W * -(-(-(-n // L)) // 128) * 2
            old_L = 16
            while old_L < 1024 and 2 * old_L * (1 << 19) <= n * W:
                old_L <<= 1
            assert cost(n, W, L, simds) <= cost(n, W, old_L, simds) + 1e-9, (curve, n, L, old_L)
            NB = 1 << (c - 1)
            per = -(-NB // S)
            assert S >= 1 and S * per >= NB and -(-S // 64) * W <= max(simds, W), (curve, n, S, per)
    if B.msm_plan(0, 1 << 22)[5] == 1024:
        assert B.msm_plan(0, 1 << 22)[2] == 128 and B.msm_plan(0, 1 << 24)[2] == 256
        assert B.msm_plan(0, 1 << 22)[4] % 1024 == 0
    assert B.msm_last_params() == before
    with pytest.raises(hip.CoSnarksHipError):
        B.msm_plan(7, 100)


def test_msm_planner_balanced_windows(hip):
    """Round 5 (msm_impl.hpp choose_windows), host-only through csh_msm_plan: the balanced plan keeps the tuned uniform plan's NUMBER of
    windows and spreads bits + 1 bits evenly over them -- W (c - 1) < bits + 1 <= W c, so no window is left with a few bits on top -- with
    c never above the uniform width; msm_balanced = 0 gives the uniform plan back; a forced W (msm_w) is honoured inside its range."""
    B = hip.bindings
    bits = {0: 254, 1: 255, 2: 254}
    for curve in (0, 1, 2):
        total = bits[curve] + 1
        for n in (1, 300, 1 << 10, 5000, 1 << 13, 1 << 14, 1 << 15, 1 << 16, 70001, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 22, 1 << 24):
            with hip.tuned(msm_balanced=0):
                cu, wu = B.msm_plan(curve, n)[:2]
            c, w = B.msm_plan(curve, n)[:2]
            assert wu == -(-total // cu)                                     # the uniform plan: W = ceil((bits + 1) / c)
            assert w == wu and c <= cu and w * (c - 1) < total <= w * c, (curve, n, c, w, cu, wu)
            assert c == -(-total // w)
        for fw in (16, 17, 23, 40, 85, 127):
            with hip.tuned(msm_w=fw):
                c, w = B.msm_plan(curve, 1 << 16)[:2]
            assert w == fw and c == -(-total // fw) and c >= 3
        with hip.tuned(msm_w=15):                                            # below ceil((bits + 1) / 16): c would exceed 16 -- ignored
            assert B.msm_plan(curve, 1 << 16)[1] != 15
        with hip.tuned(msm_c=12):                                            # a forced width means the uniform form
            c, w = B.msm_plan(curve, 1 << 16)[:2]
            assert c == 12 and w == -(-total // 12)


def test_product_does_not_import_the_oracle():
    """Static check: nothing under co-snarks_amd/ references oracle/ (the oracle is test infrastructure only)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "co-snarks_amd")
    bad = []
    for dp, _dn, fn in os.walk(pkg):
        if "/build" in dp or "/lib" in dp or "__pycache__" in dp:
            continue
        for f in fn:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".inc")):
                t = open(os.path.join(dp, f), errors="ignore").read()
                if "oracle/" in t and f != "__init__.py" or "import oracle" in t or "from oracle" in t:
                    if f in ("selftest.hip",):
                        continue
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_host_mirror_ark_wire_formats_match_the_oracle_serializer():
    """arkwire.hpp (ark-serialize uncompressed Vec<F> and G1 points) against the oracle's restatement, byte
    for byte; host-only code path (no device call)."""
    import ctypes as C
    import numpy as np
    from cosnarks_amd import groth16 as g
    from oracle import arkfmt, curves as cv
    from tests import helpers as H
    L = g.glib()
    for curve, nb in (("bn254", 32), ("bls12_381", 48)):
        F = H.FR[curve]
        G = cv.CURVES[curve][0]
        cid = H.CURVE_IDS[curve]
        r = H.rng(17)
        vals = [0, 1, F.p - 1] + H.rand_elems(F, 9, r)
        out = (C.c_uint8 * 4096)()
        n = L.cog16_ark_roundtrip(cid, 0, H.pack(F, vals).ctypes.data_as(C.c_void_p), C.c_size_t(len(vals)), out, C.c_size_t(4096))
        assert n > 0, L.cog16_last_error()
        assert bytes(out[:n]) == arkfmt.ser_vec(vals, 32)
        pts = H.rand_points(G, 7, r, with_inf=True) + [G.neg(G.gen)]
        n = L.cog16_ark_roundtrip(cid, 1, cv.pack_points(G, pts).ctypes.data_as(C.c_void_p), C.c_size_t(len(pts)), out, C.c_size_t(4096))
        assert n > 0, L.cog16_last_error()
        assert bytes(out[:n]) == b"".join(arkfmt.ser_g1(pt, G.F.p, nb) for pt in pts)


def test_header_is_plain_c():
    """The drop-in boundary must be bindable from cgo / bindgen / ctypes: the header parses as C11 and as C++17 on its own."""
    import subprocess
    from cosnarks_amd import bindings as B
    for args in (["gcc", "-std=c11", "-x", "c"], ["g++", "-std=c++17", "-x", "c++"]):
        r = subprocess.run(args + ["-fsyntax-only", "-Wall", "-Werror", B.header_path()], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr



def _build_c_example(tmp_path, name="msm_ntt_from_c"):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / name)
    libdir = os.path.join(root, "co-snarks_amd", "lib")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", name + ".c"), "-L" + libdir, "-lcosnarks_hip", "-Wl,-rpath," + libdir, "-o", exe],
                   check=True, capture_output=True)
    return exe


def test_plain_c_caller_links_and_fails_loudly_without_a_gpu(hip, tmp_path):
    """examples/msm_ntt_from_c.c -- the boundary used from C11 with gcc, no C++ and no Python in between -- compiles
    warning-free, links against the library, and on a box without a GPU exits with the library's no-device error."""
    import subprocess
    import torch
    exe = _build_c_example(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert "no CPU fallback" in r.stderr


def test_plain_c_split_msm_caller_links_and_fails_loudly_without_a_gpu(hip, tmp_path):
    """examples/msm_split_from_c.c (one MSM over every GPU of the node from a single C thread) builds warning-free and, with no
    GPU, exits with status 3."""
    import subprocess
    import torch
    exe = _build_c_example(tmp_path, "msm_split_from_c")
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr)


def test_rust_sys_crate_covers_every_header_symbol():
    """rust/cosnarks-hip-sys/src/lib.rs is generated from include/cosnarks_hip.h (tools/gen_rust_sys.py): regenerating must be a
    no-op and every declared entry point must appear in the extern block (the crate cannot be compiled here: no rustc)."""
    import re
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(ROOT, "rust", "cosnarks-hip-sys", "src", "lib.rs")
    before = open(path).read()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py")], check=True, capture_output=True)
    assert open(path).read() == before, "rust/cosnarks-hip-sys/src/lib.rs is stale: run tools/gen_rust_sys.py"
    have = set(re.findall(r"pub fn (csh_\w+)\(", before))
    from cosnarks_amd import bindings
    assert have == set(bindings.declared_symbols())
    for f in ("lib.rs", "drivers.rs", "hip_reduction.rs", "layout.rs", "bases.rs", "domain.rs", "split.rs", "error.rs"):
        assert os.path.exists(os.path.join(ROOT, "rust", "co-groth16-hip", "src", f)), f


def test_experiment_knobs_are_not_in_the_product_library():
    """knob values documented as "wrong results" (ntt_variant bits 12-13 and 16-19: skipped passes / butterflies / loads /
    stores; allow_unmasked_rep3) exist only in builds with -DCSH_EXPERIMENTS. The product library drops them from the environment and
    refuses them in csh_tune_set; the harmless A/B bits of the same knob still work."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import cosnarks_amd as hip\n"
        "from cosnarks_amd import bindings as B\n"
        "assert B.tune_get('ntt_variant') == 0x800, hex(B.tune_get('ntt_variant'))\n"
        "assert B.tune_get('allow_unmasked_rep3') == 0\n"
        "for key, v in (('ntt_variant', 0x10000), ('ntt_variant', 0x1000), ('ntt_variant', 0x2800), ('allow_unmasked_rep3', 1)):\n"
        "    try:\n"
        "        B.tune_set(key, v)\n"
        "    except hip.CoSnarksHipError as e:\n"
        "        assert 'CSH_EXPERIMENTS' in str(e), e\n"
        "    else:\n"
        "        raise SystemExit('accepted %s=%#x' % (key, v))\n"
        "B.tune_set('ntt_variant', 0x100001)\n"
        "assert B.tune_get('ntt_variant') == 0x100001\n"
        "print('ok')\n")
    env = dict(os.environ, CSH_NTT_VARIANT="0x10800", CSH_ALLOW_UNMASKED_REP3="1", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout, r.stderr)


def test_every_accumulate_kernel_is_defined_in_exactly_one_object():
    """Round 6: the pinned accumulate kernels (msm_accum_*.hip, CSH_PIN_MADS) had been shadowed since round 4 by UNPINNED copies that
    msm.hip instantiated implicitly for its occupancy query -- two code objects registering one host stub, and the runtime launched the
    slower one (3-4 % on every G1 MSM). Every k_msm_accum* instantiation must live in one object, and that object must be msm_accum_*.o."""
    import collections
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir(os.path.join(root, "co-snarks_amd", "build")):
        pytest.skip("no build directory (prebuilt libraries only)")
    out = os.path.join(root, "co-snarks_amd", "build", "_kernel_meta_accum.csv")
    subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_meta.py"), "k_msm_accum", "--csv", out], check=True, capture_output=True, timeout=600)
    rows = [l.rstrip("\n").rsplit(",", 7) for l in open(out) if l.startswith('"')]
    os.remove(out)
    assert len(rows) >= 7, rows                               # five groups + the two lane-pair kernels of the G2 groups
    where = collections.defaultdict(set)
    for r in rows:
        where[r[0]].add(r[-1])
    for kernel, objs in where.items():
        assert len(objs) == 1 and next(iter(objs)).startswith("msm_accum_"), (kernel, objs)
