"""ctypes harness over lib/libcosnarks_groth16.so (the C++ host mirror of the reference's CoGroth16 interface)."""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

from .bindings import CoSnarksHipError, lib

_HERE = os.path.dirname(os.path.abspath(__file__))
_G = None


def glib():
    global _G
    if _G is None:
        lib()  # libcosnarks_hip.so first (dependency, same HIP runtime)
        p = os.environ.get("COSNARKS_GROTH16_LIB") or os.path.join(_HERE, "lib", "libcosnarks_groth16.so")   # (the override: A/B runs of two builds)
        if not os.path.exists(p):
            raise CoSnarksHipError(f"{p} not found: run `python co-snarks_amd/build.py`")
        _G = C.CDLL(p)
        _G.cog16_last_error.restype = C.c_char_p
    return _G


def _scalar(v):
    if v is None:
        return None
    return (C.c_uint64 * 4)(*[(int(v) >> (64 * i)) & (2**64 - 1) for i in range(4)])


def prove_plain(curve: int, zkey: bytes, wtns: bytes, r=None, s=None, want_h=False, h_elems=1 << 20):
    """Groth16::plain_prove::<CircomReduction> -> (proof dict as in circom.proof, h limbs or None)."""
    out = C.create_string_buffer(8192)
    h = np.zeros(h_elems * 4, dtype=np.uint64) if want_h else None
    rc = glib().cog16_prove_plain(curve, zkey, C.c_size_t(len(zkey)), wtns, C.c_size_t(len(wtns)), _scalar(r), _scalar(s), out,
                                  C.c_size_t(len(out)), h.ctypes.data_as(C.c_void_p) if want_h else None, C.c_size_t(h_elems))
    if rc != 0:
        raise CoSnarksHipError(glib().cog16_last_error().decode())
    return json.loads(out.value.decode()), h


def prove_rep3(curve: int, zkey: bytes, wtns: bytes, seed: int, r=None, s=None, want_h=False, h_elems=1 << 20):
    """Three in-process Rep3 parties (LocalNetwork); returns the agreed proof and the parties' h half-shares."""
    out = C.create_string_buffer(8192)
    h = np.zeros(3 * h_elems * 4, dtype=np.uint64) if want_h else None
    rc = glib().cog16_prove_rep3(curve, zkey, C.c_size_t(len(zkey)), wtns, C.c_size_t(len(wtns)), C.c_uint64(seed), _scalar(r), _scalar(s),
                                 out, C.c_size_t(len(out)), h.ctypes.data_as(C.c_void_p) if want_h else None, C.c_size_t(3 * h_elems))
    if rc != 0:
        raise CoSnarksHipError(glib().cog16_last_error().decode())
    return json.loads(out.value.decode()), h


def prove_shamir(curve: int, zkey: bytes, wtns: bytes, num_parties: int, threshold: int, seed: int, r=None, s=None, bridge=False):
    """ShamirCoGroth16::prove (or Rep3CoGroth16::prove_with_shamir_bridge when bridge=True) with in-process parties."""
    out = C.create_string_buffer(8192)
    rc = glib().cog16_prove_shamir(curve, zkey, C.c_size_t(len(zkey)), wtns, C.c_size_t(len(wtns)), num_parties, threshold, C.c_uint64(seed),
                                   _scalar(r), _scalar(s), int(bridge), out, C.c_size_t(len(out)))
    if rc != 0:
        raise CoSnarksHipError(glib().cog16_last_error().decode())
    return json.loads(out.value.decode())


def split_witness(curve: int, protocol: str, wtns: bytes, num_inputs: int, seed: int, compression: int = 0, threshold: int = 1,
                  num_parties: int = 3):
    """`co-circom split-witness`: the parties' `.shared` files (bincode) as a list of bytes. protocol "rep3" | "shamir";
    num_inputs counts the public inputs and the constant 1. Host-only."""
    proto = {"rep3": 0, "shamir": 1}[protocol]
    n_files = 3 if proto == 0 else num_parties
    cap = n_files * (2 * len(wtns) + 4096)
    out = (C.c_uint8 * cap)()
    sizes = (C.c_size_t * n_files)()
    rc = glib().cog16_split_witness(curve, proto, wtns, C.c_size_t(len(wtns)), C.c_size_t(num_inputs), compression, threshold, num_parties,
                                    C.c_uint64(seed), out, C.c_size_t(cap), sizes)
    if rc < 0:
        raise CoSnarksHipError(glib().cog16_last_error().decode())
    raw, files, at = bytes(out), [], 0
    for i in range(rc):
        files.append(raw[at:at + sizes[i]])
        at += sizes[i]
    return files


def share_file_roundtrip(curve: int, protocol: str, data: bytes):
    """Parse a `.shared` file with the host mirror and serialize it again -> (bytes, variant, n_public, n_witness)."""
    out = (C.c_uint8 * (len(data) + 64))()
    variant, npub, nwit = C.c_uint32(0), C.c_size_t(0), C.c_size_t(0)
    rc = glib().cog16_share_file_roundtrip(curve, {"rep3": 0, "shamir": 1}[protocol], data, C.c_size_t(len(data)), out, C.c_size_t(len(out)),
                                           C.byref(variant), C.byref(npub), C.byref(nwit))
    if rc < 0:
        raise CoSnarksHipError(glib().cog16_last_error().decode())
    return bytes(out)[:rc], variant.value, npub.value, nwit.value


def rep3_send_many(curve: int, items: np.ndarray, points: bool = False) -> bytes:
    """Rep3NetworkExt::send_many payload (mpc-core/src/protocols/rep3/network.rs:103-109): items = Montgomery limbs of n field elements
    (4 u64 each) or n G1 affine points in the C-ABI layout -> the bytes of the one message the reference puts on the wire."""
    arr = np.ascontiguousarray(items, dtype=np.uint64)
    per = (point_words(curve) if points else 4)
    n = arr.size // per
    cap = 16 + arr.nbytes
    out = (C.c_uint8 * cap)()
    fn = glib().cog16_rep3_wire
    fn.restype = C.c_long
    rc = fn(curve, 0, int(points), arr.ctypes.data_as(C.c_void_p), C.c_size_t(n), out, C.c_size_t(cap))
    if rc < 0:
        raise CoSnarksHipError(glib().cog16_last_error().decode())
    return bytes(out)[:rc]


def rep3_recv_many(curve: int, message: bytes, points: bool = False) -> np.ndarray:
    """Rep3NetworkExt::recv_many (network.rs:152-156): one received message -> Montgomery limbs / C-ABI points."""
    per = (point_words(curve) if points else 4)
    out = np.zeros(max(1, len(message) // 8 + 8), dtype=np.uint64)
    fn = glib().cog16_rep3_wire
    fn.restype = C.c_long
    rc = fn(curve, 1, int(points), message, C.c_size_t(len(message)), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes))
    if rc < 0:
        raise CoSnarksHipError(glib().cog16_last_error().decode())
    return out[:rc * per].copy()


def point_words(curve: int) -> int:
    return 8 if curve == 0 else 12     # G1 affine x || y in u64 words


def public_inputs_json(curve: int, protocol: str, share_file: bytes) -> str:
    """The public-input JSON `generate-proof` writes (decimal strings, constant 1 skipped), from a `.shared` file."""
    out = C.create_string_buffer(64 + 80 * 4096)
    rc = glib().cog16_public_inputs_json(curve, {"rep3": 0, "shamir": 1}[protocol], share_file, C.c_size_t(len(share_file)), out, C.c_size_t(len(out)))
    if rc != 0:
        raise CoSnarksHipError(glib().cog16_last_error().decode())
    return out.value.decode()


def prove_from_shares(curve: int, protocol: str, zkey: bytes, files, threshold: int = 1, seed: int = 1, r=None, s=None):
    """`co-circom generate-proof` from the parties' `.shared` files with in-process parties -> proof dict."""
    n = len(files)
    arr = (C.c_char_p * n)(*files)
    lens = (C.c_size_t * n)(*[len(f) for f in files])
    out = C.create_string_buffer(8192)
    rc = glib().cog16_prove_from_shares(curve, {"rep3": 0, "shamir": 1}[protocol], zkey, C.c_size_t(len(zkey)), arr, lens, n, threshold,
                                        C.c_uint64(seed), _scalar(r), _scalar(s), out, C.c_size_t(len(out)))
    if rc != 0:
        raise CoSnarksHipError(glib().cog16_last_error().decode())
    return json.loads(out.value.decode())


def translate_witness(curve: int, files):
    """`co-circom translate-witness`: three Rep3 `.shared` files -> three Shamir (n = 3, t = 1) `.shared` files."""
    arr = (C.c_char_p * 3)(*files)
    lens = (C.c_size_t * 3)(*[len(f) for f in files])
    _, _, n_pub, n_wit = share_file_roundtrip(curve, "rep3", files[0])   # seeded files are tiny, the Shamir files are not
    cap = 3 * (64 + 32 * (n_pub + n_wit))
    out = (C.c_uint8 * cap)()
    sizes = (C.c_size_t * 3)()
    rc = glib().cog16_translate_witness(curve, arr, lens, out, C.c_size_t(cap), sizes)
    if rc < 0:
        raise CoSnarksHipError(glib().cog16_last_error().decode())
    raw, res, at = bytes(out), [], 0
    for i in range(3):
        res.append(raw[at:at + sizes[i]])
        at += sizes[i]
    return res


PLACE_AUTO, PLACE_BY_QUERY, PLACE_BY_RANGE = 0, 1, 2


def set_prover_devices(devices, mode=PLACE_AUTO):
    """One prover's five query MSMs (rayon_join5, groth16.rs:227-294) over several GPUs: keys built afterwards clone their queries
    onto `devices` (entry 0 = the key's home GPU; a GPU may be listed more than once). mode: whole queries per GPU (BY_QUERY), the
    k-th range of every query per GPU (BY_RANGE), or AUTO (by query up to two GPUs, by range from three on). None / one entry: off."""
    d = list(devices or [])
    arr = (C.c_int * max(1, len(d)))(*d) if d else None
    if glib().cog16_set_prover_devices_mode(arr, len(d), int(mode)) != 0:
        raise CoSnarksHipError(glib().cog16_last_error().decode())


def bench_synthetic(curve: int, log_domain: int, iters: int = 3, with_rep3: bool = False):
    """Plain Groth16 prove on a synthetic 2^log_domain circuit with a known-dlog key (closed-form check); optionally
    also three in-process Rep3 parties proving the same circuit (BASELINE config 4 at scale). `prove_ms` = prove_inner with the
    key resident; `prove_phases_ms` = host-clock phases INSIDE that prove (witness upload + device witness map, the five MSM groups
    = create_proof_with_assignment up to the join, the finish); `witness_map_ms` = the standalone host-facing
    witness_map_from_matrices (h returned to the host), a different code path."""
    ms = (C.c_double * 6)()
    ph = (C.c_double * 3)()
    tr = (C.c_double * 5)()
    r3 = (C.c_double * 26)()
    mn = (C.c_double * 3)()
    ok = C.c_int(0)
    rc = glib().cog16_bench_synthetic4(curve, log_domain, iters, ms, C.byref(ok), int(with_rep3), ph, tr, r3 if with_rep3 else None, mn)
    if rc != 0:
        raise CoSnarksHipError(glib().cog16_last_error().decode())
    out = {"log_domain": log_domain, "statistic": "median of %d runs after 2 warm-up runs (minimum beside it)" % iters,
           "witness_map_ms": ms[0], "prove_ms": ms[2], "prove_ms_min": mn[0],
           "prove_phases_ms": {"witness_upload_and_map": ph[0], "msm_groups": ph[1], "finish": ph[2]},
           # the same prove driven the way rust/co-groth16-hip drives the C ABI behind the UNCHANGED reference: host slices in and out
           # of every seam call (one csh_groth16_witness_map_masks, h on the host, five concurrent csh_msm calls with host scalars)
           "trait_path_ms": tr[0], "trait_path_ms_min": mn[1],
           "trait_path_phases_ms": {"witness_map_host_slices": tr[1], "msm_groups_host_scalars": tr[2], "finish": tr[3]},
           "trait_path_closed_form_check": bool(tr[4]),
           "key_setup_ms": ms[3], "closed_form_check": bool(ok.value)}
    if with_rep3:
        # the device-resident mirror: masks from the party's own ChaCha12 keys on the device -- NOT reachable behind the unchanged
        # reference (Rep3Rand's generators are private, mpc-core/src/protocols/rep3/rngs.rs:83-86)
        out["rep3_three_parties_prove_ms"] = ms[4]
        out["rep3_three_parties_prove_ms_min"] = mn[2]
        out["rep3_proofs_equal_plain"] = bool(ms[5])

        def mode(o):
            return {"three_parties_one_gpu_ms": o[0], "three_parties_one_gpu_ms_min": o[1],
                    "party0_phases_ms": {"mask_draw": o[2], "witness_map_host_slices_incl_masks": o[3], "to_half_share": o[4],
                                         "msm_groups_host_scalars": o[5], "finish": o[6]},
                    "proofs_equal_plain": bool(o[7]),
                    # one party with the GPU to itself (one GPU per party), its compute between the network rounds: witness map +
                    # to_half_share + five MSMs (the finish's three curve points on the wire left out)
                    "one_party_alone_ms": o[8],
                    "one_party_alone_phases_ms": {"mask_draw": o[9], "witness_map_host_slices_incl_masks": o[10], "to_half_share": o[11],
                                                  "msm_groups_host_scalars": o[12]}}

        # BASELINE config 4 through the zero-upstream-edit path: what rust/co-groth16-hip drives for a Rep3 party. "host_masks" = its
        # default (two masking_field_elements_vec draws per witness map on the host, rngs.rs:137-156, inside the timed region);
        # "seeded_device_masks" = its opt-in all-GPU-parties mode (one public Rep3Rand::random_seeds() draw, rngs.rs:233, masks generated
        # on the device; not wire-compatible with a reference CPU party in the same session)
        out["rep3_trait_path"] = {"host_masks": mode(r3[0:13]), "seeded_device_masks": mode(r3[13:26])}
    return out


def bench_rep3_party_per_gpu(curve: int, log_domain: int, devices, iters: int = 3):
    """Three in-process Rep3 parties prove the synthetic circuit, party p bound to devices[p] with its own key copy there (BASELINE
    config 4: one GPU per party; `devices` may repeat to fold the parties onto fewer GPUs)."""
    dv = (C.c_int * 3)(*[int(d) for d in devices])
    out = (C.c_double * 3)()
    if glib().cog16_bench_rep3_party_per_gpu(curve, log_domain, iters, dv, out) != 0:
        raise CoSnarksHipError(glib().cog16_last_error().decode())
    return {"log_domain": log_domain, "party_devices": [int(d) for d in devices], "three_parties_prove_ms": out[0], "proofs_equal_plain": bool(out[1]),
            "key_setup_ms_per_device": out[2]}


class trait_path:
    """`with trait_path():` -- every prove of the host mirror inside the block runs the way rust/co-groth16-hip drives the C ABI behind
    the unchanged reference: host slices in and out of every seam call (one csh_groth16_witness_map_masks per witness map, h on the
    host, five concurrent csh_msm calls with host scalars)."""

    def __init__(self, on: bool = True):
        self.on = on

    def __enter__(self):
        self.prev = glib().cog16_get_trait_path()
        glib().cog16_set_trait_path(int(self.on))   # True / 1: host masks; 2: seeded device masks for Rep3 (all parties must use it)
        return self

    def __exit__(self, *exc):
        glib().cog16_set_trait_path(self.prev)
        return False


class SynthCircuit:
    """The synthetic 2^log_domain-constraint circuit with its known-dlog proving key resident on the device (queries placed on
    the GPUs of `set_prover_devices`, if any): `prove()` = one plain `prove_inner`, `check()` = closed-form check of a proof."""

    def __init__(self, curve: int, log_domain: int):
        self.h = C.c_void_p()
        self.log_domain = log_domain
        if glib().cog16_synth_open(curve, log_domain, C.byref(self.h)) != 0:
            raise CoSnarksHipError(glib().cog16_last_error().decode())

    def prove(self):
        ph = (C.c_double * 3)()
        key = C.c_double(0)
        if glib().cog16_synth_prove(self.h, ph, C.byref(key)) != 0:
            raise CoSnarksHipError(glib().cog16_last_error().decode())
        return {"witness_upload_and_map": ph[0], "msm_groups": ph[1], "finish": ph[2], "key_setup_ms": key.value}

    def check(self) -> bool:
        ok = C.c_int(0)
        if glib().cog16_synth_check(self.h, C.byref(ok)) != 0:
            raise CoSnarksHipError(glib().cog16_last_error().decode())
        return bool(ok.value)

    def close(self):
        if self.h:
            glib().cog16_synth_close(self.h)
            self.h = C.c_void_p()


CIRCOM_REDUCTION, LIBSNARK_REDUCTION = 0, 1


def witness_map(curve: int, reduction: int, rep3: bool, matrices, n_instance: int, witness_mont: np.ndarray, seed: int = 1):
    """R1CSToQAP::witness_map_from_matrices of the host mirror (CircomReduction / LibSnarkReduction) on explicit matrices.
    matrices = (A, B, C) rows of (Montgomery coeff limbs as 4-tuple of u64 | int already in Montgomery form, column);
    C may be None for the circom reduction. witness_mont: (n_vars, 4) u64 Montgomery limbs of public || private values.
    Returns h limbs: (n, 4) plain, or (3, n, 4) half-share vectors of the three Rep3 parties."""
    n_rows = len(matrices[0])
    keep = []
    rp = (C.POINTER(C.c_uint64) * 3)()
    cl = (C.POINTER(C.c_uint32) * 3)()
    cf = (C.POINTER(C.c_uint64) * 3)()
    for k, m in enumerate(matrices):
        if m is None:
            continue
        assert len(m) == n_rows
        row_ptr = np.zeros(n_rows + 1, dtype=np.uint64)
        cols, vals = [], []
        for i, row in enumerate(m):
            for coeff, idx in row:
                cols.append(idx)
                vals.append([(int(coeff) >> (64 * j)) & (2**64 - 1) for j in range(4)])
            row_ptr[i + 1] = len(cols)
        col = np.array(cols, dtype=np.uint32)
        val = np.array(vals, dtype=np.uint64).reshape(-1, 4)
        keep += [row_ptr, col, val]
        rp[k] = row_ptr.ctypes.data_as(C.POINTER(C.c_uint64))
        cl[k] = col.ctypes.data_as(C.POINTER(C.c_uint32))
        cf[k] = val.ctypes.data_as(C.POINTER(C.c_uint64))
    w = np.ascontiguousarray(witness_mont, dtype=np.uint64).reshape(-1, 4)
    n = 1
    while n < n_rows + n_instance:
        n *= 2
    out = np.zeros((3 if rep3 else 1) * n * 4, dtype=np.uint64)
    rc = glib().cog16_witness_map(curve, reduction, int(rep3), rp, cl, cf, C.c_size_t(n_rows), C.c_size_t(n_instance),
                                  w.ctypes.data_as(C.c_void_p), C.c_size_t(len(w)), C.c_uint64(seed), out.ctypes.data_as(C.c_void_p),
                                  C.c_size_t(len(out) // 4))
    if rc < 0:
        raise CoSnarksHipError(glib().cog16_last_error().decode())
    assert rc == n
    return out.reshape(3, n, 4) if rep3 else out.reshape(n, 4)


PLAIN, REP3, SHAMIR, FAST_MSM = 0, 1, 2, 3


def driver_fft(curve: int, driver: int, data_mont: np.ndarray, domain_size: int, inverse=False, snarkjs=True, seed: int = 1):
    """CircomPlonkProver / NoirUltraHonkProver ::{fft, ifft} of the host mirror. Returns (domain, 4) limbs, or for Rep3
    (3, domain, 2, 4): every party's share vector (the values are shared inside with `seed`)."""
    d = np.ascontiguousarray(data_mont, dtype=np.uint64).reshape(-1, 4)
    n = 1
    while n < domain_size:
        n *= 2
    out = np.zeros((3 * 2 if driver == REP3 else 1) * n * 4, dtype=np.uint64)
    rc = glib().cog16_driver_fft(curve, driver, int(inverse), int(snarkjs), d.ctypes.data_as(C.c_void_p), C.c_size_t(len(d)),
                                 C.c_size_t(domain_size), C.c_uint64(seed), out.ctypes.data_as(C.c_void_p))
    if rc < 0:
        raise CoSnarksHipError(glib().cog16_last_error().decode())
    return out.reshape(3, n, 2, 4) if driver == REP3 else out.reshape(n, 4)


def driver_local_mul_vec(curve: int, driver: int, a_mont, b_mont, seed: int = 1):
    """::local_mul_vec. Plain / Shamir: (n, 4); Rep3: (3, n, 4) additive (masked) shares of the three parties."""
    a = np.ascontiguousarray(a_mont, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b_mont, dtype=np.uint64).reshape(-1, 4)
    n = len(a)
    out = np.zeros((3 if driver == REP3 else 1) * n * 4, dtype=np.uint64)
    rc = glib().cog16_driver_local_mul_vec(curve, driver, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.c_size_t(n),
                                           C.c_uint64(seed), out.ctypes.data_as(C.c_void_p))
    if rc < 0:
        raise CoSnarksHipError(glib().cog16_last_error().decode())
    return out.reshape(3, n, 4) if driver == REP3 else out.reshape(n, 4)


def driver_msm(curve: int, driver: int, points: np.ndarray, scalars_mont, seed: int = 1):
    """::msm_public_points(_g1) / HonkCurve::fast_msm (driver FAST_MSM; curve 2 = Grumpkin). Returns affine limbs:
    (point_words,) or for Rep3 (3, 2, point_words): the (a, b) point share of every party."""
    pts = np.ascontiguousarray(points, dtype=np.uint64)
    sc = np.ascontiguousarray(scalars_mont, dtype=np.uint64).reshape(-1, 4)
    pw = 12 if curve == 1 else 8
    npts = pts.size // pw
    out = np.zeros((6 if driver == REP3 else 1) * pw, dtype=np.uint64)
    rc = glib().cog16_driver_msm(curve, driver, pts.ctypes.data_as(C.c_void_p), C.c_size_t(npts), sc.ctypes.data_as(C.c_void_p),
                                 C.c_size_t(len(sc)), C.c_uint64(seed), out.ctypes.data_as(C.c_void_p))
    if rc < 0:
        raise CoSnarksHipError(glib().cog16_last_error().decode())
    return out.reshape(3, 2, pw) if driver == REP3 else out

