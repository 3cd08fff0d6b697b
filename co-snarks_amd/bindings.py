"""ctypes binding of include/cosnarks_hip.h (one Python function per C entry point, numpy u64 in/out).

Field elements travel as little-endian u64 limb arrays in arkworks (Montgomery) layout, exactly the bytes
a Rust ``&[Fr]`` holds -- see the header for conventions.  No arithmetic happens in this file.
"""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np

BN254, BLS12_381, GRUMPKIN, BLS12_377 = 0, 1, 2, 3   # GRUMPKIN: MSM only (G1); BLS12_377: MSM (G1, G2), NTT, share vectors, reductions (no host prover mirror)
G1, G2 = 0, 1

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class CoSnarksHipError(RuntimeError):
    pass


def lib_path() -> str:
    """The in-tree build; COSNARKS_HIP_LIB points A/B runs of kernel variants at another build of the same sources."""
    return os.environ.get("COSNARKS_HIP_LIB") or os.path.join(_HERE, "lib", "libcosnarks_hip.so")


def header_path() -> str:
    return os.path.join(os.path.dirname(_HERE), "include", "cosnarks_hip.h")


def declared_symbols() -> list:
    """Every function name declared in include/cosnarks_hip.h."""
    txt = open(header_path()).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(csh_[a-z0-9_]+)\s*\(", txt)))


def lib():
    """Load the HIP library (fails loudly if it has not been built)."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise CoSnarksHipError(
                f"{p} not found: build it with `python co-snarks_amd/build.py` (hipcc, gfx950). "
                "There is no CPU fallback.")
        L = C.CDLL(p)
        L.csh_last_error.restype = C.c_char_p
        L.csh_version.restype = C.c_char_p
        _LIB = L
    return _LIB


def _check(rc: int):
    if rc != 0:
        raise CoSnarksHipError(f"cosnarks_hip error {rc}: {lib().csh_last_error().decode(errors='replace')}")


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().csh_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def have_device() -> bool:
    try:
        return device_count() > 0
    except CoSnarksHipError:
        return False


def fr_bytes(curve: int) -> int:
    return 32


def fq_bytes(curve: int) -> int:
    return 48 if curve in (BLS12_381, BLS12_377) else 32


def point_bytes(curve: int, group: int) -> int:
    return 2 * fq_bytes(curve) * (2 if group == G2 else 1)


def _u64(a, copy=False):
    arr = np.ascontiguousarray(a, dtype=np.uint64)
    return arr.copy() if copy and arr is a else arr


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class DeviceBuffer:
    """Raw device allocation through the C ABI (csh_malloc / csh_memcpy_*)."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        self.ptr = C.c_void_p()
        _check(lib().csh_malloc(C.byref(self.ptr), C.c_size_t(self.nbytes)))

    @classmethod
    def from_host(cls, arr):
        a = np.ascontiguousarray(arr)
        b = cls(a.nbytes)
        _check(lib().csh_memcpy_h2d(b.ptr, _p(a), C.c_size_t(a.nbytes)))
        return b

    def to_host(self, dtype=np.uint64, count=None):
        n = self.nbytes // np.dtype(dtype).itemsize if count is None else count
        out = np.empty(n, dtype=dtype)
        _check(lib().csh_memcpy_d2h(_p(out), self.ptr, C.c_size_t(out.nbytes)))
        return out

    def free(self):
        if self.ptr:
            lib().csh_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _devptr(x):
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if x is None:
        return None
    if isinstance(x, C.c_void_p):
        return x
    return C.c_void_p(int(x))  # raw address, e.g. torch.Tensor.data_ptr()


def _stream(s):
    return C.c_void_p(int(s)) if s else None


class Bases:
    """Device-resident MSM bases (a proving-key query): csh_bases_upload / csh_msm."""

    def __init__(self, curve: int, group: int, points, stride_bytes: int = 0):
        self.curve, self.group = curve, group
        pts = _u64(points)
        pb = point_bytes(curve, group)
        stride = stride_bytes or pb
        assert pts.nbytes % stride == 0
        self.n = pts.nbytes // stride
        self.h = C.c_void_p()
        _check(lib().csh_bases_upload(curve, group, _p(pts), C.c_size_t(self.n), C.c_size_t(stride_bytes), C.byref(self.h)))

    def _out(self):
        return np.zeros(3 * point_bytes(self.curve, self.group) // 2 // 8, dtype=np.uint64)

    def precompute(self, c: int = 0, groups: int = 0):
        """csh_bases_precompute[_grouped]: fixed-base tables (merged-window MSMs on this handle); groups = 0: one row per window."""
        if groups:
            _check(lib().csh_bases_precompute_grouped(self.h, int(c), int(groups)))
        else:
            _check(lib().csh_bases_precompute(self.h, int(c)))
        return self

    def msm(self, scalars, offset: int = 0, n: int | None = None, montgomery: bool = True):
        """-> Jacobian (X, Y, Z) limbs (Z in {0, 1}). scalars: (n, 4) u64 host array."""
        sc = _u64(scalars)
        cnt = sc.size // 4 if n is None else n
        out = self._out()
        _check(lib().csh_msm(self.h, C.c_size_t(offset), C.c_size_t(cnt), _p(sc), int(montgomery), _p(out)))
        return out

    def msm_dev(self, scalars_dev, n: int, offset: int = 0, montgomery: bool = True, stream=None):
        out = self._out()
        _check(lib().csh_msm_dev(self.h, C.c_size_t(offset), C.c_size_t(n), _devptr(scalars_dev), int(montgomery), _p(out), _stream(stream)))
        return out

    def msm_partial_dev(self, scalars_dev, n: int, out_dev, offset: int = 0, montgomery: bool = True, stream=None):
        _check(lib().csh_msm_partial_dev(self.h, C.c_size_t(offset), C.c_size_t(n), _devptr(scalars_dev), int(montgomery),
                                         _devptr(out_dev), _stream(stream)))

    def free(self):
        if self.h:
            lib().csh_bases_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def msm_partial_bytes(curve: int, group: int) -> int:
    n = C.c_size_t(0)
    _check(lib().csh_msm_partial_bytes(curve, group, C.byref(n)))
    return n.value


def msm_fold_partials(curve: int, group: int, partials: np.ndarray, nparts: int):
    buf = np.ascontiguousarray(partials)
    out = np.zeros(3 * point_bytes(curve, group) // 2 // 8, dtype=np.uint64)
    _check(lib().csh_msm_fold_partials(curve, group, _p(buf), C.c_size_t(nparts), _p(out)))
    return out


def msm_plan(curve: int, n: int):
    """csh_msm_plan: (c, W, L, S, accumulate waves, SIMDs) an n-point MSM would run with; host-only."""
    out = (C.c_uint32 * 6)()
    _check(lib().csh_msm_plan(curve, C.c_size_t(n), out))
    return tuple(int(x) for x in out)


def tune_set(key: str, value: int):
    """csh_tune_set: process-wide tuning knob (tests / A-B runs)."""
    _check(lib().csh_tune_set(key.encode(), int(value)))


def tune_get(key: str) -> int:
    v = C.c_int(0)
    _check(lib().csh_tune_get(key.encode(), C.byref(v)))
    return v.value


class tuned:
    """with tuned(msm_c=11, sort_two_level=1): ...  -- sets knobs, restores the previous values on exit."""

    def __init__(self, **kv):
        self.kv = kv
        self.old = {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = tune_get(k)
            tune_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            tune_set(k, v)
        return False


SPLIT_PEER, SPLIT_HOST, SPLIT_RCCL = 0, 1, 2
COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(lib().csh_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """csh_comm_t: one rank of a split-MSM communicator (RCCL behind the C ABI; no torch)."""

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def init_rank(cls, uid, nranks: int, rank: int):
        h = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(uid) if uid is not None else None
        _check(lib().csh_comm_init_rank(buf, int(nranks), int(rank), C.byref(h)))
        return cls(h)

    @classmethod
    def init_all(cls, devices):
        k = len(devices)
        arr = (C.c_int * k)(*devices)
        hs = (C.c_void_p * k)()
        _check(lib().csh_comm_init_all(arr, k, hs))
        return [cls(C.c_void_p(hs[i])) for i in range(k)]

    def info(self):
        r, n, d = C.c_int(0), C.c_int(0), C.c_int(0)
        _check(lib().csh_comm_info(self.h, C.byref(r), C.byref(n), C.byref(d)))
        return r.value, n.value, d.value

    def msm_split_rank_dev(self, bases: "Bases", scalars_dev, n: int, offset: int = 0, montgomery: bool = True, stream=None):
        out = bases._out()
        _check(lib().csh_msm_split_rank_dev(self.h, bases.h, C.c_size_t(offset), C.c_size_t(n), _devptr(scalars_dev), int(montgomery),
                                            _p(out), _stream(stream)))
        return out

    def destroy(self):
        if self.h:
            lib().csh_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def msm_split(bases, offsets, counts, scalars_dev, montgomery: bool = True, mode: int = SPLIT_PEER, comms=None):
    """csh_msm_split: one MSM over k ranges (one thread drives every device). bases: list of Bases; scalars_dev: list of
    device pointers / DeviceBuffers."""
    k = len(bases)
    hs = (C.c_void_p * k)(*[b.h.value for b in bases])
    offs = (C.c_size_t * k)(*offsets)
    cnts = (C.c_size_t * k)(*counts)
    ptrs = (C.c_void_p * k)(*[(_devptr(s).value if s is not None else None) for s in scalars_dev])
    cm = (C.c_void_p * k)(*[c.h.value for c in comms]) if comms else None
    out = bases[0]._out()
    _check(lib().csh_msm_split(hs, offs, cnts, ptrs, C.c_size_t(k), int(montgomery), int(mode), cm, _p(out)))
    return out


def msm_last_timing():
    out = (C.c_float * 6)()
    _check(lib().csh_msm_last_timing(out))
    return list(out)


def msm_last_params():
    """[window bits c, windows W, entries per lane L, reduction segments S] of the last MSM on this thread."""
    out = (C.c_uint32 * 4)()
    _check(lib().csh_msm_last_params(out))
    return list(out)


class Domain:
    """taceo_ark_algebra::fft::Domain equivalent: csh_domain_create + transforms."""

    def __init__(self, curve: int, log_n: int, group_gen=None):
        self.curve, self.log_n, self.n = curve, log_n, 1 << log_n
        self.h = C.c_void_p()
        gg = _u64(group_gen) if group_gen is not None else None
        _check(lib().csh_domain_create(curve, C.c_uint32(log_n), _p(gg), C.byref(self.h)))

    def _host(self, fn, data, ncomp):
        d = _u64(data).copy()
        assert d.size == self.n * ncomp * 4, (d.size, self.n, ncomp)
        _check(fn(self.h, _p(d), C.c_uint32(ncomp)))
        return d

    def ifft_in_to_out(self, data, ncomp=1):
        return self._host(lib().csh_ifft_in_to_out, data, ncomp)

    def fft_out_to_in(self, data, ncomp=1):
        return self._host(lib().csh_fft_out_to_in, data, ncomp)

    def fft(self, data, ncomp=1):
        return self._host(lib().csh_fft, data, ncomp)

    def ifft(self, data, ncomp=1):
        return self._host(lib().csh_ifft, data, ncomp)

    def coset_table(self, shift):
        out = np.zeros(self.n * 4, dtype=np.uint64)
        sh = _u64(shift)
        _check(lib().csh_coset_table(self.h, _p(sh), _p(out)))
        return out

    # device-pointer variants (asynchronous on `stream`)
    def ifft_in_to_out_dev(self, data_dev, ncomp=1, stream=None):
        _check(lib().csh_ifft_in_to_out_dev(self.h, _devptr(data_dev), C.c_uint32(ncomp), _stream(stream)))

    def fft_out_to_in_dev(self, data_dev, ncomp=1, stream=None):
        _check(lib().csh_fft_out_to_in_dev(self.h, _devptr(data_dev), C.c_uint32(ncomp), _stream(stream)))

    def fft_dev(self, data_dev, ncomp=1, stream=None):
        _check(lib().csh_fft_dev(self.h, _devptr(data_dev), C.c_uint32(ncomp), _stream(stream)))

    def ifft_dev(self, data_dev, ncomp=1, stream=None):
        _check(lib().csh_ifft_dev(self.h, _devptr(data_dev), C.c_uint32(ncomp), _stream(stream)))

    def free(self):
        if self.h:
            lib().csh_domain_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def bit_reverse(curve: int, data, log_n: int, ncomp: int = 1):
    d = _u64(data).copy()
    _check(lib().csh_bit_reverse(curve, _p(d), C.c_uint32(log_n), C.c_uint32(ncomp)))
    return d


def vec_mul(curve: int, a, b):
    a, b = _u64(a), _u64(b)
    out = np.empty_like(a)
    _check(lib().csh_vec_mul(curve, _p(a), _p(b), _p(out), C.c_size_t(a.size // 4)))
    return out


def vec_add(curve: int, a, b, ncomp: int = 1):
    a, b = _u64(a), _u64(b)
    out = np.empty_like(a)
    _check(lib().csh_vec_add(curve, _p(a), _p(b), _p(out), C.c_size_t(a.size // (4 * ncomp)), C.c_uint32(ncomp)))
    return out


def vec_sub(curve: int, a, b, ncomp: int = 1):
    a, b = _u64(a), _u64(b)
    out = np.empty_like(a)
    _check(lib().csh_vec_sub(curve, _p(a), _p(b), _p(out), C.c_size_t(a.size // (4 * ncomp)), C.c_uint32(ncomp)))
    return out


def vec_mul_table(curve: int, v, table, ncomp: int = 1):
    v = _u64(v).copy()
    t = _u64(table)
    _check(lib().csh_vec_mul_table(curve, _p(v), _p(t), C.c_size_t(t.size // 4), C.c_uint32(ncomp)))
    return v


def rep3_local_mul_vec(curve: int, lhs_ab, rhs_ab, mask=None):
    l, r = _u64(lhs_ab), _u64(rhs_ab)
    n = l.size // 8
    m = _u64(mask) if mask is not None else None
    out = np.empty(n * 4, dtype=np.uint64)
    _check(lib().csh_rep3_local_mul_vec(curve, _p(l), _p(r), _p(m), _p(out), C.c_size_t(n)))
    return out


def rep3_to_shamir_vec(curve: int, in_ab, x, y):
    a = _u64(in_ab)
    n = a.size // 8
    out = np.empty(n * 4, dtype=np.uint64)
    xx, yy = _u64(x), _u64(y)
    _check(lib().csh_rep3_to_shamir_vec(curve, _p(a), _p(xx), _p(yy), _p(out), C.c_size_t(n)))
    return out


def lincomb(curve: int, shares, coeffs):
    sh = [_u64(s) for s in shares]
    k = len(sh)
    n = sh[0].size // 4
    arr = (C.c_void_p * k)(*[s.ctypes.data for s in sh])
    co = _u64(coeffs)
    out = np.empty(n * 4, dtype=np.uint64)
    _check(lib().csh_lincomb(curve, arr, _p(co), C.c_size_t(k), _p(out), C.c_size_t(n)))
    return out


def groth16_h(dom: Domain, shift, protocol: int, a, b, mask_c=None, mask_ab=None):
    a, b = _u64(a).copy(), _u64(b).copy()
    out = np.empty(dom.n * 4, dtype=np.uint64)
    sh = _u64(shift)
    mc = _u64(mask_c) if mask_c is not None else None
    mab = _u64(mask_ab) if mask_ab is not None else None
    _check(lib().csh_groth16_h(dom.h, _p(sh), int(protocol), _p(a), _p(b), _p(mc), _p(mab), _p(out)))
    return out


def groth16_witness_map_masks(dom: Domain, shift, protocol: int, party: int, a: "Matrix", b: "Matrix", num_constraints: int, public, witness,
                              mask_c=None, mask_ab=None, fresh_output: bool = True):
    """csh_groth16_witness_map_masks: CircomReduction::witness_map_from_matrices in one call, Rep3 masks handed over by the caller.
    fresh_output: h is written into memory that was allocated but never touched (np.empty), as the Rust shim does."""
    pub, wit, sh = _u64(public), _u64(witness), _u64(shift)
    comp = 2 if protocol == 1 else 1
    out = np.empty(dom.n * 4, dtype=np.uint64) if fresh_output else np.zeros(dom.n * 4, dtype=np.uint64)
    mc = _u64(mask_c) if mask_c is not None else None
    mab = _u64(mask_ab) if mask_ab is not None else None
    _check(lib().csh_groth16_witness_map_masks(dom.h, _p(sh), int(protocol), int(party), a.h, b.h, C.c_size_t(num_constraints), _p(pub),
                                               C.c_size_t(pub.size // 4), _p(wit), C.c_size_t(wit.size // (4 * comp)), _p(mc), _p(mab), _p(out)))
    return out


def groth16_witness_map_seeded(dom: Domain, shift, protocol: int, party: int, a: "Matrix", b: "Matrix", num_constraints: int, public, witness,
                               seed1=None, off1: int = 0, seed2=None, off2: int = 0):
    """csh_groth16_witness_map: the same map with the masks generated on the device from the party's two ChaCha12 keys."""
    pub, wit, sh = _u64(public), _u64(witness), _u64(shift)
    comp = 2 if protocol == 1 else 1
    out = np.empty(dom.n * 4, dtype=np.uint64)
    s1 = (C.c_uint8 * 32)(*seed1) if seed1 is not None else None
    s2 = (C.c_uint8 * 32)(*seed2) if seed2 is not None else None
    _check(lib().csh_groth16_witness_map(dom.h, _p(sh), int(protocol), int(party), a.h, b.h, C.c_size_t(num_constraints), _p(pub),
                                         C.c_size_t(pub.size // 4), _p(wit), C.c_size_t(wit.size // (4 * comp)), s1, C.c_uint64(off1), s2,
                                         C.c_uint64(off2), _p(out)))
    return out


def sync(stream=None):
    _check(lib().csh_sync(_stream(stream)))


class Event:
    def __init__(self):
        self.h = C.c_void_p()
        _check(lib().csh_event_create(C.byref(self.h)))

    def record(self, stream=None):
        _check(lib().csh_event_record(self.h, _stream(stream)))

    def elapsed_ms(self, stop: "Event") -> float:
        ms = C.c_float(0)
        _check(lib().csh_event_elapsed_ms(self.h, stop.h, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if self.h:
                lib().csh_event_destroy(self.h)
        except Exception:
            pass


class Matrix:
    """Device-resident CSR constraint-matrix side (csh_matrix_upload)."""

    def __init__(self, curve: int, rows):
        """rows: list of rows, each a list of (coeff_limbs (4 u64, Montgomery), column index)."""
        self.curve = curve
        row_ptr = np.zeros(len(rows) + 1, dtype=np.uint64)
        cols, vals = [], []
        for i, r in enumerate(rows):
            for c, idx in r:
                cols.append(idx)
                vals.append(np.asarray(c, dtype=np.uint64))
            row_ptr[i + 1] = len(cols)
        col = np.asarray(cols, dtype=np.uint32)
        val = np.concatenate(vals).astype(np.uint64) if vals else np.zeros(0, dtype=np.uint64)
        self.n_rows = len(rows)
        self.h = C.c_void_p()
        _check(lib().csh_matrix_upload(curve, _p(row_ptr), _p(col), _p(val), C.c_size_t(len(rows)), C.c_size_t(len(cols)), C.byref(self.h)))

    def evaluate(self, protocol: int, party: int, public, witness, n_out: int):
        pub, wit = _u64(public), _u64(witness)
        comp = 2 if protocol == 1 else 1
        n_wit = wit.size // (4 * comp)
        dp, dw = DeviceBuffer.from_host(pub), DeviceBuffer.from_host(wit if wit.size else np.zeros(4, dtype=np.uint64))
        out = DeviceBuffer(n_out * comp * 32)
        _check(lib().csh_evaluate_constraints_dev(self.h, protocol, party, dp.ptr, C.c_size_t(pub.size // 4), dw.ptr, C.c_size_t(n_wit), out.ptr, C.c_size_t(n_out), None))
        sync()
        return out.to_host()

    def free(self):
        if self.h:
            lib().csh_matrix_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
