"""snarkjs/circom file formats on either side of the hot path (oracle; test infrastructure only).

The reference parses these with the external crate ``taceo-circom-types 0.3.1``
(``Cargo.lock:4799``; call site ``co-circom/co-circom/src/bin/co-circom.rs:1005-1006``); the layout
below is the published iden3 binfile format, checked by parsing the reference's own
``test_vectors/Groth16`` fixtures.
"""
from __future__ import annotations

import json
import struct
from dataclasses import dataclass, field

from . import curves as cv
from . import fields as fl


def _sections(data: bytes, magic: bytes):
    assert data[:4] == magic, data[:4]
    _version, nsec = struct.unpack_from("<II", data, 4)
    off = 12
    secs = {}
    for _ in range(nsec):
        typ, ln = struct.unpack_from("<IQ", data, off)
        off += 12
        secs.setdefault(typ, []).append((off, ln))
        off += ln
    return secs


@dataclass
class ZKey:
    curve: str
    n8q: int
    n8r: int
    q: int
    r: int
    n_vars: int
    n_public: int
    domain_size: int
    alpha_g1: tuple = None
    beta_g1: tuple = None
    beta_g2: tuple = None
    gamma_g2: tuple = None
    delta_g1: tuple = None
    delta_g2: tuple = None
    ic: list = field(default_factory=list)
    coeffs: list = field(default_factory=list)  # (matrix, constraint, signal, value)
    a_query: list = field(default_factory=list)
    b_g1_query: list = field(default_factory=list)
    b_g2_query: list = field(default_factory=list)
    l_query: list = field(default_factory=list)
    h_query: list = field(default_factory=list)

    @property
    def Fr(self):
        return fl.BN254_FR if self.curve == "bn254" else fl.BLS381_FR

    @property
    def G1(self):
        return cv.CURVES[self.curve][0]

    @property
    def G2(self):
        return cv.CURVES[self.curve][1]

    # ConstraintMatrices as the reference consumes them (groth16/reduction.rs:81-82, 99-130):
    # rows of (coeff, signal) for A and B over the first num_constraints constraints.
    @property
    def num_inputs(self):
        return self.n_public + 1

    @property
    def num_constraints(self):
        mx = max((c for (_m, c, _s, _v) in self.coeffs), default=0)
        return mx - self.n_public

    def matrices(self):
        nc = self.num_constraints
        a = [[] for _ in range(nc)]
        b = [[] for _ in range(nc)]
        for (m, c, s, v) in self.coeffs:
            if c < nc:
                (a if m == 0 else b)[c].append((v, s))
        return a, b


def parse_zkey(data: bytes) -> ZKey:
    secs = _sections(data, b"zkey")
    off, _ = secs[1][0]
    assert struct.unpack_from("<I", data, off)[0] == 1, "not a groth16 zkey"
    off, _ = secs[2][0]
    n8q = struct.unpack_from("<I", data, off)[0]
    off += 4
    q = int.from_bytes(data[off:off + n8q], "little")
    off += n8q
    n8r = struct.unpack_from("<I", data, off)[0]
    off += 4
    r = int.from_bytes(data[off:off + n8r], "little")
    off += n8r
    n_vars, n_public, domain_size = struct.unpack_from("<III", data, off)
    off += 12
    if q == fl.BN254_Q:
        curve = "bn254"
    elif q == fl.BLS381_Q:
        curve = "bls12_381"
    else:
        raise ValueError("unsupported curve modulus")
    zk = ZKey(curve, n8q, n8r, q, r, n_vars, n_public, domain_size)
    Fq = zk.G1.F
    Fr = zk.Fr
    assert r == Fr.p

    def g1(o):
        x = Fq.from_mont(int.from_bytes(data[o:o + n8q], "little"))
        y = Fq.from_mont(int.from_bytes(data[o + n8q:o + 2 * n8q], "little"))
        return (None if x == 0 and y == 0 else (x, y)), o + 2 * n8q

    def g2(o):
        v = [Fq.from_mont(int.from_bytes(data[o + i * n8q:o + (i + 1) * n8q], "little")) for i in range(4)]
        P = None if all(t == 0 for t in v) else ((v[0], v[1]), (v[2], v[3]))
        return P, o + 4 * n8q

    zk.alpha_g1, off = g1(off)
    zk.beta_g1, off = g1(off)
    zk.beta_g2, off = g2(off)
    zk.gamma_g2, off = g2(off)
    zk.delta_g1, off = g1(off)
    zk.delta_g2, off = g2(off)

    def g1_list(sec):
        o, ln = secs[sec][0]
        out = []
        for _ in range(ln // (2 * n8q)):
            P, o = g1(o)
            out.append(P)
        return out

    def g2_list(sec):
        o, ln = secs[sec][0]
        out = []
        for _ in range(ln // (4 * n8q)):
            P, o = g2(o)
            out.append(P)
        return out

    zk.ic = g1_list(3)
    o, _ = secs[4][0]
    ncoef = struct.unpack_from("<I", data, o)[0]
    o += 4
    R2inv = pow(Fr.R2, -1, Fr.p)
    for _ in range(ncoef):
        m, c, s = struct.unpack_from("<III", data, o)
        o += 12
        v = int.from_bytes(data[o:o + n8r], "little") * R2inv % Fr.p  # doubly Montgomery-encoded
        o += n8r
        zk.coeffs.append((m, c, s, v))
    zk.a_query = g1_list(5)
    zk.b_g1_query = g1_list(6)
    zk.b_g2_query = g2_list(7)
    zk.l_query = g1_list(8)
    zk.h_query = g1_list(9)
    return zk


def parse_wtns(data: bytes):
    secs = _sections(data, b"wtns")
    off, _ = secs[1][0]
    n8 = struct.unpack_from("<I", data, off)[0]
    off += 4
    _prime = int.from_bytes(data[off:off + n8], "little")
    off += n8
    n = struct.unpack_from("<I", data, off)[0]
    off, _ = secs[2][0]
    return [int.from_bytes(data[off + i * n8:off + (i + 1) * n8], "little") for i in range(n)]


def _g1_json(v):
    x, y, z = (int(t) for t in v)
    return None if z == 0 else (x, y)


def _g2_json(v):
    (x0, x1), (y0, y1), (z0, z1) = ((int(a), int(b)) for a, b in v)
    return None if (z0, z1) == (0, 0) else ((x0, x1), (y0, y1))


def parse_vk(text: str):
    j = json.loads(text)
    return {
        "curve": j.get("curve"),
        "n_public": j["nPublic"],
        "alpha_g1": _g1_json(j["vk_alpha_1"]),
        "beta_g2": _g2_json(j["vk_beta_2"]),
        "gamma_g2": _g2_json(j["vk_gamma_2"]),
        "delta_g2": _g2_json(j["vk_delta_2"]),
        "ic": [_g1_json(p) for p in j["IC"]],
    }


def parse_proof(text: str):
    j = json.loads(text)
    return {"a": _g1_json(j["pi_a"]), "b": _g2_json(j["pi_b"]), "c": _g1_json(j["pi_c"])}


def parse_public(text: str):
    return [int(x) for x in json.loads(text)]


def proof_to_json(proof, curve: str) -> str:
    """The ``circom.proof`` schema (snarkjs groth16 verify would accept it unchanged)."""
    a, b, c = proof["a"], proof["b"], proof["c"]
    return json.dumps({
        "pi_a": [str(a[0]), str(a[1]), "1"],
        "pi_b": [[str(b[0][0]), str(b[0][1])], [str(b[1][0]), str(b[1][1])], ["1", "0"]],
        "pi_c": [str(c[0]), str(c[1]), "1"],
        "protocol": "groth16",
        "curve": "bn128" if curve == "bn254" else "bls12381",
    })


# ---------------------------------------------------------------------------------------------
# snarkjs PLONK zkey (protocol id 2).  The reference reads it with the same external crate and uses
# `zkey.{qm,ql,qr,qo,qc}_poly.{coeffs, evaluations}`, `zkey.s{1,2,3}_poly`, `zkey.p_tau`
# (co-circom/co-plonk/src/round3.rs:325-332, round5.rs:154, round1.rs commitments over p_tau) and the
# verifying key's Qm..S3 commitments (co-plonk/src/lib.rs:295-312 load both files of
# test_vectors/Plonk/*/multiplier2).  Each polynomial section holds n coefficients followed by their
# 4n evaluations over the extended domain whose generator is the snarkjs root `roots[pow + 2]`
# (co-plonk/src/types.rs:70-109): a raw `EvaluationDomain::fft` vector held by the reference; the
# header's Qm..S3 are the KZG commitments msm(p_tau[0..n], coeffs): a raw `msm_unchecked` vector.
# ---------------------------------------------------------------------------------------------
PLONK_POLY_SECTIONS = {"Qm": (7, 0), "Ql": (8, 0), "Qr": (9, 0), "Qo": (10, 0), "Qc": (11, 0),
                       "S1": (12, 0), "S2": (12, 1), "S3": (12, 2)}


@dataclass
class PlonkZKey:
    curve: str
    n8q: int
    n8r: int
    n_vars: int
    n_public: int
    domain_size: int
    n_additions: int
    n_constraints: int
    k1: int
    k2: int
    commitments: dict  # name -> affine G1 point (header copy of vk.Qm .. vk.S3)
    x_2: tuple
    polys: dict        # name -> (coeffs[n], evaluations[4n]) canonical integers
    lagrange: list     # [(coeffs[n], evaluations[4n])] for L_1 .. L_max(n_public,1)
    p_tau: list        # affine G1 points tau^i G

    @property
    def Fr(self):
        return fl.BN254_FR if self.curve == "bn254" else fl.BLS381_FR

    @property
    def G1(self):
        return cv.CURVES[self.curve][0]


def parse_plonk_zkey(data: bytes) -> PlonkZKey:
    secs = _sections(data, b"zkey")
    off, _ = secs[1][0]
    assert struct.unpack_from("<I", data, off)[0] == 2, "not a plonk zkey"
    off, ln = secs[2][0]
    end = off + ln
    n8q = struct.unpack_from("<I", data, off)[0]
    off += 4
    q = int.from_bytes(data[off:off + n8q], "little")
    off += n8q
    n8r = struct.unpack_from("<I", data, off)[0]
    off += 4
    r = int.from_bytes(data[off:off + n8r], "little")
    off += n8r
    n_vars, n_public, n, n_add, n_cons = struct.unpack_from("<IIIII", data, off)
    off += 20
    curve = "bn254" if q == fl.BN254_Q else "bls12_381" if q == fl.BLS381_Q else None
    if curve is None:
        raise ValueError("unsupported curve modulus")
    G1, G2 = cv.CURVES[curve]
    Fq = G1.F
    Fr = fl.BN254_FR if curve == "bn254" else fl.BLS381_FR
    assert r == Fr.p

    def fr(o):
        return Fr.from_mont(int.from_bytes(data[o:o + n8r], "little"))

    def g1(o):
        x = Fq.from_mont(int.from_bytes(data[o:o + n8q], "little"))
        y = Fq.from_mont(int.from_bytes(data[o + n8q:o + 2 * n8q], "little"))
        return None if x == 0 and y == 0 else (x, y)

    k1, k2 = fr(off), fr(off + n8r)
    off += 2 * n8r
    commitments = {}
    for nm in PLONK_POLY_SECTIONS:
        commitments[nm] = g1(off)
        off += 2 * n8q
    v = [Fq.from_mont(int.from_bytes(data[off + i * n8q:off + (i + 1) * n8q], "little")) for i in range(4)]
    x_2 = ((v[0], v[1]), (v[2], v[3]))
    off += 4 * n8q
    assert off == end, (off, end)

    def poly(sec, idx):
        o, l = secs[sec][0]
        assert l % (5 * n * n8r) == 0 and idx < l // (5 * n * n8r)
        o += idx * 5 * n * n8r
        co = [fr(o + i * n8r) for i in range(n)]
        o += n * n8r
        ev = [fr(o + i * n8r) for i in range(4 * n)]
        return co, ev

    polys = {nm: poly(*loc) for nm, loc in PLONK_POLY_SECTIONS.items()}
    n_lag = secs[13][0][1] // (5 * n * n8r)
    lagrange = [poly(13, i) for i in range(n_lag)]
    o, l = secs[14][0]
    p_tau = [g1(o + i * 2 * n8q) for i in range(l // (2 * n8q))]
    return PlonkZKey(curve, n8q, n8r, n_vars, n_public, n, n_add, n_cons, k1, k2, commitments, x_2, polys, lagrange, p_tau)


def parse_plonk_vk(text: str):
    j = json.loads(text)
    out = {nm: _g1_json(j[nm]) for nm in PLONK_POLY_SECTIONS}
    out.update({"curve": j.get("curve"), "n_public": j["nPublic"], "power": j["power"], "k1": int(j["k1"]), "k2": int(j["k2"]),
                "X_2": _g2_json(j["X_2"]), "w": int(j["w"])})
    return out
