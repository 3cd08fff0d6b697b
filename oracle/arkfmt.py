"""ark-serialize (uncompressed) readers for the data files next to the LibSnarkReduction path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py). Formats restated from ark-serialize 0.5 / 0.6 semantics as the
reference uses them at co-circom/co-groth16/src/lib.rs:257-262: ``Matrix<F> = Vec<Vec<(F, usize)>>`` is a u64 length
followed by the rows, each a u64 length followed by (field element: little-endian canonical bytes, usize as u64 LE).
The Penumbra witness files are snarkjs ``wtns`` containers whose section headers were written as zeros, so they are read
positionally (magic, version, section count, 12-byte header, n8, prime, count, 12-byte header, values).
"""
import struct


def parse_matrix(data: bytes, nbytes: int = 32):
    off = 0
    (n_rows,) = struct.unpack_from("<Q", data, off)
    off += 8
    rows = []
    for _ in range(n_rows):
        (m,) = struct.unpack_from("<Q", data, off)
        off += 8
        row = []
        for _ in range(m):
            v = int.from_bytes(data[off:off + nbytes], "little")
            off += nbytes
            (idx,) = struct.unpack_from("<Q", data, off)
            off += 8
            row.append((v, idx))
        rows.append(row)
    if off != len(data):
        raise ValueError("trailing bytes after Matrix")
    return rows


def parse_wtns_positional(data: bytes):
    """-> (prime, values)."""
    if data[:4] != b"wtns":
        raise ValueError("bad magic")
    off = 12 + 12
    (n8,) = struct.unpack_from("<I", data, off)
    off += 4
    prime = int.from_bytes(data[off:off + n8], "little")
    off += n8
    (count,) = struct.unpack_from("<I", data, off)
    off += 4 + 12
    if off + count * n8 != len(data):
        raise ValueError("unexpected wtns length")
    return prime, [int.from_bytes(data[off + i * n8:off + (i + 1) * n8], "little") for i in range(count)]


def ser_matrix(rows, nbytes: int = 32) -> bytes:
    """inverse of parse_matrix: rows of (canonical value, column index)"""
    out = bytearray(struct.pack("<Q", len(rows)))
    for row in rows:
        out += struct.pack("<Q", len(row))
        for v, idx in row:
            out += int(v).to_bytes(nbytes, "little") + struct.pack("<Q", idx)
    return bytes(out)


def ser_wtns_positional(prime: int, values, n8: int = 32) -> bytes:
    """inverse of parse_wtns_positional, section headers written as zeros like the Penumbra fixtures"""
    return (b"wtns" + bytes(8) + bytes(12) + struct.pack("<I", n8) + prime.to_bytes(n8, "little") + struct.pack("<I", len(values)) + bytes(12) +
            b"".join(int(v).to_bytes(n8, "little") for v in values))


def vk_num_instance_variables(data: bytes, g1_bytes: int, g2_bytes: int) -> int:
    """ark_groth16::VerifyingKey uncompressed: alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1: Vec<G1Affine>;
    len(gamma_abc_g1) = num_instance_variables (the reference derives the same number from the proving key)."""
    off = g1_bytes + 3 * g2_bytes
    (k,) = struct.unpack_from("<Q", data, off)
    if off + 8 + k * g1_bytes != len(data):
        raise ValueError("unexpected vk length")
    return k


# ---- points and vectors ------------------------------------------------------------------------------------------
def ser_field(v: int, nbytes: int) -> bytes:
    return int(v).to_bytes(nbytes, "little")


def ser_vec(vals, nbytes: int = 32) -> bytes:
    """Vec<F>: u64 length, canonical little-endian elements."""
    return struct.pack("<Q", len(vals)) + b"".join(ser_field(v, nbytes) for v in vals)


def ser_g1(pt, p: int, nbytes: int) -> bytes:
    """short-Weierstrass affine, Compress::No: x, then y with SWFlags in the top two bits of the last byte
    (0x80: y > p - y, 0x40: infinity with zero coordinates)."""
    if pt is None:
        b = bytearray(2 * nbytes)
        b[-1] |= 0x40
        return bytes(b)
    x, y = pt
    b = bytearray(ser_field(x, nbytes) + ser_field(y, nbytes))
    if y > (p - y) % p:
        b[-1] |= 0x80
    return bytes(b)


def parse_g1(data: bytes, off: int, p: int, nbytes: int):
    """-> (point or None, new offset); checks canonical coordinates and the y-sign flag."""
    xb = data[off:off + nbytes]
    yb = bytearray(data[off + nbytes:off + 2 * nbytes])
    flags = yb[-1] >> 6
    yb[-1] &= 0x3F
    x, y = int.from_bytes(xb, "little"), int.from_bytes(bytes(yb), "little")
    if flags & 1:
        if x or y:
            raise ValueError("infinity flag with non-zero coordinates")
        return None, off + 2 * nbytes
    if x >= p or y >= p:
        raise ValueError("coordinate not canonical")
    if bool(flags & 2) != (y > (p - y) % p):
        raise ValueError("y-sign flag does not match y")
    return (x, y), off + 2 * nbytes


def ser_g2(pt, p: int, nbytes: int) -> bytes:
    """the G2 counterpart of ser_g1: x.c0, x.c1, y.c0, y.c1; `y > -y` compares c1 first, then c0."""
    if pt is None:
        b = bytearray(4 * nbytes)
        b[-1] |= 0x40
        return bytes(b)
    (x0, x1), (y0, y1) = pt
    b = bytearray(ser_field(x0, nbytes) + ser_field(x1, nbytes) + ser_field(y0, nbytes) + ser_field(y1, nbytes))
    if (y1, y0) > ((-y1) % p, (-y0) % p):
        b[-1] |= 0x80
    return bytes(b)


def ser_groth16_proving_key(pk: dict, p: int, nbytes: int) -> bytes:
    """ark-groth16 0.6 `ProvingKey<E>` under CanonicalSerialize, Compress::No (the `circuit.pk` the reference's LibSnark tests read with
    deserialize_uncompressed_unchecked, co-circom/co-groth16/src/lib.rs:257; the files themselves are absent upstream): fields in
    declaration order -- vk {alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1: Vec}, beta_g1, delta_g1, a_query, b_g1_query,
    b_g2_query, h_query, l_query; a Vec is its u64 length followed by the items. The leading vk is byte-for-byte the layout of the
    reference's committed circuit.vk (tests/test_oracle.py parses those)."""
    g1 = lambda P: ser_g1(P, p, nbytes)
    g2 = lambda P: ser_g2(P, p, nbytes)
    vec = lambda f, v: struct.pack("<Q", len(v)) + b"".join(f(P) for P in v)
    return (g1(pk["alpha_g1"]) + g2(pk["beta_g2"]) + g2(pk["gamma_g2"]) + g2(pk["delta_g2"]) + vec(g1, pk["gamma_abc_g1"]) +
            g1(pk["beta_g1"]) + g1(pk["delta_g1"]) + vec(g1, pk["a_query"]) + vec(g1, pk["b_g1_query"]) + vec(g2, pk["b_g2_query"]) +
            vec(g1, pk["h_query"]) + vec(g1, pk["l_query"]))


def parse_groth16_proof(data: bytes, p: int, nbytes: int):
    """ark-groth16 `Proof<E>` {a: G1, b: G2, c: G1}, Compress::No."""
    a, off = parse_g1(data, 0, p, nbytes)
    b, off = parse_g2(data, off, p, nbytes)
    c, off = parse_g1(data, off, p, nbytes)
    if off != len(data):
        raise ValueError("trailing bytes after Proof")
    return {"a": a, "b": b, "c": c}


def parse_g2(data: bytes, off: int, p: int, nbytes: int):
    """The same encoding over Fp2: x.c0, x.c1, y.c0, y.c1, SWFlags in the top two bits of the last byte (of y.c1); `y > -y` compares
    c1 first, then c0 (ark-ff's Ord for quadratic extensions). -> ((x0, x1), (y0, y1)) or None, new offset."""
    c = [bytearray(data[off + i * nbytes:off + (i + 1) * nbytes]) for i in range(4)]
    flags = c[3][-1] >> 6
    c[3][-1] &= 0x3F
    x0, x1, y0, y1 = (int.from_bytes(bytes(b), "little") for b in c)
    if flags & 1:
        if x0 or x1 or y0 or y1:
            raise ValueError("infinity flag with non-zero coordinates")
        return None, off + 4 * nbytes
    if max(x0, x1, y0, y1) >= p:
        raise ValueError("coordinate not canonical")
    n0, n1 = (-y0) % p, (-y1) % p
    if bool(flags & 2) != ((y1, y0) > (n1, n0)):
        raise ValueError("y-sign flag does not match y")
    return ((x0, x1), (y0, y1)), off + 4 * nbytes


# ---- bincode witness-share files (co-circom split-witness / generate-proof) ----------------------------------------
# Restated from the published bincode 1.3 defaults (u64 little-endian lengths, u32 enum variant index) and the
# reference's serde glue: `ark_se` (mpc-core/src/serde_compat.rs:7-15) emits ONE byte string per annotated field, holding
# the ark-serialize encoding of the value. Types: CompressedRep3SharedWitness (co-circom-types/src/lib.rs:163-173),
# Rep3ShareVecType (mpc-core/src/protocols/rep3.rs:135-150; Replicated = 0, SeededReplicated = 1, Additive = 2,
# SeededAdditive = 3), SharedWitness (co-circom-types/src/lib.rs:204-218).
REP3_REPLICATED, REP3_SEEDED_REPLICATED, REP3_ADDITIVE, REP3_SEEDED_ADDITIVE = 0, 1, 2, 3


def _bincode_bytes(b: bytes) -> bytes:
    return struct.pack("<Q", len(b)) + b


def _read_bincode_bytes(data: bytes, off: int):
    (n,) = struct.unpack_from("<Q", data, off)
    off += 8
    if off + n > len(data):
        raise ValueError("byte string exceeds the input")
    return data[off : off + n], off + n


def _parse_vec(blob: bytes, nbytes: int, items: int):
    (cnt,) = struct.unpack_from("<Q", blob, 0)
    if 8 + cnt * items * nbytes != len(blob):
        raise ValueError("Vec length does not match its byte string")
    flat = [int.from_bytes(blob[8 + i * nbytes : 8 + (i + 1) * nbytes], "little") for i in range(cnt * items)]
    return flat if items == 1 else [tuple(flat[i * items : (i + 1) * items]) for i in range(cnt)]


def _ser_seeded(x, nbytes: int) -> bytes:
    """SeededType (mpc-core/src/protocols/rep3.rs:152-165): ("shares", [a]) -> variant 0 + bytes(Vec<F>);
    ("seed", seed32, len) -> variant 1 + 32 raw bytes ([u8; 32] is a serde tuple) + u64 (usize); PhantomData is empty."""
    if x[0] == "shares":
        return struct.pack("<I", 0) + _bincode_bytes(ser_vec(x[1], nbytes))
    return struct.pack("<I", 1) + bytes(x[1]) + struct.pack("<Q", x[2])


def _parse_seeded(data: bytes, off: int, nbytes: int):
    (v,) = struct.unpack_from("<I", data, off)
    off += 4
    if v == 0:
        blob, off = _read_bincode_bytes(data, off)
        return ("shares", _parse_vec(blob, nbytes, 1)), off
    if v != 1:
        raise ValueError("unknown SeededType variant %d" % v)
    seed = data[off : off + 32]
    (n,) = struct.unpack_from("<Q", data, off + 32)
    return ("seed", seed, n), off + 40


def expand_seeded(x, p: int, nbits: int, mont_r: int):
    """SeededType::expand_vec (rep3.rs:185-196): F::rand over ChaCha12Rng::from_seed(seed). ark-ff 0.6.0 (not vendored;
    restated from the published source, fields/models/fp/mod.rs `Distribution<Fp> for Standard`): four next_u64 words =
    32 keystream bytes as little-endian limbs, top limb masked to the modulus bit size, redrawn while >= p; the limbs are
    the Montgomery representation, so the canonical value is limbs * R^-1. PARITY UNPINNED (no reference vector)."""
    if x[0] == "shares":
        return list(x[1])
    from . import chacha
    seed, n = x[1], x[2]
    out, pos, rinv = [], 0, pow(mont_r, -1, p)
    stream = chacha.keystream(seed, 32 * (n + 8) + 64)
    mask = (1 << nbits) - 1
    while len(out) < n:
        if pos + 32 > len(stream):
            stream += chacha.keystream(seed, 32 * 64, start_byte=len(stream))
        v = int.from_bytes(stream[pos : pos + 32], "little") & mask
        pos += 32
        if v < p:
            out.append(v * rinv % p)
    return out


def ser_rep3_share_file(public_inputs, kind: int, shares, nbytes: int = 32) -> bytes:
    """shares: [(a, b)] for REP3_REPLICATED (Rep3PrimeFieldShare serializes a then b), [a] for REP3_ADDITIVE, a SeededType
    tuple for REP3_SEEDED_ADDITIVE, a pair of them (ReplicatedSeedType {a, b}, rep3.rs:225-238) for REP3_SEEDED_REPLICATED."""
    out = _bincode_bytes(ser_vec(public_inputs, nbytes)) + struct.pack("<I", kind)
    if kind == REP3_REPLICATED:
        body = struct.pack("<Q", len(shares)) + b"".join(ser_field(a, nbytes) + ser_field(b, nbytes) for a, b in shares)
        return out + _bincode_bytes(body)
    if kind == REP3_ADDITIVE:
        return out + _bincode_bytes(ser_vec(shares, nbytes))
    if kind == REP3_SEEDED_ADDITIVE:
        return out + _ser_seeded(shares, nbytes)
    if kind == REP3_SEEDED_REPLICATED:
        return out + _ser_seeded(shares[0], nbytes) + _ser_seeded(shares[1], nbytes)
    raise ValueError("unknown Rep3ShareVecType variant")


def parse_rep3_share_file(data: bytes, nbytes: int = 32):
    """-> (public_inputs, kind, shares) with shares as ser_rep3_share_file takes them"""
    pub, off = _read_bincode_bytes(data, 0)
    (kind,) = struct.unpack_from("<I", data, off)
    off += 4
    if kind in (REP3_REPLICATED, REP3_ADDITIVE):
        body, off = _read_bincode_bytes(data, off)
        shares = _parse_vec(body, nbytes, 2 if kind == REP3_REPLICATED else 1)
    elif kind == REP3_SEEDED_ADDITIVE:
        shares, off = _parse_seeded(data, off, nbytes)
    elif kind == REP3_SEEDED_REPLICATED:
        a, off = _parse_seeded(data, off, nbytes)
        b, off = _parse_seeded(data, off, nbytes)
        shares = (a, b)
    else:
        raise ValueError("unsupported Rep3ShareVecType variant %d" % kind)
    if off != len(data):
        raise ValueError("trailing bytes")
    return _parse_vec(pub, nbytes, 1), kind, shares


def ser_shamir_share_file(public_inputs, shares, nbytes: int = 32) -> bytes:
    return _bincode_bytes(ser_vec(public_inputs, nbytes)) + _bincode_bytes(ser_vec(shares, nbytes))


def parse_shamir_share_file(data: bytes, nbytes: int = 32):
    pub, off = _read_bincode_bytes(data, 0)
    body, off = _read_bincode_bytes(data, off)
    if off != len(data):
        raise ValueError("trailing bytes")
    return _parse_vec(pub, nbytes, 1), _parse_vec(body, nbytes, 1)
