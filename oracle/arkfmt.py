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



# ---- bincode witness-share files (co-circom split-witness / generate-proof) ----------------------------------------
# Restated from the published bincode 1.3 defaults (u64 little-endian lengths, u32 enum variant index) and the
# reference's serde glue: `ark_se` (mpc-core/src/serde_compat.rs:7-15) emits ONE byte string per annotated field, holding
# the ark-serialize encoding of the value. Types: CompressedRep3SharedWitness (co-circom-types/src/lib.rs:163-173),
# Rep3ShareVecType (mpc-core/src/protocols/rep3.rs:135-150; Replicated = 0, SeededReplicated = 1, Additive = 2,
# SeededAdditive = 3), SharedWitness (co-circom-types/src/lib.rs:204-218).
REP3_REPLICATED, REP3_ADDITIVE = 0, 2


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


def ser_rep3_share_file(public_inputs, kind: int, shares, nbytes: int = 32) -> bytes:
    """shares: [(a, b)] for REP3_REPLICATED (Rep3PrimeFieldShare serializes a then b), [a] for REP3_ADDITIVE."""
    out = _bincode_bytes(ser_vec(public_inputs, nbytes)) + struct.pack("<I", kind)
    if kind == REP3_REPLICATED:
        body = struct.pack("<Q", len(shares)) + b"".join(ser_field(a, nbytes) + ser_field(b, nbytes) for a, b in shares)
    elif kind == REP3_ADDITIVE:
        body = ser_vec(shares, nbytes)
    else:
        raise ValueError("seeded variants are not restated")
    return out + _bincode_bytes(body)


def parse_rep3_share_file(data: bytes, nbytes: int = 32):
    """-> (public_inputs, kind, shares)"""
    pub, off = _read_bincode_bytes(data, 0)
    (kind,) = struct.unpack_from("<I", data, off)
    body, off = _read_bincode_bytes(data, off + 4)
    if off != len(data):
        raise ValueError("trailing bytes")
    if kind not in (REP3_REPLICATED, REP3_ADDITIVE):
        raise ValueError("unsupported Rep3ShareVecType variant %d" % kind)
    return _parse_vec(pub, nbytes, 1), kind, _parse_vec(body, nbytes, 2 if kind == REP3_REPLICATED else 1)


def ser_shamir_share_file(public_inputs, shares, nbytes: int = 32) -> bytes:
    return _bincode_bytes(ser_vec(public_inputs, nbytes)) + _bincode_bytes(ser_vec(shares, nbytes))


def parse_shamir_share_file(data: bytes, nbytes: int = 32):
    pub, off = _read_bincode_bytes(data, 0)
    body, off = _read_bincode_bytes(data, off)
    if off != len(data):
        raise ValueError("trailing bytes")
    return _parse_vec(pub, nbytes, 1), _parse_vec(body, nbytes, 1)
