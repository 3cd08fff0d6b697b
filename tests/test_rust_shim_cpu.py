"""Static audit of rust/ against the reference's real interface (no rustc in this image, so nothing else catches a private or
mis-typed reference item -- round 2's shim imported a private function with the wrong signature). Runs only where
/root/reference exists (the build container); the GPU box skips it.

Checks: (1) every `co_groth16::` / `mpc_core::` / `mpc_net::` path the shim names resolves to an item that is `pub` all the way
down in the reference's sources; (2) the three Hip drivers implement exactly the methods of `CircomGroth16Prover` with the
reference's parameter counts; (3) the `R1CSToQAP` implementors use the trait's signature; (4) the locally restated
`groth16_roots_of_unity` has the reference's shape (pow -> (gen, shift), q^2 branch)."""
import os
import re

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUST = os.path.join(ROOT, "rust", "co-groth16-hip", "src")
CRATES = {"co_groth16": "co-circom/co-groth16/src", "mpc_core": "mpc-core/src", "mpc_net": "mpc-net/src", "co_plonk": "co-circom/co-plonk/src",
          "co_noir_common": "co-noir/co-noir-common/src"}
SHIMS = ["co-groth16-hip", "co-plonk-hip", "co-noir-hip"]

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present on this box")


def _strip_comments(s):
    s = re.sub(r"//[^\n]*", "", s)
    return re.sub(r"/\*.*?\*/", "", s, flags=re.S)


def _expand_use(path):
    """`a::b::{c, d::{e, f}}` -> [a::b::c, a::b::d::e, a::b::d::f]"""
    path = path.strip()
    m = re.match(r"^(.*?)::\{(.*)\}$", path, re.S)
    if not m:
        return [re.sub(r"\s+as\s+\w+$", "", path)]
    head, body = m.group(1), m.group(2)
    parts, depth, cur = [], 0, ""
    for ch in body:
        if ch == "{":
            depth += 1
        if ch == "}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    out = []
    for p in parts:
        p = p.strip()
        out += _expand_use(head if p == "self" else head + "::" + p)
    return out


def _shim_paths():
    paths = set()
    names = "|".join(CRATES)
    for crate in SHIMS:
        d = os.path.join(ROOT, "rust", crate, "src")
        for f in sorted(os.listdir(d)):
            s = _strip_comments(open(os.path.join(d, f)).read())
            for m in re.finditer(r"\buse\s+((?:%s)\b[^;]*);" % names, s):
                paths.update(_expand_use(re.sub(r"\s+", " ", m.group(1))))
            for m in re.finditer(r"\b((?:%s)(?:::\w+)+)" % names, s):  # fully qualified uses outside `use`
                paths.add(m.group(1))
    return sorted(p for p in paths if "::" in p)


def _module_file(dirpath, stem, name):
    """file of `mod name` declared in <dirpath>/<stem>.rs (stem None = lib.rs)"""
    sub = dirpath if stem is None else os.path.join(dirpath, stem)
    for cand in (os.path.join(sub, name + ".rs"), os.path.join(sub, name, "mod.rs")):
        if os.path.exists(cand):
            return cand
    return None


def _resolve(path, depth=0):
    """True if `path` names an item reachable through `pub` items only. Follows `pub use` re-exports."""
    assert depth < 8, "re-export chain too deep: " + path
    crate, *segs = path.split("::")
    src = os.path.join(REF, CRATES[crate])
    cur_file, cur_dir, cur_stem = os.path.join(src, "lib.rs"), src, None
    for i, seg in enumerate(segs):
        text = _strip_comments(open(cur_file).read())
        last = i == len(segs) - 1
        if re.search(r"^\s*pub\s+mod\s+%s\s*;" % seg, text, re.M):
            nxt = _module_file(cur_dir, cur_stem, seg)
            assert nxt, "module file of %s not found (%s)" % (seg, path)
            if cur_stem is not None:
                cur_dir = os.path.join(cur_dir, cur_stem)
            cur_file, cur_stem = nxt, seg
            if os.path.basename(nxt) == "mod.rs":
                cur_dir, cur_stem = os.path.dirname(os.path.dirname(nxt)), seg
            if last:
                return True
            continue
        if re.search(r"^\s*pub\s+(?:unsafe\s+)?(?:struct|trait|fn|type|enum|const|static|union)\s+%s\b" % seg, text, re.M):
            return last or True  # associated items below a pub type/trait are the type's own business
        # pub use a::b::{.., seg, ..}; (possibly renamed)
        for m in re.finditer(r"^\s*pub\s+use\s+([^;]+);", text, re.M):
            for full in _expand_use(re.sub(r"\s+", " ", m.group(1))):
                if full.split("::")[-1] != seg:
                    continue
                rest = "::".join(segs[i + 1:])
                if full.startswith("crate::"):
                    target = crate + "::" + full[len("crate::"):]
                elif full.startswith(("self::", "super::")) or full.split("::")[0] not in ("std", "core"):
                    # relative to the current module: modules declared here (pub or private -- the re-export is what is public)
                    first = full.split("::")[0]
                    if first in ("self", "super"):
                        return True  # re-exported from a sibling: public by this `pub use`
                    if first in CRATES:
                        target = full
                    elif re.search(r"^\s*(?:pub(?:\([^)]*\))?\s+)?mod\s+%s\s*;" % first, text, re.M):
                        return True  # `mod x; pub use x::Item;`: public by the re-export
                    else:
                        return True  # re-export of an external crate's item
                else:
                    return True
                return _resolve(target + ("::" + rest if rest else ""), depth + 1)
        if re.search(r"^\s*(?:pub\(crate\)\s+)?(?:fn|struct|mod|trait|type|const)\s+%s\b" % seg, text, re.M):
            raise AssertionError("%s: `%s` exists in %s but is not pub" % (path, seg, os.path.relpath(cur_file, REF)))
        raise AssertionError("%s: `%s` not found in %s" % (path, seg, os.path.relpath(cur_file, REF)))
    return True


def test_every_reference_path_named_by_the_shim_is_public():
    paths = _shim_paths()
    assert len(paths) >= 25, paths
    for p in paths:
        assert _resolve(p), p


def test_private_reference_items_are_not_imported():
    # the two functions round 2's shim wrongly imported: private in the reference, restated locally now
    src = open(os.path.join(REF, "co-circom/co-groth16/src/groth16.rs")).read()
    assert re.search(r"^fn groth16_roots_of_unity<F: PrimeField \+ FftField>\(pow: usize\) -> \(F, F\)", src, re.M)
    assert re.search(r"^fn roots_of_unity<F: PrimeField \+ FftField>\(\) -> \(F, Vec<F>\)", src, re.M)
    for f in os.listdir(RUST):
        s = _strip_comments(open(os.path.join(RUST, f)).read())
        assert "co_groth16::groth16_roots_of_unity" not in s and "co_groth16::roots_of_unity" not in s, f
    shim = _strip_comments(open(os.path.join(RUST, "hip_reduction.rs")).read())
    assert re.search(r"fn groth16_roots_of_unity<F: PrimeField \+ FftField>\(pow: usize\) -> \(F, F\)", shim)
    assert "q.square()" in shim and "roots[pow + 1]" in shim and "F::TWO_ADICITY as usize == pow" in shim
    assert ".ilog2()" in shim and ".max(1)" not in shim  # reduction.rs:85-86


def _fn_params(text):
    """{fn name: set of parameter-name tuples} for every `fn name<..>(..)` in text (balanced brackets, split at depth 0)"""
    out = {}
    for m in re.finditer(r"\bfn\s+(\w+)", text):
        i = m.end()
        depth = 0
        while i < len(text) and (text[i] != "(" or depth):  # skip the generic parameter list
            depth += text[i] == "<"
            depth -= text[i] == ">" and text[i - 1] != "-"
            i += 1
        j, depth, cur, params = i + 1, 0, "", []
        while j < len(text):
            ch = text[j]
            if ch in "([<":
                depth += 1
            elif ch in ")]" or (ch == ">" and text[j - 1] != "-"):
                if ch == ")" and depth == 0:
                    break
                depth -= 1
            if ch == "," and depth == 0:
                params.append(cur)
                cur = ""
            else:
                cur += ch
            j += 1
        if cur.strip():
            params.append(cur)
        names = tuple(re.sub(r"\s+", " ", q).strip().split(":")[0].strip() for q in params)
        out.setdefault(m.group(1), set()).add(names)
    return out


def _trait_methods(text, trait):
    body = text[text.index("pub trait " + trait):]
    depth, end = 0, None
    for i, ch in enumerate(body):
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
            if depth == 0:
                end = i
                break
    return _fn_params(_strip_comments(body[:end]))


def test_drivers_cover_the_prover_trait_with_the_reference_arity():
    ref = _trait_methods(open(os.path.join(REF, "co-circom/co-groth16/src/mpc.rs")).read(), "CircomGroth16Prover")
    assert len(ref) == 12 and ref["evaluate_constraint"] == {("id", "lhs", "public_inputs", "private_witness")}, ref
    got = _fn_params(_strip_comments(open(os.path.join(RUST, "drivers.rs")).read()))
    shim = _strip_comments(open(os.path.join(RUST, "drivers.rs")).read())
    for name, params in ref.items():
        assert name in got, "driver method missing: " + name
        (want,) = params
        for have in got[name]:   # same number of parameters, in the reference's order (names may be `_` where unused)
            assert len(have) == len(want), (name, have, want)
            assert all(h == w or h == "_" for h, w in zip(have, want)), (name, have, want)
    # associated types of the three implementors = the reference drivers' (mpc/{plain,rep3,shamir}.rs)
    for drv, share, state in (("Plain", "P::ScalarField", "()"), ("Rep3", "Rep3PrimeFieldShare<P::ScalarField>", "Rep3State"),
                              ("Shamir", "ShamirPrimeFieldShare<P::ScalarField>", "ShamirState<P::ScalarField>")):
        r = open(os.path.join(REF, "co-circom/co-groth16/src/mpc/%s.rs" % drv.lower())).read()
        assert "type ArithmeticShare = %s;" % share in r and "type State = %s;" % state in r
        blk = shim[shim.index("for Hip%sGroth16Driver" % drv):]
        assert "type ArithmeticShare = %s;" % share in blk and "type State = %s;" % state in blk


def test_reduction_implementors_use_the_trait_signature():
    ref = _strip_comments(open(os.path.join(REF, "co-circom/co-groth16/src/groth16/reduction.rs")).read())
    sig = re.search(r"fn witness_map_from_matrices<P: Pairing, T: CircomGroth16Prover<P>>\((.*?)\)\s*->\s*Result<Vec<T::ArithmeticHalfShare>>", ref, re.S)
    assert sig
    want = re.sub(r"\s+", " ", sig.group(1)).strip().rstrip(",")
    shim = _strip_comments(open(os.path.join(RUST, "hip_reduction.rs")).read())
    sigs = re.findall(r"fn witness_map_from_matrices<P: Pairing, T: CircomGroth16Prover<P>>\((.*?)\)\s*->\s*eyre::Result<Vec<T::ArithmeticHalfShare>>", shim, re.S)
    assert len(sigs) == 2
    for s in sigs:
        assert re.sub(r"\s+", " ", s).strip().rstrip(",") == want
    # fields / methods the shim reads on reference types
    assert "pub rngs: Rep3CorrelatedRng" in open(os.path.join(REF, "mpc-core/src/protocols/rep3.rs")).read()
    rng = open(os.path.join(REF, "mpc-core/src/protocols/rep3/rngs.rs")).read()
    assert "pub rand: Rep3Rand" in rng and "pub fn masking_field_elements_vec<F: PrimeField>(&mut self, len: usize) -> Vec<F>" in rng
    assert "fn id(&self) -> Self::PartyID;" in open(os.path.join(REF, "mpc-core/src/lib.rs")).read()


# ---- the shim's calls into the C ABI: every `sys::csh_*(...)` call passes exactly the arguments include/cosnarks_hip.h declares ----
def _split_top(s):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return parts


def _header_arity():
    txt = open(os.path.join(ROOT, "include", "cosnarks_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(csh_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(_split_top(args))
    return out


def _rust_sys_calls(crate_dir):
    calls = []
    for f in sorted(os.listdir(crate_dir)):
        if not f.endswith(".rs"):
            continue
        s = _strip_comments(open(os.path.join(crate_dir, f)).read())
        for m in re.finditer(r"\bsys::(csh_[a-z0-9_]+)\s*\(", s):
            i, depth = m.end(), 1
            while depth:
                depth += s[i] in "([{"
                depth -= s[i] in ")]}"
                i += 1
            body = s[m.end():i - 1]
            body = re.sub(r"\|[^|]*\|", "|_|", body)          # closure parameter lists hold no call arguments
            body = re.sub(r"::<[^()]*?>", "", body)            # turbofish commas
            calls.append((f, m.group(1), len(_split_top(body))))
    return calls


@pytest.mark.parametrize("crate", ["co-groth16-hip", "co-plonk-hip", "co-noir-hip"])
def test_every_abi_call_of_the_shim_matches_the_header_arity(crate):
    d = os.path.join(ROOT, "rust", crate, "src")
    if not os.path.isdir(d):
        pytest.skip(crate + " not present")
    arity = _header_arity()
    calls = _rust_sys_calls(d)
    assert calls, crate
    for f, name, n in calls:
        assert name in arity, "%s calls %s, which include/cosnarks_hip.h does not declare" % (f, name)
        assert n == arity[name], "%s: %s called with %d arguments, the header declares %d" % (f, name, n, arity[name])
    # and the generated FFI block declares every function the shim calls
    sysrs = open(os.path.join(ROOT, "rust", "cosnarks-hip-sys", "src", "lib.rs")).read()
    for _f, name, _n in calls:
        assert re.search(r"pub fn %s\(" % name, sysrs), name


def test_circom_reduction_is_one_abi_call_through_the_public_surface():
    """one library call per witness map, masks and party index learnt through public trait methods only."""
    shim = _strip_comments(open(os.path.join(RUST, "hip_reduction.rs")).read())
    body = shim[shim.index("impl R1CSToQAP for HipCircomReduction"):shim.index("pub struct HipLibSnarkReduction")]
    # two branches, ONE call each: the opt-in seeded mode returns right after its call, the default hands the two host masks over
    assert re.findall(r"sys::(csh_\w+)", body) == ["csh_groth16_witness_map", "csh_groth16_witness_map_masks"]
    seeded = body[body.index("seeded_masks_enabled()"):body.index("csh_groth16_witness_map_masks")]
    assert "return Ok(h);" in seeded
    assert "csh_ifft" not in shim and "csh_vec_" not in shim            # no per-step host round trips left
    assert "T::local_mul_vec(zeros(), zeros(), state)" in shim and "T::promote_to_trivial_shares(id, &[P::ScalarField::one()])" in shim
    assert "state.rngs" not in shim                                       # nothing protocol-specific in the generic reduction
    # VERDICT r4 #1b: the mask is asked for with EMPTY operands (length in a thread-local); the zero-vector form is only the fallback for a
    # foreign T, after the request came back unanswered
    dm = shim[shim.index("fn draw_mask"):shim.index("fn draw_seeds")]
    assert dm.index("MaskRequest::MaskOnly(n), || T::local_mul_vec(Vec::new(), Vec::new(), state)") < dm.index("if m.len() == n") < dm.index("zeros()")
    drv = _strip_comments(open(os.path.join(RUST, "drivers.rs")).read())
    rep3 = drv[drv.index("for HipRep3Groth16Driver"):drv.index("pub struct HipShamirGroth16Driver")]
    assert "MaskRequest::MaskOnly(n) if a.is_empty() => return state.rngs.rand.masking_field_elements_vec::<P::ScalarField>(n)" in rep3
    # VERDICT r4 #1c: the seeded mode goes through the PUBLIC Rep3Rand::random_seeds (rngs.rs:233), which exists with this signature
    assert "state.rngs.rand.random_seeds()" in rep3
    rn = open(os.path.join(REF, "mpc-core/src/protocols/rep3/rngs.rs")).read()
    assert "pub fn random_seeds(&mut self) -> ([u8; crate::SEED_SIZE], [u8; crate::SEED_SIZE])" in rn
    assert re.search(r"pub struct Rep3Rand \{\s*rng1: RngType,\s*rng2: RngType,\s*\}", rn)     # the generators themselves stay private
    assert "pub fn masking_field_elements_vec<F: PrimeField>(&mut self, len: usize) -> Vec<F>" in rn
    st = open(os.path.join(REF, "mpc-core/src/protocols/rep3.rs")).read()
    assert re.search(r"pub rngs: Rep3CorrelatedRng", st) and "pub rand: Rep3Rand" in rn
    assert "Vec::with_capacity(n)" in shim and "set_len(domain_size)" in shim
    # promote_to_trivial_share's party rule, which protocol_of decodes (rep3/arithmetic.rs)
    ar = open(os.path.join(REF, "mpc-core/src/protocols/rep3/arithmetic.rs")).read()
    m = re.search(r"pub fn promote_to_trivial_share<F: PrimeField>\(id: PartyID, public_value: F\) -> FieldShare<F> \{(.*?)\n\}", ar, re.S)
    assert m and "PartyID::ID0 => Rep3PrimeFieldShare::new(public_value, F::zero())" in m.group(1)
    assert "PartyID::ID1 => Rep3PrimeFieldShare::new(F::zero(), public_value)" in m.group(1)


# ---- seam 3: CircomPlonkProver / NoirUltraHonkProver implementors (rust/co-plonk-hip, rust/co-noir-hip) -----------------------------------
SEAM3 = [("co-plonk-hip", "co-circom/co-plonk/src/mpc.rs", "CircomPlonkProver", ["local_mul_vec", "fft", "ifft", "msm_public_points_g1"],
          [("Plain", "co-circom/co-plonk/src/mpc/plain.rs", "PlainPlonkDriver"), ("Rep3", "co-circom/co-plonk/src/mpc/rep3.rs", "Rep3PlonkDriver"),
           ("Shamir", "co-circom/co-plonk/src/mpc/shamir.rs", "ShamirPlonkDriver")], "Hip%sPlonkDriver"),
         ("co-noir-hip", "co-noir/co-noir-common/src/mpc/mod.rs", "NoirUltraHonkProver", ["local_mul_vec", "fft", "ifft", "msm_public_points"],
          [("Plain", "co-noir/co-noir-common/src/mpc/plain.rs", "PlainUltraHonkDriver"), ("Rep3", "co-noir/co-noir-common/src/mpc/rep3.rs", "Rep3UltraHonkDriver"),
           ("Shamir", "co-noir/co-noir-common/src/mpc/shamir.rs", "ShamirUltraHonkDriver")], "Hip%sUltraHonkDriver")]


def _impl_block(text, struct):
    i = text.index("for %s" % struct)
    j = text.index("{", i)
    depth, k = 0, j
    while True:
        depth += text[k] == "{"
        depth -= text[k] == "}"
        if depth == 0:
            break
        k += 1
    h = text.rindex("impl", 0, i)
    return text[h:j], text[j + 1:k]


@pytest.mark.parametrize("crate,trait_file,trait,hot,drivers,pattern", SEAM3)
def test_seam3_implementors_cover_the_trait_with_the_reference_signatures(crate, trait_file, trait, hot, drivers, pattern):
    ref = _trait_methods(open(os.path.join(REF, trait_file)).read(), trait)
    src = os.path.join(ROOT, "rust", crate, "src")
    cold = _fn_params(_strip_comments(open(os.path.join(src, "cold.rs")).read()))
    shim = _strip_comments(open(os.path.join(src, "drivers.rs")).read())
    assert set(cold) == set(ref) - set(hot), (sorted(set(ref) - set(hot) - set(cold)), sorted(set(cold) - set(ref)))
    for name, params in ref.items():
        (want,) = params
        if name in cold:
            (have,) = cold[name]
            assert len(have) == len(want) and all(h == w or w == "_" for h, w in zip(have, want)), (name, have, want)
    for proto, ref_file, ref_struct in drivers:
        hip = pattern % proto
        head, body = _impl_block(shim, hip)
        rhead, rbody = _impl_block(_strip_comments(open(os.path.join(REF, ref_file)).read()), ref_struct)
        # same impl header (generic parameters and where clause), same associated types as the reference driver
        norm = lambda t: re.sub(r"\s+", " ", t).strip().rstrip(",")
        assert norm(head.replace(hip, "X")) == norm(rhead.replace(ref_struct, "X")), (hip, head, rhead)
        assert sorted(norm(t) for t in re.findall(r"type \w+ = [^;]+;", body)) == sorted(norm(t) for t in re.findall(r"type \w+ = [^;]+;", rbody)), hip
        assert "_cold_methods!(%s);" % ref_struct in body, hip
        got = _fn_params(body)
        assert set(got) == set(hot), (hip, sorted(got))
        for name in hot:                       # the hand-written hot methods: the trait's parameter list
            (want,) = ref[name]
            (have,) = got[name]
            assert len(have) == len(want) and all(h == w or h.lstrip("_") == w or w == "_" for h, w in zip(have, want)), (hip, name, have, want)


def test_generated_delegations_are_up_to_date():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_rust_delegates", os.path.join(ROOT, "tools", "gen_rust_delegates.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    for src, trait, macro, hot, dst in gen.JOBS:
        text = gen.strip_comments(open(os.path.join(REF, src)).read())
        body = gen.emit_macro(macro, trait, gen.methods(gen.trait_body(text, trait)), set(hot))
        have = open(os.path.join(ROOT, dst)).read()
        assert have.endswith(body), dst + " is stale: run python tools/gen_rust_delegates.py"
