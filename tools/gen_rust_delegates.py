#!/usr/bin/env python3
"""Generates the "every cold method delegates to the reference's own driver" macros of the Rust shims from the reference's trait
definitions (run in the build container, where /root/reference exists; the output is committed):

    python tools/gen_rust_delegates.py

    co-circom/co-plonk/src/mpc.rs            trait CircomPlonkProver<P>     -> rust/co-plonk-hip/src/cold.rs   (plonk_cold_methods!)
    co-noir/co-noir-common/src/mpc/mod.rs    trait NoirUltraHonkProver<P>   -> rust/co-noir-hip/src/cold.rs    (honk_cold_methods!)

The hot methods (the ones that reach the GPU: local_mul_vec, fft, ifft, msm_public_points[_g1]) are left out of the macro and written by
hand in the crate's drivers.rs. Only SIGNATURES are taken from the reference (an implementor of a trait has to repeat them verbatim);
every generated body is the one-line delegation `<$inner as Trait<P>>::method(args)`. No Rust toolchain exists in this image, so the
signatures are never re-typed by hand: what the reference declares is what the shim implements (tests/test_rust_shim_cpu.py re-parses
both sides and compares them)."""
import os
import re
import sys

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def strip_comments(s):
    s = re.sub(r"//[^\n]*", "", s)
    return re.sub(r"/\*.*?\*/", "", s, flags=re.S)


def trait_body(text, trait):
    i = text.index("pub trait " + trait)
    j = text.index("{", i)
    depth, k = 0, j
    while True:
        depth += text[k] == "{"
        depth -= text[k] == "}"
        if depth == 0:
            break
        k += 1
    return text[j + 1:k]


def split_top(s, sep=","):
    parts, depth, cur = [], 0, ""
    prev = ""
    for ch in s:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}" or (ch == ">" and prev != "-"):
            depth -= 1
        if ch == sep and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
        prev = ch
    if cur.strip():
        parts.append(cur)
    return parts


def methods(body):
    """[(name, generics text incl. <>, params text, tail text (return type + where clause), has_default_body)] in declaration order"""
    out = []
    i, n = 0, len(body)
    depth = 0
    while i < n:
        m = re.compile(r"\bfn\s+(\w+)").search(body, i)
        if not m:
            break
        # only functions at depth 0 of the trait body
        depth = body[:m.start()].count("{") - body[:m.start()].count("}")
        if depth != 0:
            i = m.end()
            continue
        name = m.group(1)
        k = m.end()
        generics = ""
        if body[k] == "<":
            d, j = 0, k
            while True:
                d += body[j] == "<"
                d -= body[j] == ">" and body[j - 1] != "-"
                j += 1
                if d == 0:
                    break
            generics = body[k:j]
            k = j
        assert body[k] == "(", (name, body[k:k + 20])
        d, j = 0, k
        while True:
            d += body[j] == "("
            d -= body[j] == ")"
            j += 1
            if d == 0:
                break
        params = body[k + 1:j - 1]
        # tail: up to ';' or '{' at depth 0
        d, t = 0, j
        while True:
            ch = body[t]
            if ch in "(<[":
                d += 1
            elif ch in ")]" or (ch == ">" and body[t - 1] != "-"):
                d -= 1
            if d == 0 and ch in ";{":
                break
            t += 1
        tail = body[j:t]
        has_body = body[t] == "{"
        if has_body:
            d, e = 0, t
            while True:
                d += body[e] == "{"
                d -= body[e] == "}"
                e += 1
                if d == 0:
                    break
            i = e
        else:
            i = t + 1
        out.append((name, re.sub(r"\s+", " ", generics).strip(), params, re.sub(r"\s+", " ", tail).strip(), has_body))
    return out


def generic_names(generics):
    if not generics:
        return []
    names = []
    for item in split_top(generics[1:-1]):
        item = item.strip()
        if not item or item.startswith("'"):
            continue
        item = re.sub(r"^const\s+", "", item)
        names.append(item.split(":")[0].strip())
    return names


def emit_macro(macro, trait, meths, hot):
    lines = ["macro_rules! %s {" % macro, "    ($inner:ty) => {"]
    for name, generics, params, tail, _has_body in meths:
        if name in hot:
            continue
        plist, args = [], []
        for idx, p in enumerate(split_top(params)):
            p = re.sub(r"\s+", " ", p).strip()
            if not p:
                continue
            pat, ty = p.split(":", 1)
            pat = pat.strip()
            ident = re.sub(r"^(mut|ref)\s+", "", pat)
            if ident == "_" or not re.match(r"^\w+$", ident):
                ident = "arg%d" % idx
                pat = ident
            plist.append("%s:%s" % (pat, ty))
            args.append(ident)
        gn = generic_names(generics)
        fish = "::<%s>" % ", ".join(gn) if gn else ""
        sig = "        fn %s%s(%s)%s" % (name, generics, ", ".join(plist), (" " + tail) if tail else "")
        lines.append(sig + " {")
        lines.append("            <$inner as %s<P>>::%s%s(%s)" % (trait, name, fish, ", ".join(args)))
        lines.append("        }")
    lines += ["    };", "}", "pub(crate) use %s;" % macro, ""]
    return "\n".join(lines)


JOBS = [
    ("co-circom/co-plonk/src/mpc.rs", "CircomPlonkProver", "plonk_cold_methods", ["local_mul_vec", "fft", "ifft", "msm_public_points_g1"],
     "rust/co-plonk-hip/src/cold.rs"),
    ("co-noir/co-noir-common/src/mpc/mod.rs", "NoirUltraHonkProver", "honk_cold_methods", ["local_mul_vec", "fft", "ifft", "msm_public_points"],
     "rust/co-noir-hip/src/cold.rs"),
]


def main():
    for src, trait, macro, hot, dst in JOBS:
        text = strip_comments(open(os.path.join(REF, src)).read())
        meths = methods(trait_body(text, trait))
        for h in hot:
            assert any(m[0] == h for m in meths), (trait, h)
        head = ("//! GENERATED by tools/gen_rust_delegates.py from the reference's `%s` (%s): every method that does not reach the GPU,\n"
                "//! delegated to the reference's own driver `$inner` with the trait's own signature. Do not edit; re-run the generator.\n"
                "//! Hot methods written by hand in drivers.rs: %s.\n" % (trait, src, ", ".join(hot)))
        out = head + emit_macro(macro, trait, meths, set(hot))
        path = os.path.join(ROOT, dst)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        open(path, "w").write(out)
        print("%s: %d methods, %d delegated -> %s" % (trait, len(meths), len(meths) - len(hot), dst))


if __name__ == "__main__":
    sys.exit(main())
