"""Lexical sanity of rust/ (no rustc in this image): delimiters balance outside comments / strings / char literals, every `mod x;`
has its file, every `crate::x` path names a declared module, every crate's Cargo.toml names the crates its sources `use`."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUST = os.path.join(ROOT, "rust")


def _code_only(src):
    out, i, n = [], 0, len(src)
    while i < n:
        if src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            i = n if j < 0 else j + 2
        elif src[i] == '"':
            j = i + 1
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            i = j + 1
        elif src[i] == "'" and re.match(r"'(\\.|[^\\'])'", src[i:]):
            i += re.match(r"'(\\.|[^\\'])'", src[i:]).end()
        else:
            out.append(src[i])
            i += 1
    return "".join(out)


def test_delimiters_balance_in_every_source_file():
    files = sorted(glob.glob(os.path.join(RUST, "**", "*.rs"), recursive=True))
    assert len(files) >= 15
    pairs = {")": "(", "]": "[", "}": "{"}
    for f in files:
        stack = []
        for ch in _code_only(open(f).read()):
            if ch in "([{":
                stack.append(ch)
            elif ch in pairs:
                assert stack and stack.pop() == pairs[ch], f
        assert not stack, f


def test_modules_and_crate_paths_resolve():
    for crate in sorted(os.listdir(RUST)):
        src = os.path.join(RUST, crate, "src")
        if not os.path.isdir(src):
            continue
        lib = _code_only(open(os.path.join(src, "lib.rs")).read())
        mods = set(re.findall(r"\bmod\s+(\w+)\s*;", lib))
        for m in mods:
            assert os.path.exists(os.path.join(src, m + ".rs")) or os.path.exists(os.path.join(src, m, "mod.rs")), (crate, m)
        for f in glob.glob(os.path.join(src, "*.rs")):
            for m in re.findall(r"\bcrate::(\w+)", _code_only(open(f).read())):
                assert m in mods or re.search(r"\b(pub\s+)?(fn|struct|enum|type|use[^;]*\b)\s*" + m + r"\b", lib), (crate, os.path.basename(f), m)


def test_cargo_manifests_name_the_crates_the_sources_use():
    std = {"std", "core", "alloc", "crate", "self", "super"}
    for crate in sorted(os.listdir(RUST)):
        toml = os.path.join(RUST, crate, "Cargo.toml")
        if not os.path.exists(toml):
            continue
        manifest = open(toml).read()
        deps = {d.replace("-", "_") for d in re.findall(r"^([A-Za-z0-9_-]+)\s*=", manifest, re.M)} | {crate.replace("-", "_")}
        used = set()
        lib = os.path.join(RUST, crate, "src", "lib.rs")
        local = set(re.findall(r"\bmod\s+(\w+)\s*;", _code_only(open(lib).read()))) if os.path.exists(lib) else set()
        for f in glob.glob(os.path.join(RUST, crate, "src", "*.rs")):
            code = _code_only(open(f).read())
            used |= set(re.findall(r"^\s*(?:pub\s+)?use\s+(\w+)::", code, re.M))
            used |= set(re.findall(r"^\s*extern\s+crate\s+(\w+)", code, re.M))
        for u in used - std - local:
            assert u in deps, (crate, u, sorted(deps))
