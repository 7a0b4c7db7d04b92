"""Mechanical check of the Rust shim's `extern "C"` block (rust/src/ffi.rs) against include/kzg355.h -- the closest thing to `cargo check`
this image allows (no rustc).  For every function the shim binds: the header declares it, with the same number
of arguments, the same pointer depth and constness per argument, the same integer widths, and the same return type; `kzg355_options` has the
same fields in the same order on both sides (and in the ctypes mirror); the status constants agree; and the shim binds every entry point the
reference's `impl Kzg` needs (src/kzg.rs:983-1079)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_SCALARS = {"int": "c_int", "long": "c_long", "size_t": "usize", "uint8_t": "u8", "int32_t": "i32", "bool": "bool", "char": "c_char", "double": "f64", "void": "void",
             "kzg355_settings": "kzg355_settings", "kzg355_options": "kzg355_options", "kzg355_ticket": "kzg355_ticket"}
RUST_SCALARS = {"i32", "c_int", "c_long", "usize", "u8", "bool", "c_char", "f64", "kzg355_settings", "kzg355_options", "kzg355_ticket"}


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def c_type(decl):
    """'const uint8_t *blob' / 'uint8_t out[48]' / 'kzg355_settings **out' / 'size_t n' -> canonical Rust-style spelling"""
    decl = decl.strip()
    arr = re.search(r"\[[^\]]*\]\s*$", decl)
    if arr:
        decl = decl[:arr.start()]
    toks = re.findall(r"[A-Za-z_][A-Za-z0-9_]*|\*", decl)
    const = "const" in toks
    toks = [t for t in toks if t not in ("const", "struct")]
    base = toks[0]
    depth = toks.count("*") + (1 if arr else 0)
    assert base in C_SCALARS, decl
    t = C_SCALARS[base]
    if depth == 0:
        return t
    # in the header `const` always qualifies the pointee of the innermost pointer; outer levels (T **out) are mutable
    inner = ("*const " if const else "*mut ") + t
    return "*mut " * (depth - 1) + inner


def header_functions():
    text = strip_c_comments(open(os.path.join(ROOT, "include", "kzg355.h")).read())
    fns = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(kzg355_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argl = [] if args in ("", "void") else [c_type(a) for a in args.split(",")]
        fns[name] = (argl, c_type(ret + " x") if ret.strip() != "void" else "void")
    return fns


def rust_type(t):
    t = " ".join(t.split())
    parts = t.split(" ")
    assert parts[-1] in RUST_SCALARS and all(p in ("*const", "*mut") for p in parts[:-1]), t
    return t


def rust_functions():
    text = re.sub(r"//[^\n]*", "", open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read())
    block = re.search(r'extern\s+"C"\s*\{(.*)\}', text, flags=re.S).group(1)
    fns = {}
    for m in re.finditer(r"pub\s+fn\s+(kzg355_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block):
        name, args, ret = m.group(1), m.group(2).strip(), (m.group(3) or "void").strip()
        argl = [rust_type(a.split(":", 1)[1]) for a in args.split(",") if a.strip()]
        fns[name] = (argl, rust_type(ret) if ret != "void" else "void")
    return fns, text


def test_every_bound_function_matches_the_header():
    hdr = header_functions()
    rs, _ = rust_functions()
    assert len(rs) >= 25
    for name, (args, ret) in rs.items():
        assert name in hdr, f"{name}: bound by the shim, not declared in include/kzg355.h"
        h_args, h_ret = hdr[name]
        assert len(args) == len(h_args), f"{name}: {len(args)} arguments in ffi.rs, {len(h_args)} in the header"
        for i, (a, b) in enumerate(zip(args, h_args)):
            assert a == b, f"{name}: argument {i} is `{a}` in ffi.rs and `{b}` in the header"
        assert ret == h_ret, f"{name}: returns `{ret}` in ffi.rs and `{h_ret}` in the header"


def test_shim_binds_what_impl_kzg_needs():
    rs, _ = rust_functions()
    needed = ["kzg355_load_trusted_setup", "kzg355_load_trusted_setup_file", "kzg355_free_trusted_setup", "kzg355_blob_to_kzg_commitment",
              "kzg355_compute_kzg_proof", "kzg355_compute_blob_kzg_proof", "kzg355_verify_kzg_proof", "kzg355_verify_blob_kzg_proof",
              "kzg355_verify_blob_kzg_proof_batch"]                      # src/kzg.rs:995-1079 + Drop
    assert not [n for n in needed if n not in rs]
    # and the shim's own sources call nothing that ffi.rs does not declare
    for f in ("kzg.rs", "trusted_setup.rs", "lib.rs"):
        src = open(os.path.join(ROOT, "rust", "src", f)).read()
        for used in set(re.findall(r"ffi::(kzg355_[a-z0-9_]+)\s*\(", src)):
            assert used in rs, f"rust/src/{f} calls ffi::{used}, which ffi.rs does not declare"


def test_options_struct_is_the_same_on_all_three_sides():
    text = strip_c_comments(open(os.path.join(ROOT, "include", "kzg355.h")).read())
    body = re.search(r"typedef\s+struct\s+kzg355_options\s*\{(.*?)\}\s*kzg355_options\s*;", text, flags=re.S).group(1)
    c_fields = [(c_type(d), re.findall(r"[A-Za-z_][A-Za-z0-9_]*", d)[-1]) for d in body.split(";") if d.strip()]
    _, rs_text = rust_functions()
    rbody = re.search(r"pub\s+struct\s+kzg355_options\s*\{(.*?)\}", rs_text, flags=re.S).group(1)
    r_fields = [(rust_type(m.group(2)), m.group(1)) for m in re.finditer(r"pub\s+([a-z0-9_]+)\s*:\s*([^,]+),", rbody)]
    assert c_fields == r_fields
    assert "#[repr(C)]" in rs_text.split("pub struct kzg355_options")[0].rsplit("}", 1)[-1]
    import ctypes as C
    from kzg_rust_amd import _lib
    ct = {C.c_size_t: "usize", C.c_int: "c_int"}
    assert [(ct[t], n) for n, t in _lib.Options._fields_] == c_fields


def test_status_constants_agree():
    text = strip_c_comments(open(os.path.join(ROOT, "include", "kzg355.h")).read())
    c_consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(KZG355_[A-Z_]+)\s*=\s*(\d+)", text)}
    _, rs_text = rust_functions()
    r_consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub\s+const\s+(KZG355_[A-Z_]+)\s*:\s*c_int\s*=\s*(\d+)\s*;", rs_text)}
    assert len(c_consts) == 9 and r_consts == c_consts


def _rust_code_only(text):
    """the source with comments, string / byte-string / char literals and lifetimes blanked (delimiters inside them do not count)"""
    out, i, n = [], 0, len(text)
    while i < n:
        if text.startswith("//", i):
            j = text.find("\n", i); i = n if j < 0 else j
        elif text.startswith("/*", i):
            depth, i = 1, i + 2
            while i < n and depth:
                if text.startswith("/*", i): depth += 1; i += 2
                elif text.startswith("*/", i): depth -= 1; i += 2
                else: i += 1
        elif text[i] == '"' or (text[i] == "b" and text[i + 1:i + 2] == '"'):
            i += 1 if text[i] == '"' else 2
            while i < n and text[i] != '"':
                i += 2 if text[i] == "\\" else 1
            i += 1
        elif text[i] == "r" and re.match(r'r#*"', text[i:]):
            hashes = re.match(r'r(#*)"', text[i:]).group(1)
            j = text.find('"' + hashes, i + 2 + len(hashes)); i = n if j < 0 else j + 1 + len(hashes)
        elif text[i] == "'":
            m = re.match(r"'(\\.[^']*|[^'\\])'", text[i:])
            i += len(m.group(0)) if m else 1                     # a char literal, else a lifetime tick
        else:
            out.append(text[i]); i += 1
    return "".join(out)


def test_rust_sources_have_balanced_delimiters():
    """no compiler here: at least every (, [, { of rust/src/*.rs and rust/tests/*.rs closes in order, outside comments and literals"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "rust", "src", "*.rs")) + glob.glob(os.path.join(ROOT, "rust", "tests", "*.rs")))
    assert len(files) >= 4
    pairs = {")": "(", "]": "[", "}": "{"}
    for f in files:
        stack = []
        code = _rust_code_only(open(f).read())
        line = 1
        for ch in code:
            if ch == "\n":
                line += 1
            elif ch in "([{":
                stack.append((ch, line))
            elif ch in pairs:
                assert stack and stack[-1][0] == pairs[ch], f"{f}:{line}: unmatched {ch!r}" + (f" (open {stack[-1][0]!r} from line {stack[-1][1]})" if stack else "")
                stack.pop()
        assert not stack, f"{f}: {stack[-1][0]!r} opened at line {stack[-1][1]} never closes"
