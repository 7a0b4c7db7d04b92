"""The reference's vector-driven test logic (src/lib.rs:30-203), restated once and run against any
backend that exposes the `Kzg` surface (the CPU oracle here; the HIP product through the C-ABI in
the -m gpu tests).  Pass rule, identical for all six functions (e.g. src/lib.rs:189-201):
  * an input that fails to PARSE (bad hex / wrong length: Blob::from_hex kzg.rs:160-177,
    Bytes48::from_bytes kzg.rs:130-141, Bytes32::from_bytes kzg.rs:107-117) => output must be null;
  * otherwise call the API:  Ok(v) => v == output ;  Err(_) => output is null.
"""


class ParseError(Exception):
    pass


def hex_to_bytes(s):  # kzg.rs:82-86
    if not isinstance(s, str):
        raise ParseError("not a string")
    h = s[2:] if s.startswith("0x") else s
    try:
        return bytes.fromhex(h)
    except ValueError as e:
        raise ParseError(str(e))


def parse_fixed(s, n):
    b = hex_to_bytes(s)
    if len(b) != n:
        raise ParseError(f"length {len(b)} != {n}")
    return b


def get_blob(ref, blobs):
    if "raw" in ref:
        b = hex_to_bytes(ref["raw"])
    else:
        b = blobs[ref["blob"]]
    if len(b) != 131072:
        raise ParseError("blob length")
    return b


def hx(s):
    return hex_to_bytes(s)


def run_case(fn, case, backend, settings, blobs):
    """Returns None on pass, or a string describing the mismatch."""
    inp, exp = case["input"], case["output"]
    try:
        if fn == "blob_to_kzg_commitment":
            args = (get_blob(inp["blob"], blobs),)
        elif fn == "compute_kzg_proof":
            args = (get_blob(inp["blob"], blobs), parse_fixed(inp["z"], 32))
        elif fn == "compute_blob_kzg_proof":
            args = (get_blob(inp["blob"], blobs), parse_fixed(inp["commitment"], 48))
        elif fn == "verify_kzg_proof":
            args = (parse_fixed(inp["commitment"], 48), parse_fixed(inp["z"], 32), parse_fixed(inp["y"], 32),
                    parse_fixed(inp["proof"], 48))
        elif fn == "verify_blob_kzg_proof":
            args = (get_blob(inp["blob"], blobs), parse_fixed(inp["commitment"], 48), parse_fixed(inp["proof"], 48))
        elif fn == "verify_blob_kzg_proof_batch":
            args = ([get_blob(b, blobs) for b in inp["blobs"]], [parse_fixed(c, 48) for c in inp["commitments"]],
                    [parse_fixed(p, 48) for p in inp["proofs"]])
        else:
            raise AssertionError(fn)
    except ParseError:
        return None if exp is None else f"{case['name']}: input failed to parse but output is {exp!r}"
    try:
        res = getattr(backend, fn)(*args, settings)
    except Exception as e:  # any Err(_)
        return None if exp is None else f"{case['name']}: got Err({e}) expected {exp!r}"
    if exp is None:
        return f"{case['name']}: got Ok({res!r}) expected Err"
    if fn == "compute_kzg_proof":
        want = (hx(exp[0]), hx(exp[1]))
        got = (bytes(res[0]), bytes(res[1]))
    elif fn.startswith("verify"):
        want, got = bool(exp), bool(res)
    else:
        want, got = hx(exp), bytes(res)
    return None if want == got else f"{case['name']}: got {got!r} expected {want!r}"


def run_function(fn, vectors, backend, settings, blobs):
    cases = vectors[fn]
    assert cases
    failures = [m for m in (run_case(fn, c, backend, settings, blobs) for c in cases) if m]
    return len(cases), failures
