"""-m gpu: the reference's vector loop (src/lib.rs:30-203) replayed through the C++ host mirror include/kzg355.hpp
(`kzg355::Kzg`, `Blob`, `Bytes32`, `Bytes48`, `Result<T>`), i.e. through a compiled-language consumer of the C ABI, the
way the Rust crate would use it.  tests/native/cpp_vector_runner.cpp is a line-protocol driver around that header."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def runner(tmp_path_factory, setup_bytes):
    exe = os.path.join(NATIVE, "cpp_vector_runner")
    src = os.path.join(NATIVE, "cpp_vector_runner.cpp")
    hdr = os.path.join(ROOT, "include", "kzg355.hpp")
    if not os.path.exists(exe) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(exe):
        subprocess.run(["g++", "-std=c++17", "-O2", "-o", exe, src, "-L" + os.path.join(ROOT, "kzg_rust_amd"), "-lkzg355",
                        "-Wl,-rpath," + os.path.join(ROOT, "kzg_rust_amd")], check=True)
    g1, g2 = setup_bytes
    ts = tmp_path_factory.mktemp("ts") / "trusted_setup.txt"
    ts.write_text("4096\n65\n" + "\n".join(g1[48 * i:48 * i + 48].hex() for i in range(4096)) + "\n" +
                  "\n".join(g2[96 * i:96 * i + 96].hex() for i in range(65)) + "\n")
    p = subprocess.Popen([exe, str(ts)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1)
    assert p.stdout.readline().strip() == "ready"
    yield p
    p.stdin.close()
    p.wait(timeout=30)


def _blob_arg(ref):
    return "hex:" + ref["raw"] if "raw" in ref else os.path.join(GOLDEN, "blobs", f"blob_{ref['blob']}.bin")


def _ask(p, line):
    p.stdin.write(line + "\n")
    p.stdin.flush()
    return p.stdout.readline().strip()


def _arg(x):
    return x if x else "-"


@pytest.mark.parametrize("fn", ["blob_to_kzg_commitment", "compute_kzg_proof", "compute_blob_kzg_proof", "verify_kzg_proof",
                                "verify_blob_kzg_proof", "verify_blob_kzg_proof_batch"])
def test_reference_vectors_through_cpp_mirror(fn, runner, golden_vectors):
    failures = []
    for case in golden_vectors[fn]:
        i, exp = case["input"], case["output"]
        if fn == "blob_to_kzg_commitment":
            args = [_blob_arg(i["blob"])]
        elif fn == "compute_kzg_proof":
            args = [_blob_arg(i["blob"]), i["z"]]
        elif fn == "compute_blob_kzg_proof":
            args = [_blob_arg(i["blob"]), i["commitment"]]
        elif fn == "verify_kzg_proof":
            args = [i["commitment"], i["z"], i["y"], i["proof"]]
        elif fn == "verify_blob_kzg_proof":
            args = [_blob_arg(i["blob"]), i["commitment"], i["proof"]]
        else:
            args = [",".join(_blob_arg(b) for b in i["blobs"]), ",".join(i["commitments"]), ",".join(i["proofs"])]
        ans = _ask(runner, fn + " " + " ".join(_arg(a) for a in args))
        if exp is None:
            good = ans == "parse" or ans.startswith("err")
        elif fn == "compute_kzg_proof":
            good = ans == f"ok {exp[0][2:]} {exp[1][2:]}"
        elif fn.startswith("verify"):
            good = ans == f"ok {'true' if exp else 'false'}"
        else:
            good = ans == f"ok {exp[2:]}"
        if not good:
            failures.append(f"{case['name']}: {ans[:120]} expected {exp!r}")
    assert not failures, "\n".join(failures)


def _parses(hexes_and_sizes):
    for h, n in hexes_and_sizes:
        t = h[2:] if h.startswith("0x") else h
        try:
            if len(bytes.fromhex(t)) != n:
                return False
        except ValueError:
            return False
    return True


def test_single_proof_vectors_through_the_many_forms_of_the_cpp_mirror(runner, golden_vectors):
    """kzg355::Kzg::verify_kzg_proof_many / compute_kzg_proof_many (include/kzg355.hpp): every vector whose inputs parse, in ONE call each; a unit's Err is
    that vector's null output (src/lib.rs:189-201 per unit)."""
    cases = [c for c in golden_vectors["verify_kzg_proof"]
             if _parses([(c["input"]["commitment"], 48), (c["input"]["z"], 32), (c["input"]["y"], 32), (c["input"]["proof"], 48)])]
    assert len(cases) >= 64
    ans = _ask(runner, "verify_kzg_proof_many " + " ".join(",".join(c["input"][k] for c in cases) for k in ("commitment", "z", "y", "proof")))
    assert ans.startswith("many "), ans[:200]
    units = ans.split()[1:]
    assert len(units) == len(cases)
    bad = [c["name"] for c, u in zip(cases, units) if u != ("e" if c["output"] is None else "t" if c["output"] else "f")]
    assert not bad, bad
    cases = [c for c in golden_vectors["compute_kzg_proof"]
             if "blob" in c["input"]["blob"] and os.path.getsize(_blob_arg(c["input"]["blob"])) == 131072 and _parses([(c["input"]["z"], 32)])]
    assert len(cases) >= 30
    ans = _ask(runner, "compute_kzg_proof_many " + ",".join(_blob_arg(c["input"]["blob"]) for c in cases) + " " + ",".join(c["input"]["z"] for c in cases))
    assert ans.startswith("many "), ans[:200]
    units = ans.split()[1:]
    assert len(units) == len(cases)
    bad = [c["name"] for c, u in zip(cases, units) if u != ("e" if c["output"] is None else f"{c['output'][0][2:]}:{c['output'][1][2:]}")]
    assert not bad, bad
