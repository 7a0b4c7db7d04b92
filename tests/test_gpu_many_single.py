"""-m gpu: the *_many forms of the single-proof functions (kzg355_verify_kzg_proof_many, kzg355_verify_blob_kzg_proof_many,
kzg355_compute_kzg_proof_many and the two *_device forms) through the C ABI.

The reference's pass rule (src/lib.rs:189-201) per UNIT of one call: every vector of verify_kzg_proof (92), verify_blob_kzg_proof (24) and
compute_kzg_proof (46) whose inputs parse goes into ONE call of the *_many form; a unit whose status is non-zero is that vector's Err (expected
output null), the others must carry the expected bool / bytes.  Then the same checks against the single-call entry points and the oracle at sizes
that take the throughput kernels (k_lincomb_single: four ladder lanes per check, from 64 checks on) and the latency forms (fewer)."""
import ctypes as C

import pytest

from synth import random_blob, random_field_element
from vector_harness import ParseError, get_blob, hx, parse_fixed

pytestmark = pytest.mark.gpu

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


@pytest.fixture(scope="module")
def kz():
    import kzg_rust_amd
    return kzg_rust_amd


@pytest.fixture(scope="module")
def settings(kz, setup_bytes):
    g1, g2 = setup_bytes
    s = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
    yield s
    s.free()


def parsed_cases(cases, parse):
    """(case, args) of every vector whose inputs parse; the others must expect null (the newtypes' errors come before any FFI: kzg.rs:107-178)."""
    out = []
    for c in cases:
        try:
            out.append((c, parse(c["input"])))
        except ParseError:
            assert c["output"] is None, c["name"]
    return out


def check_units(units, results, want_of, got_of):
    failures = []
    for (case, _), res in zip(units, results):
        exp = case["output"]
        if isinstance(res, Exception):
            if exp is not None:
                failures.append(f"{case['name']}: got Err({res}) expected {exp!r}")
        elif exp is None:
            failures.append(f"{case['name']}: got Ok({res!r}) expected Err")
        elif want_of(exp) != got_of(res):
            failures.append(f"{case['name']}: got {got_of(res)!r} expected {want_of(exp)!r}")
    assert not failures, "\n".join(failures)


def test_verify_kzg_proof_vectors_in_one_call(kz, settings, golden_vectors):
    units = parsed_cases(golden_vectors["verify_kzg_proof"],
                         lambda i: (parse_fixed(i["commitment"], 48), parse_fixed(i["z"], 32), parse_fixed(i["y"], 32), parse_fixed(i["proof"], 48)))
    assert len(units) >= 64                                       # enough checks for the throughput form (k_lincomb_single)
    cs, zs, ys, ps = zip(*[a for _, a in units])
    res = kz.Kzg.verify_kzg_proof_many(cs, zs, ys, ps, settings)
    check_units(units, res, bool, bool)
    assert sum(1 for r in res if r is True) >= 30 and sum(1 for r in res if r is False) >= 30 and sum(1 for r in res if isinstance(r, Exception)) >= 10
    # the same units through the latency forms: fewer than 64 checks per call (pre-shifted linear combination), and one at a time
    for lo in range(0, len(units), 23):
        part = units[lo:lo + 23]
        c2, z2, y2, p2 = zip(*[a for _, a in part])
        check_units(part, kz.Kzg.verify_kzg_proof_many(c2, z2, y2, p2, settings), bool, bool)
    one = kz.Kzg.verify_kzg_proof_many(cs[:1], zs[:1], ys[:1], ps[:1], settings)
    check_units(units[:1], one, bool, bool)
    assert kz.Kzg.verify_kzg_proof_many([], [], [], [], settings) == []
    with pytest.raises(kz.BadArgs):
        kz.Kzg.verify_kzg_proof_many(cs, zs[:-1], ys, ps, settings)


def test_verify_blob_kzg_proof_vectors_in_one_call(kz, settings, golden_vectors, golden_blobs):
    units = parsed_cases(golden_vectors["verify_blob_kzg_proof"],
                         lambda i: (get_blob(i["blob"], golden_blobs), parse_fixed(i["commitment"], 48), parse_fixed(i["proof"], 48)))
    assert len(units) >= 12
    bl, cs, ps = zip(*[a for _, a in units])
    res = kz.Kzg.verify_blob_kzg_proof_many(bl, cs, ps, settings)
    check_units(units, res, bool, bool)
    # ... and tiled past 64 units, so that stage 2 takes the throughput form of the linear combination
    reps = 64 // len(units) + 1
    res = kz.Kzg.verify_blob_kzg_proof_many(bl * reps, cs * reps, ps * reps, settings)
    check_units(units * reps, res, bool, bool)


def test_compute_kzg_proof_vectors_in_one_call(kz, settings, golden_vectors, golden_blobs):
    units = parsed_cases(golden_vectors["compute_kzg_proof"], lambda i: (get_blob(i["blob"], golden_blobs), parse_fixed(i["z"], 32)))
    assert len(units) >= 30
    bl, zs = zip(*[a for _, a in units])
    res = kz.Kzg.compute_kzg_proof_many(bl, zs, settings)
    check_units(units, res, lambda e: (hx(e[0]), hx(e[1])), lambda r: (bytes(r[0]), bytes(r[1])))
    assert sum(1 for r in res if isinstance(r, Exception)) >= 1 and sum(1 for r in res if not isinstance(r, Exception)) >= 30
    assert kz.Kzg.compute_kzg_proof_many([], [], settings) == []


@pytest.fixture(scope="module")
def honest(kz, settings, oracle, oracle_settings):
    """200 honest (C, z, y, proof) tuples over 8 seeded blobs: commitments and proofs from the ORACLE, so the product only ever checks them."""
    blobs = [random_blob(7000 + i) for i in range(8)]
    cs = [oracle.blob_to_kzg_commitment(b, oracle_settings) for b in blobs]
    out = []
    for k in range(200):
        z = random_field_element(5000 + k)
        if k % 50 == 0:                                           # a point inside the domain now and then (kzg.rs:494-523): a power of the 4096-th root of unity
            z = pow(pow(7, (R - 1) // 4096, R), k + 1, R).to_bytes(32, "big")      # 7 generates Fr* (consts.rs:163-168)
        p, y = oracle.compute_kzg_proof(blobs[k % 8], z, oracle_settings)
        out.append((k % 8, cs[k % 8], z, y, p))
    return blobs, out


def test_many_checks_against_oracle_and_single_calls(kz, settings, honest, oracle, oracle_settings):
    blobs, tuples = honest
    cs = [t[1] for t in tuples]; zs = [t[2] for t in tuples]; ys = [t[3] for t in tuples]; ps = [t[4] for t in tuples]
    # compute_kzg_proof_many reproduces the oracle's proofs and y values, in-domain points included
    got = kz.Kzg.compute_kzg_proof_many([blobs[t[0]] for t in tuples], zs, settings)
    assert [(bytes(p), bytes(y)) for p, y in got] == [(t[4], t[3]) for t in tuples]
    assert kz.Kzg.verify_kzg_proof_many(cs, zs, ys, ps, settings) == [True] * len(tuples)
    # every third y off by one, every seventh proof swapped with its neighbour's, one non-canonical z, one point off the subgroup's encoding rules
    ys2, ps2, zs2, cs2 = list(ys), list(ps), list(zs), list(cs)
    expect = [True] * len(tuples)
    for i in range(0, len(tuples), 3):
        ys2[i] = ((int.from_bytes(ys[i], "big") + 1) % R).to_bytes(32, "big"); expect[i] = False
    for i in range(1, len(tuples) - 1, 7):
        ps2[i] = ps[i + 1]; expect[i] = expect[i] and ps[i + 1] == ps[i]
    zs2[4] = R.to_bytes(32, "big"); expect[4] = "err"             # z = r: bytes_to_bls_field fails (utils.rs:267-271)
    cs2[5] = bytes([0x9a]) + b"\xff" * 47; expect[5] = "err"       # x >= p (utils.rs:291-296)
    res = kz.Kzg.verify_kzg_proof_many(cs2, zs2, ys2, ps2, settings)
    for i, (r, e) in enumerate(zip(res, expect)):
        if e == "err":
            assert isinstance(r, kz.BadArgs), i
        else:
            assert r is e, i
    # the oracle and the single-call entry point agree on a sample of the same units
    for i in (0, 1, 2, 3, 8, 50, 100, 150):
        assert kz.Kzg.verify_kzg_proof(kz.KzgCommitment(cs2[i]), kz.Bytes32(zs2[i]), kz.Bytes32(ys2[i]), kz.KzgProof(ps2[i]), settings) is expect[i]
        assert oracle.verify_kzg_proof(cs2[i], zs2[i], ys2[i], ps2[i], oracle_settings) is expect[i]


def test_many_forms_agree_with_the_window_form_and_the_device_entry_points(kz, setup_bytes, settings, honest):
    """The throughput kernel of the n = 1 linear combination (four lanes per check) against the per-term window ladder pinned through
    kzg355_options.lincomb_form = 1, and the *_device forms against the host forms."""
    import torch
    blobs, tuples = honest
    g1, g2 = setup_bytes
    L = kz.kzg.lib()
    n = len(tuples)
    rec = bytearray()
    for k, (_, c, z, y, p) in enumerate(tuples):
        if k % 4 == 1:
            y = ((int.from_bytes(y, "big") + 2) % R).to_bytes(32, "big")
        rec += c + z + y + p
    expect = [k % 4 != 1 for k in range(n)]
    dev = torch.device("cuda", settings.device)
    t_rec = torch.frombuffer(rec, dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    ok = (C.c_bool * n)(); st = (C.c_int * n)()
    assert L.kzg355_verify_kzg_proof_many_device(ok, st, t_rec.data_ptr(), n, settings.handle) == 0
    assert [bool(x) for x in ok] == expect and not any(st)
    sw = kz.KzgSettings.load_trusted_setup_ex([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)],
                                              device=settings.device, lincomb_form=1)
    try:
        ok2 = (C.c_bool * n)()
        assert L.kzg355_verify_kzg_proof_many_device(ok2, st, t_rec.data_ptr(), n, sw.handle) == 0
        assert [bool(x) for x in ok2] == expect and not any(st)
    finally:
        sw.free()
    # compute_kzg_proof_many_device: blobs and points in HBM, proofs and y on the host
    m = 12
    tb = torch.frombuffer(bytearray(b"".join(blobs[t[0]] for t in tuples[:m])), dtype=torch.uint8).to(dev)
    tz = torch.frombuffer(bytearray(b"".join(t[2] for t in tuples[:m])), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    out = C.create_string_buffer(48 * m); ys = C.create_string_buffer(32 * m); stm = (C.c_int * m)()
    assert L.kzg355_compute_kzg_proof_many_device(out, ys, stm, tb.data_ptr(), tz.data_ptr(), m, settings.handle) == 0
    assert out.raw == b"".join(t[4] for t in tuples[:m]) and ys.raw == b"".join(t[3] for t in tuples[:m])
    # a misaligned record buffer is refused before anything is queued
    assert L.kzg355_verify_kzg_proof_many_device(ok, st, t_rec.data_ptr() + 4, n - 1, settings.handle) == 1
    assert all(x == 1 for x in st[:n - 1])


def test_minimal_preset_many_forms(kz, setup_bytes):
    """The same three entry points on a FIELD_ELEMENTS_PER_BLOB = 4 handle against tests/golden/minimal.json (oracle-derived)."""
    import json
    import os
    from kzg_rust_amd import kzg_minimal as km
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "minimal.json")))
    g2 = setup_bytes[1]
    s = km.Kzg.load_trusted_setup([bytes.fromhex(x) for x in fx["setup_g1_lagrange"]], [g2[96 * i:96 * i + 96] for i in range(65)])
    try:
        blobs = [km.Blob(bytes.fromhex(b)) for b in fx["blobs"]]
        cases = fx["compute_kzg_proof"] * 10                       # 70 units: past the 64-check threshold of the throughput form
        res = km.Kzg.compute_kzg_proof_many([blobs[c["blob"]] for c in cases], [bytes.fromhex(c["z"]) for c in cases], s)
        assert [(p.to_bytes().hex(), y.to_bytes().hex()) for p, y in res] == [(c["proof"], c["y"]) for c in cases]
        cs = [bytes.fromhex(fx["commitments"][c["blob"]]) for c in cases]
        ys = [bytes.fromhex(c["y"]) for c in cases]
        ys[3] = ((int.from_bytes(ys[3], "big") + 1) % R).to_bytes(32, "big")
        v = km.Kzg.verify_kzg_proof_many(cs, [bytes.fromhex(c["z"]) for c in cases], ys, [bytes.fromhex(c["proof"]) for c in cases], s)
        assert v == [i != 3 for i in range(len(cases))]
        nb = len(blobs)
        vb = km.Kzg.verify_blob_kzg_proof_many(blobs * 12, [bytes.fromhex(c) for c in fx["commitments"]] * 12,
                                               [bytes.fromhex(p) for p in fx["blob_proofs"]] * 12, s)
        assert vb == [True] * (12 * nb)
        bad = km.Blob(bytes.fromhex(fx["invalid_blob"]))
        vb = km.Kzg.verify_blob_kzg_proof_many([bad] + blobs[1:], [bytes.fromhex(c) for c in fx["commitments"]], [bytes.fromhex(p) for p in fx["blob_proofs"]], s)
        assert isinstance(vb[0], kz.BadArgs) and vb[1:] == [True] * (nb - 1)
    finally:
        s.free()
