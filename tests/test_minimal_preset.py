"""The minimal preset (FIELD_ELEMENTS_PER_BLOB = 4; BASELINE.json configs[0]).

CPU part (-m "not gpu"): the C oracle built with -DN_FE=4 reproduces tests/golden/minimal.json (oracle-derived, cross-checked by
oracle/pyref.py and by the monomial identity when it was generated -- tests/golden/make_minimal_fixtures.py) and pyref agrees
again here on a sample.  GPU part (-m gpu): the product's small-domain path (csrc/k_small.hip) through the C ABI."""
import ctypes as C
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


@pytest.fixture(scope="module")
def fx():
    return json.load(open(os.path.join(HERE, "golden", "minimal.json")))


@pytest.fixture(scope="module")
def mono():
    b = open(os.path.join(HERE, "golden", "setup_g1_monomial_first4.bin"), "rb").read()
    return [b[48 * i:48 * i + 48] for i in range(4)]


@pytest.fixture(scope="module")
def omin(fx, setup_bytes):
    from oracle.oracle import Oracle
    o = Oracle(preset="minimal")
    s = o.load_trusted_setup(b"".join(bytes.fromhex(x) for x in fx["setup_g1_lagrange"]), setup_bytes[1])
    yield o, s
    o.free_trusted_setup(s)


def test_oracle_minimal_matches_fixture(fx, mono, omin, setup_bytes):
    from oracle import minimal_setup, pyref
    from oracle.oracle import OracleError
    o, s = omin
    assert o.field_elements_per_blob == 4 == fx["field_elements_per_blob"]
    assert [x.hex() for x in minimal_setup.lagrange_from_monomial(o, mono)] == fx["setup_g1_lagrange"]
    assert o.roots_of_unity(s).hex() == "".join(fx["roots_of_unity_brp"])
    blobs = [bytes.fromhex(b) for b in fx["blobs"]]
    for b, c, p in zip(blobs, fx["commitments"], fx["blob_proofs"]):
        assert o.blob_to_kzg_commitment(b, s).hex() == c
        assert o.compute_blob_kzg_proof(b, bytes.fromhex(c), s).hex() == p
        assert o.verify_blob_kzg_proof(b, bytes.fromhex(c), bytes.fromhex(p), s) is True
    for case in fx["compute_kzg_proof"]:
        pr, y = o.compute_kzg_proof(blobs[case["blob"]], bytes.fromhex(case["z"]), s)
        assert (pr.hex(), y.hex()) == (case["proof"], case["y"])
    n = fx["batch"]["n"]
    cs = [bytes.fromhex(x) for x in fx["commitments"][:n]]; ps = [bytes.fromhex(x) for x in fx["blob_proofs"][:n]]
    inter = o.verify_batch_intermediates(blobs[:n], cs, ps, s)
    assert (inter["ok"], inter["r"].hex(), inter["proof_lincomb"].hex(), inter["rhs"].hex()) == (True, fx["batch"]["r"], fx["batch"]["proof_lincomb"], fx["batch"]["rhs"])
    with pytest.raises(OracleError):
        o.blob_to_kzg_commitment(bytes.fromhex(fx["invalid_blob"]), s)
    # independent big-integer restatement on a sample (pyref has its own group law and pairing)
    pyref.set_preset(4)
    try:
        g2 = setup_bytes[1]
        ps_ = pyref.Settings([bytes.fromhex(x) for x in fx["setup_g1_lagrange"]], [g2[96 * i:96 * i + 96] for i in range(65)])
        assert pyref.blob_to_kzg_commitment(blobs[1], ps_).hex() == fx["commitments"][1]
        case = fx["compute_kzg_proof"][4]          # z = w_4: inside the domain
        pr, y = pyref.compute_kzg_proof(blobs[case["blob"]], bytes.fromhex(case["z"]), ps_)
        assert (pr.hex(), y.hex()) == (case["proof"], case["y"])
        assert pyref.verify_blob_kzg_proof_batch(blobs[:3], cs[:3], ps[:3], ps_) is True
    finally:
        pyref.set_preset(4096)


def test_mainnet_oracle_rejects_a_four_point_setup(fx, setup_bytes, oracle):
    """The mainnet build keeps the reference's compile-time check n1 == FIELD_ELEMENTS_PER_BLOB (kzg.rs:843)."""
    from oracle.oracle import OracleError
    with pytest.raises(OracleError):
        oracle.load_trusted_setup(b"".join(bytes.fromhex(x) for x in fx["setup_g1_lagrange"]), setup_bytes[1])


# ------------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def kzmin(fx, setup_bytes):
    import kzg_rust_amd as kz
    from kzg_rust_amd import kzg_minimal as km
    g2 = setup_bytes[1]
    s = km.Kzg.load_trusted_setup([bytes.fromhex(x) for x in fx["setup_g1_lagrange"]], [g2[96 * i:96 * i + 96] for i in range(65)])
    yield kz, km, s
    s.free()


@pytest.mark.gpu
def test_gpu_lagrange_setup_from_monomial(fx, mono, kzmin):
    kz, km, s = kzmin
    assert [x.hex() for x in km.lagrange_setup_from_monomial(mono)] == fx["setup_g1_lagrange"]
    assert s.field_elements_per_blob == 4
    bad = list(mono); bad[2] = bytes([0x9a]) + b"\xff" * 47
    with pytest.raises(kz.BadArgs):
        km.lagrange_setup_from_monomial(bad)
    with pytest.raises(kz.BadArgs):
        km.lagrange_setup_from_monomial(mono[:3])


@pytest.mark.gpu
def test_gpu_minimal_commit_prove_verify(fx, kzmin, omin):
    kz, km, s = kzmin
    o, so = omin
    blobs = [bytes.fromhex(b) for b in fx["blobs"]]
    B = [km.Blob(b) for b in blobs]
    for b, c in zip(B, fx["commitments"]):
        assert km.Kzg.blob_to_kzg_commitment(b, s).to_bytes().hex() == c
    cs = km.Kzg.blob_to_kzg_commitment_many(B, s)
    assert [c.to_bytes().hex() for c in cs] == fx["commitments"]
    ps = km.Kzg.compute_blob_kzg_proof_many(B, cs, s)
    assert [p.to_bytes().hex() for p in ps] == fx["blob_proofs"]
    for case in fx["compute_kzg_proof"]:            # z = 0, 1, 2, r-1, w, w^3 (in the domain) and a random point
        pr, y = km.Kzg.compute_kzg_proof(B[case["blob"]], kz.Bytes32(bytes.fromhex(case["z"])), s)
        assert (pr.to_bytes().hex(), y.to_bytes().hex()) == (case["proof"], case["y"])
        assert km.Kzg.verify_kzg_proof(cs[case["blob"]], kz.Bytes32(bytes.fromhex(case["z"])), y, pr, s) is True
        wrong_y = kz.Bytes32(((int.from_bytes(y.to_bytes(), "big") + 1) % R).to_bytes(32, "big"))
        assert km.Kzg.verify_kzg_proof(cs[case["blob"]], kz.Bytes32(bytes.fromhex(case["z"])), wrong_y, pr, s) is False
    for b, c, p in zip(B, cs, ps):
        assert km.Kzg.verify_blob_kzg_proof(b, c, p, s) is True
    n = fx["batch"]["n"]
    assert km.Kzg.verify_blob_kzg_proof_batch(B[:n], cs[:n], ps[:n], s) is fx["batch"]["expect"]
    a, b2 = fx["batch"]["swapped_pair"]
    sw = list(ps[:n]); sw[a], sw[b2] = sw[b2], sw[a]
    assert km.Kzg.verify_blob_kzg_proof_batch(B[:n], cs[:n], sw, s) is fx["batch"]["expect_swapped"]
    assert km.Kzg.verify_blob_kzg_proof_batch(B, cs, ps, s) is True
    assert km.Kzg.verify_blob_kzg_proof_batch([], [], [], s) is True
    # error rules: non-canonical element -> Err; wrong blob length for this preset -> InvalidBytesLength before any FFI
    with pytest.raises(kz.BadArgs):
        km.Kzg.blob_to_kzg_commitment(km.Blob(bytes.fromhex(fx["invalid_blob"])), s)
    with pytest.raises(kz.BadArgs):
        km.Kzg.verify_blob_kzg_proof_batch([km.Blob(bytes.fromhex(fx["invalid_blob"]))] + B[1:n], cs[:n], ps[:n], s)
    with pytest.raises(kz.InvalidBytesLength):
        km.Kzg.blob_to_kzg_commitment(kz.Blob(bytes(131072)), s)
    with pytest.raises(kz.InvalidBytesLength):
        km.Blob(bytes(127))
    # random blobs beyond the fixture, against the oracle
    from synth import splitmix64_bytes
    extra = [b"".join(b"\x00" + splitmix64_bytes(900 + 4 * k + i, 31) for i in range(4)) for k in range(70)]
    ce = km.Kzg.blob_to_kzg_commitment_many([km.Blob(b) for b in extra], s)
    assert [c.to_bytes() for c in ce] == [o.blob_to_kzg_commitment(b, so) for b in extra]
    pe = km.Kzg.compute_blob_kzg_proof_many([km.Blob(b) for b in extra], ce, s)
    assert [p.to_bytes() for p in pe] == [o.compute_blob_kzg_proof(b, c.to_bytes(), so) for b, c in zip(extra, ce)]
    assert km.Kzg.verify_blob_kzg_proof_batch([km.Blob(b) for b in extra], ce, pe, s) is True       # n = 70: bucket-form lincomb, two SHA chunks of r


@pytest.mark.gpu
def test_gpu_minimal_batch_intermediates(fx, kzmin):
    """Stage 1 records (z_i, y_i) and stage 2 (r, proof_lincomb, rhs) of the 6-blob batch, byte for byte: pins u64be(4) in both
    transcripts (kzg.rs:298-339, utils.rs:449)."""
    import torch
    kz, km, s = kzmin
    L = kz.kzg.lib()
    n = fx["batch"]["n"]
    dev = torch.device("cuda", s.device)
    tb = torch.frombuffer(bytearray(b"".join(bytes.fromhex(b) for b in fx["blobs"][:n])), dtype=torch.uint8).to(dev)
    tc = torch.frombuffer(bytearray(b"".join(bytes.fromhex(b) for b in fx["commitments"][:n])), dtype=torch.uint8).to(dev)
    tp = torch.frombuffer(bytearray(b"".join(bytes.fromhex(b) for b in fx["blob_proofs"][:n])), dtype=torch.uint8).to(dev)
    rec = torch.zeros(160 * n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    st = (C.c_int * 1)(-1)
    assert L.kzg355_verify_shard_records_device(rec.data_ptr(), st, tb.data_ptr(), tc.data_ptr(), tp.data_ptr(), n, 1, s.handle) == 0 and st[0] == 0
    r = bytes(rec.cpu().numpy())
    for i in range(n):
        assert r[160 * i + 48:160 * i + 80].hex() == fx["batch"]["z"][i]
        assert r[160 * i + 80:160 * i + 112].hex() == fx["batch"]["y"][i]
    out = C.create_string_buffer(128); ok = (C.c_bool * 1)()
    assert L.kzg355_debug_batch_intermediates(out, ok, st, rec.data_ptr(), n, 1, s.handle) == 0
    d = out.raw
    assert (d[:32].hex(), d[32:80].hex(), d[80:].hex(), bool(ok[0])) == (fx["batch"]["r"], fx["batch"]["proof_lincomb"], fx["batch"]["rhs"], True)
