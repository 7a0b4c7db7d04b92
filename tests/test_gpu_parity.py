"""-m gpu: parity of the HIP engine (through the C ABI of libkzg355.so) with the reference's golden vectors and
with the CPU oracle.  Everything here runs on cuda:0 of the GPU box; /root/reference is never read."""
import ctypes as C
import os

import pytest

from synth import random_blob, random_field_element
from vector_harness import run_function

pytestmark = pytest.mark.gpu

COUNTS = {"blob_to_kzg_commitment": 10, "compute_kzg_proof": 46, "compute_blob_kzg_proof": 14,
          "verify_kzg_proof": 92, "verify_blob_kzg_proof": 24, "verify_blob_kzg_proof_batch": 22}


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


@pytest.fixture(scope="module")
def backend():
    from gpu_backend import ProductBackend
    return ProductBackend()


@pytest.mark.parametrize("fn", list(COUNTS))
def test_reference_vectors(fn, golden_vectors, golden_blobs, backend, settings):
    """The reference's own test loop (src/lib.rs:30-203) over all 208 vectors, product in place of blst."""
    n, failures = run_function(fn, golden_vectors, backend, settings, golden_blobs)
    assert n == COUNTS[fn]
    assert not failures, "\n".join(failures)


def test_load_trusted_setup_errors(kz, setup_bytes, tmp_path):
    g1, g2 = setup_bytes
    g1l = [g1[48 * i:48 * i + 48] for i in range(4096)]
    g2l = [g2[96 * i:96 * i + 96] for i in range(65)]
    with pytest.raises(kz.InvalidTrustedSetup):
        kz.Kzg.load_trusted_setup(g1l[:-1], g2l)                  # kzg.rs:49-55
    with pytest.raises(kz.InvalidTrustedSetup):
        kz.Kzg.load_trusted_setup(g1l, g2l[:-1])                  # kzg.rs:56-62
    bad = list(g1l); bad[7] = bytes([0x9a]) + b"\xff" * 47         # x >= p
    with pytest.raises(kz.BadArgs):
        kz.Kzg.load_trusted_setup(bad, g2l)                       # kzg.rs:863
    bad2 = list(g2l); bad2[64] = bytes(96)                         # uncompressed flag
    with pytest.raises(kz.BadArgs):
        kz.Kzg.load_trusted_setup(g1l, bad2)                      # kzg.rs:878
    mono = open(os.path.join(os.path.dirname(__file__), "golden", "setup_g1_monomial_first2.bin"), "rb").read()
    monol = [mono[:48], mono[48:96]] + g1l[2:]
    with pytest.raises(kz.BadArgs):
        kz.Kzg.load_trusted_setup(monol, g2l)                     # monomial form rejected, kzg.rs:823-826
    # file loader (kzg.rs:906-979)
    path = tmp_path / "ts.txt"
    path.write_text("4096\n65\n" + "\n".join(x.hex() for x in g1l) + "\n" + "\n".join(x.hex() for x in g2l) + "\n")
    s = kz.Kzg.load_trusted_setup_file(str(path))
    s.free()
    (tmp_path / "bad1.txt").write_text("4095\n65\n")
    with pytest.raises(kz.InvalidTrustedSetup):
        kz.Kzg.load_trusted_setup_file(str(tmp_path / "bad1.txt"))
    (tmp_path / "bad2.txt").write_text("4096\n64\n")
    with pytest.raises(kz.InvalidTrustedSetup):
        kz.Kzg.load_trusted_setup_file(str(tmp_path / "bad2.txt"))
    with pytest.raises(kz.InvalidTrustedSetup):
        kz.Kzg.load_trusted_setup_file(str(tmp_path / "missing.txt"))


def test_load_from_json_trusted_setup(kz, setup_bytes, golden_vectors, golden_blobs):
    """Kzg::load_trusted_setup fed from the JSON helper (src/trusted_setup.rs) gives a working handle."""
    import json
    g1, g2 = setup_bytes
    text = json.dumps({"setup_G1_lagrange": ["0x" + g1[48 * i:48 * i + 48].hex() for i in range(4096)],
                       "setup_G2": ["0x" + g2[96 * i:96 * i + 96].hex() for i in range(65)]})
    ts = kz.TrustedSetup.from_json(text)
    s = kz.Kzg.load_trusted_setup(ts.g1_points(), ts.g2_points())
    case = [c for c in golden_vectors["blob_to_kzg_commitment"] if c["output"]][0]
    got = kz.Kzg.blob_to_kzg_commitment(kz.Blob(golden_blobs[case["input"]["blob"]["blob"]]), s)
    assert got.to_bytes().hex() == case["output"][2:]
    s.free()


N_RANDOM = 8


@pytest.fixture(scope="module")
def random_set(oracle, oracle_settings):
    blobs = [random_blob(i) for i in range(N_RANDOM)]
    cs = [oracle.blob_to_kzg_commitment(b, oracle_settings) for b in blobs]
    ps = [oracle.compute_blob_kzg_proof(b, c, oracle_settings) for b, c in zip(blobs, cs)]
    return blobs, cs, ps


def test_random_commit_and_proof_match_oracle(kz, settings, random_set, oracle, oracle_settings):
    blobs, cs, ps = random_set
    got_c = kz.Kzg.blob_to_kzg_commitment_many([kz.Blob(b) for b in blobs], settings)
    assert [c.to_bytes() for c in got_c] == cs
    got_p = kz.Kzg.compute_blob_kzg_proof_many([kz.Blob(b) for b in blobs], [kz.KzgCommitment(c) for c in cs], settings)
    assert [p.to_bytes() for p in got_p] == ps
    for i in range(3):
        z = random_field_element(i)
        p, y = kz.Kzg.compute_kzg_proof(kz.Blob(blobs[i]), kz.Bytes32(z), settings)
        op, oy = oracle.compute_kzg_proof(blobs[i], z, oracle_settings)
        assert (p.to_bytes(), y.to_bytes()) == (op, oy)


def test_random_verify_batch(kz, settings, random_set, oracle, oracle_settings):
    blobs, cs, ps = random_set
    B, Cm, Pr = [kz.Blob(b) for b in blobs], [kz.KzgCommitment(c) for c in cs], [kz.KzgProof(p) for p in ps]
    assert kz.Kzg.verify_blob_kzg_proof_batch(B, Cm, Pr, settings) is True
    # swap two proofs: still valid points, wrong statement
    Pr2 = list(Pr); Pr2[1], Pr2[2] = Pr2[2], Pr2[1]
    assert kz.Kzg.verify_blob_kzg_proof_batch(B, Cm, Pr2, settings) is False
    assert oracle.verify_blob_kzg_proof_batch(blobs, cs, [p.to_bytes() for p in Pr2], oracle_settings) is False
    for n in (1, 2, 3):
        assert kz.Kzg.verify_blob_kzg_proof_batch(B[:n], Cm[:n], Pr[:n], settings) is True
    with pytest.raises(kz.BadArgs):
        kz.Kzg.verify_blob_kzg_proof_batch(B, Cm[:-1], Pr, settings)
    assert kz.Kzg.verify_blob_kzg_proof_batch([], [], [], settings) is True
    # several independent batches in one launch; one of them wrong, one with an invalid commitment
    bad_c = list(Cm[:4]); bad_c[0] = kz.KzgCommitment(bytes([0x9a]) + b"\xff" * 47)
    res = kz.Kzg.verify_blob_kzg_proof_batch_many([(B[:4], Cm[:4], Pr[:4]), (B[4:], Cm[4:], Pr[4:]), (B[:4], Cm[:4], Pr2[:4]), (B[:4], bad_c, Pr[:4])], settings)
    assert res[0] is True and res[1] is True and res[2] is False and isinstance(res[3], kz.BadArgs)


def test_stage_records_match_oracle(kz, settings, random_set, oracle, oracle_settings):
    """Stage-by-stage: the 160-byte records (C | z | y | proof) against the oracle's z_i / y_i, then stage 2 on them."""
    import torch
    blobs, cs, ps = random_set
    n = len(blobs)
    dev = torch.device("cuda", settings.device)
    t_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).to(dev)
    t_c = torch.frombuffer(bytearray(b"".join(cs)), dtype=torch.uint8).to(dev)
    t_p = torch.frombuffer(bytearray(b"".join(ps)), dtype=torch.uint8).to(dev)
    t_rec = torch.zeros(160 * n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    st = (C.c_int * 1)(-1)
    L = kz.kzg.lib()
    rc = L.kzg355_verify_shard_records_device(t_rec.data_ptr(), st, t_blobs.data_ptr(), t_c.data_ptr(), t_p.data_ptr(), n, 1, settings.handle)
    assert rc == 0 and st[0] == 0
    rec = bytes(t_rec.cpu().numpy())
    inter = oracle.verify_batch_intermediates(blobs, cs, ps, oracle_settings)
    for i in range(n):
        r = rec[160 * i:160 * i + 160]
        assert r[:48] == cs[i] and r[112:] == ps[i]
        assert r[48:80] == inter["z"][i], f"z[{i}]"
        assert r[80:112] == inter["y"][i], f"y[{i}]"
    ok = (C.c_bool * 2)(); st2 = (C.c_int * 2)()
    assert L.kzg355_verify_records_device(ok, st2, t_rec.data_ptr(), n, 1, settings.handle) == 0 and ok[0] is True
    # the same records viewed as two batches of n/2 (each half is a valid batch of its own)
    assert L.kzg355_verify_records_device(ok, st2, t_rec.data_ptr(), n // 2, 2, settings.handle) == 0 and list(ok) == [True, True]
    # device-resident batched entry point (what bench.py times)
    okm = (C.c_bool * 2)(); stm = (C.c_int * 2)()
    rc = L.kzg355_verify_blob_kzg_proof_batch_many_device(okm, stm, t_blobs.data_ptr(), t_c.data_ptr(), t_p.data_ptr(), n // 2, 2, settings.handle)
    assert rc == 0 and list(okm) == [True, True] and list(stm) == [0, 0]


def _product_commit_prove(kz, settings, blobs):
    B = [kz.Blob(b) for b in blobs]
    cs = kz.Kzg.blob_to_kzg_commitment_many(B, settings)
    ps = kz.Kzg.compute_blob_kzg_proof_many(B, cs, settings)
    return B, cs, ps


def test_eval_tree_extreme_blobs_match_oracle(kz, settings, oracle, oracle_settings):
    """k_eval's radix-4 tree (csrc/eval_core.h) at the worst case of its lazy bounds ON THE DEVICE: blobs whose values are all r - 1, or
    r - 1 / 0 patterns that maximise the sums and differences of every level.  y (and z) of the stage-1 records byte-exact against the
    oracle's evaluate_polynomial_in_evaluation_form (kzg.rs:346-389); the commitments are whatever the product computes (any G1 point
    gives a well-defined challenge)."""
    import torch
    R_ = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    top, zero = (R_ - 1).to_bytes(32, "big"), bytes(32)
    blobs = [top * 4096, (top + zero) * 2048, (zero + top) * 2048, (top + top + zero + zero) * 1024, (zero + zero + top + top) * 1024,
             (top * 16 + zero * 16) * 128, (zero * 64 + top * 64) * 32, top * 2048 + zero * 2048]
    ones = (((R_ >> 232) << 232) - 1).to_bytes(32, "big")          # every 29-bit limb below the top one at its maximum: the widest product columns
    blobs += [ones * 4096, (ones + zero) * 2048, (zero + zero + ones + ones) * 1024]
    B, cs, ps = _product_commit_prove(kz, settings, blobs)
    cs, ps = [bytes(c) for c in cs], [bytes(q) for q in ps]
    n = len(blobs)
    dev = torch.device("cuda", settings.device)
    t_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).to(dev)
    t_c = torch.frombuffer(bytearray(b"".join(cs)), dtype=torch.uint8).to(dev)
    t_p = torch.frombuffer(bytearray(b"".join(ps)), dtype=torch.uint8).to(dev)
    t_rec = torch.zeros(160 * n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    st = (C.c_int * 1)(-1)
    L = kz.kzg.lib()
    assert L.kzg355_verify_shard_records_device(t_rec.data_ptr(), st, t_blobs.data_ptr(), t_c.data_ptr(), t_p.data_ptr(), n, 1, settings.handle) == 0 and st[0] == 0
    rec = bytes(t_rec.cpu().numpy())
    inter = oracle.verify_batch_intermediates(blobs, cs, ps, oracle_settings)
    for i in range(n):
        assert rec[160 * i + 48:160 * i + 80] == inter["z"][i], f"z[{i}]"
        assert rec[160 * i + 80:160 * i + 112] == inter["y"][i], f"y[{i}]"
    assert kz.Kzg.verify_blob_kzg_proof_batch(B, [kz.Bytes48(c) for c in cs], [kz.Bytes48(q) for q in ps], settings) is True


def test_batch64_round_trip_and_oracle_agreement(kz, settings, oracle, oracle_settings):
    """The only n = 64 correctness signal the reference has is its bench (benches/kzg_benches.rs:34-41, 113-120: commit,
    prove, then unwrap() the batch verify).  Made explicit here: honest batch -> true, one swapped proof -> false, and the
    CPU oracle agrees on both; commitments / proofs of a few blobs are also compared with the oracle bit for bit."""
    blobs = [random_blob(2000 + i) for i in range(64)]
    B, cs, ps = _product_commit_prove(kz, settings, blobs)
    for i in (0, 17, 63):
        assert cs[i].to_bytes() == oracle.blob_to_kzg_commitment(blobs[i], oracle_settings)
        assert ps[i].to_bytes() == oracle.compute_blob_kzg_proof(blobs[i], cs[i].to_bytes(), oracle_settings)
    assert kz.Kzg.verify_blob_kzg_proof_batch(B, cs, ps, settings) is True
    assert oracle.verify_blob_kzg_proof_batch(blobs, [c.to_bytes() for c in cs], [p.to_bytes() for p in ps], oracle_settings) is True
    bad = list(ps); bad[40], bad[41] = bad[41], bad[40]
    assert kz.Kzg.verify_blob_kzg_proof_batch(B, cs, bad, settings) is False
    assert oracle.verify_blob_kzg_proof_batch(blobs, [c.to_bytes() for c in cs], [p.to_bytes() for p in bad], oracle_settings) is False
    # idempotence: the same call again gives the same verdicts
    assert kz.Kzg.verify_blob_kzg_proof_batch(B, cs, ps, settings) is True


def test_odd_batch_sizes(kz, settings):
    """Batch sizes that do not line up with wave / workgroup sizes (65 = one lane into a second SHA workgroup; 3n+1 GLV
    items spilling into a partial wave)."""
    blobs = [random_blob(3000 + i) for i in range(67)]
    B, cs, ps = _product_commit_prove(kz, settings, blobs)
    for n in (7, 33, 65, 67):
        assert kz.Kzg.verify_blob_kzg_proof_batch(B[:n], cs[:n], ps[:n], settings) is True
        bad = list(ps[:n]); bad[n - 1] = ps[(n - 2)]
        assert kz.Kzg.verify_blob_kzg_proof_batch(B[:n], cs[:n], bad, settings) is False


def test_skewed_blobs_msm(kz, settings, oracle, oracle_settings):
    """Digit distributions that overload single MSM buckets: a constant blob (every window puts all 4096 points in one
    bucket), all-zero, all r-1, and 31-byte-packed data (top byte 0: window 31 only sees digits 0/1)."""
    r_minus_1 = bytes.fromhex("73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000000")
    const_el = bytes.fromhex("0012ab" + "cd" * 29)
    blobs = [const_el * 4096, bytes(131072), r_minus_1 * 4096, (b"\x00" + b"\xff" * 31) * 4096,
             b"".join((i % 7).to_bytes(32, "big") for i in range(4096))]
    cs = kz.Kzg.blob_to_kzg_commitment_many([kz.Blob(b) for b in blobs], settings)
    for b, c in zip(blobs, cs):
        assert c.to_bytes() == oracle.blob_to_kzg_commitment(b, oracle_settings)
    ps = kz.Kzg.compute_blob_kzg_proof_many([kz.Blob(b) for b in blobs], cs, settings)
    for b, c, p in zip(blobs, cs, ps):
        assert p.to_bytes() == oracle.compute_blob_kzg_proof(b, c.to_bytes(), oracle_settings)
    assert kz.Kzg.verify_blob_kzg_proof_batch([kz.Blob(b) for b in blobs], cs, ps, settings) is True


def test_many_api_reports_per_unit_status(kz, settings, random_set):
    blobs, cs, ps = random_set
    bad_blob = bytearray(blobs[1]); bad_blob[64:96] = b"\xff" * 32          # element 2 >= r
    res = kz.Kzg.blob_to_kzg_commitment_many([kz.Blob(blobs[0]), kz.Blob(bytes(bad_blob)), kz.Blob(blobs[2])], settings)
    assert res[0].to_bytes() == cs[0] and isinstance(res[1], kz.BadArgs) and res[2].to_bytes() == cs[2]
    resp = kz.Kzg.compute_blob_kzg_proof_many([kz.Blob(blobs[0]), kz.Blob(blobs[1])],
                                              [kz.KzgCommitment(cs[0]), kz.KzgCommitment(bytes([0x9a]) + b"\xff" * 47)], settings)
    assert resp[0].to_bytes() == ps[0] and isinstance(resp[1], kz.BadArgs)


def test_concurrent_calls_on_one_handle(kz, settings, random_set):
    """The reference's KzgSettings is Send + Sync; the handle must serve concurrent callers (workspace pool)."""
    import threading
    blobs, cs, ps = random_set
    B, Cm, Pr = [kz.Blob(b) for b in blobs], [kz.KzgCommitment(c) for c in cs], [kz.KzgProof(p) for p in ps]
    out = [None] * 6

    def work(i):
        if i % 2 == 0:
            out[i] = kz.Kzg.verify_blob_kzg_proof_batch(B, Cm, Pr, settings)
        else:
            out[i] = kz.Kzg.blob_to_kzg_commitment(B[i], settings).to_bytes() == cs[i]
    ts = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert out == [True] * 6


def test_sharded_driver_world1_on_gpu(kz, settings, random_set):
    """kzg_rust_amd/sharded.py over the HIP engine with no process group (world = 1): same verdicts as the batch call."""
    import torch
    from kzg_rust_amd.sharded import HipEngine, verify_blob_kzg_proof_batch_sharded
    blobs, cs, ps = random_set
    n = len(blobs) // 2
    dev = torch.device("cuda", settings.device)
    bad = list(ps[n:]); bad[0], bad[1] = bad[1], bad[0]
    tb = torch.frombuffer(bytearray(b"".join(blobs[:n] + blobs[n:2 * n])), dtype=torch.uint8).to(dev)
    tc = torch.frombuffer(bytearray(b"".join(cs[:n] + cs[n:2 * n])), dtype=torch.uint8).to(dev)
    tp = torch.frombuffer(bytearray(b"".join(ps[:n] + bad)), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    ok, st = verify_blob_kzg_proof_batch_sharded(tb, tc, tp, n, 2, HipEngine(settings))
    assert ok == [True, False] and st == [0, 0]


def test_differential_verify_kzg_proof_mutations(kz, settings, oracle, oracle_settings, random_set):
    """Differential test against the oracle on mutated inputs: valid statements, wrong y / z / proof, points off the
    curve, points on the curve but outside G1, non-canonical field elements, flag-bit corruptions.  Ok/Err and the
    boolean must agree case by case."""
    import random
    from oracle.oracle import OracleError
    rnd = random.Random(99)
    blobs, cs, ps = random_set
    R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    base = []
    for i in range(4):
        z = random_field_element(40 + i)
        p, y = oracle.compute_kzg_proof(blobs[i], z, oracle_settings)
        base.append((cs[i], z, y, p))

    def mutate(t):
        c, z, y, p = t
        k = rnd.randrange(10)
        if k == 0:
            return t
        if k == 1:
            return (c, z, ((int.from_bytes(y, "big") + 1) % R).to_bytes(32, "big"), p)
        if k == 2:
            return (c, ((int.from_bytes(z, "big") + 1) % R).to_bytes(32, "big"), y, p)
        if k == 3:
            return (c, z, y, rnd.choice(ps))
        if k == 4:   # random x: usually not on the curve, sometimes on the curve but outside the subgroup
            b = bytearray(rnd.randrange(1 << 381).to_bytes(48, "big")); b[0] = (b[0] & 0x1F) | 0x80 | (0x20 if rnd.random() < .5 else 0)
            return (bytes(b), z, y, p) if rnd.random() < .5 else (c, z, y, bytes(b))
        if k == 5:
            return (c, (R + rnd.randrange(1000)).to_bytes(32, "big"), y, p)      # non-canonical z
        if k == 6:
            return (c, z, b"\xff" * 32, p)                                        # non-canonical y
        if k == 7:
            b = bytearray(c); b[0] ^= rnd.choice([0x80, 0x40, 0x20]); return (bytes(b), z, y, p)
        if k == 8:
            return (bytes([0xC0]) + bytes(47), z, y, bytes([0xC0]) + bytes(47))   # infinity / infinity
        b = bytearray(p); b[rnd.randrange(1, 48)] ^= 1 << rnd.randrange(8); return (c, z, y, bytes(b))

    n_ok = n_err = n_true = 0
    for _ in range(120):
        c, z, y, p = mutate(rnd.choice(base))
        try:
            want = oracle.verify_kzg_proof(c, z, y, p, oracle_settings)
        except OracleError:
            want = None
        try:
            got = kz.Kzg.verify_kzg_proof(kz.KzgCommitment(c), kz.Bytes32(z), kz.Bytes32(y), kz.KzgProof(p), settings)
        except kz.Error:
            got = None
        assert got == want, (c.hex(), z.hex(), y.hex(), p.hex(), got, want)
        n_ok += want is not None; n_err += want is None; n_true += want is True
    assert n_true >= 5 and n_err >= 10 and n_ok - n_true >= 10      # every outcome class was exercised


def test_bucket_lincomb_matches_windowed(kz, setup_bytes, settings, oracle, oracle_settings):
    """The two forms of the batch linear combination (per-term windowed, bucket method) must agree: honest batches verify,
    corrupted ones do not, for batch sizes around the bucket kernel's packing (8 tasks x 8 buckets per wave, 33 windows)
    and past its limit (129 falls back to the windowed form).  KZG355_LINCOMB pins the form for one settings handle."""
    g1, g2 = setup_bytes
    os.environ["KZG355_LINCOMB"] = "bucket"
    try:
        sb = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
    finally:
        del os.environ["KZG355_LINCOMB"]
    try:
        blobs = [random_blob(7000 + i) for i in range(129)]
        B, cs, ps = _product_commit_prove(kz, settings, blobs)
        for n in (8, 9, 33, 64, 128, 129):
            assert kz.Kzg.verify_blob_kzg_proof_batch(B[:n], cs[:n], ps[:n], sb) is True
            bad = list(ps[:n]); bad[n - 1] = ps[n - 2]
            assert kz.Kzg.verify_blob_kzg_proof_batch(B[:n], cs[:n], bad, sb) is False
            badc = list(cs[:n]); badc[0] = cs[1]
            assert kz.Kzg.verify_blob_kzg_proof_batch(B[:n], badc, ps[:n], sb) is False
        # points at infinity inside a bucket-form batch: the all-zero blob has C = proof = infinity (0xc0 00...), a constant blob has
        # proof = infinity; both are valid and must stay valid next to ordinary blobs
        zero = kz.Blob(bytes(131072)); const = kz.Blob((bytes(31) + b"\x05") * 4096)
        inf = bytes([0xC0]) + bytes(47)
        c_const = kz.Kzg.blob_to_kzg_commitment(const, settings)
        assert kz.Kzg.blob_to_kzg_commitment(zero, settings).to_bytes() == inf
        assert kz.Kzg.compute_blob_kzg_proof(const, c_const, settings).to_bytes() == inf
        Bi = B[:9] + [zero, const, zero]; Ci = cs[:9] + [kz.KzgCommitment(inf), c_const, kz.KzgCommitment(inf)]
        Pi = ps[:9] + [kz.KzgProof(inf), kz.KzgProof(inf), kz.KzgProof(inf)]
        assert kz.Kzg.verify_blob_kzg_proof_batch(Bi, Ci, Pi, sb) is True
        assert kz.Kzg.verify_blob_kzg_proof_batch(Bi, Ci, Pi, settings) is True
        Ci[10] = cs[0]
        assert kz.Kzg.verify_blob_kzg_proof_batch(Bi, Ci, Pi, sb) is False
        # many batches in one launch (the throughput entry point), one of them wrong
        groups = [(B[16 * g:16 * g + 16], cs[16 * g:16 * g + 16], ps[16 * g:16 * g + 16]) for g in range(8)]
        groups[5] = (groups[5][0], groups[5][1], list(reversed(groups[5][2])))
        res = kz.Kzg.verify_blob_kzg_proof_batch_many(groups, sb)
        assert [r is True for r in res] == [g != 5 for g in range(8)]
        assert oracle.verify_blob_kzg_proof_batch(blobs[:16], [c.to_bytes() for c in cs[:16]], [p.to_bytes() for p in ps[:16]], oracle_settings) is True
        # a multi-GPU sized batch (64 blobs x 8 ranks): the per-task lists no longer fit LDS and live in a global slab
        import json
        fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "batch512.json")))
        B5 = [kz.Blob(random_blob(fx["first_index"] + i)) for i in range(512)]
        C5 = [kz.KzgCommitment(bytes.fromhex(c)) for c in fx["commitments"]]; P5 = [kz.KzgProof(bytes.fromhex(p)) for p in fx["proofs"]]
        assert kz.Kzg.verify_blob_kzg_proof_batch(B5, C5, P5, sb) is True
        a, b = fx["swapped_pair"]
        P5[a], P5[b] = P5[b], P5[a]
        assert kz.Kzg.verify_blob_kzg_proof_batch(B5, C5, P5, sb) is False
    finally:
        sb.free()


def _records_of(kz, s, blobs, cs, ps):
    import torch
    n = len(blobs)
    dev = torch.device("cuda", s.device)
    t_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).to(dev)
    t_c = torch.frombuffer(bytearray(b"".join(cs)), dtype=torch.uint8).to(dev)
    t_p = torch.frombuffer(bytearray(b"".join(ps)), dtype=torch.uint8).to(dev)
    t_rec = torch.zeros(160 * n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    st = (C.c_int * 1)(-1)
    rc = kz.kzg.lib().kzg355_verify_shard_records_device(t_rec.data_ptr(), st, t_blobs.data_ptr(), t_c.data_ptr(), t_p.data_ptr(), n, 1, s.handle)
    assert rc == 0 and st[0] == 0
    return bytes(t_rec.cpu().numpy())


@pytest.mark.parametrize("form", ["1w", "2w"])
def test_challenge_kernel_forms_match_oracle(form, kz, setup_bytes, random_set, oracle, oracle_settings):
    """Both forms of the Fiat-Shamir kernel (producer/consumer pair of waves for few blobs, single wave when the card is
    full) give the oracle's z_i -- and so the same y_i -- whatever size picks between them in production.  65 + 2 blobs:
    a second, partial workgroup."""
    g1, g2 = setup_bytes
    os.environ["KZG355_CHALLENGE"] = form
    try:
        s = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
    finally:
        del os.environ["KZG355_CHALLENGE"]
    try:
        blobs, cs, ps = random_set
        reps = (65 + len(blobs) - 1) // len(blobs) + 1
        B, Cs, Ps = (blobs * reps)[:67], (cs * reps)[:67], (ps * reps)[:67]
        rec = _records_of(kz, s, B, Cs, Ps)
        inter = oracle.verify_batch_intermediates(blobs, cs, ps, oracle_settings)
        for i in range(67):
            j = i % len(blobs)
            assert rec[160 * i + 48:160 * i + 80] == inter["z"][j], f"z[{i}]"
            assert rec[160 * i + 80:160 * i + 112] == inter["y"][j], f"y[{i}]"
    finally:
        s.free()


def test_msm_bucket_form_matches_wide_table_form(kz, setup_bytes, settings, random_set, oracle, oracle_settings):
    """Commitments and proofs through every form of the fixed-base MSM equal the oracle's, for 1, 5 and all blobs per call (the launch shapes
    differ with the count): the 8-bit bucket form (what a handle falls back to when the table cannot be allocated), the GLV tables of 12-, 13-,
    15- and 16-bit windows (11 / 10 / 9 / 8 windows per 128-bit half; 16 bits: the unsigned 44,288-row top window, 143.5 GB), round 3's 256-bit
    windows (KZG355_MSM_GLV=off), and the module's default handle (table sized from the free HBM on its first commitment)."""
    import torch
    blobs, cs, ps = random_set
    B = [kz.Blob(b) for b in blobs]

    def check(s):
        for n in (1, 5, len(blobs)):
            got = kz.Kzg.blob_to_kzg_commitment_many(B[:n], s)
            assert [c.to_bytes() for c in got] == cs[:n]
            gp = kz.Kzg.compute_blob_kzg_proof_many(B[:n], [kz.KzgCommitment(c) for c in cs[:n]], s)
            assert [p.to_bytes() for p in gp] == ps[:n]
    check(settings)
    bits, windows, glv, nbytes = settings.msm_shape()
    assert glv == 1 and bits in (12, 13, 15, 16) and windows == -(-128 // bits) and settings.msm_form == bits and nbytes > 10e9
    want = {"bucket": (0, 0, 0), "12": (12, 11, 1), "13": (13, 10, 1), "15": (15, 9, 1), "16": (16, 8, 1), "glv-off-12": (12, 22, 0), "glv-off-14": (14, 19, 0)}
    free_b = torch.cuda.mem_get_info(settings.device)[0]
    for name, shape in want.items():
        if name == "16" and free_b < 165e9:
            continue                                              # (143.5 GB + the build's scratch: only on an otherwise empty card)
        env = {"KZG355_MSM": "bucket"} if name == "bucket" else {"KZG355_MSM_BITS": name.split("-")[-1]}
        if name.startswith("glv-off"):
            env["KZG355_MSM_GLV"] = "off"
        s = _handle_with_env(kz, setup_bytes, **env)
        try:
            assert s.msm_shape() == (0, 0, 0, 0)                  # not built by the load
            check(s)
            assert s.msm_shape()[:3] == shape, (name, s.msm_shape())
            assert s.msm_form == (8 if name == "bucket" else shape[0])
        finally:
            s.free()


def test_first_commitments_from_several_threads_build_the_table_once(kz, setup_bytes, random_set):
    """The MSM table is built by the first commitment / proof call on a handle (std::call_once): four threads making their first calls together on
    a fresh handle all get the oracle's bytes, the table exists afterwards, a handle that only verified has none, and kzg355_settings_build_msm_table
    builds it ahead of any call."""
    import threading
    blobs, cs, ps = random_set
    B = [kz.Blob(b) for b in blobs]
    s = _handle_with_env(kz, setup_bytes, KZG355_MSM_BITS="12")
    try:
        assert s.msm_shape() == (0, 0, 0, 0) and s.msm_form == 12
        assert kz.Kzg.verify_blob_kzg_proof_batch(B, [kz.KzgCommitment(c) for c in cs], [kz.KzgProof(p) for p in ps], s) is True
        assert s.msm_shape() == (0, 0, 0, 0)
        out, gate = [None] * 4, threading.Barrier(4)

        def work(k):
            gate.wait()
            if k % 2:
                out[k] = kz.Kzg.compute_blob_kzg_proof(B[k], kz.KzgCommitment(cs[k]), s).to_bytes()
            else:
                out[k] = kz.Kzg.blob_to_kzg_commitment(B[k], s).to_bytes()
        th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert out == [cs[0], ps[1], cs[2], ps[3]]
        assert s.msm_shape()[:3] == (12, 11, 1) and s.msm_shape()[3] == 128 * 4096 * (10 * 2048 + 256)
    finally:
        s.free()
    s = _handle_with_env(kz, setup_bytes, KZG355_MSM_BITS="13")
    try:
        s.build_msm_table()
        assert s.msm_shape()[:3] == (13, 10, 1)
        assert kz.Kzg.blob_to_kzg_commitment(B[5], s).to_bytes() == cs[5]
    finally:
        s.free()
    s = _handle_with_env(kz, setup_bytes, KZG355_VERIFY_ONLY="1")
    try:
        s.build_msm_table()                                       # nothing to build: the bucket form by request
        assert s.msm_form == 8 and s.msm_shape() == (0, 0, 0, 0)
        assert kz.Kzg.blob_to_kzg_commitment(B[6], s).to_bytes() == cs[6]
    finally:
        s.free()


@pytest.mark.parametrize("bits", [12, 13, 15, 16])
def test_glv_table_digit_extremes(bits, kz, setup_bytes, oracle, oracle_settings):
    """Scalars k = a + b x^2 whose HALVES hit the corners of the signed recoding of the GLV tables: every window of a half at 2^(bits-1)
    (digit -2^(bits-1) and a carry chain into the unsigned top window), at 2^(bits-1) - 1, alternating all-ones / 1, the largest top digit
    with a carry on top of it (a = 0xac45a400ffff...: top window 0xac45 + 1 of the 16-bit form), a = x^2 - 1, b = x^2 - 1 (k = r - 1), halves 0
    and 1 -- commitments and proofs must be the oracle's."""
    import torch
    X2 = 0xd201000000010000 ** 2
    R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    W = -(-128 // bits)
    half = 1 << (bits - 1)

    def halfval(vals):                       # little-endian window values -> a value below x^2
        v = sum(x << (bits * i) for i, x in enumerate(vals))
        return v % X2
    hs = [halfval([half] * W), halfval([half - 1] * W), halfval([2 * half - 1, 1] * W), X2 - 1, 0, 1, 0xAC45A400FFFF << 80, (0xAC45A400FFFF << 80) | ((1 << 80) - 1),
          halfval([half] * (W - 1)), halfval([0] * (W - 1) + [1]), 1 << 127, (1 << 127) - 1, halfval([half, half - 1] * W)]
    hs = [h for h in hs if h < X2]
    pats = []
    for i, a in enumerate(hs):
        for b in (hs[(i + 3) % len(hs)], hs[(2 * i + 1) % len(hs)], 0):
            k = a + b * X2
            if k < R:
                pats.append(k)
    pats += [R - 1, (X2 - 1) * X2, X2 - 1, X2, 0, 1]
    assert all(0 <= p < R for p in pats) and len(pats) > 30
    blobs = [b"".join(pats[(i + s) % len(pats)].to_bytes(32, "big") for i in range(4096)) for s in (0, 5)]
    if bits == 16 and torch.cuda.mem_get_info(0)[0] < 165e9:
        pytest.skip("143.5 GB table: only on an otherwise empty card")
    s = _handle_with_env(kz, setup_bytes, KZG355_MSM_BITS=str(bits))
    try:
        cs = kz.Kzg.blob_to_kzg_commitment_many([kz.Blob(b) for b in blobs], s)
        assert s.msm_shape()[:3] == (bits, W, 1)
        for b, c in zip(blobs, cs):
            assert c.to_bytes() == oracle.blob_to_kzg_commitment(b, oracle_settings)
        ps = kz.Kzg.compute_blob_kzg_proof_many([kz.Blob(b) for b in blobs], cs, s)
        for b, c, p in zip(blobs, cs, ps):
            assert p.to_bytes() == oracle.compute_blob_kzg_proof(b, c.to_bytes(), oracle_settings)
        one = kz.Kzg.blob_to_kzg_commitment(kz.Blob(blobs[0]), s)              # the lone-blob launch shape (two window parts)
        assert one.to_bytes() == cs[0].to_bytes()
    finally:
        s.free()


@pytest.mark.parametrize("n", [64, 512])
def test_batch_fixtures_stage_by_stage(n, kz, settings):
    """The committed n = 64 / n = 512 batches (tests/golden/batch{n}.json, oracle-derived from the seeded bench recipe; SURVEY 8c
    item 3): the product reproduces every commitment and proof bit for bit, the stage-1 records carry the fixture's z_i and
    y_i, the batch verifies, the swapped twin does not -- and the same 512 blobs as eight batches of 64 verify in one launch."""
    import json
    import torch
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", f"batch{n}.json")))
    first = fx["first_index"]
    blobs = [random_blob(first + i) for i in range(n)]
    B = [kz.Blob(b) for b in blobs]
    cs = kz.Kzg.blob_to_kzg_commitment_many(B, settings)
    assert [c.to_bytes().hex() for c in cs] == fx["commitments"]
    ps = kz.Kzg.compute_blob_kzg_proof_many(B, cs, settings)
    assert [p.to_bytes().hex() for p in ps] == fx["proofs"]
    rec = _records_of(kz, settings, blobs, [c.to_bytes() for c in cs], [p.to_bytes() for p in ps])
    for i in range(n):
        assert rec[160 * i + 48:160 * i + 80].hex() == fx["z"][i], f"z[{i}]"
        assert rec[160 * i + 80:160 * i + 112].hex() == fx["y"][i], f"y[{i}]"
    assert kz.Kzg.verify_blob_kzg_proof_batch(B, cs, ps, settings) is fx["expect"]
    a, b = fx["swapped_pair"]
    sw = list(ps); sw[a], sw[b] = sw[b], sw[a]
    assert kz.Kzg.verify_blob_kzg_proof_batch(B, cs, sw, settings) is fx["expect_swapped"]
    if n == 512:
        groups = [(B[64 * g:64 * g + 64], cs[64 * g:64 * g + 64], ps[64 * g:64 * g + 64]) for g in range(8)]
        assert kz.Kzg.verify_blob_kzg_proof_batch_many(groups, settings) == [True] * 8
        groups[a // 64] = (groups[a // 64][0], groups[a // 64][1], sw[64 * (a // 64):64 * (a // 64) + 64])
        assert kz.Kzg.verify_blob_kzg_proof_batch_many(groups, settings) == [g != a // 64 for g in range(8)]
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_differential_blob_proof_commitment_mutations(kz, settings, oracle, oracle_settings, random_set):
    """compute_blob_kzg_proof validates its commitment (kzg.rs:321-323, utils.rs:282-310) before it hashes it.  The product walks the subgroup
    ladder from x alone beside the square root of the decoding (k_subgroup_ladder_from_x_quad + k_subgroup_finish): honest commitments, the other
    square root (-C: a valid point, another challenge), infinity, random x (off the curve, or on it and outside G1), corrupted flag bits and
    bytes -- the 48 proof bytes or the Err must be the oracle's, case by case."""
    import random
    from oracle.oracle import OracleError
    rnd = random.Random(4844)
    blobs, cs, ps = random_set
    n_ok = n_err = 0
    for it in range(60):
        i = rnd.randrange(len(blobs))
        c = bytearray(cs[i])
        k = it % 6
        if k == 1:
            c[0] ^= 0x20                                            # the other square root
        elif k == 2:
            c = bytearray(bytes([0xC0]) + bytes(47))                # infinity
        elif k == 3:                                                # random x: half of them on the curve, practically none of those in G1
            c = bytearray(rnd.randrange(1 << 381).to_bytes(48, "big")); c[0] = (c[0] & 0x1F) | 0x80 | (0x20 if rnd.random() < .5 else 0)
        elif k == 4:
            c[0] ^= rnd.choice([0x80, 0x40])                        # compression / infinity flag
        elif k == 5:
            c[rnd.randrange(1, 48)] ^= 1 << rnd.randrange(8)
        c = bytes(c)
        try:
            want = oracle.compute_blob_kzg_proof(blobs[i], c, oracle_settings)
        except OracleError:
            want = None
        try:
            got = kz.Kzg.compute_blob_kzg_proof(kz.Blob(blobs[i]), kz.KzgCommitment(c), settings).to_bytes()
        except kz.Error:
            got = None
        assert got == want, (it, k, c.hex())
        n_ok += want is not None; n_err += want is None
    assert n_ok >= 15 and n_err >= 15                               # both outcomes well represented


def test_differential_batch_mutations(kz, settings, oracle, oracle_settings, random_set):
    """Differential test of verify_blob_kzg_proof_batch against the oracle on mutated batches of 1..6 blobs: honest, a field
    element of a blob pushed to r / r-1 / 2^256-1, a blob byte flipped, proofs swapped, a commitment replaced by another
    blob's, by infinity, by an off-curve / out-of-subgroup x, flag bits corrupted.  Ok(true) / Ok(false) / Err must agree."""
    import random
    from oracle.oracle import OracleError
    rnd = random.Random(4844)
    blobs, cs, ps = random_set
    R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001

    def case():
        n = rnd.randrange(1, 7)
        idx = rnd.sample(range(len(blobs)), n)
        B, Cm, Pr = [blobs[i] for i in idx], [cs[i] for i in idx], [ps[i] for i in idx]
        k = rnd.randrange(10)
        j = rnd.randrange(n)
        if k == 1:
            v = rnd.choice([R, R - 1, (1 << 256) - 1, R + 5])
            b = bytearray(B[j]); e = rnd.randrange(4096); b[32 * e:32 * e + 32] = v.to_bytes(32, "big"); B[j] = bytes(b)
        elif k == 2:
            b = bytearray(B[j]); b[rnd.randrange(131072) | 1] ^= 1 << rnd.randrange(8); B[j] = bytes(b)   # odd offsets: never the zeroed top byte
        elif k == 3 and n > 1:
            a = (j + 1) % n; Pr[j], Pr[a] = Pr[a], Pr[j]
        elif k == 4:
            Cm[j] = cs[(idx[j] + 1) % len(cs)]
        elif k == 5:
            Cm[j] = bytes([0xC0]) + bytes(47)
        elif k == 6:
            x = bytearray(rnd.randrange(1 << 381).to_bytes(48, "big")); x[0] = (x[0] & 0x1F) | 0x80 | (0x20 if rnd.random() < .5 else 0)
            if rnd.random() < .5: Cm[j] = bytes(x)
            else: Pr[j] = bytes(x)
        elif k == 7:
            x = bytearray(Pr[j]); x[0] ^= rnd.choice([0x80, 0x40, 0x20]); Pr[j] = bytes(x)
        elif k == 8:
            x = bytearray(Pr[j]); x[rnd.randrange(1, 48)] ^= 1 << rnd.randrange(8); Pr[j] = bytes(x)
        return B, Cm, Pr

    seen = {True: 0, False: 0, None: 0}
    for _ in range(60):
        B, Cm, Pr = case()
        try:
            want = oracle.verify_blob_kzg_proof_batch(B, Cm, Pr, oracle_settings)
        except OracleError:
            want = None
        try:
            got = kz.Kzg.verify_blob_kzg_proof_batch([kz.Blob(b) for b in B], [kz.KzgCommitment(c) for c in Cm], [kz.KzgProof(p) for p in Pr], settings)
        except kz.Error:
            got = None
        assert got == want, (len(B), got, want)
        seen[want] += 1
    assert seen[True] >= 3 and seen[False] >= 8 and seen[None] >= 8, seen


def test_msm_launch_shapes_agree(kz, settings, random_set, setup_bytes):
    """The wide-table MSM picks its launch shape by blob count (1 or 2 window parts, 1 / 4 / 16 scalars per lane): 1024 blobs in
    one call (16 scalars per lane, one workgroup per blob) must give the commitments and proofs of the same blobs sent in
    calls of 64 and of 200 (other shapes), and the first ones must be the oracle's (random_set)."""
    import torch
    blobs, cs, ps = random_set
    n = 1024
    dev = torch.device("cuda", settings.device)
    reps = (n + len(blobs) - 1) // len(blobs)
    tb = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).to(dev).repeat(reps)[:n * 131072].contiguous()
    g = torch.Generator(device="cpu"); g.manual_seed(7)
    noise = torch.randint(0, 256, (n * 131072,), dtype=torch.uint8, generator=g).to(dev)
    noise.view(n * 4096, 32)[:, 0] = 0                          # keep every element canonical
    tb[len(blobs) * 131072:] = noise[len(blobs) * 131072:]      # first len(blobs) blobs stay the oracle-checked ones
    torch.cuda.synchronize()                                    # torch's stream is not ordered against the library's own streams
    L = kz.kzg.lib()

    def commit(lo, hi):
        out = C.create_string_buffer(48 * (hi - lo)); st = (C.c_int * (hi - lo))()
        assert L.kzg355_blob_to_kzg_commitment_many_device(out, st, tb.data_ptr() + lo * 131072, hi - lo, settings.handle) == 0
        return out.raw

    def prove(lo, hi, c):
        tc = torch.frombuffer(bytearray(c), dtype=torch.uint8).to(dev); torch.cuda.synchronize()
        out = C.create_string_buffer(48 * (hi - lo)); st = (C.c_int * (hi - lo))()
        assert L.kzg355_compute_blob_kzg_proof_many_device(out, st, tb.data_ptr() + lo * 131072, tc.data_ptr(), hi - lo, settings.handle) == 0
        return out.raw

    c_all = commit(0, n)
    assert [c_all[48 * i:48 * i + 48] for i in range(len(blobs))] == cs
    assert b"".join(commit(lo, lo + 64) for lo in range(0, n, 64)) == c_all
    assert b"".join(commit(lo, min(lo + 200, n)) for lo in range(0, n, 200)) == c_all
    p_all = prove(0, n, c_all)
    assert [p_all[48 * i:48 * i + 48] for i in range(len(blobs))] == ps
    assert b"".join(prove(lo, lo + 64, c_all[48 * lo:48 * (lo + 64)]) for lo in range(0, n, 64)) == p_all
    # and the independent 8-bit bucket kernels give the same 1024 commitments and proofs
    g1, g2 = setup_bytes
    os.environ["KZG355_MSM"] = "bucket"
    try:
        sb = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
    finally:
        del os.environ["KZG355_MSM"]
    try:
        out = C.create_string_buffer(48 * n); st = (C.c_int * n)()
        assert L.kzg355_blob_to_kzg_commitment_many_device(out, st, tb.data_ptr(), n, sb.handle) == 0 and out.raw == c_all
        tc = torch.frombuffer(bytearray(c_all), dtype=torch.uint8).to(dev); torch.cuda.synchronize()
        assert L.kzg355_compute_blob_kzg_proof_many_device(out, st, tb.data_ptr(), tc.data_ptr(), n, sb.handle) == 0 and out.raw == p_all
    finally:
        sb.free()


def test_wide_table_digit_extremes(kz, setup_bytes, oracle, oracle_settings):
    """(Round 3's table form, KZG355_MSM_GLV=off: windows over the 256 bits of the scalar.)  Scalars built to hit the corners of the signed 12-bit recoding of the wide-table MSM: every window 0x800 (digit -2048 with a
    carry chain through all 22 windows), 0x7ff (largest positive digit), 0xfff / 0x001 alternating, a lone top-window bit, and
    r - 1 next to 0 and 1 -- commitments and proofs must be the oracle's."""
    R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    def windows(vals):                       # little-endian list of 12-bit window values -> scalar reduced below r
        return sum(v << (12 * i) for i, v in enumerate(vals)) % R
    pats = [windows([0x800] * 21), windows([0x7FF] * 21), windows([0xFFF, 0x001] * 10 + [0xFFF]), 1 << 252, (1 << 252) - 1,
            R - 1, 0, 1, windows([0x800] * 20 + [0x7FF]), windows([0] * 20 + [0x800]), (1 << 254) + (1 << 12) - 1]
    assert all(0 <= p < R for p in pats)
    blobs = [b"".join(pats[(i + s) % len(pats)].to_bytes(32, "big") for i in range(4096)) for s in (0, 3)]
    blobs.append(pats[0].to_bytes(32, "big") * 4096)
    settings = _handle_with_env(kz, setup_bytes, KZG355_MSM_BITS="12", KZG355_MSM_GLV="off")
    try:
        cs = kz.Kzg.blob_to_kzg_commitment_many([kz.Blob(b) for b in blobs], settings)
        assert settings.msm_shape()[:3] == (12, 22, 0)
        for b, c in zip(blobs, cs):
            assert c.to_bytes() == oracle.blob_to_kzg_commitment(b, oracle_settings)
        ps = kz.Kzg.compute_blob_kzg_proof_many([kz.Blob(b) for b in blobs], cs, settings)
        for b, c, p in zip(blobs, cs, ps):
            assert p.to_bytes() == oracle.compute_blob_kzg_proof(b, c.to_bytes(), oracle_settings)
        one = kz.Kzg.blob_to_kzg_commitment(kz.Blob(blobs[0]), settings)          # the lone-blob launch shape (two window parts)
        assert one.to_bytes() == cs[0].to_bytes()
    finally:
        settings.free()


def test_launch_shape_sweep(kz, settings):
    """Odd launch sizes through the device entry points: batch sizes 1..1600 x batch counts 1..200 (both lincomb forms, both hash
    forms, every MSM shape).  Honest batches verify; swapping two proofs inside the LAST batch turns exactly that verdict false;
    commitments / proofs of a prefix do not depend on how many blobs the call carries."""
    import torch
    L = kz.kzg.lib(); dev = torch.device("cuda", settings.device)
    NB = 1600
    gen = torch.Generator(device=dev); gen.manual_seed(99)
    tb = torch.randint(0, 256, (NB, 4096, 32), dtype=torch.uint8, device=dev, generator=gen); tb[:, :, 0] = 0
    tb = tb.reshape(-1).contiguous()
    torch.cuda.synchronize()                                    # (torch's stream is not ordered against the library's own streams: here and below)
    out = C.create_string_buffer(48 * NB); st = (C.c_int * NB)()
    assert L.kzg355_blob_to_kzg_commitment_many_device(out, st, tb.data_ptr(), NB, settings.handle) == 0
    cs = out.raw; tc = torch.frombuffer(bytearray(cs), dtype=torch.uint8).to(dev); torch.cuda.synchronize()
    assert L.kzg355_compute_blob_kzg_proof_many_device(out, st, tb.data_ptr(), tc.data_ptr(), NB, settings.handle) == 0
    ps = out.raw; tp = torch.frombuffer(bytearray(ps), dtype=torch.uint8).to(dev); torch.cuda.synchronize()
    for n in (1, 2, 15, 16, 17, 127, 128, 129, 1023, 1024, 1025):
        o2 = C.create_string_buffer(48 * n)
        assert L.kzg355_blob_to_kzg_commitment_many_device(o2, st, tb.data_ptr(), n, settings.handle) == 0 and o2.raw == cs[:48 * n], n
        assert L.kzg355_compute_blob_kzg_proof_many_device(o2, st, tb.data_ptr(), tc.data_ptr(), n, settings.handle) == 0 and o2.raw == ps[:48 * n], n
    for npg, G in ((1, 1), (1, 70), (2, 33), (3, 65), (7, 200), (8, 64), (9, 63), (31, 51), (64, 1), (64, 2), (64, 3), (64, 25), (65, 24),
                   (100, 16), (127, 12), (128, 12), (129, 12), (200, 8), (512, 3), (1600, 1)):
        ok = (C.c_bool * G)(); stg = (C.c_int * G)()
        rc = L.kzg355_verify_blob_kzg_proof_batch_many_device(ok, stg, tb.data_ptr(), tc.data_ptr(), tp.data_ptr(), npg, G, settings.handle)
        assert rc == 0 and all(ok[i] for i in range(G)) and not any(stg[i] for i in range(G)), (npg, G, rc)
        if npg > 1:
            bad = tp.clone(); j = (npg * G - 1) * 48; k = (npg * (G - 1)) * 48
            tmp = bad[j:j + 48].clone(); bad[j:j + 48] = bad[k:k + 48]; bad[k:k + 48] = tmp
            torch.cuda.synchronize()
            rc = L.kzg355_verify_blob_kzg_proof_batch_many_device(ok, stg, tb.data_ptr(), tc.data_ptr(), bad.data_ptr(), npg, G, settings.handle)
            assert rc == 0 and [ok[i] for i in range(G)] == [True] * (G - 1) + [False], (npg, G)


# ---------------------------------------------------------------------------------------------------------------------
# Stage 2 pinned value by value: the Fiat-Shamir batch challenge r (utils.rs:426-474), proof_lincomb
# (kzg.rs:601) and rhs (kzg.rs:618-622) read back through kzg355_debug_batch_intermediates.  A consistently wrong
# transcript (domain string, byte order, n encoding, a dropped field) keeps every boolean right; these tests do not.
def _stage2_dump(kz, s, rec, n, groups=1):
    import torch
    dev = torch.device("cuda", s.device)
    t_rec = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    out = C.create_string_buffer(128 * groups)
    ok = (C.c_bool * groups)(); st = (C.c_int * groups)()
    rc = kz.kzg.lib().kzg355_debug_batch_intermediates(out, ok, st, t_rec.data_ptr(), n, groups, s.handle)
    assert rc == 0 and not any(st), (rc, list(st))
    d = out.raw
    return [{"r": d[128 * g:128 * g + 32], "proof_lincomb": d[128 * g + 32:128 * g + 80], "rhs": d[128 * g + 80:128 * g + 128], "ok": bool(ok[g])}
            for g in range(groups)]


@pytest.fixture(scope="module")
def lincomb_handles(kz, setup_bytes):
    """One handle per form of the batch linear combination (KZG355_LINCOMB pins it when the handle is created)."""
    g1, g2 = setup_bytes
    hs = {}
    for form in ("window", "bucket", "preshift"):
        os.environ["KZG355_LINCOMB"] = form
        try:
            hs[form] = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
        finally:
            del os.environ["KZG355_LINCOMB"]
    # and one that takes the many-batch kernels whatever the batch count (by default from 1024 and 6144 batches on): r-transcripts hashed one
    # lane per batch (k_rhash_lanes), bucket form ending in one Horner chain per class (k_lc_wsum + k_lc_hchain_quad)
    os.environ["KZG355_RHASH_LANES_FROM"] = "1"; os.environ["KZG355_LC_CHAIN_FROM"] = "1"; os.environ["KZG355_LINCOMB"] = "bucket"
    try:
        hs["rhash-lanes"] = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
    finally:
        del os.environ["KZG355_RHASH_LANES_FROM"]; del os.environ["KZG355_LC_CHAIN_FROM"]; del os.environ["KZG355_LINCOMB"]
    # the batch challenge r of a lone small call is hashed on the HOST by default (records copied back, verify_stages.hip run_stage2): this
    # handle keeps the device transcript hash (k_rpowers) for every size, so that both routes meet the same fixtures
    os.environ["KZG355_HOST_RHASH"] = "off"
    try:
        hs["device-rhash"] = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
    finally:
        del os.environ["KZG355_HOST_RHASH"]
    yield hs
    for h in hs.values():
        h.free()


@pytest.mark.parametrize("n", [64, 512])
def test_stage2_intermediates_match_fixtures(n, kz, settings, lincomb_handles):
    """tests/golden/batch{n}.json carries r, proof_lincomb and rhs of the seeded n-blob batch (oracle-derived, cross-checked by
    oracle/pyref.py when the fixture was made): the HIP path must reproduce all three byte for byte, in both lincomb forms."""
    import json
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", f"batch{n}.json")))
    blobs = [random_blob(fx["first_index"] + i) for i in range(n)]
    cs = [bytes.fromhex(c) for c in fx["commitments"]]; ps = [bytes.fromhex(p) for p in fx["proofs"]]
    rec = _records_of(kz, settings, blobs, cs, ps)
    for name, s in [("auto", settings)] + list(lincomb_handles.items()):
        d = _stage2_dump(kz, s, rec, n)[0]
        assert d["r"].hex() == fx["r"], name
        assert d["proof_lincomb"].hex() == fx["proof_lincomb"], name
        assert d["rhs"].hex() == fx["rhs"], name
        assert d["ok"] is True
    # the swapped twin: same r recipe, different transcript -> different r, and the verdict flips
    a, b = fx["swapped_pair"]
    sw = list(ps); sw[a], sw[b] = sw[b], sw[a]
    d = _stage2_dump(kz, settings, _records_of(kz, settings, blobs, cs, sw), n)[0]
    assert d["r"].hex() != fx["r"] and d["ok"] is False


def test_stage2_intermediates_match_oracle(kz, settings, lincomb_handles, oracle, oracle_settings):
    """r / proof_lincomb / rhs against oracle.verify_batch_intermediates for n in {2, 6, 9, 65} (windowed and bucket forms; 65 =
    a second SHA chunk of the r-transcript and a partial second wave of powers), and for 520 batches in one launch (the
    four-waves-per-workgroup shape of k_rpowers), one of them corrupted."""
    blobs = [random_blob(9000 + i) for i in range(65)]
    B, cs, ps = _product_commit_prove(kz, settings, blobs)
    cs = [c.to_bytes() for c in cs]; ps = [p.to_bytes() for p in ps]
    for n in (2, 6, 9, 65):
        want = oracle.verify_batch_intermediates(blobs[:n], cs[:n], ps[:n], oracle_settings)
        assert want["ok"] is True
        rec = _records_of(kz, settings, blobs[:n], cs[:n], ps[:n])
        for name, s in [("auto", settings)] + list(lincomb_handles.items()):
            d = _stage2_dump(kz, s, rec, n)[0]
            assert (d["r"], d["proof_lincomb"], d["rhs"], d["ok"]) == (want["r"], want["proof_lincomb"], want["rhs"], True), (n, name)
        # a wrong-but-valid proof: every intermediate still has to be the oracle's (the boolean alone would also flip on garbage)
        bad = list(ps[:n]); bad[n - 1] = ps[n - 2]
        wantb = oracle.verify_batch_intermediates(blobs[:n], cs[:n], bad, oracle_settings)
        d = _stage2_dump(kz, settings, _records_of(kz, settings, blobs[:n], cs[:n], bad), n)[0]
        assert (d["r"], d["proof_lincomb"], d["rhs"], d["ok"]) == (wantb["r"], wantb["proof_lincomb"], wantb["rhs"], False), n
    # points at infinity among the terms (the zero blob: commitment = proof = infinity, kzg.rs:299-301) and a repeated blob (equal
    # points meet in one bucket: the doubling case of the accumulation), through every form
    inf = b"\xc0" + bytes(47)
    zb = [blobs[0], bytes(131072), blobs[1], blobs[1], bytes(131072), blobs[2], blobs[3], blobs[1], blobs[4]]
    zc = [cs[0], inf, cs[1], cs[1], inf, cs[2], cs[3], cs[1], cs[4]]
    zp = [ps[0], inf, ps[1], ps[1], inf, ps[2], ps[3], ps[1], ps[4]]
    want = oracle.verify_batch_intermediates(zb, zc, zp, oracle_settings)
    assert want["ok"] is True
    rec = _records_of(kz, settings, zb, zc, zp)
    for name, s in [("auto", settings)] + list(lincomb_handles.items()):
        d = _stage2_dump(kz, s, rec, len(zb))[0]
        assert (d["r"], d["proof_lincomb"], d["rhs"], d["ok"]) == (want["r"], want["proof_lincomb"], want["rhs"], True), name
    n, G = 6, 520
    rec = _records_of(kz, settings, blobs[:n], cs[:n], ps[:n])
    rec2 = _records_of(kz, settings, blobs[n:2 * n], cs[n:2 * n], ps[n:2 * n])
    w1 = oracle.verify_batch_intermediates(blobs[:n], cs[:n], ps[:n], oracle_settings)
    w2 = oracle.verify_batch_intermediates(blobs[n:2 * n], cs[n:2 * n], ps[n:2 * n], oracle_settings)
    many = b"".join(rec2 if g % 7 == 3 else rec for g in range(G))
    for s in (settings, lincomb_handles["bucket"], lincomb_handles["rhash-lanes"]):
        ds = _stage2_dump(kz, s, many, n, G)
        for g in range(G):
            w = w2 if g % 7 == 3 else w1
            assert (ds[g]["r"], ds[g]["proof_lincomb"], ds[g]["rhs"], ds[g]["ok"]) == (w["r"], w["proof_lincomb"], w["rhs"], True), g


def test_verify_records_checked_rejects_what_stage1_would(kz, settings, random_set):
    """kzg355_verify_records_checked_device = verify_kzg_proof_batch on untrusted records: off-subgroup points and non-canonical
    z / y are Err there, while the unchecked stage-2 entry point (precondition: stage-1 status merged by the caller) is not
    required to notice.  Honest records verify through both."""
    import torch
    blobs, cs, ps = random_set
    n = len(blobs)
    rec = bytearray(_records_of(kz, settings, blobs, cs, ps))
    L = kz.kzg.lib(); dev = torch.device("cuda", settings.device)

    def run(fn, r):
        t = torch.frombuffer(bytearray(r), dtype=torch.uint8).to(dev); torch.cuda.synchronize()
        ok = (C.c_bool * 1)(); st = (C.c_int * 1)()
        rc = fn(ok, st, t.data_ptr(), n, 1, settings.handle)
        return rc, bool(ok[0]), st[0]
    assert run(L.kzg355_verify_records_checked_device, rec) == (0, True, 0)
    assert run(L.kzg355_verify_records_device, rec) == (0, True, 0)
    bad = bytearray(rec); bad[160 * 2 + 48:160 * 2 + 80] = b"\xff" * 32             # z_2 >= r
    rc, _, st = run(L.kzg355_verify_records_checked_device, bad)
    assert rc == 1 and st == 1
    # x = 0x...cde0: on the curve, outside G1 (the reference's own not_in_G1 vector value)
    off = bytes.fromhex("8123456789abcdef0123456789abcdef0123456789abcdef0123456789abcdef0123456789abcdef0123456789abcde0")
    bad = bytearray(rec); bad[160 * 3:160 * 3 + 48] = off
    rc, _, st = run(L.kzg355_verify_records_checked_device, bad)
    assert rc == 1 and st == 1


def test_invalid_blobs_on_the_bucket_msm_handle(kz, setup_bytes, golden_vectors, golden_blobs, backend):
    """The reference's invalid-blob vectors (non-canonical field elements, top byte >= 0x81 among them) on a KZG355_MSM=bucket
    handle: Err as on the table path, valid blobs after them still give the right commitments (no out-of-range bucket index)."""
    g1, g2 = setup_bytes
    os.environ["KZG355_MSM"] = "bucket"
    try:
        sb = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
    finally:
        del os.environ["KZG355_MSM"]
    try:
        for fn in ("blob_to_kzg_commitment", "compute_blob_kzg_proof"):
            n, failures = run_function(fn, golden_vectors, backend, sb, golden_blobs)
            assert n == COUNTS[fn] and not failures, failures
        hi = bytearray(random_blob(123)); hi[32 * 77:32 * 77 + 32] = bytes([0x90]) + bytes(31)      # digit 0x90 in the top window
        good = random_blob(124)
        res = kz.Kzg.blob_to_kzg_commitment_many([kz.Blob(bytes(hi)), kz.Blob(good), kz.Blob(b"\xff" * 131072)], sb)
        assert isinstance(res[0], kz.BadArgs) and isinstance(res[2], kz.BadArgs)
        assert res[1].to_bytes() == kz.Kzg.blob_to_kzg_commitment(kz.Blob(good), sb).to_bytes()
    finally:
        sb.free()


def _handle_with_env(kz, setup_bytes, **env):
    g1, g2 = setup_bytes
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.parametrize("staging", ["direct", "ring"])
def test_host_pipeline_chunks_keep_order_and_status(staging, kz, setup_bytes, settings, random_set):
    """Host-buffer *_many calls cut into many chunks (KZG355_CHUNK_MB=1: one 8-blob batch per chunk, three workspaces in flight), in both
    staging modes: verdicts and per-unit statuses come back in call order, an Err in a middle chunk neither stops nor reorders the
    others, and commitments / proofs of a chunked call equal the unchunked ones."""
    blobs, cs, ps = random_set
    s = _handle_with_env(kz, setup_bytes, KZG355_CHUNK_MB="1", KZG355_STAGING=staging, KZG355_MSM="bucket")
    try:
        B, Cm, Pr = [kz.Blob(b) for b in blobs], [kz.KzgCommitment(c) for c in cs], [kz.KzgProof(p) for p in ps]
        sw = list(Pr); sw[0], sw[1] = sw[1], sw[0]
        bad_c = list(Cm); bad_c[5] = kz.KzgCommitment(bytes([0x9a]) + b"\xff" * 47)
        bb = bytearray(blobs[2]); bb[32 * 100:32 * 101] = b"\xff" * 32
        bad_b = list(B); bad_b[2] = kz.Blob(bytes(bb))
        groups, want = [], []
        for g in range(11):
            if g == 3:
                groups.append((B, Cm, sw)); want.append(False)
            elif g == 6:
                groups.append((B, bad_c, Pr)); want.append("err")
            elif g == 9:
                groups.append((bad_b, Cm, Pr)); want.append("err")
            else:
                groups.append((B, Cm, Pr)); want.append(True)
        res = kz.Kzg.verify_blob_kzg_proof_batch_many(groups, s)
        assert [("err" if isinstance(r, kz.Error) else r) for r in res] == want
        many = [kz.Blob(random_blob(8800 + i)) for i in range(21)]               # 21 blobs at 8 per chunk: 3 chunks, the last one short
        many[13] = kz.Blob(bytes(bb))
        ref_c = kz.Kzg.blob_to_kzg_commitment_many(many, settings)
        got_c = kz.Kzg.blob_to_kzg_commitment_many(many, s)
        assert [isinstance(x, kz.Error) for x in got_c] == [i == 13 for i in range(21)]
        assert [x.to_bytes() for i, x in enumerate(got_c) if i != 13] == [x.to_bytes() for i, x in enumerate(ref_c) if i != 13]
        cm_ok = [ref_c[i] if i != 13 else ref_c[0] for i in range(21)]
        ref_p = kz.Kzg.compute_blob_kzg_proof_many(many, cm_ok, settings)
        got_p = kz.Kzg.compute_blob_kzg_proof_many(many, cm_ok, s)
        assert [isinstance(x, kz.Error) for x in got_p] == [i == 13 for i in range(21)]
        assert [x.to_bytes() for i, x in enumerate(got_p) if i != 13] == [x.to_bytes() for i, x in enumerate(ref_p) if i != 13]
    finally:
        s.free()


def test_overlapped_launch_sets_keep_order(kz, setup_bytes, random_set):
    """Device-resident verify calls run as several launch sets on several streams (default: two sets from 262,144 blobs; here
    KZG355_SPLIT=3,2 on a small call): 13 batches dealt 5 + 4 + 4 over two streams, two sets queued on one workspace -- verdicts and
    statuses must land at their batch's index."""
    import torch
    blobs, cs, ps = random_set
    n = len(blobs)
    s = _handle_with_env(kz, setup_bytes, KZG355_SPLIT="3,2", KZG355_MSM="bucket")
    try:
        dev = torch.device("cuda", s.device)
        G = 13
        P = [list(ps) for _ in range(G)]; Cc = [list(cs) for _ in range(G)]
        P[4][0], P[4][1] = P[4][1], P[4][0]                                   # false, first set
        P[7][2], P[7][3] = P[7][3], P[7][2]                                   # false, second set
        Cc[11][1] = bytes([0x9a]) + b"\xff" * 47                               # Err, third set (second one on its stream)
        tb = torch.frombuffer(bytearray(b"".join(blobs) * G), dtype=torch.uint8).to(dev)
        tc = torch.frombuffer(bytearray(b"".join(b"".join(c) for c in Cc)), dtype=torch.uint8).to(dev)
        tp = torch.frombuffer(bytearray(b"".join(b"".join(p) for p in P)), dtype=torch.uint8).to(dev)
        torch.cuda.synchronize()
        ok = (C.c_bool * G)(); st = (C.c_int * G)()
        rc = kz.kzg.lib().kzg355_verify_blob_kzg_proof_batch_many_device(ok, st, tb.data_ptr(), tc.data_ptr(), tp.data_ptr(), n, G, s.handle)
        assert rc == 1
        assert [st[g] for g in range(G)] == [1 if g == 11 else 0 for g in range(G)]
        assert [ok[g] for g in range(G) if g != 11] == [g not in (4, 7) for g in range(G) if g != 11]
    finally:
        s.free()


@pytest.mark.parametrize("mode", ["by-size", "pipeline", "sets"])
def test_submit_collect_keeps_sets_apart(mode, kz, setup_bytes, random_set):
    """kzg355_verify_blob_kzg_proof_batch_many_device_submit / kzg355_verify_collect: one thread keeps four launch sets in flight on four
    streams of the handle (sets of 5, 3, 1 and 4 batches with a false verdict, an Err and an honest set among them) and collects them out of
    order: every verdict and status lands at its set's batch index; an empty set and a zero-blob set answer without a device; a misaligned
    pointer is refused at submit; the synchronous call agrees with the collected verdicts."""
    import torch
    blobs, cs, ps = random_set
    n = len(blobs)
    # by-size: these small sets go each to the stream of its own workspace; pipeline: the two-stage software pipeline large sets take (stage 2 of a
    # set queued behind the next set's hash, or by its collect), forced here on small ones; sets: form 1 forced
    settings = _handle_with_env(kz, setup_bytes, KZG355_VERIFY_ONLY="1", **({} if mode == "by-size" else {"KZG355_SUBMIT": mode}))
    try:
        _submit_collect_body(kz, settings, blobs, cs, ps, n)
    finally:
        settings.free()


def _submit_collect_body(kz, settings, blobs, cs, ps, n):
    import torch
    L = kz.kzg.lib(); dev = torch.device("cuda", settings.device)
    sizes = [5, 3, 1, 4]
    sets = []
    for k, G in enumerate(sizes):
        P = [list(ps) for _ in range(G)]; Cc = [list(cs) for _ in range(G)]
        expect_ok, expect_st = [True] * G, [0] * G
        if k == 0:
            P[3][0], P[3][1] = P[3][1], P[3][0]; expect_ok[3] = False
        if k == 1:
            Cc[2][5] = bytes([0x9a]) + b"\xff" * 47; expect_st[2] = 1
        if k == 3:
            P[0][6], P[0][7] = P[0][7], P[0][6]; expect_ok[0] = False
        tb = torch.frombuffer(bytearray(b"".join(blobs) * G), dtype=torch.uint8).to(dev)
        tc = torch.frombuffer(bytearray(b"".join(b"".join(c) for c in Cc)), dtype=torch.uint8).to(dev)
        tp = torch.frombuffer(bytearray(b"".join(b"".join(p) for p in P)), dtype=torch.uint8).to(dev)
        sets.append((G, tb, tc, tp, expect_ok, expect_st))
    torch.cuda.synchronize()
    tickets = []
    for G, tb, tc, tp, _, _ in sets:
        tk = C.c_void_p()
        assert L.kzg355_verify_blob_kzg_proof_batch_many_device_submit(C.byref(tk), tb.data_ptr(), tc.data_ptr(), tp.data_ptr(), n, G, settings.handle) == 0
        assert tk.value
        tickets.append(tk)
    for k in (2, 0, 3, 1):                                                    # out of order
        G, tb, tc, tp, expect_ok, expect_st = sets[k]
        ok = (C.c_bool * G)(); st = (C.c_int * G)()
        rc = L.kzg355_verify_collect(tickets[k], ok, st)
        assert rc == (1 if any(expect_st) else 0), (k, rc)
        assert [st[g] for g in range(G)] == expect_st, k
        assert [ok[g] for g in range(G) if not expect_st[g]] == [e for e, f in zip(expect_ok, expect_st) if not f], k
        ok2 = (C.c_bool * G)(); st2 = (C.c_int * G)()
        L.kzg355_verify_blob_kzg_proof_batch_many_device(ok2, st2, tb.data_ptr(), tc.data_ptr(), tp.data_ptr(), n, G, settings.handle)
        assert list(st2) == list(st) and [ok2[g] for g in range(G) if not st[g]] == [ok[g] for g in range(G) if not st[g]]
    # nothing to queue: no batches, or batches of no blobs (kzg.rs:653-655: Ok(true))
    G, tb, tc, tp, _, _ = sets[0]
    tk = C.c_void_p()
    assert L.kzg355_verify_blob_kzg_proof_batch_many_device_submit(C.byref(tk), None, None, None, 0, 3, settings.handle) == 0
    ok = (C.c_bool * 3)(); st = (C.c_int * 3)(7, 7, 7)
    assert L.kzg355_verify_collect(tk, ok, st) == 0 and list(ok) == [True] * 3 and list(st) == [0] * 3
    tk = C.c_void_p()
    assert L.kzg355_verify_blob_kzg_proof_batch_many_device_submit(C.byref(tk), tb.data_ptr(), tc.data_ptr(), tp.data_ptr(), n, 0, settings.handle) == 0
    assert L.kzg355_verify_collect(tk, None, None) == 0
    tk = C.c_void_p()
    assert L.kzg355_verify_blob_kzg_proof_batch_many_device_submit(C.byref(tk), tb.data_ptr() + 4, tc.data_ptr(), tp.data_ptr(), n, G, settings.handle) == 1
    assert not tk.value
    assert L.kzg355_verify_collect(None, ok, st) == 1


def test_device_entry_points_refuse_misaligned_pointers(kz, settings, random_set):
    """The device-resident entry points read blobs and records 16 bytes at a time: a pointer that is not 16-byte aligned is BadArgs,
    not a faulting kernel."""
    import torch
    blobs, cs, ps = random_set
    n = len(blobs)
    L = kz.kzg.lib(); dev = torch.device("cuda", settings.device)
    tb = torch.zeros(131072 * n + 64, dtype=torch.uint8, device=dev)
    tc = torch.frombuffer(bytearray(b"".join(cs)), dtype=torch.uint8).to(dev); tp = torch.frombuffer(bytearray(b"".join(ps)), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    ok = (C.c_bool * 1)(); st = (C.c_int * 1)()
    assert L.kzg355_verify_blob_kzg_proof_batch_many_device(ok, st, tb.data_ptr() + 4, tc.data_ptr(), tp.data_ptr(), n, 1, settings.handle) == 1
    out = C.create_string_buffer(48 * n); sts = (C.c_int * n)()
    assert L.kzg355_blob_to_kzg_commitment_many_device(out, sts, tb.data_ptr() + 8, n, settings.handle) == 1
    rec = torch.zeros(160 * n + 64, dtype=torch.uint8, device=dev)
    assert L.kzg355_verify_records_device(ok, st, rec.data_ptr() + 4, n, 1, settings.handle) == 1
    assert L.kzg355_verify_shard_records_device(rec.data_ptr() + 2, st, tb.data_ptr(), tc.data_ptr(), tp.data_ptr(), n, 1, settings.handle) == 1


def test_many_small_batches_take_the_many_batch_kernels(kz, settings, lincomb_handles, random_set):
    """2100 batches of 8 blobs in one launch set: past the threshold of the lane-per-batch transcript hash (1024) with the default
    handle, and through the single-chain Horner tail (default from 6144 batches) with the handle that pins both thresholds to 1;
    short and empty bucket lists (16 items of class 0 over 16 buckets).  Four batches carry two swapped proofs and must be the only
    false verdicts; one carries an off-curve commitment and must be the only Err."""
    import torch
    blobs, cs, ps = random_set
    n, G = len(blobs), 2100
    L = kz.kzg.lib(); dev = torch.device("cuda", settings.device)
    tb = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).to(dev).repeat(G)
    good_c, good_p = b"".join(cs), b"".join(ps)
    sw = list(ps); sw[2], sw[5] = sw[5], sw[2]
    bad_p = b"".join(sw)
    false_at, err_at = {5, 2047, 2048, 2099}, 1500
    bad_c = bytearray(good_c); bad_c[48 * 3:48 * 4] = bytes([0x9a]) + b"\xff" * 47        # x >= p: bytes_to_kzg_commitment fails
    tc = torch.frombuffer(bytearray(b"".join(bytes(bad_c) if g == err_at else good_c for g in range(G))), dtype=torch.uint8).to(dev)
    tp = torch.frombuffer(bytearray(b"".join(bad_p if g in false_at else good_p for g in range(G))), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    for s in (settings, lincomb_handles["rhash-lanes"]):
        ok = (C.c_bool * G)(); st = (C.c_int * G)()
        rc = L.kzg355_verify_blob_kzg_proof_batch_many_device(ok, st, tb.data_ptr(), tc.data_ptr(), tp.data_ptr(), n, G, s.handle)
        assert rc == 1
        assert [g for g in range(G) if st[g] != 0] == [err_at]
        assert sorted(g for g in range(G) if st[g] == 0 and not ok[g]) == sorted(false_at)


def test_sharded_exchange_on_rccl_with_one_rank(kz, settings, random_set):
    """The collectives of the sharded path on the backend the driver uses (nccl = RCCL) with device tensors: a one-rank process group
    still goes through torch's all_to_all_single (uneven-split API) and all_reduce(MAX) on the GPU.  Verdicts as the batch call gives."""
    import torch
    import torch.distributed as dist
    from kzg_rust_amd.sharded import HipEngine, verify_blob_kzg_proof_batch_sharded
    blobs, cs, ps = random_set
    n = len(blobs) // 2
    dev = torch.device("cuda", settings.device)
    bad = list(ps[n:]); bad[0], bad[1] = bad[1], bad[0]
    tb = torch.frombuffer(bytearray(b"".join(blobs[:n] + blobs[n:2 * n] + blobs[:n])), dtype=torch.uint8).to(dev)
    tc = torch.frombuffer(bytearray(b"".join(cs[:n] + cs[n:2 * n] + cs[:n])), dtype=torch.uint8).to(dev)
    tp = torch.frombuffer(bytearray(b"".join(ps[:n] + bad + ps[:n])), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1, device_id=dev)
    try:
        ok, st = verify_blob_kzg_proof_batch_sharded(tb, tc, tp, n, 3, HipEngine(settings), force_exchange=True)
    finally:
        if created:
            dist.destroy_process_group()
    assert ok == [True, False, True] and st == [0, 0, 0]


def test_full_size_launch_set_verdict_positions(kz, settings):
    """BASELINE.json's measured configuration as ONE launch set: 8192 batches of 64 blobs (524,288 blobs, 69 GB resident).  The batch
    equation is checked per batch, so the verdict vector is a size-independent property: every batch is true except the three whose
    proofs were permuted (first, middle, last), and the one with a non-canonical field element in a blob is the only Err."""
    import torch
    n, G = 64, 8192
    L = kz.kzg.lib(); dev = torch.device("cuda", settings.device)
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 90 * 2**30:
        pytest.skip("needs ~75 GB of free device memory")
    blobs = [random_blob(31000 + i) for i in range(n)]
    B = [kz.Blob(b) for b in blobs]
    cs = kz.Kzg.blob_to_kzg_commitment_many(B, settings)
    ps = kz.Kzg.compute_blob_kzg_proof_many(B, cs, settings)
    cb, pb = [c.to_bytes() for c in cs], [p.to_bytes() for p in ps]
    base = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).to(dev)
    tb = base.repeat(G)
    tc = torch.frombuffer(bytearray(b"".join(cb)), dtype=torch.uint8).to(dev).repeat(G)
    good_p = torch.frombuffer(bytearray(b"".join(pb)), dtype=torch.uint8).to(dev)
    sw = list(pb); sw[7], sw[40] = sw[40], sw[7]
    bad_p = torch.frombuffer(bytearray(b"".join(sw)), dtype=torch.uint8).to(dev)
    tp = good_p.repeat(G)
    false_at, err_at = [0, 4095, 8191], 6000
    for g in false_at:
        tp[48 * n * g:48 * n * (g + 1)] = bad_p
    tb[131072 * (n * err_at + 5) + 32 * 100:131072 * (n * err_at + 5) + 32 * 101] = 0xff       # element 100 of blob 5 of that batch: >= r
    torch.cuda.synchronize()
    ok = (C.c_bool * G)(); st = (C.c_int * G)()
    rc = L.kzg355_verify_blob_kzg_proof_batch_many_device(ok, st, tb.data_ptr(), tc.data_ptr(), tp.data_ptr(), n, G, settings.handle)
    assert rc == 1
    import numpy as np
    stv = np.frombuffer(st, dtype=np.int32); okv = np.frombuffer(ok, dtype=np.uint8)
    assert np.flatnonzero(stv).tolist() == [err_at]
    assert np.flatnonzero((stv == 0) & (okv == 0)).tolist() == false_at
    del tb, tc, tp
    torch.cuda.empty_cache()


def test_concurrent_mixed_calls_stress(kz, settings, random_set):
    """Eight host threads, 15 rounds each of every entry point on ONE handle (workspace pool, the shared side stream, the per-handle
    kernel statistics): every result must equal the single-threaded one."""
    import threading
    blobs, cs, ps = random_set
    B, Cm, Pr = [kz.Blob(b) for b in blobs], [kz.KzgCommitment(c) for c in cs], [kz.KzgProof(p) for p in ps]
    z = kz.Bytes32(random_field_element(3))
    proof0, y0 = kz.Kzg.compute_kzg_proof(B[0], z, settings)
    bad = list(Pr); bad[0], bad[1] = bad[1], bad[0]
    errors = []

    def work(t):
        try:
            for it in range(15):
                k = (t + it) % len(B)
                assert kz.Kzg.verify_blob_kzg_proof_batch(B, Cm, Pr, settings) is True
                assert kz.Kzg.verify_blob_kzg_proof_batch(B, Cm, bad, settings) is False
                assert kz.Kzg.blob_to_kzg_commitment(B[k], settings).to_bytes() == cs[k]
                assert kz.Kzg.compute_blob_kzg_proof(B[k], Cm[k], settings).to_bytes() == ps[k]
                assert kz.Kzg.verify_blob_kzg_proof(B[k], Cm[k], Pr[k], settings) is True
                p, y = kz.Kzg.compute_kzg_proof(B[0], z, settings)
                assert p.to_bytes() == proof0.to_bytes() and y.to_bytes() == y0.to_bytes()
                assert kz.Kzg.verify_kzg_proof(Cm[0], z, y0, proof0, settings) is True
                assert kz.Kzg.verify_kzg_proof(Cm[1], z, y0, proof0, settings) is False
        except Exception as e:           # noqa: BLE001 -- collected and re-raised on the main thread
            errors.append((t, repr(e)))
    ts = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:3]
