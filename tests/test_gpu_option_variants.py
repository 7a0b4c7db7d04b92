"""-m gpu: every dispatch variant a caller can reach through kzg355_options (include/kzg355.h) runs the reference's 22 batch vectors and the committed
n = 64 fixture's intermediates ("test or delete every reachable variant").  One handle per variant, loaded with
kzg355_load_trusted_setup_ex -- the explicit form, which reads no environment variable -- so the test names the option, not a KZG355_* string.

Per variant (pass rule of src/lib.rs:189-201):
  * the 22 verify_blob_kzg_proof_batch vectors (true / false / Err) through the drop-in entry point;
  * tests/golden/batch64.json through the C ABI from host slices: verdict true, swapped twin false;
  * the same batch's stage-1 records -> stage 2: r, proof_lincomb and rhs byte for byte against the fixture (kzg.rs:601-622; oracle-derived, the oracle
    is pinned by the reference's vectors), once as ONE batch and once as 80 copies in one launch set (the sizes from which the many-batch kernel forms
    apply are options too, and the variants below pull them down to 1);
  * for the commitment-side options: the 10 + 14 commitment / blob-proof vectors, byte-exact.
Variants that only differ from the default in SPEED must be bit-identical in everything observable; that is what this file pins."""
import ctypes as C
import json
import os

import pytest

from synth import random_blob
from vector_harness import run_function

pytestmark = pytest.mark.gpu

# (id, options).  Every value of every enumerated option appears at least once; the thresholds appear at both ends (1 = from the first batch on, -1 = never).
VARIANTS = [
    ("defaults", {}),
    ("challenge_form=1w", {"challenge_form": 1, "host_hash": -1}),
    ("challenge_form=2w", {"challenge_form": 2, "host_hash": -1}),
    ("lincomb_form=window", {"lincomb_form": 1}),
    ("lincomb_form=bucket,lc_chain_from=1", {"lincomb_form": 2, "lc_chain_from": 1}),
    ("lincomb_form=bucket,lc_chain_from=never", {"lincomb_form": 2, "lc_chain_from": 1 << 24}),
    ("lincomb_form=preshift", {"lincomb_form": 3}),
    ("pairing_lane", {"pairing_lane": 1}),
    ("pairing_two_wave_upto=never,hard12=never", {"pairing_two_wave_upto": -1, "pairing_hard12_from": -1}),
    ("pairing_two_wave_upto=always", {"pairing_two_wave_upto": 1 << 20}),
    ("pairing_hard12_from=1", {"pairing_hard12_from": 1, "pairing_two_wave_upto": -1}),
    ("miller_segments=1", {"miller_segments": 1}),
    ("miller_segments=4", {"miller_segments": 4}),
    ("rhash_lanes_from=1,host_rhash=off", {"rhash_lanes_from": 1, "host_rhash": -1}),
    ("host_rhash_max_records=8", {"host_rhash_max_records": 8}),
    ("host_hash=on,portable sha", {"host_hash": 1, "host_sha": 1, "host_threads": 3}),
    ("host_hash_max_blobs=8", {"host_hash_max_blobs": 8}),
    ("host_hash=off,device max=never", {"host_hash": -1, "host_hash_device_max_blobs": -1}),
    ("beside_max_blobs=1", {"beside_max_blobs": 1}),
    ("split_parts=3", {"split_parts": 3, "split_streams": 2}),
    ("submit_sets=1", {"submit_sets": 1}),
    ("submit_sets=2", {"submit_sets": 2}),
    ("staging_ring,chunk_mb=1", {"staging_ring": 1, "chunk_mb": 1, "chunks_in_flight": 2}),
    ("verify_only", {"verify_only": 1}),
    ("msm_bits=8", {"msm_bits": 8}),
    ("msm_eager,msm_bits=12", {"msm_eager": 1, "msm_bits": 12}),
    ("msm_glv=off,msm_bits=12,require_wide", {"msm_glv": -1, "msm_bits": 12, "msm_require_wide": 1}),
    ("quotient_form=6", {"quotient_form": 6, "msm_bits": 8}),
    ("self_test=0", {"self_test": 0}),
]
MSM_SIDE = {"verify_only", "msm_bits=8", "msm_eager,msm_bits=12", "msm_glv=off,msm_bits=12,require_wide", "quotient_form=6"}


def test_every_option_field_is_exercised():
    """No field of kzg355_options without a variant above (device / exchange / force_* belong to tests/test_gpu_multi_device.py)."""
    from kzg_rust_amd import _lib
    fields = {n for n, _ in _lib.Options._fields_} - {"struct_size", "device", "exchange", "force_multi", "force_sharded"}
    used = set()
    for _, o in VARIANTS:
        used |= set(o)
    assert fields - used == set(), sorted(fields - used)


@pytest.fixture(scope="module")
def kz():
    import kzg_rust_amd
    return kzg_rust_amd


@pytest.fixture(scope="module")
def fx64():
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "batch64.json")))
    n = fx["n"]
    blobs = [random_blob(fx["first_index"] + i) for i in range(n)]
    cs = [bytes.fromhex(c) for c in fx["commitments"]]
    ps = [bytes.fromhex(p) for p in fx["proofs"]]
    return fx, blobs, cs, ps


def _intermediates(L, s, t_rec, n, groups):
    out = C.create_string_buffer(128 * groups)
    ok = (C.c_bool * groups)(); st = (C.c_int * groups)()
    rc = L.kzg355_debug_batch_intermediates(out, ok, st, t_rec.data_ptr(), n, groups, s.handle)
    assert rc == 0 and not any(st), (rc, list(st)[:4])
    return [(out.raw[128 * g:128 * g + 32].hex(), out.raw[128 * g + 32:128 * g + 80].hex(), out.raw[128 * g + 80:128 * g + 128].hex(), bool(ok[g])) for g in range(groups)]


@pytest.mark.parametrize("name,opts", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_variant_runs_the_batch_vectors_and_the_fixture(name, opts, kz, setup_bytes, golden_vectors, golden_blobs, fx64):
    import torch
    from gpu_backend import ProductBackend
    g1, g2 = setup_bytes
    fx, blobs, cs, ps = fx64
    n = fx["n"]
    L = kz.kzg.lib()
    s = kz.KzgSettings.load_trusted_setup_ex([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)], **opts)
    try:
        # 1. the reference's batch vectors
        cnt, failures = run_function("verify_blob_kzg_proof_batch", golden_vectors, ProductBackend(), s, golden_blobs)
        assert cnt == 22 and not failures, "\n".join(failures)
        # 2. the n = 64 fixture from host slices, and its swapped twin
        B = [kz.Blob(b) for b in blobs]
        assert kz.Kzg.verify_blob_kzg_proof_batch(B, cs, ps, s) is fx["expect"]
        a, b = fx["swapped_pair"]
        sw = list(ps); sw[a], sw[b] = sw[b], sw[a]
        assert kz.Kzg.verify_blob_kzg_proof_batch(B, cs, sw, s) is fx["expect_swapped"]
        # 3. stage 1 records (z_i, y_i against the fixture), stage 2 intermediates: one batch, then 80 copies in one launch set
        dev = torch.device("cuda", s.device)
        G = 80
        tb = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).to(dev)
        tc = torch.frombuffer(bytearray(b"".join(cs)), dtype=torch.uint8).to(dev)
        tp = torch.frombuffer(bytearray(b"".join(ps)), dtype=torch.uint8).to(dev)
        rec = torch.zeros(160 * n, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        st1 = (C.c_int * 1)(-1)
        assert L.kzg355_verify_shard_records_device(rec.data_ptr(), st1, tb.data_ptr(), tc.data_ptr(), tp.data_ptr(), n, 1, s.handle) == 0 and st1[0] == 0
        r = bytes(rec.cpu().numpy())
        assert [r[160 * i + 48:160 * i + 80].hex() for i in range(n)] == fx["z"]
        assert [r[160 * i + 80:160 * i + 112].hex() for i in range(n)] == fx["y"]
        want = (fx["r"], fx["proof_lincomb"], fx["rhs"], True)
        assert _intermediates(L, s, rec, n, 1) == [want]
        many = rec.repeat(G).contiguous()
        torch.cuda.synchronize()
        assert _intermediates(L, s, many, n, G) == [want] * G
        # the device-resident entry point the bench times, G copies of the batch, and its submit / collect halves
        tbb, tcc, tpp = tb.repeat(4).contiguous(), tc.repeat(4).contiguous(), tp.repeat(4).contiguous()
        tpp[48 * (n + a):48 * (n + a) + 48] = tp[48 * b:48 * b + 48]       # batch 1: the proof of blob a replaced by that of blob b
        torch.cuda.synchronize()
        ok = (C.c_bool * 4)(); st = (C.c_int * 4)()
        assert L.kzg355_verify_blob_kzg_proof_batch_many_device(ok, st, tbb.data_ptr(), tcc.data_ptr(), tpp.data_ptr(), n, 4, s.handle) == 0
        assert [bool(x) for x in ok] == [True, False, True, True] and not any(st)
        tk = C.c_void_p()
        assert L.kzg355_verify_blob_kzg_proof_batch_many_device_submit(C.byref(tk), tbb.data_ptr(), tcc.data_ptr(), tpp.data_ptr(), n, 4, s.handle) == 0
        ok2 = (C.c_bool * 4)()
        assert L.kzg355_verify_collect(tk, ok2, st) == 0 and [bool(x) for x in ok2] == [True, False, True, True]
        # 4. the commitment side, where the variant touches it
        if name in MSM_SIDE:
            for fn, want_n in (("blob_to_kzg_commitment", 10), ("compute_blob_kzg_proof", 14), ("compute_kzg_proof", 46)):
                cnt, failures = run_function(fn, golden_vectors, ProductBackend(), s, golden_blobs)
                assert cnt == want_n and not failures, "\n".join(failures)
            got = kz.Kzg.blob_to_kzg_commitment_many(B[:5], s)
            assert [c.to_bytes() for c in got] == cs[:5]
            got = kz.Kzg.compute_blob_kzg_proof_many(B[:5], cs[:5], s)
            assert [p.to_bytes() for p in got] == ps[:5]
            if "msm_bits" in opts:
                assert s.msm_form == (8 if opts["msm_bits"] == 8 or opts.get("verify_only") else opts["msm_bits"])
            if opts.get("verify_only"):
                assert s.msm_form == 8 and s.msm_shape() == (0, 0, 0, 0)
    finally:
        s.free()
