"""-m gpu: randomised differential test of stage 1 (challenge z, evaluation y) against the oracle on blobs built to stress the lazy
arithmetic of k_eval's tree (csrc/eval_core.h): every field element drawn from a per-blob mixture of uniform values, tiny values,
values just below r, single bits, all-ones limb patterns and zeros.  z and y of the 160-byte records must be byte-exact
(reference: compute_challenge kzg.rs:298-339, evaluate_polynomial_in_evaluation_form kzg.rs:346-389).
KZG355_FUZZ_BATCHES (default 4) batches of 64 blobs; a long run (200 batches) is recorded in profiles/r03/eval_fuzz.txt."""
import ctypes as C
import os
import random

import pytest

pytestmark = pytest.mark.gpu
R_ = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


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


def _element(rnd, kind):
    if kind == 0:
        return rnd.randrange(R_)
    if kind == 1:
        return rnd.randrange(1 << 16)
    if kind == 2:
        return R_ - 1 - rnd.randrange(1 << 16)
    if kind == 3:
        return 1 << rnd.randrange(254)
    if kind == 4:                                    # runs of all-ones 29-bit limbs
        v = 0
        for limb in range(8):
            if rnd.random() < 0.7:
                v |= ((1 << 29) - 1) << (29 * limb)
        return v | (rnd.randrange((R_ >> 232)) << 232)
    return 0


def _blob(rnd):
    weights = [rnd.random() ** 2 for _ in range(6)]
    kinds = rnd.choices(range(6), weights=weights, k=4096)
    if rnd.random() < 0.3:                           # long constant stretches: the sums of a whole subtree at their extremes
        k0 = rnd.randrange(6); v0 = _element(rnd, k0)
        a = rnd.randrange(4096); b = rnd.randrange(a, 4097)
        return b"".join((v0 if a <= i < b else _element(rnd, kinds[i])).to_bytes(32, "big") for i in range(4096))
    return b"".join(_element(rnd, k).to_bytes(32, "big") for k in kinds)


def test_stage1_z_and_y_match_oracle_on_adversarial_blobs(kz, settings, oracle, oracle_settings):
    import torch
    batches = int(os.environ.get("KZG355_FUZZ_BATCHES", "4"))
    rnd = random.Random(int(os.environ.get("KZG355_FUZZ_SEED", "20261004")))
    L = kz.kzg.lib()
    dev = torch.device("cuda", settings.device)
    n = 64
    for bi in range(batches):
        blobs = [_blob(rnd) for _ in range(n)]
        B = [kz.Blob(b) for b in blobs]
        cs = [bytes(c) for c in kz.Kzg.blob_to_kzg_commitment_many(B, settings)]
        ps = cs                                      # any valid G1 points: z and y do not depend on the proofs
        t_blobs = torch.frombuffer(bytearray(b"".join(blobs)), dtype=torch.uint8).to(dev)
        t_c = torch.frombuffer(bytearray(b"".join(cs)), dtype=torch.uint8).to(dev)
        t_rec = torch.zeros(160 * n, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        st = (C.c_int * 1)(-1)
        assert L.kzg355_verify_shard_records_device(t_rec.data_ptr(), st, t_blobs.data_ptr(), t_c.data_ptr(), t_c.data_ptr(), n, 1, settings.handle) == 0 and st[0] == 0
        rec = bytes(t_rec.cpu().numpy())
        inter = oracle.verify_batch_intermediates(blobs, cs, ps, oracle_settings)
        for i in range(n):
            assert rec[160 * i + 48:160 * i + 80] == inter["z"][i], f"batch {bi} z[{i}]"
            assert rec[160 * i + 80:160 * i + 112] == inter["y"][i], f"batch {bi} y[{i}]"
