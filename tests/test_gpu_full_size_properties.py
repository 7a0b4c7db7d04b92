"""-m gpu: size-independent properties at the FULL launch sizes of BASELINE.json's configurations (16,384 blobs per commitment / proof launch; 8192 batches of 64
per verify launch set), on blobs that are all different -- the oracle cannot follow to these sizes in seconds, the algebra can:

* linearity: commitment(a + b) = commitment(a) + commitment(b) for 5,461 triples of one launch (kzg.rs:392-407 is a linear map of the blob's field elements);
  the G1 addition is the oracle's (one point addition per triple);
* round trip: 16,384 blobs -> commitments -> blob proofs -> verify_blob_kzg_proof_batch over 256 batches of 64 in one call: every verdict true; a blob byte
  flipped in one batch, two commitments swapped in another, two proofs swapped in a third: exactly those three batches false (kzg.rs:527-666);
* the same round trip over a FULL verify launch set: 524,288 different blobs (68.7 GB), five disturbed batches found at their positions.

Inputs are made on the device (torch), every field element with its top byte below 0x73: canonical by construction."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu

N_LAUNCH = 16384


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


def _random_blobs(torch, dev, n, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    b = torch.randint(0, 256, (n, 4096, 32), dtype=torch.uint8, device=dev, generator=g)
    b[:, :, 0] = torch.randint(0, 0x73, (n, 4096), dtype=torch.uint8, device=dev, generator=g)        # big-endian top byte < 0x73: below r
    return b


def _commit(L, s, torch, blobs):
    n = blobs.shape[0]
    out = C.create_string_buffer(48 * n); st = (C.c_int * n)()
    assert L.kzg355_blob_to_kzg_commitment_many_device(out, st, blobs.data_ptr(), n, s.handle) == 0
    assert not any(st)
    return out.raw


def test_commitments_are_linear_over_a_full_launch(kz, settings, oracle):
    import torch
    L = kz.kzg.lib(); dev = torch.device("cuda", settings.device)
    t = N_LAUNCH // 3                                                        # 5461 triples (a, b, a + b) = 16,383 blobs: one launch
    g = torch.Generator(device=dev); g.manual_seed(6001)
    raw = torch.randint(0, 256, (2, t, 4096, 32), dtype=torch.uint8, device=dev, generator=g)
    # a keeps the even byte positions, b the odd ones, both with a zero top byte: a + b has no carry anywhere, so its bytes are a | b, and all three are < 2^248
    even = (torch.arange(32, device=dev) % 2 == 0) & (torch.arange(32, device=dev) > 0)
    odd = torch.arange(32, device=dev) % 2 == 1
    a = raw[0] * even.to(torch.uint8); b = raw[1] * odd.to(torch.uint8)
    blobs = torch.cat([a, b, a | b]).contiguous()
    cs = _commit(L, settings, torch, blobs)
    ca, cb, cc = (cs[48 * t * k:48 * t * (k + 1)] for k in range(3))
    one = (1).to_bytes(32, "big")
    bad = [i for i in range(t) if oracle.g1_mul_add(ca[48 * i:48 * i + 48], one, cb[48 * i:48 * i + 48]) != cc[48 * i:48 * i + 48]]
    assert not bad, bad[:8]
    assert len(set(cc[48 * i:48 * i + 48] for i in range(t))) == t           # (all different: not one constant answer)


def test_round_trip_of_a_full_launch_of_distinct_blobs(kz, settings):
    import torch
    L = kz.kzg.lib(); dev = torch.device("cuda", settings.device)
    n, per = N_LAUNCH, 64
    blobs = _random_blobs(torch, dev, n, 6002)
    cs = _commit(L, settings, torch, blobs)
    d_cs = torch.frombuffer(bytearray(cs), dtype=torch.uint8).to(dev)
    out = C.create_string_buffer(48 * n); st = (C.c_int * n)()
    assert L.kzg355_compute_blob_kzg_proof_many_device(out, st, blobs.data_ptr(), d_cs.data_ptr(), n, settings.handle) == 0
    assert not any(st)
    d_ps = torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev)
    groups = n // per
    ok = (C.c_bool * groups)(); sg = (C.c_int * groups)()

    def verdicts():
        torch.cuda.synchronize()
        assert L.kzg355_verify_blob_kzg_proof_batch_many_device(ok, sg, blobs.data_ptr(), d_cs.data_ptr(), d_ps.data_ptr(), per, groups, settings.handle) == 0
        assert not any(sg)
        return [bool(ok[i]) for i in range(groups)]
    assert all(verdicts())
    blobs[per * 17 + 5, 1234, 31] ^= 1                                       # batch 17: a field element of one blob off by one
    c = d_cs.view(n, 48); c[[per * 101 + 3, per * 101 + 40]] = c[[per * 101 + 40, per * 101 + 3]]      # batch 101: two commitments swapped
    p = d_ps.view(n, 48); p[[per * 255, per * 255 + 63]] = p[[per * 255 + 63, per * 255]]              # batch 255: two proofs swapped
    got = verdicts()
    assert [i for i, v in enumerate(got) if not v] == [17, 101, 255]


def test_round_trip_of_a_full_verify_launch_set_of_distinct_blobs(kz, settings):
    """BASELINE.json's verify configuration at its full size with NO repeated input: 8192 batches x 64 = 524,288 different blobs (68.7 GB), their commitments and
    proofs made by the product's own *_many_device calls, verified in ONE call: every verdict true; after a blob byte, a commitment and a proof are disturbed in
    five batches spread over the launch set, exactly those five are false."""
    import torch
    L = kz.kzg.lib(); dev = torch.device("cuda", settings.device)
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 200e9:
        pytest.skip(f"needs ~200 GB of free HBM (blobs 68.7 GB + the MSM table a default handle sizes + scratch), {free / 1e9:.0f} GB free")
    per, groups = 64, 8192
    n = per * groups
    blobs = torch.empty((n, 4096, 32), dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(6003)
    for lo in range(0, n, N_LAUNCH):
        blobs[lo:lo + N_LAUNCH].random_(0, 256, generator=g)
        blobs[lo:lo + N_LAUNCH, :, 0].random_(0, 0x73, generator=g)
    out = C.create_string_buffer(48 * n); st = (C.c_int * n)()
    assert L.kzg355_blob_to_kzg_commitment_many_device(out, st, blobs.data_ptr(), n, settings.handle) == 0
    assert not any(st)
    d_cs = torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev)
    assert L.kzg355_compute_blob_kzg_proof_many_device(out, st, blobs.data_ptr(), d_cs.data_ptr(), n, settings.handle) == 0
    assert not any(st)
    d_ps = torch.frombuffer(bytearray(out.raw), dtype=torch.uint8).to(dev)
    ok = (C.c_bool * groups)(); sg = (C.c_int * groups)()

    def verdicts():
        torch.cuda.synchronize()
        assert L.kzg355_verify_blob_kzg_proof_batch_many_device(ok, sg, blobs.data_ptr(), d_cs.data_ptr(), d_ps.data_ptr(), per, groups, settings.handle) == 0
        assert not any(sg)
        return [bool(ok[i]) for i in range(groups)]
    assert all(verdicts())
    c = d_cs.view(n, 48); p = d_ps.view(n, 48)
    blobs[per * 0 + 0, 0, 31] ^= 1                                           # the first field element of the first blob
    blobs[per * 8191 + 63, 4095, 31] ^= 1                                    # the last field element of the last blob
    c[[per * 4096, per * 4096 + 1]] = c[[per * 4096 + 1, per * 4096]]        # batch 4096: two commitments swapped
    p[[per * 2047 + 7, per * 2047 + 8]] = p[[per * 2047 + 8, per * 2047 + 7]]      # batch 2047: two proofs swapped
    p[per * 6000 + 31] = p[per * 6001 + 31]                                  # batch 6000: a proof of another batch's blob
    got = verdicts()
    assert [i for i, v in enumerate(got) if not v] == [0, 2047, 4096, 6000, 8191]
    del blobs
    torch.cuda.empty_cache()
