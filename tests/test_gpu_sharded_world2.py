"""-m gpu: the sharded driver (kzg_rust_amd/sharded.py) over the HIP engine with TWO ranks.  The GPU box has one card, so both
ranks use cuda:0 and the collectives run on gloo (staged through the host); everything else -- stage 1 on the rank's blocks, the
permute of the gathered records into transcript order, stage 2 on the rank's share of the batches, the merged verdicts -- is the
code `bench.py --gpus N` runs with RCCL.  Results must equal the single-process verdicts."""
import os
import socket
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_TOTAL, GROUPS = 8, 3


def _worker(rank, world, port, blobs, cs, ps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["KZG355_MSM"] = "bucket"
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import kzg_rust_amd as kz
    from kzg_rust_amd.sharded import HipEngine, partition, verify_blob_kzg_proof_batch_sharded
    g = os.path.join(ROOT, "tests", "golden")
    g1 = open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read()
    s = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
    lo, hi = partition(N_TOTAL, world)[rank]
    n_local = hi - lo
    dev = torch.device("cuda", 0)
    batches = []
    for gi in range(GROUPS):
        b, c, p = list(blobs[lo:hi]), list(cs[lo:hi]), list(ps[lo:hi])
        if gi == 1 and rank == 1:
            p[0], p[1] = p[1], p[0]                        # valid points, wrong statement -> false, from the second block
        if gi == 2 and rank == 0:
            c[2] = bytes([0x9A]) + b"\xff" * 47            # Err on rank 0 only
        batches.append((b, c, p))
    tb = torch.frombuffer(bytearray(b"".join(x for bt in batches for x in bt[0])), dtype=torch.uint8).to(dev)
    tc = torch.frombuffer(bytearray(b"".join(x for bt in batches for x in bt[1])), dtype=torch.uint8).to(dev)
    tp = torch.frombuffer(bytearray(b"".join(x for bt in batches for x in bt[2])), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    eng = HipEngine(s)
    res = []
    for exchange in ("alltoall", "allgather", "allgather_split"):      # stage 2 split by batch / BASELINE.json's single all-gather with stage 2 replicated / the all-gather with stage 2 split
        ok, st = verify_blob_kzg_proof_batch_sharded(tb, tc, tp, n_local, GROUPS, eng, exchange=exchange)
        ok1, st1 = verify_blob_kzg_proof_batch_sharded(tb[:n_local * 131072], tc[:n_local * 48], tp[:n_local * 48], n_local, 1, eng, exchange=exchange)    # one batch: rank 0's share is empty
        res.append((exchange, ok, st, ok1, st1))

    # a whole-call failure of stage 2 on ONE rank (here: rank 1 reports NO_MEMORY for every batch it was given) must raise on EVERY rank in both
    # exchanges -- in the all-gather form stage 2 is replicated and nothing else would tell rank 0
    class FailingStage2(HipEngine):
        def verify_records_words(self, records, points, n, groups, words):
            if rank == 1:
                words.fill_(1 + 256 * 7)
            else:
                super().verify_records_words(records, points, n, groups, words)
    raised = []
    for exchange in ("alltoall", "allgather", "allgather_split"):
        try:
            verify_blob_kzg_proof_batch_sharded(tb, tc, tp, n_local, GROUPS, FailingStage2(s), exchange=exchange)
            raised.append(False)
        except RuntimeError:
            raised.append(True)
    # ... and the ranks are still in step afterwards: an honest call goes through
    ok, st = verify_blob_kzg_proof_batch_sharded(tb, tc, tp, n_local, GROUPS, eng, exchange="allgather")
    q.put((rank, res, raised, ok, st))
    s.free()
    dist.destroy_process_group()


def test_sharded_driver_two_ranks_on_the_hip_engine():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import kzg_rust_amd as kz
    from synth import random_blob
    g = os.path.join(ROOT, "tests", "golden")
    g1 = open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(); g2 = open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read()
    os.environ["KZG355_MSM"] = "bucket"
    try:
        s = kz.Kzg.load_trusted_setup([g1[48 * i:48 * i + 48] for i in range(4096)], [g2[96 * i:96 * i + 96] for i in range(65)])
    finally:
        del os.environ["KZG355_MSM"]
    blobs = [random_blob(7700 + i) for i in range(N_TOTAL)]
    B = [kz.Blob(b) for b in blobs]
    cs = kz.Kzg.blob_to_kzg_commitment_many(B, s)
    ps = kz.Kzg.compute_blob_kzg_proof_many(B, cs, s)
    assert kz.Kzg.verify_blob_kzg_proof_batch(B, cs, ps, s) is True
    s.free()
    cb, pb = [c.to_bytes() for c in cs], [p.to_bytes() for p in ps]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, blobs, cb, pb, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, per_exchange, raised, ok_after, st_after in res:
        assert raised == [True, True, True], (rank, raised)
        assert ok_after == [True, False, False] and st_after[:2] == [0, 0], (rank, ok_after, st_after)
        for exchange, ok, st, ok1, st1 in per_exchange:
            assert ok == [True, False, False], (rank, exchange, ok)
            assert st[0] == 0 and st[1] == 0 and st[2] != 0, (rank, exchange, st)
            assert (ok1, st1) == ([True], [0]), (rank, exchange, ok1, st1)
