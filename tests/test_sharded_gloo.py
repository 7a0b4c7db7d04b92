"""world_size-2 and world_size-8 gloo tests (CPU) of the sharded verify orchestration in kzg_rust_amd/sharded.py: partitioning, the single
all-to-all of 160-byte records, gather order (= transcript order), status merging.  The compute stages are played by the
CPU oracle here (test infrastructure); on the GPU box the same driver runs over HipEngine (tests/test_gpu_parity.py and
bench.py --gpus N)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_LOCAL = 2        # blobs of every batch per rank (BASELINE config 5 has 64 per rank: same code path, the oracle is the slow part)
GROUPS = 3         # batch 0 honest, batch 1 has swapped proofs (false), batch 2 has an invalid commitment on one rank (Err)


class OracleEngine:
    def __init__(self):
        from oracle.oracle import Oracle
        self.o = Oracle()
        g = os.path.join(ROOT, "tests", "golden")
        self.s = self.o.load_trusted_setup(open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(),
                                           open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read())

    def shard_records(self, blobs, commitments, proofs, n_local, groups):
        from oracle.oracle import OracleError
        b, c, p = bytes(blobs.numpy()), bytes(commitments.numpy()), bytes(proofs.numpy())
        out, st = bytearray(groups * n_local * 160), []
        for g in range(groups):
            lo = g * n_local
            try:
                rec = self.o.shard_records([b[(lo + i) * 131072:(lo + i + 1) * 131072] for i in range(n_local)],
                                           [c[(lo + i) * 48:(lo + i + 1) * 48] for i in range(n_local)],
                                           [p[(lo + i) * 48:(lo + i + 1) * 48] for i in range(n_local)], self.s)
                out[lo * 160:(lo + n_local) * 160] = rec
                st.append(0)
            except OracleError as e:
                st.append(e.code)
        return torch.frombuffer(out, dtype=torch.uint8).clone(), None, st       # (no decoded points: the oracle's stage 2 decodes the records)

    def verify_records(self, records, points, n, groups):
        from oracle.oracle import OracleError
        r = bytes(records.numpy())
        ok, st = [], []
        for g in range(groups):
            try:
                ok.append(self.o.verify_records(r[g * n * 160:(g + 1) * n * 160], self.s)); st.append(0)
            except OracleError as e:
                ok.append(False); st.append(e.code)
        return ok, st


class OracleWordsEngine(OracleEngine):
    """The interface HipEngine gives the N-GPU run (shard_records_words / verify_records_words: per-batch results left as int32 words in a tensor the
    collectives carry) played by the oracle on HOST tensors: `_sharded_device_words` -- the exchange, the permutes into transcript order, the merge, the
    failure word of the all-gather form -- runs on CPU exactly as `bench.py --gpus N` runs it.  fail_stage2_on: the rank whose stage 2 reports a whole-call
    device failure (status 7 in every word), as an out-of-memory there would."""
    words_on_host = True

    def __init__(self, fail_stage2_on=None):
        super().__init__()
        self.fail_stage2_on = fail_stage2_on

    def shard_records_words(self, blobs, commitments, proofs, n_local, groups, words):
        rec, _, st = self.shard_records(blobs, commitments, proofs, n_local, groups)
        words.copy_(torch.tensor(st, dtype=torch.int32))
        return rec, torch.zeros(groups * 2 * n_local * 112, dtype=torch.uint8)      # (decoded points: opaque to the orchestration; the oracle's stage 2 decodes the records)

    def verify_records_words(self, records, points, n, groups, words):
        if self.fail_stage2_on is not None and dist.get_rank() == self.fail_stage2_on:
            words.fill_(1 + 256 * 7)
            return
        ok, st = self.verify_records(records, None, n, groups)
        words.copy_(torch.tensor([1 + int(o) + 256 * int(c) for o, c in zip(ok, st)], dtype=torch.int32))


def _inputs(n_total):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle.oracle import Oracle
    from synth import random_blob
    o = Oracle()
    g = os.path.join(ROOT, "tests", "golden")
    s = o.load_trusted_setup(open(os.path.join(g, "trusted_setup_g1.bin"), "rb").read(), open(os.path.join(g, "trusted_setup_g2.bin"), "rb").read())
    blobs = [random_blob(500 + i) for i in range(n_total)]
    cs = [o.blob_to_kzg_commitment(b, s) for b in blobs]
    ps = [o.compute_blob_kzg_proof(b, c, s) for b, c in zip(blobs, cs)]
    return blobs, cs, ps


def _worker(rank, world, port, blobs, cs, ps, q, exchange, engine_kind="records"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from kzg_rust_amd.sharded import partition, verify_blob_kzg_proof_batch_sharded
    lo, hi = partition(N_LOCAL * world, world)[rank]
    n_local = hi - lo
    batches = []
    for g in range(GROUPS):
        b, c, p = blobs[lo:hi], list(cs[lo:hi]), list(ps[lo:hi])
        if g == 1 and rank == 0:
            p[0], p[1] = p[1], p[0]                        # valid points, wrong statement -> false
        if g == 2 and rank == world - 3 + 2 * (world == 2):
            c[0] = bytes([0x9A]) + b"\xff" * 47            # x >= p -> Err on one rank only (rank 1 of 2, rank 5 of 8)
        batches.append((b, c, p))
    tb = torch.frombuffer(bytearray(b"".join(x for bt in batches for x in bt[0])), dtype=torch.uint8)
    tc = torch.frombuffer(bytearray(b"".join(x for bt in batches for x in bt[1])), dtype=torch.uint8)
    tp = torch.frombuffer(bytearray(b"".join(x for bt in batches for x in bt[2])), dtype=torch.uint8)
    eng = OracleEngine() if engine_kind == "records" else OracleWordsEngine()
    timings = {}
    ok, st = verify_blob_kzg_proof_batch_sharded(tb, tc, tp, n_local, GROUPS, eng, exchange=exchange, timings=timings)
    assert timings.get("stage1_ms", 0) > 0 and "exchange_ms" in timings and "merge_ms" in timings, timings
    # ONE batch over the ranks (BASELINE config 5's shape): every share of the batches but the last rank's is EMPTY
    # (groups * r // world == groups * (r + 1) // world), the last rank runs stage 2 alone, and the all-reduce still has to hand
    # the others the verdict -- honest first, then with rank 0's proofs swapped, then with
    # an invalid proof on rank 0 (Err raised by a rank that verifies nothing itself)
    one = lambda t, per: t[:n_local * per].clone()
    ok1, st1 = verify_blob_kzg_proof_batch_sharded(one(tb, 131072), one(tc, 48), one(tp, 48), n_local, 1, eng, exchange=exchange)
    sw = one(tp, 48)
    if rank == 0:
        tmp = sw[:48].clone(); sw[:48] = sw[48:96]; sw[48:96] = tmp
    ok2, st2 = verify_blob_kzg_proof_batch_sharded(one(tb, 131072), one(tc, 48), sw, n_local, 1, eng, exchange=exchange)
    bad = one(tp, 48)
    if rank == 0:
        bad[:48] = torch.frombuffer(bytearray(bytes([0x9A]) + b"\xff" * 47), dtype=torch.uint8)
    ok3, st3 = verify_blob_kzg_proof_batch_sharded(one(tb, 131072), one(tc, 48), bad, n_local, 1, eng, exchange=exchange)
    raised = None
    if engine_kind == "words":
        # a whole-call failure of ONE rank's stage 2 must raise on EVERY rank in both forms (the all-gather form replicates stage 2: only its failure word tells
        # the others) -- and the ranks must still be in step afterwards
        try:
            verify_blob_kzg_proof_batch_sharded(tb, tc, tp, n_local, GROUPS, OracleWordsEngine(fail_stage2_on=world - 1), exchange=exchange)
            raised = False
        except RuntimeError:
            raised = True
        cap = {}
        ok4, st4 = verify_blob_kzg_proof_batch_sharded(tb, tc, tp, n_local, GROUPS, eng, exchange=exchange, capture=cap)
        assert ok4 == ok and st4 == st
        lo_g, hi_g = cap.get("share", (0, 0))
        assert cap.get("records") is None or cap["records"].numel() == (hi_g - lo_g) * n_local * world * 160      # the gathered batches, whole
    q.put((rank, ok, st, (ok1, st1, ok2, st2, ok3, st3), raised))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,exchange,engine_kind", [(2, "alltoall", "records"), (2, "allgather", "records"), (8, "alltoall", "records"), (8, "allgather", "records"),
                                                       (2, "alltoall", "words"), (2, "allgather", "words"), (4, "allgather", "words"), (4, "alltoall", "words"),
                                                       (2, "allgather_split", "words"), (4, "allgather_split", "words"), (2, "allgather_split", "records"),
                                                       (3, "alltoall", "words"), (3, "allgather_split", "words")])      # an odd world: one batch per rank
def test_sharded_verify_gloo(world, exchange, engine_kind):
    """both exchanges: the all-to-all with stage 2 split by batch, and BASELINE.json's single all-gather with stage 2 replicated; at world 8
    (config 5's rank count) the three batches fall to ranks 2, 5 and 7 -- five ranks with an empty share of the batches"""
    blobs, cs, ps = _inputs(N_LOCAL * world)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, blobs, cs, ps, q, exchange, engine_kind)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, st, single, raised in res:                  # identical verdicts on every rank
        assert raised is (True if engine_kind == "words" else None), (rank, raised)
        assert ok == [True, False, False], (rank, ok)
        assert st[0] == 0 and st[1] == 0 and st[2] != 0, (rank, st)
        ok1, st1, ok2, st2, ok3, st3 = single
        assert (ok1, st1) == ([True], [0]), (rank, ok1, st1)
        assert (ok2, st2) == ([False], [0]), (rank, ok2, st2)
        assert ok3 == [False] and st3[0] != 0, (rank, ok3, st3)


def test_partition():
    from kzg_rust_amd.sharded import partition
    assert partition(512, 8)[3] == (192, 256)
    with pytest.raises(ValueError):
        partition(10, 4)
