"""Sharded verify_blob_kzg_proof_batch: one process per GPU, batches split into contiguous per-rank shards.

The reference has no distributed layer (single-threaded; "Potentially paralellizable" is a comment at src/kzg.rs:661).
The per-blob loop (kzg.rs:671-683) is independent per blob; the only coupling between blobs is the random-linear-
combination challenge r, a hash over ALL (C_i, z_i, y_i, proof_i) (utils.rs:454-463), and the final sums + one pairing
(kzg.rs:601-625).  So:

  stage 1 (per rank, its shard)        -> 160-byte records  C_i | z_i | y_i | proof_i      (no communication)
  ONE all-to-all of the records          (torch.distributed; backend "nccl" = RCCL over xGMI): stage 2 is split by batch, so rank j
                                         needs the records of ITS batches only, from every rank -- 1 / world of what an all-gather
                                         would move, pairwise over the direct links instead of around a ring; the decoded points
                                         of those records (224 bytes per blob) ride in the same buffer, so that stage 2 does not
                                         take 2n square roots to decompress what stage 1 already decoded
  stage 2 (per rank, its share of the BATCHES, now complete) -> r-powers, the linear combinations,
                                         the pairing.  On the GPU stage 2 costs about a third of stage 1, so replicating it on
                                         every rank would cap the scaling; splitting it by batch keeps the work per rank constant.
  ONE all-reduce(MAX) of the per-batch verdict / status words, so every rank returns every verdict and an Err on any rank
                                         is an Err everywhere (the `?` semantics)

BASELINE.json's north_star names the other form -- a single all-gather of the records with stage 2 replicated on every rank; it is
here as exchange="allgather" (bench.py --exchange allgather), so that an 8-GPU run can compare the two.

The compute stages are delegated to an `engine` with two methods, so that the orchestration (partitioning, gather
order, status merging) is testable on CPU with gloo; the product engine is HipEngine (C ABI of libkzg355.so).
"""
import ctypes as C

RECORD = 160
BLOB = 131072


POINT = 112       # KZG355_BYTES_PER_POINT: one validated affine point as stage 1 leaves it (opaque)


class HipEngine:
    """Stages on the HIP engine; tensors are uint8 CUDA tensors on the settings' device.  Stage 1 also hands out the decoded points of
    its shard and stage 2 takes the gathered ones, so that no rank decompresses (a square root per point) what another rank already has."""

    def __init__(self, settings):
        from . import kzg
        self.s = settings
        self.L = kzg.lib()

    def shard_records(self, blobs, commitments, proofs, n_local, groups):
        import torch
        # torch.empty, not zeros: a fill kernel would be queued on torch's stream, which nothing orders against the engine's own streams
        # (a late fill could wipe records stage 1 has already written).  A call that writes nothing reports it in every status.
        rec = torch.empty(groups * n_local * RECORD, dtype=torch.uint8, device=blobs.device)
        pts = torch.empty(groups * 2 * n_local * POINT, dtype=torch.uint8, device=blobs.device)
        st = (C.c_int * max(groups, 1))()
        rc = self.L.kzg355_verify_shard_records_points_device(rec.data_ptr(), pts.data_ptr(), st, blobs.data_ptr(), commitments.data_ptr(), proofs.data_ptr(),
                                                              n_local, groups, self.s.handle)
        st = _ints(st, groups)
        # rc == 1 (BADARGS) is a per-batch status when some batch carries it.  A whole-call refusal (misaligned pointer, too many blobs:
        # nothing was written to rec / pts) marks EVERY batch, so no batch of such a call is ever read as verified; a
        # non-zero rc with every batch OK cannot come from this library and is raised
        if rc not in (0, 1) or (rc != 0 and not st.any()):
            raise RuntimeError(f"kzg355_verify_shard_records_points_device: status {rc}")
        return rec, pts, st

    # The same two stages with the per-batch results left on the device (int32 tensors): nothing is copied back between the stages and the
    # collectives; a whole-call failure (a refused call, a device error) is folded into EVERY batch's word instead of being raised, so that
    # all ranks go through the same collectives and see the failure together.
    def shard_records_words(self, blobs, commitments, proofs, n_local, groups, words):
        """stage 1; words (int32[groups], device): the KZG355 status of every batch on this rank's shard"""
        import torch
        rec = torch.empty(groups * n_local * RECORD, dtype=torch.uint8, device=blobs.device)
        pts = torch.empty(groups * 2 * n_local * POINT, dtype=torch.uint8, device=blobs.device)
        rc = self.L.kzg355_verify_shard_records_points_words_device(rec.data_ptr(), pts.data_ptr(), words.data_ptr(), blobs.data_ptr(), commitments.data_ptr(),
                                                                    proofs.data_ptr(), n_local, groups, self.s.handle)
        if rc != 0:
            words.fill_(rc)
        return rec, pts

    def verify_records_words(self, records, points, n, groups, words):
        """stage 2; words (int32[groups], device): 1 + ok + 256 * status of every batch"""
        rc = self.L.kzg355_verify_records_points_words_device(words.data_ptr(), records.data_ptr(), points.data_ptr(), n, groups, self.s.handle)
        if rc != 0:
            words.fill_(1 + 256 * rc)

    def batch_intermediates(self, records, n, groups):
        """audit readback of stage 2 on gathered records (kzg355_debug_batch_intermediates): per batch (r, proof_lincomb, rhs, ok, status) -- what
        bench.py's N > 1 parity gate compares with the committed fixture / the single-device run (transcript order of utils.rs:454-463)"""
        out = C.create_string_buffer(128 * max(groups, 1))
        ok = (C.c_bool * max(groups, 1))()
        st = (C.c_int * max(groups, 1))()
        rc = self.L.kzg355_debug_batch_intermediates(out, ok, st, records.data_ptr(), n, groups, self.s.handle)
        if rc not in (0, 1):
            raise RuntimeError(f"kzg355_debug_batch_intermediates: status {rc}")
        return [(out.raw[128 * g:128 * g + 32], out.raw[128 * g + 32:128 * g + 80], out.raw[128 * g + 80:128 * g + 128], bool(ok[g]), int(st[g])) for g in range(groups)]

    def verify_records(self, records, points, n, groups):
        ok = (C.c_bool * max(groups, 1))()
        st = (C.c_int * max(groups, 1))()
        if points is None:
            rc = self.L.kzg355_verify_records_device(ok, st, records.data_ptr(), n, groups, self.s.handle)
        else:
            rc = self.L.kzg355_verify_records_points_device(ok, st, records.data_ptr(), points.data_ptr(), n, groups, self.s.handle)
        st = _ints(st, groups)
        if rc not in (0, 1) or (rc != 0 and not st.any()):
            raise RuntimeError(f"kzg355_verify_records_device: status {rc}")
        return _bools(ok, groups), st


def _ints(c_array, n):
    """ctypes int array -> numpy (one copy; a Python loop over 8192 ctypes elements costs ~1 ms per step)"""
    import numpy as np
    return np.frombuffer(c_array, dtype=np.int32, count=n).copy()


def _bools(c_array, n):
    import numpy as np
    return np.frombuffer(c_array, dtype=np.uint8, count=n) != 0


def partition(n_total, world):
    """Contiguous blocks: rank g owns blobs [g*n/G, (g+1)*n/G); gathered records are then already in transcript order."""
    if n_total % world:
        raise ValueError("batch size must be a multiple of the world size")
    n_local = n_total // world
    return [(r * n_local, (r + 1) * n_local) for r in range(world)]


def _on_host(group):
    """gloo moves host memory: with device tensors the two collectives are staged through the host (CPU tests, and the two-ranks-on-
    one-GPU test of the HIP engine); nccl = RCCL takes the device tensors as they are."""
    import torch.distributed as dist
    return dist.get_backend(group) == "gloo"


def _tick(timings, key, t0, device=None):
    """accumulate the wall time since t0 under `key` (the engine calls are synchronous; torch work is synchronised first)"""
    import time
    if timings is None:
        return time.perf_counter()
    if device is not None and device.type == "cuda":
        import torch
        torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    timings[key] = timings.get(key, 0.0) + (t1 - t0) * 1e3
    return t1


def verify_blob_kzg_proof_batch_sharded(local_blobs, local_commitments, local_proofs, n_local, groups, engine, group=None, force_exchange=False,
                                        exchange=None, timings=None, capture=None):
    """`groups` independent batches; this rank holds n_local blobs of each (group-major uint8 tensors).
    Returns (ok[groups], status[groups]) -- identical on every rank.  status != 0 <=> the reference returns Err.

    exchange (default "alltoall"; an ARGUMENT only -- ranks that read it from their environments could disagree and hang in different collectives):
      "alltoall"   stage 2 split by batch: ONE all-to-all brings every rank the records (+ decoded points) of its share of the batches,
                   one small all-reduce(MAX) spreads the verdicts and merges the statuses;
      "allgather"  BASELINE.json's north_star form: ONE all-gather of every rank's records (+ points + stage-1 statuses), then EVERY rank
                   runs stage 2 on all batches (replicated: no second collective, world x the stage-2 work).
      "allgather_split"  the same single all-gather, but stage 2 is NOT replicated: every rank verifies its share of the batches out of the gathered
                   buffer and one small all-reduce(MAX) spreads the verdict words, as in "alltoall" -- north_star's collective with the all-to-all's
                   partition of the work (what it costs over "alltoall" is the traffic: every rank receives world x what it needs).  Engines without the
                   words interface take "allgather" for it (same verdicts).
    timings (dict or None): accumulates stage1_ms / exchange_ms / stage2_ms / merge_ms of this rank, so that a scaling curve can be
    attributed (bench.py reports them per rank in config.exchange).
    capture (dict or None; device-words path only): receives "records" -- the gathered records this rank's stage 2 ran on, in transcript order
    (all batches for "allgather", this rank's share for "alltoall": "share" = (first batch, end batch)) -- for bench.py's parity gate."""
    import time
    import numpy as np
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if n_local == 0:
        return [True] * groups, [0] * groups                       # kzg.rs:653-655
    mode = exchange or "alltoall"
    if mode not in ("alltoall", "allgather", "allgather_split"):
        raise ValueError(f"exchange must be alltoall, allgather or allgather_split, not {mode!r}")
    dev = local_blobs.device
    t0 = time.perf_counter()
    # (words_on_host: a CPU stand-in engine of tests/test_sharded_gloo.py that keeps its per-batch words in host tensors -- the SAME orchestration, so that
    # the exchange, the merge and the failure handling the N-GPU run uses are exercised without a GPU)
    words_path = hasattr(engine, "shard_records_words") and (local_blobs.is_cuda or getattr(engine, "words_on_host", False))
    if words_path and (world > 1 or (force_exchange and dist.is_initialized())):
        return _sharded_device_words(local_blobs, local_commitments, local_proofs, n_local, groups, engine, group, mode, timings, world, capture)
    rec, pts, st_local = engine.shard_records(local_blobs, local_commitments, local_proofs, n_local, groups)
    t0 = _tick(timings, "stage1_ms", t0)
    if world == 1 and not (force_exchange and dist.is_initialized()):      # (force_exchange: run the collectives of a one-rank group too -- test hook)
        ok, st2 = engine.verify_records(rec, pts, n_local, groups)
        _tick(timings, "stage2_ms", t0)
        st1, st2, ok = np.asarray(st_local, dtype=np.int64), np.asarray(st2, dtype=np.int64), np.asarray(ok, dtype=bool)
        status = np.where(st1 != 0, st1, st2)
        return (ok & (status == 0)).tolist(), status.tolist()
    rank = dist.get_rank(group)
    rec_b = n_local * RECORD                                    # bytes per batch: records, points
    pts_b = 0 if pts is None else 2 * n_local * POINT
    rec2 = rec.view(groups, rec_b)
    pts2 = None if pts is None else pts.view(groups, pts_b)
    on_host = _on_host(group) and rec.is_cuda
    if mode in ("allgather", "allgather_split"):
        # ONE all-gather: [records | points | stage-1 statuses (int32 per batch)] of every rank to every rank
        st_bytes = torch.from_numpy(np.asarray(st_local, dtype=np.int32).view(np.uint8).copy())
        parts = [rec.reshape(-1)] + ([pts.reshape(-1)] if pts is not None else [])
        send = torch.cat(parts + [st_bytes.to(dev)])
        per = send.numel()
        if on_host:
            host = torch.empty(world * per, dtype=torch.uint8)
            dist.all_gather_into_tensor(host, send.cpu(), group=group)
            got = host.to(dev)
        else:
            got = torch.empty(world * per, dtype=torch.uint8, device=dev)
            dist.all_gather_into_tensor(got, send, group=group)
        per_src = got.view(world, per)
        # [rank][batch][n_local*160] -> [batch][rank][n_local*160]: transcript order (contiguous blocks of blobs per rank)
        recs = per_src[:, :groups * rec_b].reshape(world, groups, rec_b).permute(1, 0, 2).contiguous().view(-1)
        points = None
        if pts is not None:     # [rank][batch][C | proofs][n_local] -> [batch][C | proofs][rank][n_local]
            points = per_src[:, groups * rec_b:groups * (rec_b + pts_b)].reshape(world, groups, 2, n_local * POINT).permute(1, 2, 0, 3).contiguous().view(-1)
        st_all = per_src[:, groups * (rec_b + pts_b):].contiguous().cpu().numpy().view(np.int32).reshape(world, groups).astype(np.int64)     # (also orders the stream: the permutes are done)
        t0 = _tick(timings, "exchange_ms", t0, dev)
        ok, st2 = engine.verify_records(recs, points, n_local * world, groups)           # every rank: all the batches
        t0 = _tick(timings, "stage2_ms", t0)
        st1 = st_all.max(axis=0)                                    # an Err on any rank's shard is an Err of the batch (the `?`s of kzg.rs:673-682)
        st2, ok = np.asarray(st2, dtype=np.int64), np.asarray(ok, dtype=bool)
        status = np.where(st1 != 0, st1, st2)
        _tick(timings, "merge_ms", t0)
        return (ok & (status == 0)).tolist(), status.tolist()
    # The ONE data-path collective: an all-to-all.  Stage 2 is split by batch, so rank j needs the records (and decoded points) of
    # the batches in ITS share only, from every rank: rank i sends rank j the slice [g_lo_j, g_hi_j) of its records | points.  An
    # all-gather would deliver every rank's whole shard to everybody -- world x the bytes, over a ring; here every pair of ranks
    # exchanges 1 / world of a shard over its own xGMI link (at 8 ranks and 8192 batches per step: 25 MB per link instead of
    # 1.4 GB around the ring).
    shares = [((groups * r) // world, (groups * (r + 1)) // world) for r in range(world)]
    g_lo, g_hi = shares[rank]
    mine = g_hi - g_lo
    parts = []
    for lo, hi in shares:
        parts.append(rec2[lo:hi].reshape(-1))
        if pts2 is not None:
            parts.append(pts2[lo:hi].reshape(-1))
    send = torch.cat(parts)
    in_splits = [(hi - lo) * (rec_b + pts_b) for lo, hi in shares]
    out_splits = [mine * (rec_b + pts_b)] * world
    if on_host:
        host = torch.empty(sum(out_splits), dtype=torch.uint8)
        dist.all_to_all_single(host, send.cpu(), out_splits, in_splits, group=group)
        got = host.to(dev)
    else:
        got = torch.empty(sum(out_splits), dtype=torch.uint8, device=dev)
        dist.all_to_all_single(got, send, out_splits, in_splits, group=group)
    # [0:G] stage-1 status, [G:2G] 1 + ok + 256 * stage-2 status of this rank's share: assembled on the host, ONE upload
    code_h = np.zeros(2 * groups, dtype=np.int32)
    code_h[:groups] = np.asarray(st_local, dtype=np.int32)
    if mine > 0:
        per_src = got.view(world, mine * (rec_b + pts_b))          # from rank i: [records of my batches | points of my batches]
        # [rank][batch][n_local*160] -> [batch][rank][n_local*160]: transcript order (contiguous blocks of blobs per rank)
        recs = per_src[:, :mine * rec_b].reshape(world, mine, rec_b).permute(1, 0, 2).contiguous().view(-1)
        points = None
        if pts is not None:     # [rank][batch][C | proofs][n_local] -> [batch][C | proofs][rank][n_local]
            points = per_src[:, mine * rec_b:].reshape(world, mine, 2, n_local * POINT).permute(1, 2, 0, 3).contiguous().view(-1)
        if rec.is_cuda:
            torch.cuda.synchronize(dev)                            # the permutes ran on torch's stream, the engine has its own
        t0 = _tick(timings, "exchange_ms", t0)
        ok, st2 = engine.verify_records(recs, points, n_local * world, mine)
        t0 = _tick(timings, "stage2_ms", t0)
        code_h[groups + g_lo:groups + g_hi] = 1 + np.asarray(ok, dtype=np.int32) + 256 * np.asarray(st2, dtype=np.int32)
    else:
        t0 = _tick(timings, "exchange_ms", t0, dev)
    code = torch.from_numpy(code_h)
    if not on_host and rec.is_cuda:
        code = code.to(dev)
    dist.all_reduce(code, op=dist.ReduceOp.MAX, group=group)        # verdicts of every share + status merge, one small collective
    code = code.cpu().numpy().astype(np.int64)                      # the one read-back
    st1, enc = code[:groups], code[groups:]
    status = np.where(st1 != 0, st1, enc >> 8)
    _tick(timings, "merge_ms", t0)
    return (((enc & 0xFF) == 2) & (status == 0)).tolist(), status.tolist()


def _sharded_device_words(local_blobs, local_commitments, local_proofs, n_local, groups, engine, group, mode, timings, world, capture=None):
    """The N > 1 path over an engine that leaves its per-batch results on the device (HipEngine): statuses and verdicts travel as int32 words in
    device memory -- through the all-gather's buffer or the all-reduce -- and are read back ONCE, at the end; a failure of a whole engine call is
    a status on every batch of that rank, seen by all ranks after the merge (nobody raises in the middle of a collective sequence).
    Raises RuntimeError on every rank together when a merged status is a device-level failure (KZG355_NO_DEVICE and above): the all-to-all form
    merges all words in its all-reduce; the all-gather form carries the stage-1 words in the gathered buffer and all-reduces ONE word for stage 2
    (replicated there: only a failure needs telling)."""
    import time
    import numpy as np
    import torch
    import torch.distributed as dist
    dev = local_blobs.device
    rank = dist.get_rank(group)
    on_host = _on_host(group)
    rec_b, pts_b = n_local * RECORD, 2 * n_local * POINT
    t0 = time.perf_counter()
    # [0:G] stage-1 status of this rank's shard, [G:2G] 1 + ok + 256 * stage-2 status of this rank's share of the batches (0 elsewhere)
    code = torch.zeros(2 * groups, dtype=torch.int32, device=dev)
    on_gpu = dev.type == "cuda"
    if on_gpu:
        torch.cuda.current_stream(dev).synchronize()               # the fill is on torch's stream, the engine writes on its own
    rec, pts = engine.shard_records_words(local_blobs, local_commitments, local_proofs, n_local, groups, code[:groups])
    t0 = _tick(timings, "stage1_ms", t0)

    def collective(fn, out_numel, send, *splits):
        if on_host and not on_gpu:                                 # host tensors over gloo: as they are
            got = torch.empty(out_numel, dtype=send.dtype)
            fn(got, send, *splits, group=group)
            return got
        if on_host:                                                # gloo: staged through host memory (CPU tests, rehearsals on one GPU)
            host = torch.empty(out_numel, dtype=send.dtype)
            fn(host, send.cpu(), *splits, group=group)
            return host.to(dev)
        got = torch.empty(out_numel, dtype=send.dtype, device=dev)
        fn(got, send, *splits, group=group)
        return got

    if mode == "allgather_split":
        send = torch.cat([rec, pts, code[:groups].view(torch.uint8)])
        per = send.numel()
        per_src = collective(dist.all_gather_into_tensor, world * per, send).view(world, per)
        shares = [((groups * r) // world, (groups * (r + 1)) // world) for r in range(world)]
        g_lo, g_hi = shares[rank]
        mine = g_hi - g_lo
        code[:groups] = per_src[:, groups * (rec_b + pts_b):].contiguous().view(torch.int32).view(world, groups).max(dim=0).values     # stage-1 statuses, merged: the same on every rank
        if mine > 0:                                               # only this rank's batches are permuted into transcript order
            recs = per_src[:, :groups * rec_b].reshape(world, groups, rec_b)[:, g_lo:g_hi].permute(1, 0, 2).contiguous().view(-1)
            points = per_src[:, groups * rec_b:groups * (rec_b + pts_b)].reshape(world, groups, 2, n_local * POINT)[:, g_lo:g_hi].permute(1, 2, 0, 3).contiguous().view(-1)
            if on_gpu:
                torch.cuda.synchronize(dev)
            t0 = _tick(timings, "exchange_ms", t0)
            if capture is not None:
                capture["records"], capture["share"] = recs, (g_lo, g_hi)
            engine.verify_records_words(recs, points, n_local * world, mine, code[groups + g_lo:groups + g_hi])
            t0 = _tick(timings, "stage2_ms", t0)
        else:
            if on_gpu:
                torch.cuda.synchronize(dev)
            t0 = _tick(timings, "exchange_ms", t0, dev)
        if on_host:
            host = code.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
            merged = host.numpy().astype(np.int64)
        else:
            dist.all_reduce(code, op=dist.ReduceOp.MAX, group=group)
            merged = code.cpu().numpy().astype(np.int64)
    elif mode == "allgather":
        send = torch.cat([rec, pts, code[:groups].view(torch.uint8)])
        per = send.numel()
        per_src = collective(dist.all_gather_into_tensor, world * per, send).view(world, per)
        recs = per_src[:, :groups * rec_b].reshape(world, groups, rec_b).permute(1, 0, 2).contiguous().view(-1)
        points = per_src[:, groups * rec_b:groups * (rec_b + pts_b)].reshape(world, groups, 2, n_local * POINT).permute(1, 2, 0, 3).contiguous().view(-1)
        st1 = per_src[:, groups * (rec_b + pts_b):].contiguous().view(torch.int32).view(world, groups).max(dim=0).values
        if on_gpu:
            torch.cuda.synchronize(dev)                            # the permutes ran on torch's stream, the engine has its own
        t0 = _tick(timings, "exchange_ms", t0)
        if capture is not None:
            capture["records"], capture["share"] = recs, (0, groups)
        engine.verify_records_words(recs, points, n_local * world, groups, code[groups:])      # every rank: all the batches
        t0 = _tick(timings, "stage2_ms", t0)
        # Stage 2 is replicated, so its verdicts need no merge -- but a whole-call failure of ONE rank's stage 2 (out of memory: every rank runs
        # world x the stage-2 work) would be that rank's alone, and it would raise while the others walk into the next collective.
        # One 4-byte word travels: the largest device-level status (>= 6) any rank's stage 2 left; every rank then raises, or none does.
        w2 = code[groups:] >> 8
        fail = torch.where(w2 >= 6, w2, torch.zeros_like(w2)).max().reshape(1)
        if on_host:
            fail_h = fail.cpu()
            dist.all_reduce(fail_h, op=dist.ReduceOp.MAX, group=group)
            fail = fail_h.to(dev)
        else:
            dist.all_reduce(fail, op=dist.ReduceOp.MAX, group=group)
        merged = torch.cat([st1, code[groups:], fail]).cpu().numpy().astype(np.int64)          # the one read-back
        if merged[-1] >= 6:
            raise RuntimeError(f"sharded verification: engine failure in stage 2 on a rank (status {int(merged[-1])}); every rank raises")
        merged = merged[:-1]
    else:
        shares = [((groups * r) // world, (groups * (r + 1)) // world) for r in range(world)]
        g_lo, g_hi = shares[rank]
        mine = g_hi - g_lo
        rec2, pts2 = rec.view(groups, rec_b), pts.view(groups, pts_b)
        send = torch.cat([t for lo, hi in shares for t in (rec2[lo:hi].reshape(-1), pts2[lo:hi].reshape(-1))])
        in_splits = [(hi - lo) * (rec_b + pts_b) for lo, hi in shares]
        out_splits = [mine * (rec_b + pts_b)] * world
        got = collective(dist.all_to_all_single, sum(out_splits), send, out_splits, in_splits)
        if mine > 0:
            per_src = got.view(world, mine * (rec_b + pts_b))      # from rank i: [records of my batches | points of my batches]
            recs = per_src[:, :mine * rec_b].reshape(world, mine, rec_b).permute(1, 0, 2).contiguous().view(-1)
            points = per_src[:, mine * rec_b:].reshape(world, mine, 2, n_local * POINT).permute(1, 2, 0, 3).contiguous().view(-1)
            if on_gpu:
                torch.cuda.synchronize(dev)
            t0 = _tick(timings, "exchange_ms", t0)
            if capture is not None:
                capture["records"], capture["share"] = recs, (g_lo, g_hi)
            engine.verify_records_words(recs, points, n_local * world, mine, code[groups + g_lo:groups + g_hi])
            t0 = _tick(timings, "stage2_ms", t0)
        else:
            t0 = _tick(timings, "exchange_ms", t0, dev)
        if on_host:
            host = code.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
            merged = host.numpy().astype(np.int64)
        else:
            dist.all_reduce(code, op=dist.ReduceOp.MAX, group=group)    # verdicts of every share + status merge, one small collective on the device
            merged = code.cpu().numpy().astype(np.int64)                # the one read-back
    st1, enc = merged[:groups], merged[groups:]
    status = np.where(st1 != 0, st1, enc >> 8)
    _tick(timings, "merge_ms", t0)
    # KZG355_NO_DEVICE / NO_MEMORY / DEVICE_ERROR on some rank, in either stage (a stage-2 failure of a batch that already carries a stage-1 Err must not hide
    # behind it): every rank has the same merged words
    device_level = np.concatenate([st1[st1 >= 6], (enc >> 8)[(enc >> 8) >= 6]])
    if device_level.size:
        raise RuntimeError(f"sharded verification: engine failure on a rank, merged per-batch statuses {sorted(set(device_level.tolist()))}")
    return (((enc & 0xFF) == 2) & (status == 0)).tolist(), status.tolist()
