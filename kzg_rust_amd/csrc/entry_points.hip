// entry_points.hip -- the C entry points of include/kzg355.h: submit / collect, device-resident calls, host-buffer calls (the drop-in surface) (host side of
// libkzg355.so; see engine.h).
#include "engine.h"

namespace kzg355_impl {

// Stage 2 of a submitted set, onto the handle's tail stream: behind the set's own stage 1 (ev_stage) and, when the next set's hash has just
// been queued, behind that as well (after).  Called with s->pipe_mu held.
int queue_tail(kzg355_ticket *t, hipEvent_t after) {
    Workspace *w = t->w;
    t->tail_queued = true;
    w->stream = t->s->pipe_tail;
    HIPCHK(hipStreamWaitEvent(w->stream, w->ev_stage, 0));
    if (after) HIPCHK(hipStreamWaitEvent(w->stream, after, 0));
    int rc = verify_enqueue_stage2(t->s, w, *t->tm, (int)t->npg, (int)t->groups);
    if (rc) return rc;
    HIPCHK(hipEventRecord(w->ev_done, w->stream));
    return KZG355_OK;
}

// d_words (device, or null): the per-batch statuses are left on the device instead of being copied to `status` (host, then null)
static int shard_records_impl(uint8_t *d_records, uint8_t *d_points, int *status, const uint8_t *d_blobs, const uint8_t *d_commitments,
                              const uint8_t *d_proofs, size_t n_local, size_t groups, const kzg355_settings *cs, int32_t *d_words = nullptr) {
    if (!cs || (!status && !d_words)) return KZG355_BADARGS;
    std::vector<int> unused;
    if (!status) { unused.assign(groups ? groups : 1, KZG355_OK); status = unused.data(); }
    for (size_t i = 0; i < groups; i++) status[i] = KZG355_OK;
    if (n_local == 0 || groups == 0) return KZG355_OK;
    // a refusal of the call as a whole writes nothing to d_records / d_points: every batch carries the status, so that a caller that
    // reads per-batch statuses cannot mistake it for success
    auto refuse = [&](int code) { for (size_t i = 0; i < groups; i++) status[i] = code; return code; };
    if (n_local > (size_t)1 << 24 || groups > (size_t)1 << 24 || n_local * groups > (size_t)1 << 24) return refuse(KZG355_BADARGS);
    if (!d_records || ((uintptr_t)d_records & 15) || ((uintptr_t)d_points & 3) || !d_blobs || ((uintptr_t)d_blobs & 15) || !d_commitments ||
            !d_proofs || ((uintptr_t)d_words & 3)) return refuse(KZG355_BADARGS);      // (before anything is queued: a refusal leaves no work in flight)
    WsGuard g(cs);
    if (!g.w) return KZG355_NO_DEVICE;
    kzg355_settings *s = g.s; Workspace *w = g.w;
    int rc;
    if ((rc = w->err.ensure(sizeof(int) * groups))) return rc;
    if ((rc = w->h_err.ensure(sizeof(int) * groups))) return rc;
    w->in_flight = true;
    HIPCHK(hipMemsetAsync(w->err.p, 0, sizeof(int) * groups, w->stream));
    Timed tm(s, w);
    // any error poisons its batch, as the `?`s at kzg.rs:673-682 do for the call
    // (no window shifts here: stage 2 runs on the gathered batch, on whichever rank gets it)
    // (a small shard -- BASELINE config 5 gives every rank 64 blobs of the one batch -- takes the host-hash route of the device-resident calls:
    // the 3.7 ms device hash chain would be the whole of such a rank's stage 1)
    HostFront hf;
    const bool via_host = device_call_hashes_on_host(s, n_local * groups);
    if (via_host) { hf.from_device = true; hf.d_commitments = d_commitments; hf.n_blobs = n_local * groups; }
    if ((rc = run_stage1(s, w, tm, d_blobs, d_commitments, d_proofs, (int)(n_local * groups), (int)n_local, d_records, reinterpret_cast<G1Affine *>(d_points),
            w->err.as<int>(), false,
                         via_host ? &hf : nullptr))) return rc;
    if ((rc = join_side(w))) return rc;
    if (d_words) {                                                // statuses stay on the device: no copy back, the caller reads them after its merge
        launch_status_words(w->err.as<int>(), nullptr, d_words, (int)groups, w->stream);
        HIPCHK(hipStreamSynchronize(w->stream));
        w->in_flight = false;
        tm.collect();
        return KZG355_OK;
    }
    HIPCHK(hipMemcpyAsync(w->h_err.p, w->err.p, sizeof(int) * groups, hipMemcpyDeviceToHost, w->stream));
    HIPCHK(hipStreamSynchronize(w->stream));
    w->in_flight = false;
    tm.collect();
    int first = KZG355_OK;
    for (size_t i = 0; i < groups; i++) {
        status[i] = status_from_err(w->h_err.as<int>()[i]);
        if (status[i] != KZG355_OK && first == KZG355_OK) first = status[i];
    }
    return first;
}

// stage 2 over gathered records; `dump` (host, groups*128 bytes or null) receives r | proof_lincomb | rhs per batch; d_points (or
// null): the validated affine points of the records as stage 1 produced them ([batch][commitments, proofs]), sparing their decompression
static int verify_records_impl(bool *ok, int *status, uint8_t *dump, const uint8_t *d_records, size_t n, size_t groups, int validate, const kzg355_settings *cs,
                        const uint8_t *d_points = nullptr, int32_t *d_words = nullptr) {
    if (!cs || (!ok && !d_words)) return KZG355_BADARGS;
    if (groups == 0) return KZG355_OK;
    auto refuse = [&](int code) { if (status) for (size_t i = 0; i < groups; i++) status[i] = code; return code; };      // whole-call refusals mark every batch
    if (n == 0) return refuse(KZG355_BADARGS);                   // verify_kzg_proof_batch: n == 0 is an error (kzg.rs:588-592)
    if (n > (size_t)1 << 24 || groups > (size_t)1 << 24 || n * groups > (size_t)1 << 24) return refuse(KZG355_BADARGS);
    // the kernels read the records 16 bytes at a time
    if (!d_records || ((uintptr_t)d_records & 15) || ((uintptr_t)d_points & 3) || ((uintptr_t)d_words & 3)) return refuse(KZG355_BADARGS);
    WsGuard g(cs);
    if (!g.w) return KZG355_NO_DEVICE;
    kzg355_settings *s = g.s; Workspace *w = g.w;
    int rc;
    const int G = (int)groups;
    if (!d_points && (rc = w->pts.ensure(sizeof(G1Affine) * 2 * n * groups))) return rc;
    const G1Affine *pts = d_points ? reinterpret_cast<const G1Affine *>(d_points) : w->pts.as<G1Affine>();
    if ((rc = w->err.ensure(sizeof(int) * groups))) return rc;
    if ((rc = w->ok.ensure(sizeof(int) * groups))) return rc;
    if ((rc = w->h_ok.ensure(sizeof(int) * groups))) return rc;
    if ((rc = w->h_err.ensure(sizeof(int) * groups))) return rc;
    if (dump && ((rc = w->out48.ensure(128 * groups)) || (rc = w->h_out.ensure(128 * groups)))) return rc;
    w->in_flight = true;
    HIPCHK(hipMemsetAsync(w->err.p, 0, sizeof(int) * groups, w->stream));
    Timed tm(s, w);
    if (d_points) {
        // nothing to decode
    } else if (validate) {   // full validate_kzg_g1 on C_i / proof_i (decompression + subgroup test) straight from the records
        tm.begin("validate_points");
        launch_validate_points(d_records, d_records + 112, (int)(n * groups), (int)n, w->pts.as<G1Affine>(), w->err.as<int>(), w->stream, RECORD_BYTES);
        tm.end();
    } else {
        tm.begin("points_from_records"); launch_points_from_records(d_records, (int)(n * groups), (int)n, w->pts.as<G1Affine>(), w->err.as<int>(), w->stream);
                tm.end();
    }
    if ((rc = run_stage2(s, w, tm, d_records, (int)n, G, validate, pts, w->err.as<int>(), w->ok.as<int>()))) return rc;
    if (dump) {
        launch_dump_intermediates(w->scal_a.as<uint32_t>(), w->pair_pts.as<PairPt>(), (int)n, G, w->out48.as<uint8_t>(), w->stream);
        HIPCHK(hipMemcpyAsync(w->h_out.p, w->out48.p, 128 * groups, hipMemcpyDeviceToHost, w->stream));
    }
    // 1 + ok + 256 * status per batch, left on the device (the sharded path's all-reduce takes them from there)
    if (d_words) {
        launch_status_words(w->err.as<int>(), w->ok.as<int>(), d_words, G, w->stream);
        HIPCHK(hipStreamSynchronize(w->stream));
        w->in_flight = false;
        tm.collect();
        return KZG355_OK;
    }
    HIPCHK(hipMemcpyAsync(w->h_ok.p, w->ok.p, sizeof(int) * groups, hipMemcpyDeviceToHost, w->stream));
    HIPCHK(hipMemcpyAsync(w->h_err.p, w->err.p, sizeof(int) * groups, hipMemcpyDeviceToHost, w->stream));
    HIPCHK(hipStreamSynchronize(w->stream));
    w->in_flight = false;
    tm.collect();
    if (dump) memcpy(dump, w->h_out.p, 128 * groups);
    int first = KZG355_OK;
    for (int i = 0; i < G; i++) {
        int st = status_from_err(w->h_err.as<int>()[i]);
        if (status) status[i] = st;
        if (st == KZG355_OK) ok[i] = w->h_ok.as<int>()[i] != 0;
        else if (first == KZG355_OK) first = st;
    }
    return first;
}

// n independent verify_kzg_proof checks from host arrays, on the one device of `cs` (a replica of a multi-device handle included).  The four arrays
// become the 160-byte records C | z | y | proof (the layout stage 2 reads: utils.rs:454-463) in a pinned slot, in launch sets of at most 2^17 checks
// (21 MB of records, ~0.8 GB of scratch); the staging workspace only lends its buffers, the checks run on a second one.
int verify_proofs_on_one_device(bool *ok, int *status, const uint8_t *commitments, const uint8_t *zs, const uint8_t *ys, const uint8_t *proofs, size_t n,
                                const kzg355_settings *cs) {
    WsGuard g(cs);
    if (!g.w) return KZG355_NO_DEVICE;
    Workspace *w = g.w;
    const size_t SET = (size_t)1 << 17;
    int first = KZG355_OK, rc;
    for (size_t lo = 0; lo < n; lo += SET) {
        const size_t cnt = n - lo < SET ? n - lo : SET;
        if ((rc = w->h_records.ensure((size_t)RECORD_BYTES * cnt)) || (rc = w->records.ensure((size_t)RECORD_BYTES * cnt))) return rc;
        uint8_t *rec = w->h_records.as<uint8_t>();
        for (size_t i = 0; i < cnt; i++, rec += RECORD_BYTES) {
            memcpy(rec, commitments + 48 * (lo + i), 48); memcpy(rec + 48, zs + 32 * (lo + i), 32); memcpy(rec + 80, ys + 32 * (lo + i), 32);
            memcpy(rec + 112, proofs + 48 * (lo + i), 48);
        }
        w->in_flight = true;
        HIPCHK(hipMemcpyAsync(w->records.p, w->h_records.p, (size_t)RECORD_BYTES * cnt, hipMemcpyHostToDevice, w->stream));
        HIPCHK(hipStreamSynchronize(w->stream));
        w->in_flight = false;
        rc = verify_records_impl(ok + lo, status ? status + lo : nullptr, nullptr, w->records.as<uint8_t>(), 1, cnt, 1, cs);
        if (rc == KZG355_NO_DEVICE || rc == KZG355_NO_MEMORY || rc == KZG355_DEVICE_ERROR || rc == KZG355_INTERNAL) return rc;
        if (rc != KZG355_OK && first == KZG355_OK) first = rc;
    }
    return first;
}

}  // namespace kzg355_impl

extern "C" {
#pragma GCC visibility push(default)

int kzg355_verify_blob_kzg_proof_batch_many_device_submit(kzg355_ticket **ticket, const uint8_t *d_blobs, const uint8_t *d_commitments,
                                                          const uint8_t *d_proofs, size_t n_per_group, size_t groups, const kzg355_settings *cs) {
    if (!ticket || !cs) return KZG355_BADARGS;
    *ticket = nullptr;
    // (each factor first: the product must not wrap)
    if (n_per_group > (size_t)1 << 24 || groups > (size_t)1 << 24 || n_per_group * groups > (size_t)1 << 24) return KZG355_BADARGS;
    std::unique_ptr<kzg355_ticket> t(new kzg355_ticket());
    kzg355_settings *s = t->s = const_cast<kzg355_settings *>(cs);
    t->npg = n_per_group; t->groups = groups;
    if (groups == 0 || n_per_group == 0) { t->immediate = true; *ticket = t.release(); return KZG355_OK; }
    if (!d_blobs || !d_commitments || !d_proofs || ((uintptr_t)d_blobs & 15) || ((uintptr_t)d_commitments & 3) ||
            ((uintptr_t)d_proofs & 3)) return KZG355_BADARGS;
    DeviceScope scope;
    if (!scope.enter(s->device)) return KZG355_NO_DEVICE;
    Workspace *w = t->w = ws_acquire(s);
    if (!w) return KZG355_NO_DEVICE;
    t->tm.reset(new Timed(s, w));
    auto fail = [&](int rc) { w->quiesce(); ws_release(s, w); return rc; };
    // Small sets (up to two launch sets' worth of the "point kernels beside the hash" regime: 512 batches of 64 on 256 CUs) are chains of
    // kernels that leave most of the card idle: each on the stream of its own workspace, they simply overlap (measured, blobs/s with one /
    // four sets in flight: 256 batches 1.54 -> 1.89 M, 512 batches 2.17 -> 2.80 M; the two-stage pipeline below: 1.86 M, 2.59 M).
    if (s->submit_mode == 1 || (s->submit_mode == 0 && n_per_group * groups <= 2 * (size_t)s->beside_max_blobs)) {
        int rc = verify_enqueue(s, w, *t->tm, d_blobs, d_commitments, d_proofs, (int)n_per_group, (int)groups);
        if (rc == KZG355_OK && hipEventRecord(w->ev_done, w->stream) != hipSuccess) rc = KZG355_DEVICE_ERROR;
        if (rc) return fail(rc);
        t->tail_queued = true;
        s->tickets_out++;
        *ticket = t.release();
        return KZG355_OK;
    }
    // Two-stage software pipeline.  Stage 1 of the submitted sets runs in submission order on ONE stream (two hash or evaluation kernels
    // side by side gain nothing: each fills the card).  Stage 2 of set k goes to a second stream, and it is queued LATE: when set k + 1 is
    // submitted, right behind that set's Fiat-Shamir kernel -- so it runs beside the evaluation and point kernels of set k + 1 and never
    // beside the hash, whose 11 KB loop the instruction streams of the point-arithmetic kernels evict from the instruction cache
    // (measured, profiles/r04/pipeline_sweep.txt: hash 5.9 -> 12.3 ms next to the bucket kernel, a step slower than the two in a row) --
    // or when set k is collected, whichever comes first.
    std::lock_guard<std::mutex> lk(s->pipe_mu);
    if (!s->pipe_main) {
        if (hipStreamCreateWithFlags(&s->pipe_main, hipStreamNonBlocking) != hipSuccess) { s->pipe_main = nullptr; (void)hipGetLastError();
                return fail(KZG355_DEVICE_ERROR); }
        if (hipStreamCreateWithFlags(&s->pipe_tail, hipStreamNonBlocking) != hipSuccess) { s->pipe_tail = nullptr; (void)hipGetLastError();
                return fail(KZG355_DEVICE_ERROR); }
    }
    w->stream = s->pipe_main; w->borrowed[0] = s->pipe_main; w->borrowed[1] = s->pipe_tail;
    kzg355_ticket *prev = s->pending_tail;
    const std::function<int()> after_challenge = [&]() -> int {
        if (!prev) return KZG355_OK;
        s->pending_tail = nullptr;
        HIPCHK(hipEventRecord(w->ev_fork2, w->stream));           // this set's hash is queued up to here
        prev->tail_rc = queue_tail(prev, w->ev_fork2);            // (a failure is reported when that set is collected)
        return KZG355_OK;
    };
    int rc = verify_enqueue_stage1(s, w, *t->tm, d_blobs, d_commitments, d_proofs, (int)n_per_group, (int)groups, 0, nullptr, &after_challenge);
    if (rc == KZG355_OK && hipEventRecord(w->ev_stage, w->stream) != hipSuccess) rc = KZG355_DEVICE_ERROR;
    if (rc) {
        // the earlier set must not wait for a set that never came
        if (s->pending_tail == prev && prev) { s->pending_tail = nullptr; prev->tail_rc = queue_tail(prev, nullptr); }
        return fail(rc);
    }
    s->pending_tail = t.get();
    s->tickets_out++;
    *ticket = t.release();
    return KZG355_OK;
}

int kzg355_verify_collect(kzg355_ticket *ticket, bool *ok, int *status) {
    if (!ticket) return KZG355_BADARGS;
    std::unique_ptr<kzg355_ticket> t(ticket);                    // the ticket is consumed whatever happens
    if (t->immediate) {
        if (!ok && t->groups) return KZG355_BADARGS;
        for (size_t g = 0; g < t->groups; g++) { ok[g] = true; if (status) status[g] = KZG355_OK; }
        return KZG355_OK;
    }
    kzg355_settings *s = t->s; Workspace *w = t->w;
    DeviceScope scope;
    const bool entered = scope.enter(s->device);
    int rc = !entered ? KZG355_NO_DEVICE : KZG355_OK;
    {
        std::lock_guard<std::mutex> lk(s->pipe_mu);
        if (s->pending_tail == t.get()) {                         // no later set came: stage 2 goes out now
            s->pending_tail = nullptr;
            if (rc == KZG355_OK) t->tail_rc = queue_tail(t.get(), nullptr);
        }
    }
    if (rc == KZG355_OK) rc = t->tail_rc;
    if (rc == KZG355_OK && !ok) rc = KZG355_BADARGS;
    if (rc == KZG355_OK && w->borrowed[0]) {
        // the set sits on the handle's shared pipeline streams: wait for ITS end (later sets may be queued behind it), then hand the
        // workspace its own stream back, so that verify_collect's stream wait returns at once
        if (hipEventSynchronize(w->ev_done) != hipSuccess) rc = KZG355_DEVICE_ERROR;
        else { w->borrowed[0] = w->borrowed[1] = nullptr; w->stream = w->own_stream; }
    }
    if (rc == KZG355_OK) rc = verify_collect(w, *t->tm, ok, status, (int)t->groups);
    w->quiesce();
    ws_release(s, w);
    t.reset();
    bool free_now;
    { std::lock_guard<std::mutex> lk(s->pipe_mu); free_now = --s->tickets_out == 0 && s->free_deferred; }
    if (free_now) free_single(s);                                 // the handle was freed while this ticket was out
    return rc;
}

// ---- device-resident entry points ---------------------------------------------------------------------
int kzg355_verify_blob_kzg_proof_batch_many_device(bool *ok, int *status, const uint8_t *d_blobs, const uint8_t *d_commitments,
                                                   const uint8_t *d_proofs, size_t n_per_group, size_t groups, const kzg355_settings *s) {
    return verify_many_device_impl(ok, status, d_blobs, d_commitments, d_proofs, n_per_group, groups, s);
}
int kzg355_blob_to_kzg_commitment_many_device(uint8_t *out, int *status, const uint8_t *d_blobs, size_t n, const kzg355_settings *s) {
    return msm_op_many_device_impl(out, status, d_blobs, nullptr, n, s);
}
int kzg355_compute_blob_kzg_proof_many_device(uint8_t *out, int *status, const uint8_t *d_blobs, const uint8_t *d_commitments, size_t n,
                                              const kzg355_settings *s) {
    if (!d_commitments) return KZG355_BADARGS;
    return msm_op_many_device_impl(out, status, d_blobs, d_commitments, n, s);
}

int kzg355_verify_shard_records_device(uint8_t *d_records, int *status, const uint8_t *d_blobs, const uint8_t *d_commitments,
                                       const uint8_t *d_proofs, size_t n_local, size_t groups, const kzg355_settings *cs) {
    return shard_records_impl(d_records, nullptr, status, d_blobs, d_commitments, d_proofs, n_local, groups, cs);
}
int kzg355_verify_shard_records_points_words_device(uint8_t *d_records, uint8_t *d_points, int32_t *d_status_words, const uint8_t *d_blobs,
                                                    const uint8_t *d_commitments, const uint8_t *d_proofs, size_t n_local, size_t groups,
                                                            const kzg355_settings *cs) {
    if (!d_points || !d_status_words) return KZG355_BADARGS;
    return shard_records_impl(d_records, d_points, nullptr, d_blobs, d_commitments, d_proofs, n_local, groups, cs, d_status_words);
}
int kzg355_verify_shard_records_points_device(uint8_t *d_records, uint8_t *d_points, int *status, const uint8_t *d_blobs, const uint8_t *d_commitments,
                                              const uint8_t *d_proofs, size_t n_local, size_t groups, const kzg355_settings *cs) {
    if (!d_points) return KZG355_BADARGS;
    return shard_records_impl(d_records, d_points, status, d_blobs, d_commitments, d_proofs, n_local, groups, cs);
}

int kzg355_verify_records_device(bool *ok, int *status, const uint8_t *d_records, size_t n, size_t groups, const kzg355_settings *cs) {
    return verify_records_impl(ok, status, nullptr, d_records, n, groups, 0, cs);
}
int kzg355_verify_records_points_device(bool *ok, int *status, const uint8_t *d_records, const uint8_t *d_points, size_t n, size_t groups,
        const kzg355_settings *cs) {
    if (!d_points) return KZG355_BADARGS;
    return verify_records_impl(ok, status, nullptr, d_records, n, groups, 0, cs, d_points);
}
int kzg355_verify_records_points_words_device(int32_t *d_words, const uint8_t *d_records, const uint8_t *d_points, size_t n, size_t groups,
        const kzg355_settings *cs) {
    if (!d_points || !d_words) return KZG355_BADARGS;
    return verify_records_impl(nullptr, nullptr, nullptr, d_records, n, groups, 0, cs, d_points, d_words);
}
int kzg355_verify_records_checked_device(bool *ok, int *status, const uint8_t *d_records, size_t n, size_t groups, const kzg355_settings *cs) {
    return verify_records_impl(ok, status, nullptr, d_records, n, groups, 1, cs);
}
int kzg355_debug_batch_intermediates(uint8_t *out, bool *ok, int *status, const uint8_t *d_records, size_t n, size_t groups, const kzg355_settings *cs) {
    if (!out) return KZG355_BADARGS;
    return verify_records_impl(ok, status, out, d_records, n, groups, 1, cs);
}

// ---- host-buffer entry points (the drop-in surface) ------------------------------------------------------
int kzg355_verify_blob_kzg_proof_batch_many(bool *ok, int *status, const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs,
                                            size_t n_per_group, size_t groups, const kzg355_settings *cs) {
    if (!cs || !ok) return KZG355_BADARGS;
    const size_t n = n_per_group * groups;
    if (n == 0) return verify_many_device_impl(ok, status, nullptr, nullptr, nullptr, n_per_group, groups, cs);
    if (!blobs || !commitments || !proofs) return KZG355_BADARGS;
    if (n > (size_t)1 << 24) return KZG355_BADARGS;
    if (cs->multi) return multi_verify_many(ok, status, blobs, commitments, proofs, n_per_group, groups, cs);
    return single_verify_many(ok, status, blobs, commitments, proofs, n_per_group, groups, cs);
}

int kzg355_debug_verify_host_records(uint8_t *records_out, bool *ok, int *status, const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs,
                                     size_t n_per_group, size_t groups, const kzg355_settings *cs) {
    if (!cs || !ok || !records_out || !blobs || !commitments || !proofs || n_per_group == 0 || groups == 0 || cs->multi) return KZG355_BADARGS;
    // one chunk
    if (n_per_group > (size_t)1 << 24 || groups > (size_t)1 << 24 || n_per_group * groups > (size_t)1 << 24 ||
            blob_bytes_of(cs) * n_per_group * groups > ((size_t)64 << 20)) return KZG355_BADARGS;
    HostCall hc{0, blobs, commitments, proofs, n_per_group, ok, nullptr, status};
    hc.records_out = records_out;
    return host_pipeline(hc, groups, cs);
}

int kzg355_debug_verify_sharded_intermediates(uint8_t *out, bool *ok, int *status, const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs,
                                              size_t n_per_group, size_t groups, const kzg355_settings *cs) {
    if (!cs || !ok || !out || !blobs || !commitments || !proofs || !cs->multi || groups == 0) return KZG355_BADARGS;
    // every device gets a block of every batch
    if (n_per_group < cs->multi->rep.size() || n_per_group > (size_t)1 << 24 || groups > (size_t)1 << 24 ||
            n_per_group * groups > (size_t)1 << 24) return KZG355_BADARGS;
    return multi_verify_sharded(ok, status, blobs, commitments, proofs, n_per_group, groups, cs, out);
}

int kzg355_verify_blob_kzg_proof_batch(bool *ok, const uint8_t *blobs, size_t n_blobs, const uint8_t *commitments, size_t n_commitments,
                                       const uint8_t *proofs, size_t n_proofs, const kzg355_settings *s) {
    if (!s || !ok) return KZG355_BADARGS;
    if (n_blobs != n_commitments || n_commitments != n_proofs) return KZG355_BADARGS;   // kzg.rs:644-651
    if (n_blobs == 0) { *ok = true; return KZG355_OK; }                                   // kzg.rs:653-655
    // n == 1 is the single-blob path in the reference (kzg.rs:658-660); the batch equation with r^0 = 1 is the same check
    bool r = false; int st = KZG355_OK;
    int rc = kzg355_verify_blob_kzg_proof_batch_many(&r, &st, blobs, commitments, proofs, n_blobs, 1, s);
    if (rc == KZG355_OK) *ok = r;
    return rc;
}

int kzg355_verify_blob_kzg_proof(bool *ok, const uint8_t *blob, const uint8_t commitment[48], const uint8_t proof[48], const kzg355_settings *s) {
    return kzg355_verify_blob_kzg_proof_batch(ok, blob, 1, commitment, 1, proof, 1, s);   // kzg.rs:547-569
}

int kzg355_verify_kzg_proof(bool *ok, const uint8_t commitment[48], const uint8_t z_bytes[32], const uint8_t y_bytes[32], const uint8_t proof[48],
                            const kzg355_settings *cs) {
    if (!cs || !ok || !commitment || !z_bytes || !y_bytes || !proof) return KZG355_BADARGS;
    WsGuard g(cs);
    if (!g.w) return KZG355_NO_DEVICE;
    kzg355_settings *s = g.s; Workspace *w = g.w;
    uint8_t rec[RECORD_BYTES];
    memcpy(rec, commitment, 48); memcpy(rec + 48, z_bytes, 32); memcpy(rec + 80, y_bytes, 32); memcpy(rec + 112, proof, 48);
    int rc;
    if ((rc = w->records.ensure(RECORD_BYTES))) return rc;
    if ((rc = w->pts.ensure(sizeof(G1Affine) * 2))) return rc;
    if ((rc = w->err.ensure(sizeof(int)))) return rc;
    if ((rc = w->ok.ensure(sizeof(int)))) return rc;
    if ((rc = w->h_ok.ensure(sizeof(int)))) return rc;
    if ((rc = w->h_err.ensure(sizeof(int)))) return rc;
    w->in_flight = true;                                          // (before the first copy from caller memory: a failure below drains the streams)
    HIPCHK(hipMemcpyAsync(w->records.p, rec, RECORD_BYTES, hipMemcpyHostToDevice, w->stream));
    HIPCHK(hipMemsetAsync(w->err.p, 0, sizeof(int), w->stream));
    Timed tm(s, w);
    const uint8_t *d_rec = w->records.as<uint8_t>();
    // bytes_to_kzg_commitment / bytes_to_kzg_proof (kzg.rs:436, 439): full validation incl. subgroup.  The decoding, the subgroup test (it
    // only feeds the error word) and the window shifts of the linear combination (from x alone) run beside each other on the side streams;
    // the main stream has the canonical checks of z and y (kzg.rs:437-438: k_rpowers with check_zy = 1; with one record r^0 = 1 and there is
    // no transcript to hash), then the linear combination C + [z] proof - [y] G and the pairing.
    w->shift_ready = false;
    if (ensure_side(s, w)) {
        if ((rc = enqueue_points_beside(s, w, tm, d_rec, d_rec + 112, 1, 1, w->pts.as<G1Affine>(), w->err.as<int>(), true, RECORD_BYTES))) return rc;
    } else {
        tm.begin("validate_points"); launch_validate_points(d_rec, d_rec + 112, 1, 1, w->pts.as<G1Affine>(), w->err.as<int>(), w->stream, RECORD_BYTES);
                tm.end();
    }
    if ((rc = run_stage2(s, w, tm, d_rec, 1, 1, 1, w->pts.as<G1Affine>(), w->err.as<int>(), w->ok.as<int>()))) return rc;
    if ((rc = join_side(w))) return rc;                           // the subgroup verdict, before the error word goes back
    HIPCHK(hipMemcpyAsync(w->h_ok.p, w->ok.p, sizeof(int), hipMemcpyDeviceToHost, w->stream));
    HIPCHK(hipMemcpyAsync(w->h_err.p, w->err.p, sizeof(int), hipMemcpyDeviceToHost, w->stream));
    HIPCHK(hipStreamSynchronize(w->stream));
    w->in_flight = false;
    tm.collect();
    int st = status_from_err(w->h_err.as<int>()[0]);
    if (st == KZG355_OK) *ok = w->h_ok.as<int>()[0] != 0;
    return st;
}

// ---- *_many forms of the single-proof functions (the reference benches one proof per call, benches/kzg_benches.rs:58-91; a GPU wants many)
int kzg355_verify_kzg_proof_many_device(bool *ok, int *status, const uint8_t *d_records, size_t n, const kzg355_settings *cs) {
    // n "batches" of one record each, full input validation: verify_kzg_proof (kzg.rs:429-443) per record
    return verify_records_impl(ok, status, nullptr, d_records, 1, n, 1, cs);
}

int kzg355_verify_kzg_proof_many(bool *ok, int *status, const uint8_t *commitments, const uint8_t *zs, const uint8_t *ys, const uint8_t *proofs, size_t n,
                                 const kzg355_settings *cs) {
    if (!cs || !ok) return KZG355_BADARGS;
    if (n == 0) return KZG355_OK;
    if (!commitments || !zs || !ys || !proofs || n > (size_t)1 << 24) return KZG355_BADARGS;
    if (cs->multi && cs->multi->rep.size() > 1 && n >= cs->multi->rep.size()) {      // independent checks: contiguous ranges per device
        MultiDev *m = cs->multi;
        return fan_out(m->rep.size(), n, [&](size_t d, size_t i0, size_t cnt) -> int {
            if (hipSetDevice(m->rep[d]->device) != hipSuccess) return KZG355_NO_DEVICE;
            return verify_proofs_on_one_device(ok + i0, status ? status + i0 : nullptr, commitments + 48 * i0, zs + 32 * i0, ys + 32 * i0, proofs + 48 * i0,
                    cnt,
                    m->rep[d]);
        });
    }
    return verify_proofs_on_one_device(ok, status, commitments, zs, ys, proofs, n, cs);
}

int kzg355_verify_blob_kzg_proof_many(bool *ok, int *status, const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs, size_t n,
                                      const kzg355_settings *cs) {
    // n batches of one blob: the batch equation with r^0 = 1 is verify_blob_kzg_proof's check (kzg.rs:547-569, 658-660)
    return kzg355_verify_blob_kzg_proof_batch_many(ok, status, blobs, commitments, proofs, 1, n, cs);
}

int kzg355_compute_kzg_proof_many_device(uint8_t *proofs_out, uint8_t *ys_out, int *status, const uint8_t *d_blobs, const uint8_t *d_zs, size_t n,
                                         const kzg355_settings *cs) {
    if (!d_zs || !ys_out) return KZG355_BADARGS;
    return msm_op_many_device_impl(proofs_out, status, d_blobs, nullptr, n, cs, d_zs, ys_out);
}

int kzg355_compute_kzg_proof_many(uint8_t *proofs_out, uint8_t *ys_out, int *status, const uint8_t *blobs, const uint8_t *zs, size_t n,
                                  const kzg355_settings *cs) {
    if (!cs || !proofs_out || !ys_out) return KZG355_BADARGS;
    if (n == 0) return KZG355_OK;
    if (!blobs || !zs || n > (size_t)1 << 20) return KZG355_BADARGS;
    if (cs->multi && cs->multi->rep.size() > 1 && n >= cs->multi->rep.size()) {
        MultiDev *m = cs->multi;
        const size_t BB = blob_bytes_of(cs);
        return fan_out(m->rep.size(), n, [&](size_t d, size_t i0, size_t cnt) -> int {
            if (hipSetDevice(m->rep[d]->device) != hipSuccess) return KZG355_NO_DEVICE;
            HostCall hc{3, blobs + BB * i0, nullptr, nullptr, 1, nullptr, proofs_out + 48 * i0, status ? status + i0 : nullptr};
            hc.zs = zs + 32 * i0; hc.ys_out = ys_out + 32 * i0;
            return host_pipeline(hc, cnt, m->rep[d]);
        });
    }
    HostCall hc{3, blobs, nullptr, nullptr, 1, nullptr, proofs_out, status};
    hc.zs = zs; hc.ys_out = ys_out;
    return host_pipeline(hc, n, cs);
}

int kzg355_blob_to_kzg_commitment_many(uint8_t *out, int *status, const uint8_t *blobs, size_t n, const kzg355_settings *cs) {
    if (!cs || !out) return KZG355_BADARGS;
    if (n == 0) return KZG355_OK;
    if (!blobs) return KZG355_BADARGS;
    if (n > (size_t)1 << 20) return KZG355_BADARGS;
    if (cs->multi && cs->multi->rep.size() > 1 && n >= cs->multi->rep.size()) {      // independent blobs: contiguous ranges per device
        MultiDev *m = cs->multi;
        const size_t BB = blob_bytes_of(cs);
        return fan_out(m->rep.size(), n, [&](size_t d, size_t i0, size_t cnt) -> int {
            if (hipSetDevice(m->rep[d]->device) != hipSuccess) return KZG355_NO_DEVICE;
            HostCall hc{1, blobs + BB * i0, nullptr, nullptr, 1, nullptr, out + 48 * i0, status ? status + i0 : nullptr};
            return host_pipeline(hc, cnt, m->rep[d]);
        });
    }
    HostCall hc{1, blobs, nullptr, nullptr, 1, nullptr, out, status};
    return host_pipeline(hc, n, cs);
}
int kzg355_blob_to_kzg_commitment(uint8_t out[48], const uint8_t *blob, const kzg355_settings *s) {
    int st = KZG355_OK;
    uint8_t tmp[48];
    int rc = kzg355_blob_to_kzg_commitment_many(tmp, &st, blob, 1, s);
    if (rc == KZG355_OK) memcpy(out, tmp, 48);
    return rc;
}

int kzg355_compute_blob_kzg_proof_many(uint8_t *out, int *status, const uint8_t *blobs, const uint8_t *commitments, size_t n, const kzg355_settings *cs) {
    if (!cs || !out) return KZG355_BADARGS;
    if (n == 0) return KZG355_OK;
    if (!blobs || !commitments) return KZG355_BADARGS;
    if (n > (size_t)1 << 20) return KZG355_BADARGS;
    if (cs->multi && cs->multi->rep.size() > 1 && n >= cs->multi->rep.size()) {
        MultiDev *m = cs->multi;
        const size_t BB = blob_bytes_of(cs);
        return fan_out(m->rep.size(), n, [&](size_t d, size_t i0, size_t cnt) -> int {
            if (hipSetDevice(m->rep[d]->device) != hipSuccess) return KZG355_NO_DEVICE;
            HostCall hc{2, blobs + BB * i0, commitments + 48 * i0, nullptr, 1, nullptr, out + 48 * i0, status ? status + i0 : nullptr};
            return host_pipeline(hc, cnt, m->rep[d]);
        });
    }
    HostCall hc{2, blobs, commitments, nullptr, 1, nullptr, out, status};
    return host_pipeline(hc, n, cs);
}
int kzg355_compute_blob_kzg_proof(uint8_t proof_out[48], const uint8_t *blob, const uint8_t commitment[48], const kzg355_settings *s) {
    int st = KZG355_OK;
    uint8_t tmp[48];
    int rc = kzg355_compute_blob_kzg_proof_many(tmp, &st, blob, commitment, 1, s);
    if (rc == KZG355_OK) memcpy(proof_out, tmp, 48);
    return rc;
}

int kzg355_compute_kzg_proof(uint8_t proof_out[48], uint8_t y_out[32], const uint8_t *blob, const uint8_t z_bytes[32], const kzg355_settings *cs) {
    if (!cs || !proof_out || !y_out || !blob || !z_bytes) return KZG355_BADARGS;
    WsGuard g(cs);
    if (!g.w) return KZG355_NO_DEVICE;
    kzg355_settings *s = g.s; Workspace *w = g.w;
    int rc;
    if ((rc = w->err.ensure(sizeof(int)))) return rc;
    if ((rc = w->h_err.ensure(sizeof(int)))) return rc;
    if ((rc = w->z.ensure(sizeof(Fr)))) return rc;
    if ((rc = w->records.ensure(64))) return rc;
    if ((rc = w->h_ok.ensure(64))) return rc;
    w->in_flight = true;                                          // (before the first copy from caller memory: a failure below drains the stream)
    if ((rc = stage_to_device(w, w->blobs, blob, blob_bytes_of(cs)))) return rc;
    if ((rc = stage_to_device(w, w->small, z_bytes, 32))) return rc;
    HIPCHK(hipMemsetAsync(w->err.p, 0, sizeof(int), w->stream));
    Timed tm(s, w);
    launch_fr_from_bytes(w->small.as<uint8_t>(), 1, w->z.as<Fr>(), w->err.as<int>(), w->stream);       // kzg.rs:452
    if (is_small(s)) {
        if ((rc = w->out48.ensure(48))) return rc;
        if ((rc = w->h_out.ensure(48))) return rc;
        launch_small_proof(w->blobs.as<uint8_t>(), nullptr, w->z.as<Fr>(), 1, s->t, w->out48.as<uint8_t>(), w->records.as<uint8_t>(), w->err.as<int>(),
                w->stream);
        HIPCHK(hipMemcpyAsync(w->h_out.p, w->out48.p, 48, hipMemcpyDeviceToHost, w->stream));
    } else {
        if ((rc = prove_common(s, w, tm, w->blobs.as<uint8_t>(), 1))) return rc;
        launch_fr_to_bytes(w->y.as<Fr>(), 1, w->records.as<uint8_t>(), w->stream);                     // kzg.rs:455
    }
    HIPCHK(hipMemcpyAsync(w->h_ok.p, w->records.p, 32, hipMemcpyDeviceToHost, w->stream));
    HIPCHK(hipMemcpyAsync(w->h_err.p, w->err.p, sizeof(int), hipMemcpyDeviceToHost, w->stream));
    HIPCHK(hipStreamSynchronize(w->stream));
    w->in_flight = false;
    tm.collect();
    int st = status_from_err(w->h_err.as<int>()[0]);
    if (st == KZG355_OK) { memcpy(proof_out, w->h_out.p, 48); memcpy(y_out, w->h_ok.p, 32); }
    return st;
}

#pragma GCC visibility pop
}  // extern "C"
