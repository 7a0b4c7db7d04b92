// msm_ops.hip -- the fixed-base MSM table of a handle and the commitment / proof chains (host side of libkzg355.so; see engine.h).
#include "engine.h"

namespace kzg355_impl {

thread_local bool tl_msm_inner = false;
thread_local const kzg355_settings::WidePub *tl_wide_candidate = nullptr;
thread_local bool tl_force_bucket = false;

// The fixed-base MSM table of the handle, built the first time a commitment or proof is asked for.  Width: the explicit msm_bits, else
// the widest GLV form whose table (plus the ~7.5 GB the build parks its Jacobian runs in) fits HALF of the HBM that is free at that
// moment: 16-bit windows 143.5 GB (16 rows per scalar), 15: 68.9 GB (18), 13: 20.1 GB (20), 12: 10.9 GB (22).  Then a check of the new
// table against the bucket form on two known blobs; a table that fails it is dropped (bucket form from then on, said on stderr).
int ensure_wide_table(kzg355_settings *s) {
    if (s->msm_bits_wanted == 8 || tl_msm_inner || is_small(s)) return KZG355_OK;
    std::call_once(s->wide_once, [s] {
        DeviceScope scope;
        if (!scope.enter(s->device)) { s->wide_rc = KZG355_NO_DEVICE; s->wide_table_failed = true; return; }
        int bits = s->msm_bits_wanted;
        if (bits == 0) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
            const size_t scratch = (size_t)15 << 29;
            for (int c : {16, 15, 13, 12}) {
                const WideShape ws = wide_shape(c, s->msm_glv != 0 || c == 16);
                if (!s->msm_glv && c == 16) continue;
                if (wide_table_bytes(ws) + scratch <= free_b / 2) { bits = c; break; }
            }
            if (bits == 0) { s->wide_table_failed = true; s->wide_rc = KZG355_NO_MEMORY; }
        }
        if (bits) {
            const WideShape shape = wide_shape(bits, s->msm_glv != 0);
            if (s->wide.ensure(wide_table_bytes(shape)) != KZG355_OK) { s->wide_table_failed = true; s->wide_rc = KZG355_NO_MEMORY; }
            else {
                DeviceTables t = s->t;
                t.wide = shape;
                t.wide_table = s->wide.as<WideRow>();
                s->wide_store = kzg355_settings::WidePub{shape, t.wide_table};
                if (build_wide_table(t, nullptr)) { s->wide.release(); s->wide_table_failed = true; s->wide_rc = KZG355_DEVICE_ERROR; }
                else {
                    // the new table against the bucket form, bit for bit, on three blobs: all ones; (w_0, .., w_{N-1}) -- 255-bit elements, both GLV
                    // halves of every scalar busy; and a blob of extreme digits: r - 1 - i at even positions (the largest canonical elements), at odd
                    // positions 0x0080 0x8000 ... (every 16-bit digit of both halves at the sign boundary of the recoding) with i folded in
                    const size_t BB = blob_bytes_of(s);
                    DevBuf blobs;
                    uint8_t c_wide[144], c_bucket[144]; int st[3] = {0, 0, 0};
                    int rc = blobs.ensure(3 * BB);
                    if (rc == KZG355_OK) {
                        std::vector<uint8_t> ones(BB, 0), ext(BB, 0);
                        static const uint8_t R_BE[32] = {0x73, 0xed, 0xa7, 0x53, 0x29, 0x9d, 0x7d, 0x48, 0x33, 0x39, 0xd8, 0x08, 0x09, 0xa1, 0xd8, 0x05,
                                                         0x53, 0xbd, 0xa4, 0x02, 0xff, 0xfe, 0x5b, 0xfe, 0xff, 0xff, 0xff, 0xff, 0x00, 0x00, 0x00, 0x01};
                        for (size_t i = 0; i < (size_t)s->t.n_fe; i++) {
                            ones[32 * i + 31] = 1;
                            uint8_t *e = ext.data() + 32 * i;
                            if (i & 1) { for (int k = 0; k < 32; k++) e[k] = (k & 1) ? 0x00 : 0x80; e[0] = 0x00; e[30] ^= (uint8_t)(i >> 8);
                                    e[31] ^= (uint8_t)i; }
                            else {      // r - 1 - i: big-endian subtraction of 1 + i with borrow
                                memcpy(e, R_BE, 32);
                                uint32_t sub = 1 + (uint32_t)i;
                                for (int k = 31; k >= 0 && sub; k--) { const uint32_t d = sub & 0xff; sub >>= 8; if (e[k] >= d) e[k] = (uint8_t)(e[k] - d);
                                        else { e[k] = (uint8_t)(e[k] + 256 - d); sub += 1; } }
                            }
                        }
                        if (hipMemcpy(blobs.p, ones.data(), BB, hipMemcpyHostToDevice) != hipSuccess) rc = KZG355_DEVICE_ERROR;
                        if (hipMemcpy(blobs.as<uint8_t>() + 2 * BB, ext.data(), BB, hipMemcpyHostToDevice) != hipSuccess) rc = KZG355_DEVICE_ERROR;
                        launch_fr_to_bytes(s->t.roots, s->t.n_fe, blobs.as<uint8_t>() + BB, nullptr);
                        if (hipDeviceSynchronize() != hipSuccess) rc = KZG355_DEVICE_ERROR;
                    }
                    tl_msm_inner = true;                      // (the commitments below must not come back here)
                    // nothing published yet: bucket form
                    if (rc == KZG355_OK) rc = msm_op_many_device_impl(c_bucket, st, blobs.as<uint8_t>(), nullptr, 3, s);
                    tl_wide_candidate = &s->wide_store;
                    if (rc == KZG355_OK) rc = msm_op_many_device_impl(c_wide, st, blobs.as<uint8_t>(), nullptr, 3, s);
                    tl_wide_candidate = nullptr;
                    tl_msm_inner = false;
                    blobs.release();
                    if (rc != KZG355_OK || memcmp(c_wide, c_bucket, 144) != 0) {
                        fprintf(stderr, "kzg355: the wide-window MSM table failed its check against the bucket form (status %d): dropped\n", rc);
                        s->wide.release(); s->wide_table_failed = true; s->wide_rc = rc != KZG355_OK ? rc : KZG355_INTERNAL;
                    } else s->wide_pub.store(&s->wide_store, std::memory_order_release);
                }
            }
        }
        if (s->wide_table_failed) {
            (void)hipGetLastError();
            fprintf(stderr,
                    "kzg355: the wide-window MSM table could not be %s; commitments / proofs take the 8-bit bucket form (about 3x slower, same results)\n",
                    s->wide_rc == KZG355_NO_MEMORY ? "allocated" : "built");
        }
    });
    return s->msm_required ? s->wide_rc : KZG355_OK;
}

// MSM -> 48-byte outputs on the host.  Wide-window table form if the handle has the table (scalars straight from the blobs,
// or from Montgomery field elements for the quotient), else the 8-bit bucket form over a digit buffer.
int msm_to_host(kzg355_settings *s, Workspace *w, Timed &tm, int n, const uint8_t *d_blobs, const Fr *d_scalars) {
    int rc;
    if ((rc = ensure_wide_table(s))) return rc;
    if ((rc = w->partials.ensure(sizeof(G1Jac) * (size_t)n * MSM_WINDOWS))) return rc;
    if ((rc = w->out48.ensure(48 * (size_t)n))) return rc;
    if ((rc = w->h_out.ensure(48 * (size_t)n))) return rc;
    const kzg355_settings::WidePub *wp = tl_force_bucket ? nullptr : tl_wide_candidate ? tl_wide_candidate : s->wide_pub.load(std::memory_order_acquire);
    if (wp) {
        DeviceTables t = s->t;
        t.wide = wp->shape; t.wide_table = wp->rows;
        tm.begin("msm_wide"); launch_msm_wide(d_blobs, d_scalars, t, n, w->partials.as<G1Jac>(), w->err.as<int>(), w->stream); tm.end();
        tm.begin("msm_finalize"); launch_msm_finalize(w->partials.as<G1Jac>(), n, w->out48.as<uint8_t>(), w->stream, msm_wide_partials_per_blob(n)); tm.end();
    } else {
        if ((rc = w->digits.ensure((size_t)BLOB_BYTES * n))) return rc;
        tm.begin("digits");
        if (d_scalars) launch_digits_from_fr(d_scalars, n, w->digits.as<uint8_t>(), w->stream);
        else launch_digits_from_blobs(d_blobs, n, w->digits.as<uint8_t>(), w->err.as<int>(), w->stream);
        tm.end();
        tm.begin("msm_bucket"); launch_msm_bucket(w->digits.as<uint8_t>(), s->t, n, w->partials.as<G1Jac>(), w->stream); tm.end();
        tm.begin("msm_finalize"); launch_msm_finalize(w->partials.as<G1Jac>(), n, w->out48.as<uint8_t>(), w->stream); tm.end();
    }
    HIPCHK(hipMemcpyAsync(w->h_out.p, w->out48.p, 48 * (size_t)n, hipMemcpyDeviceToHost, w->stream));
    return KZG355_OK;
}

// proofs for n blobs at challenge points already in w->z (Montgomery); err accumulates per blob
int prove_common(kzg355_settings *s, Workspace *w, Timed &tm, const uint8_t *d_blobs, int n) {
    int rc;
    if ((rc = w->y.ensure(sizeof(Fr) * (size_t)n))) return rc;
    if ((rc = w->q.ensure((size_t)BLOB_BYTES * n))) return rc;                  // the quotient in the blob format: the MSM reads it like a blob
    if ((rc = w->qprep.ensure(quotient_scratch_bytes(n)))) return rc;
    tm.begin("quotient");
    if (launch_quotient(d_blobs, w->z.as<Fr>(), s->t, n, w->y.as<Fr>(), w->q.as<uint8_t>(), w->qprep.p, w->err.as<int>(), w->stream,
            s->quotient_form)) return KZG355_DEVICE_ERROR;
    tm.end();
    return msm_to_host(s, w, tm, n, w->q.as<uint8_t>(), nullptr);
}

// n commitments (d_c == null) or n blob proofs against the commitments d_c: enqueue on w->stream ...
int msm_op_enqueue(kzg355_settings *s, Workspace *w, Timed &tm, const uint8_t *d_blobs, const uint8_t *d_c, size_t n, HostFront *hf, const uint8_t *d_zs) {
    int rc;
    if ((rc = w->err.ensure(sizeof(int) * n))) return rc;
    if ((rc = w->h_err.ensure(sizeof(int) * n))) return rc;
    if ((d_c || d_zs) && (rc = w->z.ensure(sizeof(Fr) * n))) return rc;
    w->in_flight = true;
    HIPCHK(hipMemsetAsync(w->err.p, 0, sizeof(int) * n, w->stream));
    if (d_zs) {
        // n independent compute_kzg_proof calls (kzg.rs:446-457): z_i -> field element (non-canonical: that unit's error, kzg.rs:452), then the quotient
        // and its MSM as for blob proofs; y_i = p_i(z_i) leaves as 32 big-endian bytes (kzg.rs:455) through w->records / w->h_records
        if ((rc = w->records.ensure(32 * n)) || (rc = w->h_records.ensure(32 * n))) return rc;
        launch_fr_from_bytes(d_zs, (int)n, w->z.as<Fr>(), w->err.as<int>(), w->stream);
        if (is_small(s)) {
            if ((rc = w->out48.ensure(48 * n)) || (rc = w->h_out.ensure(48 * n))) return rc;
            tm.begin("small_proof"); launch_small_proof(d_blobs, nullptr, w->z.as<Fr>(), (int)n, s->t, w->out48.as<uint8_t>(), w->records.as<uint8_t>(),
                    w->err.as<int>(), w->stream); tm.end();
            HIPCHK(hipMemcpyAsync(w->h_out.p, w->out48.p, 48 * n, hipMemcpyDeviceToHost, w->stream));
        } else {
            if ((rc = prove_common(s, w, tm, d_blobs, (int)n))) return rc;
            launch_fr_to_bytes(w->y.as<Fr>(), (int)n, w->records.as<uint8_t>(), w->stream);
        }
        HIPCHK(hipMemcpyAsync(w->h_records.p, w->records.p, 32 * n, hipMemcpyDeviceToHost, w->stream));
        HIPCHK(hipMemcpyAsync(w->h_err.p, w->err.p, sizeof(int) * n, hipMemcpyDeviceToHost, w->stream));
        return KZG355_OK;
    }
    if (is_small(s)) {
        if ((rc = w->out48.ensure(48 * n))) return rc;
        if ((rc = w->h_out.ensure(48 * n))) return rc;
        if (!d_c) { tm.begin("small_commit"); launch_small_commit(d_blobs, (int)n, s->t, w->out48.as<uint8_t>(), w->err.as<int>(), w->stream); tm.end(); }
        else {
            tm.begin("validate_points"); launch_validate_points(d_c, nullptr, (int)n, 1, nullptr, w->err.as<int>(), w->stream); tm.end();
            tm.begin("small_proof"); launch_small_proof(d_blobs, d_c, nullptr, (int)n, s->t, w->out48.as<uint8_t>(), nullptr, w->err.as<int>(), w->stream);
                    tm.end();
        }
        HIPCHK(hipMemcpyAsync(w->h_out.p, w->out48.p, 48 * n, hipMemcpyDeviceToHost, w->stream));
    } else if (!d_c) {
        if ((rc = msm_to_host(s, w, tm, (int)n, d_blobs, nullptr))) return rc;
    } else {
        // compute_challenge validates the commitment (kzg.rs:321-323); one "group" per blob so errors stay per blob
        // (the validation only feeds the error word: for few blobs it runs on the side stream, beside the hash chain -- 1.5 ms for one
        // point against 3.7 ms for one hash -- and is joined before the statuses are copied back)
        if (n <= (size_t)s->beside_max_blobs && ensure_side(s, w)) {
            HIPCHK(hipEventRecord(w->ev_fork, w->stream));       // after the memset of the error words
            HIPCHK(hipStreamWaitEvent(w->side, w->ev_fork, 0));
            w->side_pending = true;
            // (few blobs: the decoding and the subgroup test as two kernels -- neither spills, 0.45 + 1.0 ms for a lone point against 1.7 ms fused)
            if ((rc = w->pts.ensure(sizeof(G1Affine) * 2 * n))) return rc;
            // few points: the subgroup ladder starts from x alone on a stream of its own, beside the square root (k_subgroup_ladder_from_x_quad) -- the
            // validation of the commitment is what compute_blob_kzg_proof waits for: 0.45 + 0.65 ms in a row became max(0.45, 0.65)
            const bool ladder_beside = 2 * n <= 1024 && s->calls_in_flight.load() * 3 <= s->hw_queues && w->shifts.ensure(sizeof(G1Jac) * n) == KZG355_OK &&
                    ensure_side2(s, w);
            if (ladder_beside) {
                HIPCHK(hipStreamWaitEvent(w->side2, w->ev_fork, 0));
                w->shift_pending = true;                         // (side2 has work: quiesce() drains it; join_side() waits for ev_shift)
                tm.begin("validate_points", w->side2); launch_subgroup_ladder_from_x(d_c, 48, (int)n, w->shifts.as<G1Jac>(), w->side2); tm.end(w->side2);
                HIPCHK(hipEventRecord(w->ev_shift, w->side2));
            }
            tm.begin("decompress_points", w->side); launch_decompress_points(d_c, nullptr, (int)n, 1, w->pts.as<G1Affine>(), w->err.as<int>(), w->side);
                    tm.end(w->side);
            if (ladder_beside) {
                HIPCHK(hipStreamWaitEvent(w->side, w->ev_shift, 0));
                launch_subgroup_finish(w->pts.as<G1Affine>(), w->shifts.as<G1Jac>(), (int)n, w->err.as<int>(), w->side);
            } else { tm.begin("validate_points", w->side); launch_subgroup_points(w->pts.as<G1Affine>(), (int)n, 1, w->err.as<int>(), w->side, 1);
                    tm.end(w->side); }
            HIPCHK(hipEventRecord(w->ev_join, w->side));
        } else { tm.begin("validate_points"); launch_validate_points(d_c, nullptr, (int)n, 1, nullptr, w->err.as<int>(), w->stream); tm.end(); }
        if (hf) {                                                 // challenges hashed on the host (see run_stage1)
            if (hf->from_device) { if ((rc = host_hash_from_device(s, w, hf, d_blobs))) return rc; }
            else HIPCHK(hipMemcpyAsync(const_cast<uint8_t *>(d_blobs), hf->h_blobs, hf->bytes, hipMemcpyHostToDevice, w->stream));
            hf->finish();
            if (!hf->ok()) return KZG355_DEVICE_ERROR;
            HIPCHK(hipMemcpyAsync(w->digests.p, w->h_digests.p, 32 * n, hipMemcpyHostToDevice, w->stream));
            tm.begin("challenge_from_digest"); launch_challenges_from_digests(w->digests.as<uint8_t>(), d_c, nullptr, (int)n, w->z.as<Fr>(), nullptr, nullptr,
                    w->stream); tm.end();
        } else { tm.begin("challenge"); launch_challenges(d_blobs, d_c, nullptr, (int)n, w->z.as<Fr>(), nullptr, nullptr, w->stream,
                s->challenge_form ? s->challenge_form : (int)n <= s->challenge_two_wave_upto ? 2 : 1); tm.end(); }
        if ((rc = prove_common(s, w, tm, d_blobs, (int)n))) return rc;
        if ((rc = join_side(w))) return rc;
    }
    HIPCHK(hipMemcpyAsync(w->h_err.p, w->err.p, sizeof(int) * n, hipMemcpyDeviceToHost, w->stream));
    return KZG355_OK;
}

// ... and wait for it: 48-byte outputs / statuses of its n blobs.  Returns the first non-OK status.
int msm_op_collect(Workspace *w, Timed &tm, uint8_t *out, int *status, size_t n, uint8_t *ys_out) {
    HIPCHK(hipStreamSynchronize(w->stream));
    w->in_flight = false;
    tm.collect();
    int first = KZG355_OK;
    for (size_t i = 0; i < n; i++) {
        int st = status_from_err(w->h_err.as<int>()[i]);
        if (status) status[i] = st;
        if (st == KZG355_OK) {
            memcpy(out + 48 * i, w->h_out.as<uint8_t>() + 48 * i, 48);
            if (ys_out) memcpy(ys_out + 32 * i, w->h_records.as<uint8_t>() + 32 * i, 32);
        } else if (first == KZG355_OK) first = st;
    }
    return first;
}

int msm_op_many_device_impl(uint8_t *out, int *status, const uint8_t *d_blobs, const uint8_t *d_c, size_t n, const kzg355_settings *cs, const uint8_t *d_zs,
                            uint8_t *ys_out) {
    if (!cs || !out || (d_zs && (d_c || !ys_out))) return KZG355_BADARGS;
    if (n == 0) return KZG355_OK;
    if (n > (size_t)1 << 20) return KZG355_BADARGS;
    if (!d_blobs || ((uintptr_t)d_blobs & 15) || ((uintptr_t)d_c & 3)) return KZG355_BADARGS;
    WsGuard g(cs);
    if (!g.w) return KZG355_NO_DEVICE;
    Timed tm(g.s, g.w);
    HostFront hf;
    const bool via_host = d_c && device_call_hashes_on_host(g.s, n);      // blob proofs: the challenge hashes the blob (kzg.rs:298-339)
    struct InFlight { std::atomic<int> *n; ~InFlight() { if (n) (*n)--; } } in_flight{nullptr};
    if (via_host) { hf.from_device = true; hf.d_commitments = d_c; hf.n_blobs = n; g.s->calls_in_flight++; in_flight.n = &g.s->calls_in_flight; }
    int rc = msm_op_enqueue(g.s, g.w, tm, d_blobs, d_c, n, via_host ? &hf : nullptr, d_zs);
    if (rc) return rc;
    return msm_op_collect(g.w, tm, out, status, n, ys_out);
}

}  // namespace kzg355_impl

