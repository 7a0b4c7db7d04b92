// verify_stages.hip -- stage drivers of verification (all asynchronous on the workspace's streams), launch sets, the device-resident call (host side of
// libkzg355.so; see engine.h).
#include "engine.h"

namespace kzg355_impl {

// A small DEVICE-RESIDENT call (one 64-blob batch already in HBM took 5.3 ms against 1.97 ms for the same batch arriving in host
// memory, because only host-buffer calls had the host hash).  The blobs go back over PCIe in up to eight chunks (8 MiB = 0.16 ms for 64 blobs), each
// followed by an event; the hashing job's index k (blobs 2k, 2k + 1, interleaved) waits for the event of its chunk and hashes out of the pinned
// slot -- the first pairs are being hashed while the later chunks are still on the wire.  Queued on w->stream AFTER the point kernels have forked
// off it (their side streams do not wait for the copies).  Leaves hf->running set; the caller joins the job and uploads the digests.
int host_hash_from_device(kzg355_settings *s, Workspace *w, HostFront *hf, const uint8_t *d_blobs) {
    int rc;
    const size_t nb = hf->n_blobs, BB = blob_bytes_of(s);
    if ((rc = w->h_stage.ensure(BB * nb)) || (rc = w->h_stage_cp.ensure(48 * nb)) || (rc = w->h_digests.ensure(32 * nb)) ||
            (rc = w->digests.ensure(32 * nb))) return rc;
    size_t nch = (nb + 1) / 2 < 8 ? (nb + 1) / 2 : 8;
    size_t per = (nb + nch - 1) / nch;
    per += per & 1;                                               // whole pairs per chunk
    nch = (nb + per - 1) / per;
    for (size_t c = 0; c < nch; c++)
        if (!w->ev_d2h[c] && hipEventCreateWithFlags(&w->ev_d2h[c], hipEventDisableTiming) != hipSuccess) { w->ev_d2h[c] = nullptr; (void)hipGetLastError();
                return KZG355_DEVICE_ERROR; }
    HIPCHK(hipMemcpyAsync(w->h_stage_cp.p, hf->d_commitments, 48 * nb, hipMemcpyDeviceToHost, w->stream));
    for (size_t c = 0; c < nch; c++) {
        const size_t lo = c * per, cnt = nb - lo < per ? nb - lo : per;
        HIPCHK(hipMemcpyAsync(w->h_stage.as<uint8_t>() + BB * lo, d_blobs + BB * lo, BB * cnt, hipMemcpyDeviceToHost, w->stream));
        HIPCHK(hipEventRecord(w->ev_d2h[c], w->stream));
    }
    uint8_t *dig = w->h_digests.as<uint8_t>();
    const uint8_t *hb = w->h_stage.as<uint8_t>(), *hcm = w->h_stage_cp.as<uint8_t>();
    const uint64_t n_fe = (uint64_t)s->t.n_fe; const int impl = s->sha_impl;
    hipEvent_t evs[8];
    for (size_t c = 0; c < 8; c++) evs[c] = w->ev_d2h[c];
    struct Ev8 { hipEvent_t e[8]; } ev8; memcpy(ev8.e, evs, sizeof evs);
    hf->failed = std::make_shared<std::atomic<int>>(0);
    std::shared_ptr<std::atomic<int>> failed = hf->failed;
    auto job = [=](size_t k) {
        // (chunks hold whole pairs: both blobs of the pair are behind this event)
        if (hipEventSynchronize(ev8.e[(2 * k) / per]) != hipSuccess) { (void)hipGetLastError(); failed->store(1); return; }
        kzg_host::challenge_digests(dig + 64 * k, hb + BB * 2 * k, BB, hcm + 96 * k, nb - 2 * k < 2 ? nb - 2 * k : 2, n_fe, impl);
    };
    hf->job = s->host_pool->begin((nb + 1) / 2, job);
    hf->pool = s->host_pool; hf->running = true;
    s->n_host_hashed++;
    return KZG355_OK;
}

// The point work of stage 1 depends on nothing but the inputs, so for small calls it runs BESIDE the main chain:
//   side    point validation (utils.rs:282-310); in the pre-shifted form of the linear combination as two kernels, so that the decoded
//           points (ev_pts) are out before the subgroup verdicts (ev_join), which only feed the error words;
//   side2   the window shifts of the pre-shifted form (k_ps_shift), straight from the compressed bytes: they need x only and run next to
//           the square roots of the decoding instead of after them (ev_shift).
// The main stream waits where the results are first needed: join_points() in front of the linear combination, join_side() in front of
// the copy of the error words.
int enqueue_points_beside(kzg355_settings *s, Workspace *w, Timed &tm, const uint8_t *d_c, const uint8_t *d_p, int n_total, int npg, G1Affine *d_pts,
                          int *d_err, bool allow_preshift, int stride /* bytes between consecutive inputs: 48 packed, 160 inside records */) {
    int rc;
    HIPCHK(hipEventRecord(w->ev_fork, w->stream));
    const bool pre = allow_preshift && d_pts && d_p && lincomb_form(s, npg, n_total / npg) == LC_FORM_PRESHIFT;
    bool shift_on_side2 = false;
    if (pre) {
        if ((rc = w->shifts.ensure(lincomb_preshift_bytes(npg, n_total / npg)))) return rc;
        if (s->calls_in_flight.load() * 3 <= s->hw_queues && ensure_side2(s, w)) {
            HIPCHK(hipStreamWaitEvent(w->side2, w->ev_fork, 0));
            w->shift_pending = true;
            tm.begin("lincomb_shift", w->side2); launch_lincomb_preshift_bytes(d_c, d_p, stride, npg, n_total / npg, w->shifts.as<G1Jac>(), w->side2);
                    tm.end(w->side2);
            HIPCHK(hipEventRecord(w->ev_shift, w->side2));
            shift_on_side2 = true;
        }
    }
    HIPCHK(hipStreamWaitEvent(w->side, w->ev_fork, 0));
    w->side_pending = true;                                       // (from here on a failing call has to drain the side streams: quiesce())
    if (pre) {
        tm.begin("decompress_points", w->side); launch_decompress_points(d_c, d_p, n_total, npg, d_pts, d_err, w->side, stride); tm.end(w->side);
        // (no second side stream: the shifts follow the decoding on this one, and ev_pts -- what join_points() waits for -- covers both)
        if (!shift_on_side2) { tm.begin("lincomb_shift", w->side); launch_lincomb_preshift(d_pts, npg, n_total / npg, w->shifts.as<G1Jac>(), w->side);
                tm.end(w->side); }
        HIPCHK(hipEventRecord(w->ev_pts, w->side));
        w->pts_pending = true;
        tm.begin("validate_points", w->side); launch_subgroup_points(d_pts, n_total, npg, d_err, w->side); tm.end(w->side);
        w->shift_ready = true;
    } else {
        tm.begin("validate_points", w->side); launch_validate_points(d_c, d_p, n_total, npg, d_pts, d_err, w->side, stride); tm.end(w->side);
    }
    HIPCHK(hipEventRecord(w->ev_join, w->side));
    return KZG355_OK;
}

int join_shifts(Workspace *w) {       // the window shifts of the pre-shifted form, when they run on a stream of their own
    if (w->shift_pending) { w->shift_pending = false; HIPCHK(hipStreamWaitEvent(w->stream, w->ev_shift, 0)); }
    return KZG355_OK;
}

int join_points(Workspace *w, bool shifts_too) {       // the decoded points and (unless the caller joins them later) their window shifts
    if (shifts_too) { const int rc = join_shifts(w); if (rc) return rc; }
    if (w->pts_pending) { w->pts_pending = false; HIPCHK(hipStreamWaitEvent(w->stream, w->ev_pts, 0)); }
    else if (w->side_pending) { w->side_pending = false; HIPCHK(hipStreamWaitEvent(w->stream, w->ev_join, 0)); }
    return KZG355_OK;
}

int join_side(Workspace *w) {         // everything the side streams were given, the validation verdicts included
    int rc = join_points(w);
    if (rc) return rc;
    if (w->side_pending) { w->side_pending = false; HIPCHK(hipStreamWaitEvent(w->stream, w->ev_join, 0)); }
    return KZG355_OK;
}

// after_challenge (submit / collect pipeline): called right after the Fiat-Shamir kernel is queued, which then goes FIRST -- the stage 2 of the
// previously submitted set is queued from there, so that it runs beside this set's evaluation and point kernels and never beside the hash
int run_stage1(kzg355_settings *s, Workspace *w, Timed &tm, const uint8_t *d_blobs, const uint8_t *d_c, const uint8_t *d_p, int n_total,
               int npg, uint8_t *d_records, G1Affine *d_pts, int *d_err, bool allow_preshift, HostFront *hf,
               const std::function<int()> *after_challenge) {
    int rc;
    w->shift_ready = false;
    if ((rc = w->z.ensure(sizeof(Fr) * (size_t)n_total))) return rc;
    if (!is_small(s) && (rc = w->zpow.ensure(sizeof(Fr) * EVAL_ZPOWERS * (size_t)n_total))) return rc;
    if (is_small(s)) {   // minimal preset: one lane per blob does conversion, challenge and evaluation (k_small.hip)
        tm.begin("validate_points"); launch_validate_points(d_c, d_p, n_total, npg, d_pts, d_err, w->stream); tm.end();
        tm.begin("small_records"); launch_small_records(d_blobs, d_c, d_p, n_total, npg, s->t, w->z.as<Fr>(), d_records, d_err, w->stream); tm.end();
        return KZG355_OK;
    }
    // While the card is far from full (few batches) the point work runs on the side streams next to the challenge -> evaluation
    // (-> r powers) chain.  With many batches in flight every kernel fills the card on its own and sharing the SIMDs only slows the
    // challenge kernel's producer/consumer hand-off (measured per 65,536 blobs: 6.7 + 4.2 ms apart, 19 ms together: one-wave workgroups
    // of a latency-bound kernel land unevenly on SIMDs that another grid is filling).
    const bool beside = n_total <= s->beside_max_blobs && ensure_side(s, w);
    if (beside) {
        if ((rc = enqueue_points_beside(s, w, tm, d_c, d_p, n_total, npg, d_pts, d_err, allow_preshift))) return rc;
    } else if (!after_challenge) {
        tm.begin("validate_points"); launch_validate_points(d_c, d_p, n_total, npg, d_pts, d_err, w->stream); tm.end();
    }
    if (hf) {
        // the blobs follow the point kernels into their queues (the copy from pageable memory holds the host thread for its duration);
        // then the digests the host threads have been computing meanwhile: 32 bytes per blob instead of a 2050-compression chain
        if (hf->from_device) { if ((rc = host_hash_from_device(s, w, hf, d_blobs))) return rc; }
        else if (hf->h_blobs) HIPCHK(hipMemcpyAsync(const_cast<uint8_t *>(d_blobs), hf->h_blobs, hf->bytes, hipMemcpyHostToDevice, w->stream));
        hf->finish();                                             // (h_blobs == null: the caller has queued the copies of its blobs itself)
        if (!hf->ok()) return KZG355_DEVICE_ERROR;
        HIPCHK(hipMemcpyAsync(w->digests.p, w->h_digests.p, 32 * (size_t)n_total, hipMemcpyHostToDevice, w->stream));
        tm.begin("challenge_from_digest"); launch_challenges_from_digests(w->digests.as<uint8_t>(), d_c, d_p, n_total, w->z.as<Fr>(), w->zpow.as<Fr>(),
                d_records, w->stream); tm.end();
    } else {
        tm.begin("challenge"); launch_challenges(d_blobs, d_c, d_p, n_total, w->z.as<Fr>(), w->zpow.as<Fr>(), d_records, w->stream,
                s->challenge_form ? s->challenge_form : n_total <= s->challenge_two_wave_upto ? 2 : 1); tm.end();
    }
    if (after_challenge) {
        if ((rc = (*after_challenge)())) return rc;
        if (!beside) { tm.begin("validate_points"); launch_validate_points(d_c, d_p, n_total, npg, d_pts, d_err, w->stream); tm.end(); }
    }
    tm.begin("eval"); launch_eval(d_blobs, w->z.as<Fr>(), w->zpow.as<Fr>(), s->t, n_total, npg, nullptr, d_records, d_err, w->stream); tm.end();
    return KZG355_OK;
}

// lone_call: this launch set is the whole of a synchronous call with nothing else of the caller in flight (a single-chunk host-buffer call,
// a one-set device-resident call) -- the only place where a host round trip in the middle of the chain costs nobody anything.
int run_stage2(kzg355_settings *s, Workspace *w, Timed &tm, const uint8_t *d_records, int npg, int groups, int check_zy, const G1Affine *d_pts,
               int *d_err, int *d_ok, bool lone_call) {
    int rc;
    const size_t n_total = (size_t)npg * groups;
    const bool shift_ready = w->shift_ready;                      // consumed here whatever happens below
    w->shift_ready = false;
    if ((rc = w->scal_a.ensure(32 * n_total))) return rc;
    if ((rc = w->scal_b.ensure(32 * n_total))) return rc;
    if ((rc = w->scal_c.ensure(32 * (size_t)groups))) return rc;
    if ((rc = w->pair_pts.ensure(sizeof(PairPt) * 2 * (size_t)groups))) return rc;
    // The batch challenge r hashes every record of the batch (utils.rs:439-473): one serial SHA-256 chain per batch -- 161 compressions
    // for 64 records, 0.33 ms on a lone GPU lane whatever else the card does.  For a lone small call the records go to the host instead
    // (10 KB for 64), a host core hashes them in microseconds (host_sha256.h) and 32 bytes per batch come back: ~60 us of round trip
    // in place of the chain.  Large or many-batch calls keep the device forms (k_rpowers / k_rhash_lanes), and so does every launch set that
    // is one of several in flight (split / pipelined sets, the stage-2 entry points of the sharded path): the wait below would serialise them.
    bool host_rhash = lone_call && s->host_rhash >= 0 && npg > 1 && n_total <= (size_t)s->host_rhash_max_records && !is_small(s);
    if (host_rhash && ((rc = w->h_records.ensure((size_t)RECORD_BYTES * n_total)) || (rc = w->h_rdig.ensure(32 * (size_t)groups)))) return rc;
    if (host_rhash) {
        HIPCHK(hipMemcpyAsync(w->h_records.p, d_records, (size_t)RECORD_BYTES * n_total, hipMemcpyDeviceToHost, w->stream));
        HIPCHK(hipStreamSynchronize(w->stream));
        std::vector<uint8_t> msg(32 + (size_t)RECORD_BYTES * npg);
        memcpy(msg.data(), "RCKZGBATCH___V1_", 16);                            // RANDOM_CHALLENGE_KZG_BATCH_DOMAIN (consts.rs:25)
        for (int k = 0; k < 8; k++) { msg[16 + k] = (uint8_t)((uint64_t)s->t.n_fe >> (56 - 8 * k)); msg[24 + k] = (uint8_t)((uint64_t)npg >> (56 - 8 * k)); }
        for (int g = 0; g < groups; g++) {
            memcpy(msg.data() + 32, w->h_records.as<uint8_t>() + (size_t)RECORD_BYTES * npg * g, (size_t)RECORD_BYTES * npg);
            uint8_t dg[32];
            kzg_host::sha256(dg, msg.data(), msg.size(), s->sha_impl);
            // the digest as an integer: 8 little-endian 32-bit words (what k_rhash_lanes leaves)
            uint32_t *dst = w->h_rdig.as<uint32_t>() + 8 * (size_t)g;
            for (int k = 0; k < 8; k++) dst[k] = ((uint32_t)dg[28 - 4 * k] << 24) | ((uint32_t)dg[29 - 4 * k] << 16) | ((uint32_t)dg[30 - 4 * k] << 8)
                    | dg[31 - 4 * k];
        }
        HIPCHK(hipMemcpyAsync(w->scal_c.p, w->h_rdig.p, 32 * (size_t)groups, hipMemcpyHostToDevice, w->stream));
    }
    tm.begin("rpowers"); launch_rpowers(d_records, npg, groups, check_zy, w->scal_a.as<uint32_t>(), w->scal_b.as<uint32_t>(), w->scal_c.as<uint32_t>(), d_err,
            w->stream, s->t.n_fe, s->rhash_lanes_from, host_rhash ? 1 : 0); tm.end();
    const int form = lincomb_form(s, npg, groups);
    const bool buckets = form == LC_FORM_BUCKET;
    if ((rc = w->lc_partials.ensure(form == LC_FORM_WINDOW ? lincomb_partials_bytes(npg, groups) : form == LC_FORM_SINGLE ? lincomb_single_bytes(groups) :
            lincomb_buckets_scratch_bytes(npg, groups)))) return rc;
    if ((rc = join_points(w, form != LC_FORM_PRESHIFT))) return rc;       // the decoded points (and their shifts) are needed from here on
    if (form == LC_FORM_PRESHIFT) {
        if (!shift_ready) {                                       // entry points without a stage 1 (single proofs, gathered records)
            if ((rc = w->shifts.ensure(lincomb_preshift_bytes(npg, groups)))) return rc;
            tm.begin("lincomb_shift"); launch_lincomb_preshift(d_pts, npg, groups, w->shifts.as<G1Jac>(), w->stream); tm.end();
        }
        // the digits need the points and the r powers only: they are made while the shift chain (the longest piece in front of the sums: 0.52 ms
        // for a 64-blob call, against 0.47 ms until the r powers are there) is still walking
        tm.begin("lincomb");
        launch_lincomb_preshifted(d_pts, w->shifts.as<G1Jac>(), w->scal_a.as<uint32_t>(), w->scal_b.as<uint32_t>(), w->scal_c.as<uint32_t>(), npg, groups,
                w->lc_partials.p,
                                  w->pair_pts.as<PairPt>(), w->stream, 1);
        if ((rc = join_shifts(w))) return rc;
        launch_lincomb_preshifted(d_pts, w->shifts.as<G1Jac>(), w->scal_a.as<uint32_t>(), w->scal_b.as<uint32_t>(), w->scal_c.as<uint32_t>(), npg, groups,
                w->lc_partials.p,
                                  w->pair_pts.as<PairPt>(), w->stream, 2);
    } else if (buckets) {
        static const char *names[3] = {"lincomb_prep", "lincomb", "lincomb_horner"};       // "lincomb" = the bucket kernel itself
        for (int stage = 1; stage <= 3; stage++) {
            if (stage > 1) tm.end();
            tm.begin(names[stage - 1]);
            launch_lincomb_buckets(d_pts, w->scal_a.as<uint32_t>(), w->scal_b.as<uint32_t>(), w->scal_c.as<uint32_t>(), npg, groups, w->lc_partials.p,
                    w->pair_pts.as<PairPt>(), w->stream, stage, s->lc_chain_from);
        }
    } else if (form == LC_FORM_SINGLE) {
        tm.begin("lincomb");
        launch_lincomb_single(d_pts, w->scal_b.as<uint32_t>(), w->scal_c.as<uint32_t>(), groups, w->lc_partials.p, w->pair_pts.as<PairPt>(), w->stream);
    } else {
        tm.begin("lincomb");
        launch_lincomb(d_pts, w->scal_a.as<uint32_t>(), w->scal_b.as<uint32_t>(), w->scal_c.as<uint32_t>(), npg, groups, w->lc_partials.as<G1Jac>(),
                w->pair_pts.as<PairPt>(), w->stream);
    }
    tm.end();
    tm.begin("pairing");
    if (s->lane_pairing) launch_pairing_lane(w->pair_pts.as<PairPt>(), s->t, groups, d_ok, w->stream);
    else {
        Fp *f12 = nullptr;
        // f between the two kernels of a check: many batches (hard part twelve lanes per check) and few (Miller loops in segments on several waves)
        if (((s->pairing_hard12_from > 0 && groups >= s->pairing_hard12_from) || groups <= s->pairing_two_wave_upto) &&
                w->pair_f.ensure(pairing_f12_bytes(groups)) == KZG355_OK)
            f12 = w->pair_f.as<Fp>();
        launch_pairing(w->pair_pts.as<PairPt>(), s->t, groups, d_ok, w->stream, s->pairing_two_wave_upto, f12, s->pairing_hard12_from, s->miller_segments);
    }
    tm.end();
    return KZG355_OK;
}

// Enqueue one launch set on w->stream (no host synchronisation) ...
// res_off / res_cap: several launch sets queued on one workspace (stream order keeps the device scratch safe) park their
// verdicts at different offsets of the pinned result buffers, sized res_cap entries up front
int verify_enqueue_stage1(kzg355_settings *s, Workspace *w, Timed &tm, const uint8_t *d_blobs, const uint8_t *d_c, const uint8_t *d_p, int npg, int G,
                          size_t res_cap, HostFront *hf, const std::function<int()> *after_challenge) {
    const int n_total = npg * G;
    int rc;
    if (res_cap < (size_t)G) res_cap = (size_t)G;
    if ((rc = w->records.ensure((size_t)RECORD_BYTES * n_total))) return rc;
    if ((rc = w->pts.ensure(sizeof(G1Affine) * 2 * (size_t)n_total))) return rc;
    if ((rc = w->err.ensure(sizeof(int) * (size_t)G))) return rc;
    if ((rc = w->ok.ensure(sizeof(int) * (size_t)G))) return rc;
    if ((rc = w->h_ok.ensure(sizeof(int) * res_cap))) return rc;
    if ((rc = w->h_err.ensure(sizeof(int) * res_cap))) return rc;
    w->in_flight = true;
    HIPCHK(hipMemsetAsync(w->err.p, 0, sizeof(int) * (size_t)G, w->stream));
    return run_stage1(s, w, tm, d_blobs, d_c, d_p, n_total, npg, w->records.as<uint8_t>(), w->pts.as<G1Affine>(), w->err.as<int>(), true, hf, after_challenge);
}

int verify_enqueue_stage2(kzg355_settings *s, Workspace *w, Timed &tm, int npg, int G, size_t res_off, bool lone_call) {
    int rc;
    if ((rc = run_stage2(s, w, tm, w->records.as<uint8_t>(), npg, G, 0, w->pts.as<G1Affine>(), w->err.as<int>(), w->ok.as<int>(), lone_call))) return rc;
    if ((rc = join_side(w))) return rc;                           // the subgroup verdicts, before the error words go back
    HIPCHK(hipMemcpyAsync(w->h_ok.as<int>() + res_off, w->ok.p, sizeof(int) * (size_t)G, hipMemcpyDeviceToHost, w->stream));
    HIPCHK(hipMemcpyAsync(w->h_err.as<int>() + res_off, w->err.p, sizeof(int) * (size_t)G, hipMemcpyDeviceToHost, w->stream));
    return KZG355_OK;
}

int verify_enqueue(kzg355_settings *s, Workspace *w, Timed &tm, const uint8_t *d_blobs, const uint8_t *d_c, const uint8_t *d_p, int npg, int G,
                   size_t res_off, size_t res_cap, HostFront *hf, bool lone_call) {
    int rc = verify_enqueue_stage1(s, w, tm, d_blobs, d_c, d_p, npg, G, res_cap, hf);
    if (rc) return rc;
    return verify_enqueue_stage2(s, w, tm, npg, G, res_off, lone_call);
}

// ... and wait for it: verdicts / statuses of its G batches.  Returns the first non-OK status.
int verify_collect(Workspace *w, Timed &tm, bool *ok, int *status, int G, size_t res_off) {
    HIPCHK(hipStreamSynchronize(w->stream));
    w->in_flight = false;
    tm.collect();
    int first = KZG355_OK;
    for (int i = 0; i < G; i++) {
        int st = status_from_err(w->h_err.as<int>()[res_off + i]);
        if (status) status[i] = st;
        if (st == KZG355_OK) ok[i] = w->h_ok.as<int>()[res_off + i] != 0;
        else if (first == KZG355_OK) first = st;
    }
    return first;
}

// a device-resident call small enough that bringing its blobs back and hashing them on the host threads beats the device's 3.7 ms hash chain; not while
// submitted sets are in flight (their caller is feeding a pipeline from one thread: the join of the hashing job would stall it)
bool device_call_hashes_on_host(const kzg355_settings *s, size_t n_blobs) {
    if (tl_msm_inner) return false;                               // the load-time self-test checks the DEVICE kernels: its calls keep the device hash
    return !is_small(s) && s->host_pool && s->host_hash >= 0 && s->host_hash_device_max > 0 && n_blobs <= (size_t)s->host_hash_device_max &&
           s->tickets_out.load() == 0;
}

int verify_many_device_impl(bool *ok, int *status, const uint8_t *d_blobs, const uint8_t *d_c, const uint8_t *d_p, size_t npg, size_t groups,
                            const kzg355_settings *cs) {
    if (!cs || !ok) return KZG355_BADARGS;
    if (groups == 0) return KZG355_OK;
    if (npg == 0) {   // kzg.rs:653-655
        for (size_t g = 0; g < groups; g++) { ok[g] = true; if (status) status[g] = KZG355_OK; }
        return KZG355_OK;
    }
    if (npg > (size_t)1 << 24 || groups > (size_t)1 << 24 || npg * groups > (size_t)1 << 24) return KZG355_BADARGS;
    // 16-byte loads of the blobs
    if (!d_blobs || !d_c || !d_p || ((uintptr_t)d_blobs & 15) || ((uintptr_t)d_c & 3) || ((uintptr_t)d_p & 3)) return KZG355_BADARGS;
    // Optional (KZG355_SPLIT=parts[,streams]): cut the call into `parts` launch sets dealt round-robin to `streams` workspaces, so
    // that the narrow kernels of one set (r powers, Horner tail) run under the wide kernels of another.  Measured
    // (profiles/r02/split_sweep.txt): what counts is the SIZE of a launch set -- 2048 batches 3.13 M blobs/s, 4096 3.44 M, 8192
    // 3.63 M, whether or not the sets overlap (4096 as 2 x 2048 overlapped: 3.44 M; 8192 as 2 x 4096: 3.64 M) -- and splitting a
    // set only costs (2048 as 2 / 4 / 8 parts: -2 / -20 / -44 %).  So the default is one set per call, as large as the caller makes it.
    size_t parts = 1, lanes = 1;
    if (!cs->timing && cs->split_parts > 1 && groups >= (size_t)cs->split_parts) {
        parts = (size_t)cs->split_parts;
        lanes = (size_t)cs->split_streams < parts ? (size_t)cs->split_streams : parts;
    }
    if (parts == 1) {
        WsGuard g(cs);
        if (!g.w) return KZG355_NO_DEVICE;
        Timed tm(g.s, g.w);
        HostFront hf;
        const bool via_host = device_call_hashes_on_host(g.s, npg * groups);
        struct InFlight { std::atomic<int> *n; ~InFlight() { if (n) (*n)--; } } in_flight{nullptr};
        if (via_host) { hf.from_device = true; hf.d_commitments = d_c; hf.n_blobs = npg * groups; g.s->calls_in_flight++; in_flight.n = &g.s->calls_in_flight; }
        int rc = verify_enqueue(g.s, g.w, tm, d_blobs, d_c, d_p, (int)npg, (int)groups, 0, 0, via_host ? &hf : nullptr, true);
        if (rc) return rc;
        return verify_collect(g.w, tm, ok, status, (int)groups);
    }
    std::vector<WsGuard *> gs;
    struct Cleanup { std::vector<WsGuard *> &g; ~Cleanup() { for (size_t i = g.size(); i-- > 0;) delete g[i]; } } cleanup{gs};
    std::vector<Timed> tms;
    for (size_t l = 0; l < lanes; l++) {
        gs.push_back(new WsGuard(cs));
        if (!gs.back()->w) return KZG355_NO_DEVICE;
        tms.emplace_back(gs.back()->s, gs.back()->w);
    }
    const size_t BB = blob_bytes_of(cs);
    std::vector<size_t> g0(parts + 1, 0), res_off(parts, 0), lane_fill(lanes, 0);
    // larger parts first: a later set never outgrows the scratch
    for (size_t k = 0; k < parts; k++) g0[k + 1] = g0[k] + groups / parts + (k < groups % parts ? 1 : 0);
    const size_t cap = (groups / parts + 1) * ((parts + lanes - 1) / lanes);
    for (size_t k = 0; k < parts; k++) {
        const size_t l = k % lanes, cnt = g0[k + 1] - g0[k];
        res_off[k] = lane_fill[l]; lane_fill[l] += cnt;
        int rc = verify_enqueue(gs[l]->s, gs[l]->w, tms[l], d_blobs + BB * npg * g0[k], d_c + 48 * npg * g0[k], d_p + 48 * npg * g0[k], (int)npg, (int)cnt,
                res_off[k], cap);
        if (rc) return rc;                        // (the guards wait for whatever is in flight)
    }
    int first = KZG355_OK;
    std::vector<bool> synced(lanes, false);
    for (size_t k = 0; k < parts; k++) {          // a lane's stream is synchronised once; its sets are then read back in order
        const size_t l = k % lanes, cnt = g0[k + 1] - g0[k];
        const int rc = verify_collect(gs[l]->w, tms[l], ok + g0[k], status ? status + g0[k] : nullptr, (int)cnt, res_off[k]);
        if (rc == KZG355_NO_DEVICE || rc == KZG355_NO_MEMORY || rc == KZG355_DEVICE_ERROR) return rc;
        if (rc != KZG355_OK && first == KZG355_OK) first = rc;
    }
    return first;
}

}  // namespace kzg355_impl

