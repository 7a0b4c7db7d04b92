// multi_device.hip -- handles over several devices: replicas, RCCL bound at run time, the record exchange of a sharded batch (host side of libkzg355.so; see
// engine.h).
#include "engine.h"

namespace kzg355_impl {

// ---- handles over several devices --------------------------------------------------------------------------------------
// "handle owns 1..8 devices; multi-GPU calls are collective inside the library, invisible to the caller".  The owner
// handle keeps one full replica of the settings per device (the tables are per-GPU constants) and the host-buffer entry points
// spread their work over them:
//   * many independent units (batches for verify, blobs for commit / proof): contiguous ranges of units per device, no
//     exchange at all -- every device runs the single-device pipeline on its range from its own host thread;
//   * fewer batches than devices (the 512-blob batch over 8 GPUs of BASELINE.json): every batch is cut into contiguous blocks
//     of blobs, one per device: stage 1 per block -> ONE all-gather of the 160-byte records (RCCL ncclAllGather
//     over xGMI on a persistent communicator set, or peer copies when RCCL is unavailable / the blocks are ragged) -> stage 2
//     for each batch on one device.
// RCCL is bound at run time (dlopen of the librccl the process already has, or /opt/rocm's), so the library carries no link
// dependency on it and single-device users never load it.
std::vector<kzg355_settings *> replicas_of(kzg355_settings *s) { return s->multi ? s->multi->rep : std::vector<kzg355_settings *>{s}; }

static int load_devices(const uint8_t *g1_bytes, size_t n1, const uint8_t *g2_bytes, size_t n2, const int *devices, size_t n_devices, const kzg355_options &opt,
                        kzg355_settings **out) {
    if (!out || !devices || n_devices == 0 || n_devices > 64) return KZG355_BADARGS;
    DeviceScope keep; keep.hold();               // the peer-access loop and the communicator set-up visit every device: the caller's current device comes back
    std::vector<kzg355_settings *> rep;
    auto fail = [&](int code) { for (auto *r : rep) free_single(r); return code; };
    for (size_t i = 0; i < n_devices; i++) {
        kzg355_settings *r = nullptr;
        int rc = load_on_device(g1_bytes, n1, g2_bytes, n2, devices[i], opt, &r);
        if (rc) return fail(rc);
        rep.push_back(r);
    }
    if (n_devices == 1 && !opt.force_multi) { *out = rep[0]; return KZG355_OK; }     // (force_multi: test hook, a one-device "multi" handle)
    MultiDev *m = new MultiDev();
    m->rep = rep;
    bool distinct = true;
    for (size_t i = 0; i < n_devices; i++) for (size_t j = 0; j < i; j++) distinct = distinct && devices[i] != devices[j];
    for (size_t i = 0; i < n_devices; i++)                       // peer access speeds up the record copies; not required
        for (size_t j = 0; j < n_devices; j++)
            if (devices[i] != devices[j] && hipSetDevice(devices[i]) == hipSuccess) { (void)hipDeviceEnablePeerAccess(devices[j], 0); (void)hipGetLastError(); }
    const bool want_rccl = opt.exchange != 1;
    if (want_rccl && distinct && m->rccl.load()) {
        m->comms.assign(n_devices, nullptr);
        if (m->rccl.CommInitAll(m->comms.data(), (int)n_devices, devices) == 0) m->exchange = 1;
        else m->comms.clear();
    }
    if (opt.exchange == 2 && m->exchange != 1) { delete m; return fail(KZG355_DEVICE_ERROR); }
    rep[0]->multi = m;
    *out = rep[0];
    return KZG355_OK;
}

// ---- multi-device execution of the host-buffer entry points ------------------------------------------------------------

int single_verify_many(bool *ok, int *status, const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs, size_t npg, size_t groups,
                       const kzg355_settings *cs) {
    HostCall hc{0, blobs, commitments, proofs, npg, ok, nullptr, status};
    return host_pipeline(hc, groups, cs);
}

// contiguous ranges of `units` over D devices; the work of device d is fn(d, first unit, count) on its own host thread
int fan_out(size_t D, size_t units, const std::function<int(size_t, size_t, size_t)> &fn) {
    std::vector<std::future<int>> fut;
    for (size_t d = 0; d < D; d++) {
        const size_t lo = units * d / D, hi = units * (d + 1) / D;
        if (hi > lo) fut.push_back(std::async(std::launch::async, fn, d, lo, hi - lo));
    }
    int first = KZG355_OK;
    for (auto &f : fut) { const int rc = f.get(); if (rc != KZG355_OK && first == KZG355_OK) first = rc; }
    return first;
}

// Fewer batches than devices: every batch is sharded over the devices in contiguous blocks of blobs.
// `dump` (host, groups * 128 bytes or null; kzg355_debug_verify_sharded_intermediates): r | proof_lincomb | rhs of every batch, read back from
// the device that ran its stage 2.
int multi_verify_sharded(bool *ok, int *status, const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs, size_t npg, size_t groups,
                         const kzg355_settings *cs, uint8_t *dump) {
    MultiDev *m = cs->multi;
    const size_t D = m->rep.size(), BB = blob_bytes_of(cs);
    DeviceScope keep; keep.hold();               // this thread visits every replica's device below
    std::vector<size_t> cnt(D), off(D);
    for (size_t d = 0; d < D; d++) { off[d] = npg * d / D; cnt[d] = npg * (d + 1) / D - off[d]; }
    std::vector<WsGuard *> gs(D, nullptr);
    struct Cleanup { std::vector<WsGuard *> &g; ~Cleanup() { for (size_t i = g.size(); i-- > 0;) delete g[i]; } } cleanup{gs};
    std::vector<std::vector<int>> st1(D, std::vector<int>(groups, KZG355_OK));
    // stage 1: device d takes blobs [off_d, off_d + cnt_d) of every batch (records in transcript order within the block)
    {
        std::vector<std::future<int>> fut;
        for (size_t d = 0; d < D; d++) {
            gs[d] = new WsGuard(m->rep[d]);
            if (!gs[d]->w) return KZG355_NO_DEVICE;
            fut.push_back(std::async(std::launch::async, [&, d]() -> int {
                kzg355_settings *rs = gs[d]->s; Workspace *w = gs[d]->w;
                if (hipSetDevice(rs->device) != hipSuccess) return KZG355_NO_DEVICE;
                const size_t n_loc = cnt[d], n_tot = n_loc * groups;
                int rc;
                if ((rc = w->records.ensure((size_t)RECORD_BYTES * (n_tot ? n_tot : 1)))) return rc;
                if ((rc = w->pts.ensure(sizeof(G1Affine) * 2 * (n_tot ? n_tot : 1)))) return rc;       // the decoded points travel with the records
                if (n_loc == 0) return KZG355_OK;
                if ((rc = w->blobs.ensure(BB * n_tot)) || (rc = w->commitments.ensure(48 * n_tot)) || (rc = w->proofs.ensure(48 * n_tot))) return rc;
                if ((rc = w->err.ensure(sizeof(int) * groups)) || (rc = w->h_err.ensure(sizeof(int) * groups))) return rc;
                // A small block (BASELINE config 5: 64 blobs per device) has its Fiat-Shamir challenges hashed on this replica's host threads straight
                // out of the caller's memory while the copies and the point kernels run, as a small single-device call has (host_pipeline.hip): the
                // device's hash is a 3.7 ms chain whatever the block size.  Job k: the pair of blobs 2p, 2p + 1 of batch g, k = g * pairs + p.
                HostFront hf;
                const bool host_hash = !is_small(rs) && rs->host_pool && (rs->host_hash > 0 || (rs->host_hash == 0 && n_tot <= (size_t)rs->host_hash_max));
                if (host_hash) {
                    if ((rc = w->h_digests.ensure(32 * n_tot)) || (rc = w->digests.ensure(32 * n_tot))) return rc;
                    uint8_t *dig = w->h_digests.as<uint8_t>();
                    const size_t pairs = (n_loc + 1) / 2, o_d = off[d];
                    const uint64_t n_fe = (uint64_t)rs->t.n_fe; const int impl = rs->sha_impl;
                    auto job = [=](size_t k) {
                        const size_t g = k / pairs, p = k % pairs, src = g * npg + o_d + 2 * p;
                        kzg_host::challenge_digests(dig + 32 * (g * n_loc + 2 * p), blobs + BB * src, BB, commitments + 48 * src,
                                n_loc - 2 * p < 2 ? n_loc - 2 * p : 2, n_fe, impl);
                    };
                    hf.job = rs->host_pool->begin(pairs * groups, job);
                    hf.pool = rs->host_pool; hf.running = true;
                    rs->n_host_hashed++;
                }
                for (size_t g = 0; g < groups; g++) {
                    const size_t src = g * npg + off[d], dst = g * n_loc;
                    HIPCHK(hipMemcpyAsync(w->blobs.as<uint8_t>() + BB * dst, blobs + BB * src, BB * n_loc, hipMemcpyHostToDevice, w->stream));
                    HIPCHK(hipMemcpyAsync(w->commitments.as<uint8_t>() + 48 * dst, commitments + 48 * src, 48 * n_loc, hipMemcpyHostToDevice, w->stream));
                    HIPCHK(hipMemcpyAsync(w->proofs.as<uint8_t>() + 48 * dst, proofs + 48 * src, 48 * n_loc, hipMemcpyHostToDevice, w->stream));
                }
                w->in_flight = true;
                HIPCHK(hipMemsetAsync(w->err.p, 0, sizeof(int) * groups, w->stream));
                Timed tm(rs, w);
                if ((rc = run_stage1(rs, w, tm, w->blobs.as<uint8_t>(), w->commitments.as<uint8_t>(), w->proofs.as<uint8_t>(), (int)n_tot, (int)n_loc,
                                     w->records.as<uint8_t>(), w->pts.as<G1Affine>(), w->err.as<int>(), false, host_hash ? &hf : nullptr))) return rc;
                if ((rc = join_side(w))) return rc;
                HIPCHK(hipMemcpyAsync(w->h_err.p, w->err.p, sizeof(int) * groups, hipMemcpyDeviceToHost, w->stream));
                HIPCHK(hipStreamSynchronize(w->stream));
                w->in_flight = false;
                tm.collect();
                for (size_t g = 0; g < groups; g++) st1[d][g] = status_from_err(w->h_err.as<int>()[g]);
                return KZG355_OK;
            }));
        }
        int first = KZG355_OK;
        for (auto &f : fut) { const int rc = f.get(); if (rc != KZG355_OK && first == KZG355_OK) first = rc; }
        if (first != KZG355_OK) return first;
    }
    // the exchange: ONE all-gather of the records (equal blocks: RCCL over xGMI) or peer copies into the batch's stage-2 device
    const bool equal_blocks = npg % D == 0;
    const bool use_rccl = m->exchange == 1 && equal_blocks;
    const size_t shard_bytes = (size_t)RECORD_BYTES * cnt[0] * groups;
    if (use_rccl) {
        for (size_t d = 0; d < D; d++) {
            if (hipSetDevice(gs[d]->s->device) != hipSuccess) return KZG355_NO_DEVICE;
            int rc = gs[d]->w->small.ensure(shard_bytes * D);                   // [rank][batch][block] as the collective delivers it
            if (rc) return rc;
        }
        std::lock_guard<std::mutex> lk(m->ex_mu);
        for (size_t d = 0; d < D; d++) gs[d]->w->in_flight = true;     // every replica's stream carries the collective: quiesce() drains them on any exit
        int bad = m->rccl.GroupStart();
        for (size_t d = 0; d < D && !bad; d++) {
            if (hipSetDevice(gs[d]->s->device) != hipSuccess) { bad = 1; break; }
            bad = m->rccl.AllGather(gs[d]->w->records.p, gs[d]->w->small.p, shard_bytes, /* ncclUint8 */ 1, m->comms[d], gs[d]->w->stream);
        }
        if (m->rccl.GroupEnd() != 0 || bad) return KZG355_DEVICE_ERROR;
        m->n_allgathers++;
    } else m->n_peer_exchanges++;
    // stage 2: batch g on device g mod D, over the records of all blocks in transcript order
    std::vector<int> rc_dev(D, KZG355_OK);
    {
        std::vector<std::future<int>> fut;
        for (size_t t = 0; t < D && t < groups; t++) {
            fut.push_back(std::async(std::launch::async, [&, t]() -> int {
                kzg355_settings *rs = gs[t]->s; Workspace *w = gs[t]->w;
                if (hipSetDevice(rs->device) != hipSuccess) return KZG355_NO_DEVICE;
                std::vector<size_t> mine;
                for (size_t g = t; g < groups; g += D) mine.push_back(g);
                const size_t G = mine.size();
                int rc;
                DevBuf &gath = w->q;                                               // the gathered records of this device's batches
                if ((rc = gath.ensure((size_t)RECORD_BYTES * npg * G))) return rc;
                for (size_t k = 0; k < G; k++) {
                    const size_t g = mine[k];
                    for (size_t d = 0; d < D; d++) {
                        if (!cnt[d]) continue;
                        uint8_t *dst = gath.as<uint8_t>() + (size_t)RECORD_BYTES * (k * npg + off[d]);
                        const size_t bytes = (size_t)RECORD_BYTES * cnt[d];
                        if (use_rccl) {           // local permute out of the all-gathered [rank][batch][block] layout
                            const uint8_t *src = w->small.as<uint8_t>() + shard_bytes * d + (size_t)RECORD_BYTES * cnt[d] * g;
                            HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, w->stream));
                        } else {
                            const uint8_t *src = gs[d]->w->records.as<uint8_t>() + (size_t)RECORD_BYTES * cnt[d] * g;
                            if (gs[d]->s->device == rs->device) HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, w->stream));
                            else HIPCHK(hipMemcpyPeerAsync(dst, rs->device, src, gs[d]->s->device, bytes, w->stream));
                        }
                    }
                }
                // the decoded points of every block, into [batch][commitments of all blocks | proofs of all blocks] (the stage-2 layout)
                DevBuf &gpts = w->partials;
                if ((rc = gpts.ensure(sizeof(G1Affine) * 2 * npg * G))) return rc;
                for (size_t k = 0; k < G; k++) {
                    const size_t g = mine[k];
                    for (size_t d = 0; d < D; d++) {
                        if (!cnt[d]) continue;
                        for (int half = 0; half < 2; half++) {
                            G1Affine *dst = gpts.as<G1Affine>() + (k * 2 + half) * npg + off[d];
                            const G1Affine *src = gs[d]->w->pts.as<G1Affine>() + (g * 2 + half) * cnt[d];
                            const size_t bytes = sizeof(G1Affine) * cnt[d];
                            if (gs[d]->s->device == rs->device) HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, w->stream));
                            else HIPCHK(hipMemcpyPeerAsync(dst, rs->device, src, gs[d]->s->device, bytes, w->stream));
                        }
                    }
                }
                if ((rc = w->err.ensure(sizeof(int) * G)) || (rc = w->ok.ensure(sizeof(int) * G))) return rc;
                if ((rc = w->h_ok.ensure(sizeof(int) * G)) || (rc = w->h_err.ensure(sizeof(int) * G))) return rc;
                w->in_flight = true;
                HIPCHK(hipMemsetAsync(w->err.p, 0, sizeof(int) * G, w->stream));
                Timed tm(rs, w);
                // (lone call: this thread owns the device's share of a synchronous call -- the batch challenge of few records may take its host round trip)
                if ((rc = run_stage2(rs, w, tm, gath.as<uint8_t>(), (int)npg, (int)G, 0, gpts.as<G1Affine>(), w->err.as<int>(), w->ok.as<int>(),
                        true))) return rc;
                if (dump) {
                    if ((rc = w->out48.ensure(128 * G)) || (rc = w->h_out.ensure(128 * G))) return rc;
                    launch_dump_intermediates(w->scal_a.as<uint32_t>(), w->pair_pts.as<PairPt>(), (int)npg, (int)G, w->out48.as<uint8_t>(), w->stream);
                    HIPCHK(hipMemcpyAsync(w->h_out.p, w->out48.p, 128 * G, hipMemcpyDeviceToHost, w->stream));
                }
                HIPCHK(hipMemcpyAsync(w->h_ok.p, w->ok.p, sizeof(int) * G, hipMemcpyDeviceToHost, w->stream));
                HIPCHK(hipMemcpyAsync(w->h_err.p, w->err.p, sizeof(int) * G, hipMemcpyDeviceToHost, w->stream));
                HIPCHK(hipStreamSynchronize(w->stream));
                w->in_flight = false;
                tm.collect();
                for (size_t k = 0; k < G; k++) {
                    const size_t g = mine[k];
                    if (dump) memcpy(dump + 128 * g, w->h_out.as<uint8_t>() + 128 * k, 128);
                    int st = status_from_err(w->h_err.as<int>()[k]);
                    // an Err on any block is an Err of the batch (the `?`s of kzg.rs:673-682)
                    for (size_t d = 0; d < D; d++) if (st == KZG355_OK) st = st1[d][g];
                    if (status) status[g] = st;
                    if (st == KZG355_OK) ok[g] = w->h_ok.as<int>()[k] != 0;
                    else if (rc_dev[t] == KZG355_OK) rc_dev[t] = st;
                }
                return KZG355_OK;
            }));
        }
        int first = KZG355_OK;
        for (auto &f : fut) { const int rc = f.get(); if (rc != KZG355_OK && first == KZG355_OK) first = rc; }
        if (first != KZG355_OK) return first;
    }
    for (size_t g = 0; g < groups; g++) {            // first failing batch in batch order
        int st = KZG355_OK;
        if (status) st = status[g];
        else for (size_t t = 0; t < D; t++) if (rc_dev[t] != KZG355_OK) st = rc_dev[t];
        if (st != KZG355_OK) return st;
    }
    return KZG355_OK;
}

int multi_verify_many(bool *ok, int *status, const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs, size_t npg, size_t groups,
                      const kzg355_settings *cs) {
    MultiDev *m = cs->multi;
    const size_t D = m->rep.size(), BB = blob_bytes_of(cs);
    const bool force_sharded = cs->force_sharded && npg >= D;         // (test hook: kzg355_options.force_sharded)
    // enough independent batches (or batches too small to cut): ranges of batches, no exchange
    if (!force_sharded && (groups >= D || npg < 2 * D)) {
        return fan_out(D, groups, [&](size_t d, size_t g0, size_t n) -> int {
            if (hipSetDevice(m->rep[d]->device) != hipSuccess) return KZG355_NO_DEVICE;
            return single_verify_many(ok + g0, status ? status + g0 : nullptr, blobs + BB * npg * g0, commitments + 48 * npg * g0, proofs + 48 * npg * g0, npg,
                    n, m->rep[d]);
        });
    }
    return multi_verify_sharded(ok, status, blobs, commitments, proofs, npg, groups, cs);
}


}  // namespace kzg355_impl

extern "C" {
#pragma GCC visibility push(default)

int kzg355_load_trusted_setup_devices(const uint8_t *g1_bytes, size_t n1, const uint8_t *g2_bytes, size_t n2, const int *devices, size_t n_devices,
                                      kzg355_settings **out) {
    kzg355_options o;
    kzg355_options_from_env(&o);
    return load_devices(g1_bytes, n1, g2_bytes, n2, devices, n_devices, o, out);
}
int kzg355_load_trusted_setup_ex(const uint8_t *g1_bytes, size_t n1, const uint8_t *g2_bytes, size_t n2, const int *devices, size_t n_devices,
                                 const kzg355_options *options, kzg355_settings **out) {
    const kzg355_options o = options_of(options);
    if (devices && n_devices) return load_devices(g1_bytes, n1, g2_bytes, n2, devices, n_devices, o, out);
    return load_on_device(g1_bytes, n1, g2_bytes, n2, -1, o, out);
}

int kzg355_load_trusted_setup(const uint8_t *g1_bytes, size_t n1, const uint8_t *g2_bytes, size_t n2, kzg355_settings **out) {
    kzg355_options o;
    kzg355_options_from_env(&o);
    // KZG355_DEVICES=0,1,...: the handle spans those devices; otherwise KZG355_DEVICE / the current device
    const std::vector<int> devs = env_device_list();
    if (!devs.empty()) return load_devices(g1_bytes, n1, g2_bytes, n2, devs.data(), devs.size(), o, out);
    return load_on_device(g1_bytes, n1, g2_bytes, n2, -1, o, out);
}
int kzg355_settings_device_count(const kzg355_settings *s) { return !s ? 0 : s->multi ? (int)s->multi->rep.size() : 1; }
int kzg355_settings_exchange_stats(const kzg355_settings *s, long *allgathers, long *peer_exchanges) {
    if (!s || !allgathers || !peer_exchanges) return KZG355_BADARGS;
    *allgathers = s->multi ? s->multi->n_allgathers.load() : 0L;
    *peer_exchanges = s->multi ? s->multi->n_peer_exchanges.load() : 0L;
    return s->multi ? s->multi->exchange : -1;
}

void kzg355_free_trusted_setup(kzg355_settings *s) {
    if (!s) return;
    if (MultiDev *m = s->multi) {
        s->multi = nullptr;
        for (size_t i = 0; i < m->comms.size(); i++) if (m->comms[i]) m->rccl.CommDestroy(m->comms[i]);
        for (size_t i = 1; i < m->rep.size(); i++) free_single(m->rep[i]);
        delete m;
    }
    free_single(s);
}


#pragma GCC visibility pop
}  // extern "C"
