// host_pipeline.hip -- host buffers through chunks and workspaces: H2D, kernels and results overlapped (host side of libkzg355.so; see engine.h).
#include "engine.h"

namespace kzg355_impl {

int stage_to_device(Workspace *w, DevBuf &dst, const uint8_t *src, size_t bytes) {
    int rc = dst.ensure(bytes);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(dst.p, src, bytes, hipMemcpyHostToDevice, w->stream));
    return KZG355_OK;
}

// Caller memory -> this workspace's pinned slot (parallel host copy) -> device (asynchronous DMA on w->stream).
// The workspace must be idle (its previous launch set collected): the pinned slot is reused.
int stage_via_pinned(kzg355_settings *s, Workspace *w, PinBuf &pin, size_t pin_off, DevBuf &dst, const uint8_t *src, size_t bytes) {
    int rc;
    if ((rc = dst.ensure(bytes))) return rc;
    if (s->host_pool) s->host_pool->copy(pin.as<uint8_t>() + pin_off, src, bytes);
    else memcpy(pin.as<uint8_t>() + pin_off, src, bytes);
    HIPCHK(hipMemcpyAsync(dst.p, pin.as<uint8_t>() + pin_off, bytes, hipMemcpyHostToDevice, w->stream));
    return KZG355_OK;
}

int host_pipeline(const HostCall &hc, size_t units, const kzg355_settings *cs) {
    kzg355_settings *s = const_cast<kzg355_settings *>(cs);
    struct InFlight { std::atomic<int> &n; InFlight(std::atomic<int> &c) : n(c) { n++; } ~InFlight() { n--; } } in_flight(s->calls_in_flight);
    const size_t BB = blob_bytes_of(cs);
    const size_t unit_bytes = BB * hc.npg;
    size_t upc = s->chunk_bytes / unit_bytes;                                       // units per full-size chunk
    if (upc < 1) upc = 1;
    if (upc > units) upc = units;
    // One launch set is a ~10 ms chain of latency-bound kernels whatever its size (up to ~1000 batches), and the chains of
    // successive chunks mostly serialise on the card: a chunk must carry more than 10 ms of PCIe traffic (~600 MiB) for the link,
    // not the chain, to set the pace.  So: full-size chunks (1 GiB by default), except a small first one (64 MiB) that gets the
    // card started after a ~2 ms copy; its chain runs under the H2D of the second chunk.
    std::vector<size_t> sizes;
    {
        size_t head = ((size_t)64 << 20) / unit_bytes;
        if (head < 1) head = 1;
        size_t left = units;
        if (left > upc) { const size_t c = head < left ? head : left; sizes.push_back(c); left -= c; }
        while (left) { const size_t c = upc < left ? upc : left; sizes.push_back(c); left -= c; }
    }
    const size_t nchunks = sizes.size();
    const int W = (int)(nchunks < (size_t)s->chunks_in_flight ? nchunks : (size_t)s->chunks_in_flight);
    // a call that fits one small chunk (a single 64-blob batch is 8 MiB) goes straight from caller memory: the runtime's own
    // staged copy moves it at link speed (measured 52 GB/s for 8 MiB), one pass over the bytes instead of two
    const bool direct = !s->pinned_ring || (nchunks == 1 && unit_bytes * units <= ((size_t)32 << 20));
    // Fiat-Shamir challenges hashed on the host (host_sha256.h): single-chunk verify / blob-proof calls of the mainnet preset, up to the
    // measured crossover (the device hash is a 3.7 ms chain for ANY call of up to 32,768 blobs; T host threads take ~35 us x blobs / T)
    const bool host_hash = nchunks == 1 && (hc.kind == 0 || hc.kind == 2) && !is_small(cs) && s->host_pool &&
                           (s->host_hash > 0 || (s->host_hash == 0 && units * hc.npg <= (size_t)s->host_hash_max));
    std::vector<WsGuard *> guards;
    struct Cleanup { std::vector<WsGuard *> &g; ~Cleanup() { for (size_t i = g.size(); i-- > 0;) delete g[i]; } } cleanup{guards};
    std::vector<Timed> tms;
    for (int i = 0; i < W; i++) {
        guards.push_back(new WsGuard(cs));
        if (!guards.back()->w) return KZG355_NO_DEVICE;
        tms.emplace_back(s, guards.back()->w);
    }
    struct Pending { size_t u0, cnt; };
    std::vector<Pending> pend(W, Pending{0, 0});
    int first = KZG355_OK;
    auto collect = [&](int slot) -> int {
        if (!pend[slot].cnt) return KZG355_OK;
        Workspace *w = guards[slot]->w;
        const size_t u0 = pend[slot].u0, cnt = pend[slot].cnt;
        pend[slot].cnt = 0;
        int rc = hc.kind == 0 ? verify_collect(w, tms[slot], hc.ok + u0, hc.status ? hc.status + u0 : nullptr, (int)cnt)
                              : msm_op_collect(w, tms[slot], hc.out48 + 48 * u0, hc.status ? hc.status + u0 : nullptr, cnt,
                                      hc.ys_out ? hc.ys_out + 32 * u0 : nullptr);
        if (hc.records_out && hc.kind == 0 && hipMemcpy(hc.records_out + (size_t)RECORD_BYTES * hc.npg * u0, w->records.p, (size_t)RECORD_BYTES * hc.npg * cnt,
                hipMemcpyDeviceToHost) != hipSuccess) return KZG355_DEVICE_ERROR;
        if (rc == KZG355_NO_DEVICE || rc == KZG355_NO_MEMORY || rc == KZG355_DEVICE_ERROR) return rc;
        if (rc != KZG355_OK && first == KZG355_OK) first = rc;
        return KZG355_OK;
    };
    const bool dbg = debug_pipe();
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_wait = 0, t_stage = 0, t_enq = 0, t_alloc = 0;
    std::vector<hipEvent_t> dev_ev;                                  // debug only: H2D start / H2D end / kernels end per chunk
    if (dbg) { dev_ev.resize(3 * nchunks); for (auto &e : dev_ev) (void)hipEventCreate(&e); }
    const double t_begin = now();
    size_t u0 = 0;
    for (size_t k = 0; k < nchunks; u0 += sizes[k], k++) {
        const size_t cnt = sizes[k];
        const int slot = (int)(k % W);
        Workspace *w = guards[slot]->w;
        int rc;
        double t0 = now();
        if ((rc = collect(slot))) return rc;                                       // frees this slot's pinned and device buffers
        t_wait += now() - t0; t0 = now();
        const size_t nb = cnt * hc.npg, off = u0 * hc.npg;
        HostFront hf;
        if (direct && host_hash) {
            // Small call whose challenges need a hash of every blob (verify, blob proof): the host threads hash while the copies and the
            // point kernels are queued; the blobs themselves go to the device inside the enqueue below, behind the point kernels.
            if ((rc = w->h_digests.ensure(32 * nb)) || (rc = w->digests.ensure(32 * nb)) || (rc = w->blobs.ensure(BB * nb))) return rc;
            uint8_t *dig = w->h_digests.as<uint8_t>();
            const uint8_t *hb = hc.blobs + BB * off, *hcm = hc.commitments + 48 * off;
            const uint64_t n_fe = (uint64_t)s->t.n_fe; const int impl = s->sha_impl;
            auto job = [=](size_t k) { kzg_host::challenge_digests(dig + 64 * k, hb + BB * 2 * k, BB, hcm + 96 * k, nb - 2 * k < 2 ? nb - 2 * k : 2, n_fe,
                    impl); };
            if (s->host_pool) {                                  // (every call its own job object on the shared workers)
                hf.job = s->host_pool->begin((nb + 1) / 2, job);
                hf.pool = s->host_pool; hf.h_blobs = hb; hf.bytes = BB * nb; hf.running = true;
                s->n_host_hashed++;
            }
        }
        if (hf.running) {
            if ((rc = stage_to_device(w, w->commitments, hc.commitments + 48 * off, 48 * nb))) return rc;
            if (hc.proofs && (rc = stage_to_device(w, w->proofs, hc.proofs + 48 * off, 48 * nb))) return rc;
        } else if (direct) {
            if ((rc = stage_to_device(w, w->blobs, hc.blobs + BB * off, BB * nb))) return rc;
            if (hc.commitments && (rc = stage_to_device(w, w->commitments, hc.commitments + 48 * off, 48 * nb))) return rc;
            if (hc.proofs && (rc = stage_to_device(w, w->proofs, hc.proofs + 48 * off, 48 * nb))) return rc;
        } else {
            if ((rc = w->h_stage.ensure(BB * nb))) return rc;
            if ((rc = w->h_stage_cp.ensure(96 * nb))) return rc;
            if ((rc = w->blobs.ensure(BB * nb))) return rc;
            t_alloc += now() - t0; t0 = now();
            if (dbg) (void)hipEventRecord(dev_ev[3 * k], w->stream);
            if ((rc = stage_via_pinned(s, w, w->h_stage, 0, w->blobs, hc.blobs + BB * off, BB * nb))) return rc;
            if (dbg) (void)hipEventRecord(dev_ev[3 * k + 1], w->stream);
            if (hc.commitments && (rc = stage_via_pinned(s, w, w->h_stage_cp, 0, w->commitments, hc.commitments + 48 * off, 48 * nb))) return rc;
            if (hc.proofs && (rc = stage_via_pinned(s, w, w->h_stage_cp, 48 * nb, w->proofs, hc.proofs + 48 * off, 48 * nb))) return rc;
        }
        if (hc.kind == 3 && (rc = stage_to_device(w, w->small, hc.zs + 32 * off, 32 * nb))) return rc;      // the evaluation points (kzg.rs:446-457)
        t_stage += now() - t0; t0 = now();
        HostFront *hfp = hf.running ? &hf : nullptr;
        if (hc.kind == 0) rc = verify_enqueue(s, w, tms[slot], w->blobs.as<uint8_t>(), w->commitments.as<uint8_t>(), w->proofs.as<uint8_t>(), (int)hc.npg,
                (int)cnt, 0, 0, hfp, nchunks == 1);
        else rc = msm_op_enqueue(s, w, tms[slot], w->blobs.as<uint8_t>(), hc.kind == 2 ? w->commitments.as<uint8_t>() : nullptr, cnt, hfp,
                hc.kind == 3 ? w->small.as<uint8_t>() : nullptr);
        if (rc) return rc;
        if (dbg && !direct) (void)hipEventRecord(dev_ev[3 * k + 2], w->stream);
        t_enq += now() - t0;
        pend[slot] = Pending{u0, cnt};
    }
    const double t_loop = now();
    for (size_t j = 0; j < (size_t)W; j++) {                                       // remaining chunks, oldest first
        int rc = collect((int)((nchunks + j) % W));
        if (rc) return rc;
    }
    if (dbg && !direct) {
        for (size_t k = 0; k < nchunks; k++) {
            float h2d = 0, ker = 0, since = 0;
            (void)hipEventElapsedTime(&h2d, dev_ev[3 * k], dev_ev[3 * k + 1]); (void)hipEventElapsedTime(&ker, dev_ev[3 * k + 1], dev_ev[3 * k + 2]);
            (void)hipEventElapsedTime(&since, dev_ev[0], dev_ev[3 * k]);
            fprintf(stderr, "  chunk %zu (%zu units): H2D starts at %.1f ms, takes %.1f ms (%.1f GB/s), kernels %.1f ms\n", k, sizes[k], since, h2d,
                    sizes[k] * unit_bytes / (h2d * 1e6), ker);
        }
    }
    for (auto &e : dev_ev) (void)hipEventDestroy(e);
    if (dbg) fprintf(stderr, "kzg355 pipe: %zu chunks, W %d: alloc %.1f ms, host copy + H2D enqueue %.1f ms, kernel enqueue %.1f ms, waits in loop %.1f ms, "
                             "drain %.1f ms, total %.1f ms\n",
                     nchunks, W, t_alloc, t_stage, t_enq, t_wait, now() - t_loop, now() - t_begin);
    return first;
}

}  // namespace kzg355_impl

