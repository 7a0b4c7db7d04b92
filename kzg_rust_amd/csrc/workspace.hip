// workspace.hip -- workspace pool, side streams, dispatch helpers (host side of libkzg355.so; see engine.h).
#include "engine.h"

namespace kzg355_impl {

Workspace *ws_acquire(kzg355_settings *s) {
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (!s->pool.empty()) { Workspace *w = s->pool.back(); s->pool.pop_back(); return w; }
    }
    Workspace *w = new Workspace();
    if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) { w->stream = nullptr; delete w; return nullptr; }
    w->own_stream = w->stream;
    // (ONE side stream per handle, created on first use and shared by its workspaces: HIP multiplexes streams onto a handful of
    // hardware queues -- 4 unless GPU_MAX_HW_QUEUES says otherwise -- and two workspaces whose main streams land on the same queue
    // run their launch sets one after the other.  Measured: with a side stream per workspace no more than two calls overlapped.)
    for (hipEvent_t *e : {&w->ev_fork, &w->ev_join, &w->ev_pts, &w->ev_shift, &w->ev_stage, &w->ev_done, &w->ev_fork2})
        if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) { *e = nullptr; delete w; return nullptr; }
    bool ok = true;
    for (auto &e : w->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    w->ev_ok = ok;
    if (!ok) { delete w; return nullptr; }
    return w;
}

void ws_release(kzg355_settings *s, Workspace *w) {
    std::lock_guard<std::mutex> lk(s->mu);
    s->pool.push_back(w);
}

// the handle's shared side stream, created on first use
bool ensure_side(kzg355_settings *s, Workspace *w) {
    if (!w->side) {
        if (hipStreamCreateWithFlags(&w->side, hipStreamNonBlocking) != hipSuccess) { w->side = nullptr; (void)hipGetLastError(); }
        else w->owns_side = true;
        if (w->side) return true;                                  // (else: fall back to the handle's shared stream)
    }
    if (!w->side) {
        std::lock_guard<std::mutex> lk(s->mu);
        if (!s->side_stream && hipStreamCreateWithFlags(&s->side_stream, hipStreamNonBlocking) != hipSuccess) { s->side_stream = nullptr;
                (void)hipGetLastError(); }
        w->side = s->side_stream;
    }
    return w->side != nullptr;
}

bool ensure_side2(kzg355_settings *s, Workspace *w) {
    // More calls in flight on the handle than a third of the runtime's hardware queues: the window shifts follow the decoding on the first
    // side stream (enqueue_points_beside's one-side-stream form: +0.4 ms on the call's critical path) -- two streams that share a queue run
    // one after the other whatever their calls are, and a call whose main chain waits behind another call's side work loses milliseconds.
    if (!w->side2 && s->calls_in_flight.load() * 3 > s->hw_queues) return false;
    if (!w->side2) {
        if (hipStreamCreateWithFlags(&w->side2, hipStreamNonBlocking) != hipSuccess) { w->side2 = nullptr; (void)hipGetLastError(); }
        else w->owns_side2 = true;
        if (w->side2) return true;
    }
    if (!w->side2) {
        std::lock_guard<std::mutex> lk(s->mu);
        if (!s->side2_stream && hipStreamCreateWithFlags(&s->side2_stream, hipStreamNonBlocking) != hipSuccess) { s->side2_stream = nullptr;
                (void)hipGetLastError(); }
        w->side2 = s->side2_stream;
    }
    return w->side2 != nullptr;
}

int lincomb_form(const kzg355_settings *s, int npg, int groups) {
    const bool bucket_ok = npg >= 8 && npg <= 4096, pre_ok = lincomb_preshift_fits(npg, groups);
    if (s->lincomb_mode == LC_FORM_PRESHIFT) return pre_ok ? LC_FORM_PRESHIFT : LC_FORM_WINDOW;
    if (s->lincomb_mode == LC_FORM_BUCKET) return bucket_ok ? LC_FORM_BUCKET : LC_FORM_WINDOW;
    if (s->lincomb_mode == LC_FORM_WINDOW) return LC_FORM_WINDOW;
    // many independent single-proof checks (the *_many forms of verify_kzg_proof / verify_blob_kzg_proof): four ladder lanes per check
    if (npg == 1 && !pre_ok) return LC_FORM_SINGLE;
    // few batches: shift every point under the hash, finish with ~25 additions; many: least issue work (buckets); in between: per-term ladders
    if (pre_ok) return LC_FORM_PRESHIFT;
    return bucket_ok && groups >= 64 ? LC_FORM_BUCKET : LC_FORM_WINDOW;
}

int status_from_err(int err) {
    if (err == 0) return KZG355_OK;
    return KZG355_BADARGS;   // validate_kzg_g1 / bytes_to_bls_field failures are Error::BadArgs (utils.rs:268, 292, 304)
}

}  // namespace kzg355_impl

