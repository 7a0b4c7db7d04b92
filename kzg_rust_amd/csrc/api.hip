// api.hip -- host side of libkzg355.so: the C ABI of include/kzg355.h on top of the HIP kernels.
// Mirrors the control flow of the reference's `impl Kzg` forwards and the functions behind them
// (src/kzg.rs:401-693, 833-979): argument checks and early exits happen here, all arithmetic on the device.
// There is no CPU fallback: without a usable HIP device every entry point returns KZG355_NO_DEVICE (a HIP call that fails on a
// device that exists: KZG355_DEVICE_ERROR).
#include "../../include/kzg355.h"
#include "kernels.h"
#include "host_sha256.h"
#include "host_pool.h"

#include <sched.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <dlfcn.h>
#include <functional>
#include <future>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace kzg;

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            if (getenv("KZG355_DEBUG")) fprintf(stderr, "kzg355: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return _e == hipErrorOutOfMemory ? KZG355_NO_MEMORY : (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice) ? KZG355_NO_DEVICE : KZG355_DEVICE_ERROR; \
        }                                                                                              \
    } while (0)

// The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue run one
// after the other.  A small call of this library is a chain on three streams, so the default lets one such call run at full speed and
// serialises the streams of concurrent ones (threads of n = 64 calls on one handle, median ms per call at 1 / 2 / 4 / 8 threads: 2.2 / 3.0 / 5.0 / 6.9
// with 4 queues, 2.2 / 2.3 / 2.9 / 3.3 with 24; profiles/r04/concurrent_small_calls.txt).  The variable belongs to the PROCESS: a host that serves
// concurrent small calls exports GPU_MAX_HW_QUEUES=24 before its first HIP call (kzg_rust_amd/_lib.py and bench.py do).  The library
// does not touch the environment (round 4 set it from a constructor: ADVICE r4); it reads the variable once per handle and otherwise assumes 4.

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return KZG355_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes < 256 ? 256 : bytes;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; (void)hipGetLastError(); return KZG355_NO_MEMORY; }
        cap = want;
        return KZG355_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return KZG355_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        size_t want = bytes < 256 ? 256 : bytes;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; (void)hipGetLastError(); return KZG355_NO_MEMORY; }
        cap = want;
        return KZG355_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Per-call scratch: a private stream plus grow-on-demand device buffers.  One workspace serves one call at a time;
// concurrent host threads get different workspaces from the pool in the settings handle.
struct Workspace {
    hipStream_t stream = nullptr, side = nullptr, side2 = nullptr;     // side, side2: kernels independent of the main chain (point validation; window shifts)
    hipStream_t own_stream = nullptr;                                  // `stream` is this one except while a submitted set borrows the handle's pipeline streams
    hipEvent_t ev_stage = nullptr, ev_done = nullptr, ev_fork2 = nullptr;   // submit / collect: stage 1 queued on pipe_main is done; the whole set is done; this set's hash is done
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_pts = nullptr, ev_shift = nullptr;
    // work on the side streams that the main stream has not waited for yet: everything (ev_join: the validation verdicts are in the error
    // words), the decoded points alone (ev_pts; recorded only when they are ready before the verdicts), the window shifts (ev_shift)
    bool side_pending = false, pts_pending = false, shift_pending = false;
    DevBuf blobs, commitments, proofs, records, z, y, pts, scal_a, scal_b, scal_c, pair_pts, pair_f, ok, err, digits, partials, q, out48, small, lc_partials, shifts, digests, zpow, qprep;
    bool shift_ready = false;        // stage 1 has queued the window shifts of this launch set's points (pre-shifted lincomb)
    PinBuf h_ok, h_err, h_out, h_digests, h_records, h_rdig;
    PinBuf h_stage, h_stage_cp;      // pinned staging of caller memory (blobs; commitments | proofs): slot of the host pipeline
    hipEvent_t ev[32];
    bool ev_ok = false;
    hipEvent_t ev_d2h[8] = {};       // device-resident small calls hashed on the host: one event per chunk of the blobs' way back (host_hash_from_device)
    bool in_flight = false;          // a launch set has been enqueued on `stream` and not collected yet
    bool owns_side = false, owns_side2 = false;
    hipStream_t borrowed[2] = {nullptr, nullptr};   // the handle's pipeline streams while a submitted set of this workspace is on them
    // Wait for everything this workspace has in flight (a call that fails midway must not hand a busy workspace back to the pool).
    void quiesce() {
        if (in_flight || side_pending || pts_pending || shift_pending) {
            if (side) (void)hipStreamSynchronize(side);
            if (side2) (void)hipStreamSynchronize(side2);
            for (hipStream_t st : borrowed) if (st) (void)hipStreamSynchronize(st);
            if (stream) (void)hipStreamSynchronize(stream);
        }
        borrowed[0] = borrowed[1] = nullptr;
        stream = own_stream;
        in_flight = false; side_pending = false; pts_pending = false; shift_pending = false; shift_ready = false;
    }
    ~Workspace() {
        for (DevBuf *b : {&blobs, &commitments, &proofs, &records, &z, &y, &pts, &scal_a, &scal_b, &scal_c, &pair_pts, &pair_f, &ok, &err, &digits, &partials, &q, &out48, &small, &lc_partials, &shifts, &digests, &zpow, &qprep}) b->release();
        h_ok.release(); h_err.release(); h_out.release(); h_stage.release(); h_stage_cp.release(); h_digests.release(); h_records.release(); h_rdig.release();
        if (ev_ok) for (auto &e : ev) (void)hipEventDestroy(e);
        for (hipEvent_t e : {ev_fork, ev_join, ev_pts, ev_shift, ev_stage, ev_done, ev_fork2}) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_d2h) if (e) (void)hipEventDestroy(e);
        if (own_stream) (void)hipStreamDestroy(own_stream);       // (side is the handle's shared stream unless owns_side)
        if (owns_side && side) (void)hipStreamDestroy(side);
        if (owns_side2 && side2) (void)hipStreamDestroy(side2);
    }
};

}  // namespace

struct MultiDev;
struct kzg355_settings {
    int device = 0;
    MultiDev *multi = nullptr;      // handles created over several devices: the replicas and the exchange (owner handle only)
    DeviceTables t{};
    DevBuf roots, eval_tab, wide, msm_table, lines, lines_inf, g1_first2, lines_w, frob, prog, scheds;
    bool lane_pairing = false;
    int split_parts = 1, split_streams = 2;   // KZG355_SPLIT=parts[,streams]: device-resident verify calls as several overlapped launch sets (default: one)
    int challenge_form = 0;   // 0 by size, 1 one-wave kernel, 2 two-wave kernel (KZG355_CHALLENGE=1w|2w)
    int quotient_form = 0;    // k_quotient_tree: 0 by size, 2 / 4 / 6 = 2^form leaves per lane (KZG355_QUOTIENT_FORM: tuning knob and test hook, not an option)
    int lc_chain_from = 1024;      // batches per launch set from which the bucket form ends in one Horner chain per class (KZG355_LC_CHAIN_FROM).  Round 4, with the chain walked by quads and sets kept in flight (blobs/s, three sets in flight, 16 chains per class against one): 512 batches 2.81 M either way, 1024 3.99 -> 4.03 M, 2048 4.14 -> 4.26 M, 4096 4.25 -> 4.36 M (profiles/r04/chain_from_sweep.txt); one set at a time it is within +-2 % from 512 to 4096
    int rhash_lanes_from = 1024;   // batches per launch set from which the r-transcripts are hashed one lane per batch (KZG355_RHASH_LANES_FROM); measured: 1024 batches of 512 records 6.75 -> 3.47 ms, 8192 of 64: 2.44 -> 0.57 ms
    int lincomb_mode = 0;     // 0 auto, 1 windowed per-term, 2 bucket method, 3 pre-shifted (KZG355_LINCOMB=window|bucket|preshift)
    int beside_max_blobs = 16384;  // blobs per launch set up to which the point kernels run on side streams beside the hash chain (64 per CU)
    int cu_count = 256;            // compute units of the device: the thresholds above and below are multiples of it (load_on_device)
    int pairing_two_wave_upto = 256;     // batches per launch set up to which a pairing runs its two Miller loops on two waves (1 per CU)
    int miller_segments = 0;             // ... and on how many segments per loop (k_pairing_coop_split; 0: launch_pairing's default; KZG355_MILLER_SEGMENTS=1..4)
    int pairing_hard12_from = 4096;      // batches per launch set from which the final exponentiation's hard part runs twelve lanes per check (16 per CU; KZG355_PAIRING_HARD12_FROM, 0: never)
    int challenge_two_wave_upto = 32768; // blobs per launch set up to which the Fiat-Shamir hash runs as producer / consumer wave pairs (2 workgroups per CU)
    std::mutex mu;
    hipStream_t side_stream = nullptr, side2_stream = nullptr;   // shared by the workspaces (point validation / window shifts of small calls next to the main chain)
    hipStream_t pipe_main = nullptr, pipe_tail = nullptr;         // submit / collect: stage 1 of every submitted set in order on pipe_main, stage 2 on pipe_tail
    std::mutex pipe_mu;                                           // orders the submits / collects that queue work on the two
    struct kzg355_ticket *pending_tail = nullptr;                 // the submitted set whose stage 2 is not queued yet (it goes out behind the next set's hash)
    std::atomic<int> tickets_out{0};                              // submitted and not yet collected
    bool free_deferred = false;                                   // kzg355_free_trusted_setup came while tickets were out (the caller's bug): the handle stays alive until the last of them is collected (under pipe_mu)
    bool own_side_streams = true;    // side streams per workspace (round 4; KZG355_SIDE=shared: one pair per handle, round 3's form) -- measured with
                                     // 4 threads of n = 64 calls: median call 5.0-7.9 ms shared, 3.6-5.2 ms own (4 hardware queues), 2.5-3.9 ms own with 8 queues
    std::atomic<int> calls_in_flight{0};   // host-buffer calls inside host_pipeline right now
    int hw_queues = 4;                     // GPU_MAX_HW_QUEUES as the process has it when the handle is loaded (the runtime's default is 4)
    int submit_mode = 0;             // 0 by size; 1: every submitted set on its workspace's own stream; 2: two-stage software pipeline over pipe_main / pipe_tail (KZG355_SUBMIT=sets|pipeline)
    int host_hash = 0;               // Fiat-Shamir hashing of host-buffer calls on host threads: 0 by size (<= host_hash_max blobs), 1 always, -1 never (KZG355_HOST_HASH=auto|on|off)
    int host_hash_max = 4096;        // blobs per call up to which the host hashes (KZG355_HOST_HASH_MAX): measured, profiles/r03/host_hash_crossover_v4.txt: host route ahead up to 4096 blobs (17.1 against 18.4 ms), level at 8192
    int host_hash_device_max = 512;  // device-resident verify / blob-proof calls of up to this many blobs copy them BACK and hash on the host threads (0.16 ms of D2H per 64 blobs + ~35 us x blobs / threads against the 3.7 ms device chain); KZG355_HOST_HASH_DEVICE_MAX, 0 in the options = 512, -1 never
    int host_rhash = 0;              // batch challenge r of lone small calls hashed on the host (records copied back): 0 by size, -1 never (KZG355_HOST_RHASH=off)
    int host_rhash_loaded = 0;       // ... as the handle was loaded: kzg355_settings_set_host_hash(-1) forces -1, any other mode puts this back
    int host_rhash_max_records = 256;    // records per call up to which that is done
    int sha_impl = 0;                // host SHA-256 form: 0 auto (SHA extensions when the CPU has them), 1 portable, 2 SHA extensions (KZG355_HOST_SHA=portable|shani)
    std::atomic<long> n_host_hashed{0};   // introspection: host-buffer calls whose challenges were hashed on the host
    std::vector<Workspace *> pool;
    HostPool *host_pool = nullptr;  // created with the handle: host threads for the Fiat-Shamir hashing of small host-buffer calls and the staging copies
    size_t chunk_bytes = (size_t)1024 << 20;  // blobs per chunk of a host-buffer call (KZG355_CHUNK_MB): 1 GiB = 18 ms of PCIe traffic, more than
                                              // the ~11 ms kernel chain of a chunk even when the chains of successive chunks end up on one hardware queue
    int chunks_in_flight = 3;                 // workspaces (pinned slot + device buffers + stream) a host-buffer call rotates over
    bool pinned_ring = false;                 // KZG355_STAGING=ring: stage caller memory through the workspaces' pinned slots; default: let the
                                              // runtime lock the caller's pages and DMA from them (measured on MI355X hosts: 56 GB/s, no CPU copy)
    bool wide_table_failed = false;           // the wide-window MSM table was wanted but could not be allocated / built
    // The table is built on the first commitment / proof call (or at load: kzg355_options.msm_eager, msm_require_wide) -- a handle that
    // only ever verifies never pays for it.  msm_bits_wanted: 0 = sized from the free HBM at that moment, 8 = never, else the digit width.
    int msm_bits_wanted = 0, msm_glv = 1;
    bool msm_required = false;
    std::once_flag wide_once;
    // The table is PUBLISHED, not written into `t`: launches copy `t` by value while another thread may be building, so `t` stays as the load
    // left it and msm_to_host takes shape + rows from here (release-stored once the new table has passed its check against the bucket form).
    struct WidePub { WideShape shape; WideRow *rows; };
    WidePub wide_store{};
    std::atomic<const WidePub *> wide_pub{nullptr};
    int wide_rc = KZG355_OK;                  // what building it returned (msm_require_wide: a failure fails the calls that need it)
    bool timing = false;
    struct KStat { double last = -1, total = 0; long count = 0; };
    std::map<std::string, KStat> last_ms;
};

namespace {

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
// Makes `dev` the calling thread's current device and puts the previous one back on scope exit: an entry point of this library
// leaves the caller's current device as it found it (a host program may be driving other devices from the same thread).
struct DeviceScope {
    int prev = -1; bool changed = false;
    bool enter(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
        if (prev == dev) return true;
        if (hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); return false; }
        changed = prev >= 0;
        return true;
    }
    void hold() { if (hipGetDevice(&prev) == hipSuccess) changed = true; else (void)hipGetLastError(); }   // put the current device back whatever happens in between
    ~DeviceScope() { if (changed) (void)hipSetDevice(prev); }
};
struct WsGuard {
    kzg355_settings *s; Workspace *w;
    DeviceScope scope;
    WsGuard(const kzg355_settings *cs) : s(const_cast<kzg355_settings *>(cs)), w(nullptr) {
        if (s && scope.enter(s->device)) w = ws_acquire(s);
    }
    ~WsGuard() { if (w) { w->quiesce(); ws_release(s, w); } }     // (scope is destroyed after this body: the device goes back last)
};

// Optional per-kernel-family timing with HIP events on the launch stream (kzg355_set_kernel_timing).
struct Timed {
    kzg355_settings *s; Workspace *w; std::vector<std::pair<std::string, int>> marks; int n = 0;
    Timed(kzg355_settings *s_, Workspace *w_) : s(s_), w(w_) {}
    // Event pairs are recorded on the stream the kernel is launched on; timing never changes the schedule.
    void begin(const char *name, hipStream_t st = nullptr) {
        if (!s->timing || n + 2 > 32) return;
        (void)hipEventRecord(w->ev[n], st ? st : w->stream); marks.push_back({name, n}); n++;
    }
    void end(hipStream_t st = nullptr) {
        if (!s->timing || marks.empty() || n >= 32) return;
        (void)hipEventRecord(w->ev[n], st ? st : w->stream); n++;
    }
    void collect() {   // call after the stream has been synchronised
        if (!s->timing) return;
        std::lock_guard<std::mutex> lk(s->mu);
        for (auto &m : marks) {
            float ms = 0;
            if (m.second + 1 < n && hipEventElapsedTime(&ms, w->ev[m.second], w->ev[m.second + 1]) == hipSuccess) {
                auto &k = s->last_ms[m.first]; k.last = ms; k.total += ms; k.count++;
            }
        }
        marks.clear(); n = 0;             // the object is reused for the next chunk of a chunked call
    }
};

// the handle's shared side stream, created on first use
bool ensure_side(kzg355_settings *s, Workspace *w) {
    if (!w->side && s->own_side_streams) {
        if (hipStreamCreateWithFlags(&w->side, hipStreamNonBlocking) != hipSuccess) { w->side = nullptr; (void)hipGetLastError(); }
        else w->owns_side = true;
        if (w->side) return true;                                  // (else: fall back to the handle's shared stream)
    }
    if (!w->side) {
        std::lock_guard<std::mutex> lk(s->mu);
        if (!s->side_stream && hipStreamCreateWithFlags(&s->side_stream, hipStreamNonBlocking) != hipSuccess) { s->side_stream = nullptr; (void)hipGetLastError(); }
        w->side = s->side_stream;
    }
    return w->side != nullptr;
}
bool ensure_side2(kzg355_settings *s, Workspace *w) {
    // More calls in flight on the handle than a third of the runtime's hardware queues: the window shifts follow the decoding on the first
    // side stream (enqueue_points_beside's one-side-stream form: +0.4 ms on the call's critical path) -- two streams that share a queue run
    // one after the other whatever their calls are, and a call whose main chain waits behind another call's side work loses milliseconds.
    if (!w->side2 && s->calls_in_flight.load() * 3 > s->hw_queues) return false;
    if (!w->side2 && s->own_side_streams) {
        if (hipStreamCreateWithFlags(&w->side2, hipStreamNonBlocking) != hipSuccess) { w->side2 = nullptr; (void)hipGetLastError(); }
        else w->owns_side2 = true;
        if (w->side2) return true;
    }
    if (!w->side2) {
        std::lock_guard<std::mutex> lk(s->mu);
        if (!s->side2_stream && hipStreamCreateWithFlags(&s->side2_stream, hipStreamNonBlocking) != hipSuccess) { s->side2_stream = nullptr; (void)hipGetLastError(); }
        w->side2 = s->side2_stream;
    }
    return w->side2 != nullptr;
}
inline bool is_small(const kzg355_settings *s) { return s->t.n_fe != N_FE; }
inline size_t blob_bytes_of(const kzg355_settings *s) { return (size_t)32 * s->t.n_fe; }

// which form the batch linear combination takes (KZG355_LINCOMB pins it: 1 window, 2 bucket, 3 pre-shifted)
enum { LC_FORM_WINDOW = 1, LC_FORM_BUCKET = 2, LC_FORM_PRESHIFT = 3 };
int lincomb_form(const kzg355_settings *s, int npg, int groups) {
    const bool bucket_ok = npg >= 8 && npg <= 4096, pre_ok = lincomb_preshift_fits(npg, groups);
    if (s->lincomb_mode == LC_FORM_PRESHIFT) return pre_ok ? LC_FORM_PRESHIFT : LC_FORM_WINDOW;
    if (s->lincomb_mode == LC_FORM_BUCKET) return bucket_ok ? LC_FORM_BUCKET : LC_FORM_WINDOW;
    if (s->lincomb_mode == LC_FORM_WINDOW) return LC_FORM_WINDOW;
    // few batches: shift every point under the hash, finish with ~25 additions; many: least issue work (buckets); in between: per-term ladders
    if (pre_ok) return LC_FORM_PRESHIFT;
    return bucket_ok && groups >= 64 ? LC_FORM_BUCKET : LC_FORM_WINDOW;
}

int status_from_err(int err) {
    if (err == 0) return KZG355_OK;
    return KZG355_BADARGS;   // validate_kzg_g1 / bytes_to_bls_field failures are Error::BadArgs (utils.rs:268, 292, 304)
}

// ---- stage drivers (all asynchronous on w->stream) -------------------------------------------------
// Host-hashed challenges of a small host-buffer call (host_sha256.h): a job on the handle's host threads is writing the digests of
// the call's challenge transcripts to w->h_digests while the caller queues copies and kernels.  finish() joins it (the calling
// thread takes what is left); the destructor does the same on every error path -- the job reads caller memory.
struct HostFront {
    HostPool *pool = nullptr;
    std::shared_ptr<HostPool::Job> job;
    const uint8_t *h_blobs = nullptr;   // the call's blobs in caller memory: copied to the device AFTER the point kernels are queued
    size_t bytes = 0;
    bool running = false;
    // device-resident form (host_hash_from_device): the blobs are in HBM already; they come BACK in chunks behind the point kernels' fork and the
    // hashing job is started there (run_stage1 / msm_op_enqueue), not by the caller
    const uint8_t *d_commitments = nullptr;
    size_t n_blobs = 0;
    bool from_device = false;
    void finish() { if (running) { running = false; pool->finish(job); job.reset(); } }
    ~HostFront() { finish(); }
};

// A small DEVICE-RESIDENT call (one 64-blob batch already in HBM took 5.3 ms against 1.97 ms for the same batch arriving in host
// memory, because only host-buffer calls had the host hash).  The blobs go back over PCIe in up to eight chunks (8 MiB = 0.16 ms for 64 blobs), each
// followed by an event; the hashing job's index k (blobs 2k, 2k + 1, interleaved) waits for the event of its chunk and hashes out of the pinned
// slot -- the first pairs are being hashed while the later chunks are still on the wire.  Queued on w->stream AFTER the point kernels have forked
// off it (their side streams do not wait for the copies).  Leaves hf->running set; the caller joins the job and uploads the digests.
int host_hash_from_device(kzg355_settings *s, Workspace *w, HostFront *hf, const uint8_t *d_blobs) {
    int rc;
    const size_t nb = hf->n_blobs, BB = blob_bytes_of(s);
    if ((rc = w->h_stage.ensure(BB * nb)) || (rc = w->h_stage_cp.ensure(48 * nb)) || (rc = w->h_digests.ensure(32 * nb)) || (rc = w->digests.ensure(32 * nb))) return rc;
    size_t nch = (nb + 1) / 2 < 8 ? (nb + 1) / 2 : 8;
    size_t per = (nb + nch - 1) / nch;
    per += per & 1;                                               // whole pairs per chunk
    nch = (nb + per - 1) / per;
    for (size_t c = 0; c < nch; c++)
        if (!w->ev_d2h[c] && hipEventCreateWithFlags(&w->ev_d2h[c], hipEventDisableTiming) != hipSuccess) { w->ev_d2h[c] = nullptr; (void)hipGetLastError(); return KZG355_DEVICE_ERROR; }
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
    auto job = [=](size_t k) {
        (void)hipEventSynchronize(ev8.e[(2 * k) / per]);          // (chunks hold whole pairs: both blobs of the pair are behind this event)
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
                          int *d_err, bool allow_preshift, int stride = 48 /* bytes between consecutive inputs: 48 packed, 160 inside records */) {
    int rc;
    HIPCHK(hipEventRecord(w->ev_fork, w->stream));
    const bool pre = allow_preshift && d_pts && d_p && lincomb_form(s, npg, n_total / npg) == LC_FORM_PRESHIFT;
    bool shift_on_side2 = false;
    if (pre) {
        if ((rc = w->shifts.ensure(lincomb_preshift_bytes(npg, n_total / npg)))) return rc;
        if (s->calls_in_flight.load() * 3 <= s->hw_queues && ensure_side2(s, w)) {
            HIPCHK(hipStreamWaitEvent(w->side2, w->ev_fork, 0));
            w->shift_pending = true;
            tm.begin("lincomb_shift", w->side2); launch_lincomb_preshift_bytes(d_c, d_p, stride, npg, n_total / npg, w->shifts.as<G1Jac>(), w->side2); tm.end(w->side2);
            HIPCHK(hipEventRecord(w->ev_shift, w->side2));
            shift_on_side2 = true;
        }
    }
    HIPCHK(hipStreamWaitEvent(w->side, w->ev_fork, 0));
    w->side_pending = true;                                       // (from here on a failing call has to drain the side streams: quiesce())
    if (pre) {
        tm.begin("decompress_points", w->side); launch_decompress_points(d_c, d_p, n_total, npg, d_pts, d_err, w->side, stride); tm.end(w->side);
        // (no second side stream: the shifts follow the decoding on this one, and ev_pts -- what join_points() waits for -- covers both)
        if (!shift_on_side2) { tm.begin("lincomb_shift", w->side); launch_lincomb_preshift(d_pts, npg, n_total / npg, w->shifts.as<G1Jac>(), w->side); tm.end(w->side); }
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
int join_points(Workspace *w, bool shifts_too = true) {       // the decoded points and (unless the caller joins them later) their window shifts
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
               int npg, uint8_t *d_records, G1Affine *d_pts, int *d_err, bool allow_preshift = true, HostFront *hf = nullptr,
               const std::function<int()> *after_challenge = nullptr) {
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
        else HIPCHK(hipMemcpyAsync(const_cast<uint8_t *>(d_blobs), hf->h_blobs, hf->bytes, hipMemcpyHostToDevice, w->stream));
        hf->finish();
        HIPCHK(hipMemcpyAsync(w->digests.p, w->h_digests.p, 32 * (size_t)n_total, hipMemcpyHostToDevice, w->stream));
        tm.begin("challenge_from_digest"); launch_challenges_from_digests(w->digests.as<uint8_t>(), d_c, d_p, n_total, w->z.as<Fr>(), w->zpow.as<Fr>(), d_records, w->stream); tm.end();
    } else {
        tm.begin("challenge"); launch_challenges(d_blobs, d_c, d_p, n_total, w->z.as<Fr>(), w->zpow.as<Fr>(), d_records, w->stream, s->challenge_form ? s->challenge_form : n_total <= s->challenge_two_wave_upto ? 2 : 1); tm.end();
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
               int *d_err, int *d_ok, bool lone_call = false) {
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
            uint32_t *dst = w->h_rdig.as<uint32_t>() + 8 * (size_t)g;            // the digest as an integer: 8 little-endian 32-bit words (what k_rhash_lanes leaves)
            for (int k = 0; k < 8; k++) dst[k] = ((uint32_t)dg[28 - 4 * k] << 24) | ((uint32_t)dg[29 - 4 * k] << 16) | ((uint32_t)dg[30 - 4 * k] << 8) | dg[31 - 4 * k];
        }
        HIPCHK(hipMemcpyAsync(w->scal_c.p, w->h_rdig.p, 32 * (size_t)groups, hipMemcpyHostToDevice, w->stream));
    }
    tm.begin("rpowers"); launch_rpowers(d_records, npg, groups, check_zy, w->scal_a.as<uint32_t>(), w->scal_b.as<uint32_t>(), w->scal_c.as<uint32_t>(), d_err, w->stream, s->t.n_fe, s->rhash_lanes_from, host_rhash ? 1 : 0); tm.end();
    const int form = lincomb_form(s, npg, groups);
    const bool buckets = form == LC_FORM_BUCKET;
    if ((rc = w->lc_partials.ensure(form == LC_FORM_WINDOW ? lincomb_partials_bytes(npg, groups) : lincomb_buckets_scratch_bytes(npg, groups)))) return rc;
    if ((rc = join_points(w, form != LC_FORM_PRESHIFT))) return rc;       // the decoded points (and their shifts) are needed from here on
    if (form == LC_FORM_PRESHIFT) {
        if (!shift_ready) {                                       // entry points without a stage 1 (single proofs, gathered records)
            if ((rc = w->shifts.ensure(lincomb_preshift_bytes(npg, groups)))) return rc;
            tm.begin("lincomb_shift"); launch_lincomb_preshift(d_pts, npg, groups, w->shifts.as<G1Jac>(), w->stream); tm.end();
        }
        // the digits need the points and the r powers only: they are made while the shift chain (the longest piece in front of the sums: 0.52 ms
        // for a 64-blob call, against 0.47 ms until the r powers are there) is still walking
        tm.begin("lincomb");
        launch_lincomb_preshifted(d_pts, w->shifts.as<G1Jac>(), w->scal_a.as<uint32_t>(), w->scal_b.as<uint32_t>(), w->scal_c.as<uint32_t>(), npg, groups, w->lc_partials.p,
                                  w->pair_pts.as<PairPt>(), w->stream, 1);
        if ((rc = join_shifts(w))) return rc;
        launch_lincomb_preshifted(d_pts, w->shifts.as<G1Jac>(), w->scal_a.as<uint32_t>(), w->scal_b.as<uint32_t>(), w->scal_c.as<uint32_t>(), npg, groups, w->lc_partials.p,
                                  w->pair_pts.as<PairPt>(), w->stream, 2);
    } else if (buckets) {
        static const char *names[3] = {"lincomb_prep", "lincomb", "lincomb_horner"};       // "lincomb" = the bucket kernel itself
        for (int stage = 1; stage <= 3; stage++) {
            if (stage > 1) tm.end();
            tm.begin(names[stage - 1]);
            launch_lincomb_buckets(d_pts, w->scal_a.as<uint32_t>(), w->scal_b.as<uint32_t>(), w->scal_c.as<uint32_t>(), npg, groups, w->lc_partials.p, w->pair_pts.as<PairPt>(), w->stream, stage, s->lc_chain_from);
        }
    } else {
        tm.begin("lincomb");
        launch_lincomb(d_pts, w->scal_a.as<uint32_t>(), w->scal_b.as<uint32_t>(), w->scal_c.as<uint32_t>(), npg, groups, w->lc_partials.as<G1Jac>(), w->pair_pts.as<PairPt>(), w->stream);
    }
    tm.end();
    tm.begin("pairing");
    if (s->lane_pairing) launch_pairing_lane(w->pair_pts.as<PairPt>(), s->t, groups, d_ok, w->stream);
    else {
        Fp *f12 = nullptr;
        // f between the two kernels of a check: many batches (hard part twelve lanes per check) and few (Miller loops in segments on several waves)
        if (((s->pairing_hard12_from > 0 && groups >= s->pairing_hard12_from) || groups <= s->pairing_two_wave_upto) && w->pair_f.ensure(pairing_f12_bytes(groups)) == KZG355_OK)
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
                          size_t res_cap = 0, HostFront *hf = nullptr, const std::function<int()> *after_challenge = nullptr) {
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
int verify_enqueue_stage2(kzg355_settings *s, Workspace *w, Timed &tm, int npg, int G, size_t res_off = 0, bool lone_call = false) {
    int rc;
    if ((rc = run_stage2(s, w, tm, w->records.as<uint8_t>(), npg, G, 0, w->pts.as<G1Affine>(), w->err.as<int>(), w->ok.as<int>(), lone_call))) return rc;
    if ((rc = join_side(w))) return rc;                           // the subgroup verdicts, before the error words go back
    HIPCHK(hipMemcpyAsync(w->h_ok.as<int>() + res_off, w->ok.p, sizeof(int) * (size_t)G, hipMemcpyDeviceToHost, w->stream));
    HIPCHK(hipMemcpyAsync(w->h_err.as<int>() + res_off, w->err.p, sizeof(int) * (size_t)G, hipMemcpyDeviceToHost, w->stream));
    return KZG355_OK;
}
int verify_enqueue(kzg355_settings *s, Workspace *w, Timed &tm, const uint8_t *d_blobs, const uint8_t *d_c, const uint8_t *d_p, int npg, int G,
                   size_t res_off = 0, size_t res_cap = 0, HostFront *hf = nullptr, bool lone_call = false) {
    int rc = verify_enqueue_stage1(s, w, tm, d_blobs, d_c, d_p, npg, G, res_cap, hf);
    if (rc) return rc;
    return verify_enqueue_stage2(s, w, tm, npg, G, res_off, lone_call);
}
// ... and wait for it: verdicts / statuses of its G batches.  Returns the first non-OK status.
int verify_collect(Workspace *w, Timed &tm, bool *ok, int *status, int G, size_t res_off = 0) {
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

thread_local bool tl_msm_inner = false;      // this thread is inside the build (or the load-time self-test): its MSM calls take the handle as it is
thread_local const kzg355_settings::WidePub *tl_wide_candidate = nullptr;   // ... and, while the builder checks it, the table that is not published yet
thread_local bool tl_force_bucket = false;   // the load-time self-test's second opinion: the bucket form although a table is published
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
    if (!d_blobs || !d_c || !d_p || ((uintptr_t)d_blobs & 15) || ((uintptr_t)d_c & 3) || ((uintptr_t)d_p & 3)) return KZG355_BADARGS;   // 16-byte loads of the blobs
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
    for (size_t k = 0; k < parts; k++) g0[k + 1] = g0[k] + groups / parts + (k < groups % parts ? 1 : 0);      // larger parts first: a later set never outgrows the scratch
    const size_t cap = (groups / parts + 1) * ((parts + lanes - 1) / lanes);
    for (size_t k = 0; k < parts; k++) {
        const size_t l = k % lanes, cnt = g0[k + 1] - g0[k];
        res_off[k] = lane_fill[l]; lane_fill[l] += cnt;
        int rc = verify_enqueue(gs[l]->s, gs[l]->w, tms[l], d_blobs + BB * npg * g0[k], d_c + 48 * npg * g0[k], d_p + 48 * npg * g0[k], (int)npg, (int)cnt, res_off[k], cap);
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

int msm_op_many_device_impl(uint8_t *out, int *status, const uint8_t *d_blobs, const uint8_t *d_c, size_t n, const kzg355_settings *cs);
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
                            if (i & 1) { for (int k = 0; k < 32; k++) e[k] = (k & 1) ? 0x00 : 0x80; e[0] = 0x00; e[30] ^= (uint8_t)(i >> 8); e[31] ^= (uint8_t)i; }
                            else {      // r - 1 - i: big-endian subtraction of 1 + i with borrow
                                memcpy(e, R_BE, 32);
                                uint32_t sub = 1 + (uint32_t)i;
                                for (int k = 31; k >= 0 && sub; k--) { const uint32_t d = sub & 0xff; sub >>= 8; if (e[k] >= d) e[k] = (uint8_t)(e[k] - d); else { e[k] = (uint8_t)(e[k] + 256 - d); sub += 1; } }
                            }
                        }
                        if (hipMemcpy(blobs.p, ones.data(), BB, hipMemcpyHostToDevice) != hipSuccess) rc = KZG355_DEVICE_ERROR;
                        if (hipMemcpy(blobs.as<uint8_t>() + 2 * BB, ext.data(), BB, hipMemcpyHostToDevice) != hipSuccess) rc = KZG355_DEVICE_ERROR;
                        launch_fr_to_bytes(s->t.roots, s->t.n_fe, blobs.as<uint8_t>() + BB, nullptr);
                        if (hipDeviceSynchronize() != hipSuccess) rc = KZG355_DEVICE_ERROR;
                    }
                    tl_msm_inner = true;                      // (the commitments below must not come back here)
                    if (rc == KZG355_OK) rc = msm_op_many_device_impl(c_bucket, st, blobs.as<uint8_t>(), nullptr, 3, s);        // nothing published yet: bucket form
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
            fprintf(stderr, "kzg355: the wide-window MSM table could not be %s; commitments / proofs take the 8-bit bucket form (about 3x slower, same results)\n",
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
    if (launch_quotient(d_blobs, w->z.as<Fr>(), s->t, n, w->y.as<Fr>(), w->q.as<uint8_t>(), w->qprep.p, w->err.as<int>(), w->stream, s->quotient_form)) return KZG355_DEVICE_ERROR;
    tm.end();
    return msm_to_host(s, w, tm, n, w->q.as<uint8_t>(), nullptr);
}

// n commitments (d_c == null) or n blob proofs against the commitments d_c: enqueue on w->stream ...
int msm_op_enqueue(kzg355_settings *s, Workspace *w, Timed &tm, const uint8_t *d_blobs, const uint8_t *d_c, size_t n, HostFront *hf = nullptr) {
    int rc;
    if ((rc = w->err.ensure(sizeof(int) * n))) return rc;
    if ((rc = w->h_err.ensure(sizeof(int) * n))) return rc;
    if (d_c && (rc = w->z.ensure(sizeof(Fr) * n))) return rc;
    w->in_flight = true;
    HIPCHK(hipMemsetAsync(w->err.p, 0, sizeof(int) * n, w->stream));
    if (is_small(s)) {
        if ((rc = w->out48.ensure(48 * n))) return rc;
        if ((rc = w->h_out.ensure(48 * n))) return rc;
        if (!d_c) { tm.begin("small_commit"); launch_small_commit(d_blobs, (int)n, s->t, w->out48.as<uint8_t>(), w->err.as<int>(), w->stream); tm.end(); }
        else {
            tm.begin("validate_points"); launch_validate_points(d_c, nullptr, (int)n, 1, nullptr, w->err.as<int>(), w->stream); tm.end();
            tm.begin("small_proof"); launch_small_proof(d_blobs, d_c, nullptr, (int)n, s->t, w->out48.as<uint8_t>(), nullptr, w->err.as<int>(), w->stream); tm.end();
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
            const bool ladder_beside = 2 * n <= 1024 && s->calls_in_flight.load() * 3 <= s->hw_queues && w->shifts.ensure(sizeof(G1Jac) * n) == KZG355_OK && ensure_side2(s, w);
            if (ladder_beside) {
                HIPCHK(hipStreamWaitEvent(w->side2, w->ev_fork, 0));
                w->shift_pending = true;                         // (side2 has work: quiesce() drains it; join_side() waits for ev_shift)
                tm.begin("validate_points", w->side2); launch_subgroup_ladder_from_x(d_c, 48, (int)n, w->shifts.as<G1Jac>(), w->side2); tm.end(w->side2);
                HIPCHK(hipEventRecord(w->ev_shift, w->side2));
            }
            tm.begin("decompress_points", w->side); launch_decompress_points(d_c, nullptr, (int)n, 1, w->pts.as<G1Affine>(), w->err.as<int>(), w->side); tm.end(w->side);
            if (ladder_beside) {
                HIPCHK(hipStreamWaitEvent(w->side, w->ev_shift, 0));
                launch_subgroup_finish(w->pts.as<G1Affine>(), w->shifts.as<G1Jac>(), (int)n, w->err.as<int>(), w->side);
            } else { tm.begin("validate_points", w->side); launch_subgroup_points(w->pts.as<G1Affine>(), (int)n, 1, w->err.as<int>(), w->side, 1); tm.end(w->side); }
            HIPCHK(hipEventRecord(w->ev_join, w->side));
        } else { tm.begin("validate_points"); launch_validate_points(d_c, nullptr, (int)n, 1, nullptr, w->err.as<int>(), w->stream); tm.end(); }
        if (hf) {                                                 // challenges hashed on the host (see run_stage1)
            if (hf->from_device) { if ((rc = host_hash_from_device(s, w, hf, d_blobs))) return rc; }
            else HIPCHK(hipMemcpyAsync(const_cast<uint8_t *>(d_blobs), hf->h_blobs, hf->bytes, hipMemcpyHostToDevice, w->stream));
            hf->finish();
            HIPCHK(hipMemcpyAsync(w->digests.p, w->h_digests.p, 32 * n, hipMemcpyHostToDevice, w->stream));
            tm.begin("challenge_from_digest"); launch_challenges_from_digests(w->digests.as<uint8_t>(), d_c, nullptr, (int)n, w->z.as<Fr>(), nullptr, nullptr, w->stream); tm.end();
        } else { tm.begin("challenge"); launch_challenges(d_blobs, d_c, nullptr, (int)n, w->z.as<Fr>(), nullptr, nullptr, w->stream, s->challenge_form ? s->challenge_form : (int)n <= s->challenge_two_wave_upto ? 2 : 1); tm.end(); }
        if ((rc = prove_common(s, w, tm, d_blobs, (int)n))) return rc;
        if ((rc = join_side(w))) return rc;
    }
    HIPCHK(hipMemcpyAsync(w->h_err.p, w->err.p, sizeof(int) * n, hipMemcpyDeviceToHost, w->stream));
    return KZG355_OK;
}
// ... and wait for it: 48-byte outputs / statuses of its n blobs.  Returns the first non-OK status.
int msm_op_collect(Workspace *w, Timed &tm, uint8_t *out, int *status, size_t n) {
    HIPCHK(hipStreamSynchronize(w->stream));
    w->in_flight = false;
    tm.collect();
    int first = KZG355_OK;
    for (size_t i = 0; i < n; i++) {
        int st = status_from_err(w->h_err.as<int>()[i]);
        if (status) status[i] = st;
        if (st == KZG355_OK) memcpy(out + 48 * i, w->h_out.as<uint8_t>() + 48 * i, 48);
        else if (first == KZG355_OK) first = st;
    }
    return first;
}

int msm_op_many_device_impl(uint8_t *out, int *status, const uint8_t *d_blobs, const uint8_t *d_c, size_t n, const kzg355_settings *cs) {
    if (!cs || !out) return KZG355_BADARGS;
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
    int rc = msm_op_enqueue(g.s, g.w, tm, d_blobs, d_c, n, via_host ? &hf : nullptr);
    if (rc) return rc;
    return msm_op_collect(g.w, tm, out, status, n);
}

// The host-buffer pipeline shared by the three *_many entry points.  `units` independent units of work (batches of npg blobs
// for verify, single blobs for commit / proof) are cut into chunks of <= chunk_bytes of blobs; chunk k goes through workspace
// k mod W (W = chunks_in_flight): H2D on its stream, kernels, results.  Two ways to move the bytes:
//   direct (default)  hipMemcpyAsync straight from the caller's pageable memory: the runtime locks the pages and DMAs from them
//                     (no CPU copy; 56.5 GB/s = the PCIe 5 x16 link on the MI355X hosts measured).  The call blocks the host
//                     thread for the duration of the copy, which is exactly the pacing wanted: the next chunk's copy is issued the
//                     moment the link is free, while the kernels of the previous chunks run on their own streams.
//   ring (KZG355_STAGING=ring)  parallel host copy into the workspace's pinned slot, then an asynchronous H2D from there.
//                     Measured slower here (300 k against 395 k blobs/s on an 8 GiB call): the CPU copy and the DMA compete
//                     for host memory bandwidth; kept for hosts where page locking is expensive.
// Results are collected in chunk order; a failure waits for everything in flight before the workspaces go back to the pool.
struct HostCall {
    int kind;                        // 0 verify, 1 commit, 2 blob proof
    const uint8_t *blobs, *commitments, *proofs;
    size_t npg;                      // blobs per unit
    bool *ok; uint8_t *out48; int *status;
    uint8_t *records_out = nullptr;  // verify, single-chunk calls only (kzg355_debug_verify_host_records): the stage-1 records, copied back after the chunk
};
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
                              : msm_op_collect(w, tms[slot], hc.out48 + 48 * u0, hc.status ? hc.status + u0 : nullptr, cnt);
        if (hc.records_out && hc.kind == 0 && hipMemcpy(hc.records_out + (size_t)RECORD_BYTES * hc.npg * u0, w->records.p, (size_t)RECORD_BYTES * hc.npg * cnt, hipMemcpyDeviceToHost) != hipSuccess) return KZG355_DEVICE_ERROR;
        if (rc == KZG355_NO_DEVICE || rc == KZG355_NO_MEMORY || rc == KZG355_DEVICE_ERROR) return rc;
        if (rc != KZG355_OK && first == KZG355_OK) first = rc;
        return KZG355_OK;
    };
    const bool dbg = getenv("KZG355_DEBUG_PIPE") != nullptr;
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
            auto job = [=](size_t k) { kzg_host::challenge_digests(dig + 64 * k, hb + BB * 2 * k, BB, hcm + 96 * k, nb - 2 * k < 2 ? nb - 2 * k : 2, n_fe, impl); };
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
        t_stage += now() - t0; t0 = now();
        HostFront *hfp = hf.running ? &hf : nullptr;
        if (hc.kind == 0) rc = verify_enqueue(s, w, tms[slot], w->blobs.as<uint8_t>(), w->commitments.as<uint8_t>(), w->proofs.as<uint8_t>(), (int)hc.npg, (int)cnt, 0, 0, hfp, nchunks == 1);
        else rc = msm_op_enqueue(s, w, tms[slot], w->blobs.as<uint8_t>(), hc.kind == 2 ? w->commitments.as<uint8_t>() : nullptr, cnt, hfp);
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
    if (dbg) fprintf(stderr, "kzg355 pipe: %zu chunks, W %d: alloc %.1f ms, host copy + H2D enqueue %.1f ms, kernel enqueue %.1f ms, waits in loop %.1f ms, drain %.1f ms, total %.1f ms\n",
                     nchunks, W, t_alloc, t_stage, t_enq, t_wait, now() - t_loop, now() - t_begin);
    return first;
}

int hexval(int ch) { return ch >= '0' && ch <= '9' ? ch - '0' : ch >= 'a' && ch <= 'f' ? ch - 'a' + 10 : ch >= 'A' && ch <= 'F' ? ch - 'A' + 10 : -1; }

}  // namespace

// =====================================================================================================
extern "C" {
#pragma GCC visibility push(default)

const char *kzg355_version(void) { return "kzg355 0.1 (gfx950, 29-bit-limb Montgomery, fixed-base Pippenger, precomputed-line pairing)"; }

static int device_self_test(kzg355_settings *s);
static std::vector<kzg355_settings *> replicas_of(kzg355_settings *s);

// ---- options ---------------------------------------------------------------------------------------------------------------------
// Everything a deployment may want to pin is a field of kzg355_options (include/kzg355.h) passed to kzg355_load_trusted_setup_ex; the
// dispatch thresholds default to multiples of the device's CU count (0 = auto).  The KZG355_* environment variables are what the tests
// and experiments use to override them: kzg355_options_from_env folds them into a struct, and the plain load functions use that.
void kzg355_options_default(kzg355_options *o) {
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->struct_size = sizeof *o;
    o->device = -1;
    o->self_test = 1;
}
void kzg355_options_from_env(kzg355_options *o) {
    if (!o) return;
    kzg355_options_default(o);
    auto num = [](const char *name, long lo, long hi, int *dst) { if (const char *e = getenv(name)) { const long v = atol(e); if (v >= lo && v <= hi) *dst = (int)v; } };
    num("KZG355_DEVICE", 0, 1023, &o->device);
    if (const char *e = getenv("KZG355_MSM")) { if (strcmp(e, "bucket") == 0) o->msm_bits = 8; else if (strcmp(e, "wide") == 0) o->msm_require_wide = 1; }
    if (o->msm_bits != 8) num("KZG355_MSM_BITS", 10, 16, &o->msm_bits);
    if (const char *e = getenv("KZG355_MSM_GLV")) o->msm_glv = strcmp(e, "off") == 0 ? -1 : 0;
    num("KZG355_MSM_EAGER", 0, 1, &o->msm_eager);
    num("KZG355_SELFTEST", 0, 1, &o->self_test);
    num("KZG355_COPY_THREADS", 1, 64, &o->host_threads);
    num("KZG355_HOST_THREADS", 1, 64, &o->host_threads);
    if (const char *e = getenv("KZG355_HOST_HASH")) o->host_hash = strcmp(e, "on") == 0 ? 1 : strcmp(e, "off") == 0 ? -1 : 0;
    num("KZG355_HOST_HASH_MAX", 1, 1 << 24, &o->host_hash_max_blobs);
    num("KZG355_HOST_HASH_DEVICE_MAX", 0, 1 << 24, &o->host_hash_device_max_blobs);
    if (getenv("KZG355_HOST_HASH_DEVICE_MAX") && o->host_hash_device_max_blobs == 0) o->host_hash_device_max_blobs = -1;      // "0": never
    if (const char *e = getenv("KZG355_HOST_SHA")) o->host_sha = strcmp(e, "portable") == 0 ? 1 : strcmp(e, "shani") == 0 ? 2 : 0;
    if (const char *e = getenv("KZG355_HOST_RHASH")) o->host_rhash = strcmp(e, "off") == 0 ? -1 : 0;
    num("KZG355_HOST_RHASH_MAX", 1, 1 << 20, &o->host_rhash_max_records);
    num("KZG355_CHUNK_MB", 1, 16384, &o->chunk_mb);
    if (const char *e = getenv("KZG355_STAGING")) o->staging_ring = strcmp(e, "ring") == 0;
    num("KZG355_CHUNKS_IN_FLIGHT", 1, 8, &o->chunks_in_flight);
    if (const char *e = getenv("KZG355_PAIRING")) o->pairing_lane = strcmp(e, "lane") == 0;
    num("KZG355_PAIRING_2W_UPTO", 0, 1 << 24, &o->pairing_two_wave_upto);
    if (getenv("KZG355_PAIRING_2W_UPTO") && o->pairing_two_wave_upto == 0) o->pairing_two_wave_upto = -1;      // "0": never
    if (const char *e = getenv("KZG355_SPLIT")) {
        int a = 1, b = 2;
        const int got = sscanf(e, "%d,%d", &a, &b);
        if (got >= 1 && a >= 1 && a <= 64) o->split_parts = a;
        if (got >= 2 && b >= 1 && b <= 8) o->split_streams = b;
    }
    num("KZG355_PAIRING_HARD12_FROM", 0, 1 << 24, &o->pairing_hard12_from);
    if (getenv("KZG355_PAIRING_HARD12_FROM") && o->pairing_hard12_from == 0) o->pairing_hard12_from = -1;      // "0": never
    num("KZG355_LC_CHAIN_FROM", 1, 1 << 24, &o->lc_chain_from);
    num("KZG355_RHASH_LANES_FROM", 1, 1 << 24, &o->rhash_lanes_from);
    if (const char *e = getenv("KZG355_CHALLENGE")) o->challenge_form = strcmp(e, "1w") == 0 ? 1 : strcmp(e, "2w") == 0 ? 2 : 0;
    if (const char *e = getenv("KZG355_LINCOMB")) o->lincomb_form = strcmp(e, "bucket") == 0 ? 2 : strcmp(e, "window") == 0 ? 1 : strcmp(e, "preshift") == 0 ? 3 : 0;
    if (const char *e = getenv("KZG355_EXCHANGE")) o->exchange = strcmp(e, "peer") == 0 ? 1 : strcmp(e, "rccl") == 0 ? 2 : 0;
    num("KZG355_VERIFY_ONLY", 0, 1, &o->verify_only);
    if (const char *e = getenv("KZG355_SUBMIT")) o->submit_sets = strcmp(e, "sets") == 0 ? 1 : strcmp(e, "pipeline") == 0 ? 2 : 0;
}
// the caller's struct may be older (smaller) than this library's: fields beyond its struct_size keep their defaults
static kzg355_options options_of(const kzg355_options *opt) {
    kzg355_options o;
    kzg355_options_default(&o);
    if (opt && opt->struct_size >= sizeof(size_t)) {
        const size_t n = opt->struct_size < sizeof o ? opt->struct_size : sizeof o;
        memcpy(&o, opt, n);
        o.struct_size = sizeof o;
    }
    return o;
}

static int load_on_device(const uint8_t *g1_bytes, size_t n1, const uint8_t *g2_bytes, size_t n2, int dev_or_minus1, const kzg355_options &opt, kzg355_settings **out) {
    if (!out || !g1_bytes || !g2_bytes) return KZG355_BADARGS;
    // FIELD_ELEMENTS_PER_BLOB is a compile-time constant of the reference (consts.rs:13: 4096; its README's minimal preset: 4); here
    // it is a property of the handle, taken from the number of G1 points: 4096, or a power of two in [4, 64] for the small path
    const bool small = n1 >= (size_t)SMALL_N_MIN && n1 <= (size_t)SMALL_N_MAX && (n1 & (n1 - 1)) == 0;
    if ((n1 != (size_t)N_FE && !small) || n2 != (size_t)N_G2) return KZG355_INVALID_TRUSTED_SETUP;   // kzg.rs:49-62 (843: BadArgs)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return KZG355_NO_DEVICE;
    int dev = 0;
    if (dev_or_minus1 >= 0) dev = dev_or_minus1;
    else if (opt.device >= 0) dev = opt.device;
    else if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev < 0 || dev >= ndev) return KZG355_NO_DEVICE;
    DeviceScope scope;
    if (!scope.enter(dev)) return KZG355_NO_DEVICE;
    kzg355_settings *s = new kzg355_settings();
    s->device = dev;
    DevBuf g1b, g2b, err;
    int rc = KZG355_OK;
    auto fail = [&](int code) { g1b.release(); g2b.release(); err.release(); kzg355_free_trusted_setup(s); return code; };
    s->t.n_fe = (int)n1;
    if ((rc = s->roots.ensure(sizeof(Fr) * n1))) return fail(rc);
    if (!small && (rc = s->eval_tab.ensure(sizeof(Fr) * EVAL_TAB_ENTRIES))) return fail(rc);
    if ((rc = s->msm_table.ensure(sizeof(G1Affine) * n1 * (small ? 1 : MSM_WINDOWS)))) return fail(rc);
    if ((rc = s->lines.ensure(sizeof(LineCoeff) * 3 * N_LINES))) return fail(rc);
    if ((rc = s->lines_inf.ensure(sizeof(int) * 3))) return fail(rc);
    if ((rc = s->g1_first2.ensure(sizeof(G1Affine) * 2))) return fail(rc);
    if ((rc = s->lines_w.ensure(sizeof(LineW) * 3 * N_LINES))) return fail(rc);
    if ((rc = s->frob.ensure(sizeof(FrobTables)))) return fail(rc);
    if ((rc = g1b.ensure(48 * n1))) return fail(rc);
    if ((rc = g2b.ensure(96 * n2))) return fail(rc);
    if ((rc = err.ensure(sizeof(int)))) return fail(rc);
    s->t.roots = s->roots.as<Fr>();
    s->t.eval_tab = s->eval_tab.as<Fr>();
    s->t.msm_table = s->msm_table.as<G1Affine>();
    s->t.lines = s->lines.as<LineCoeff>();
    s->t.lines_inf = s->lines_inf.as<int>();
    s->t.g1_first2 = s->g1_first2.as<G1Affine>();
    s->t.lines_w = s->lines_w.as<LineW>();
    s->t.frob = s->frob.as<FrobTables>();
    {
        static const uint32_t A1[12][NFP] = FROBW_A1_INIT, B1[12][NFP] = FROBW_B1_INIT, A2[12][NFP] = FROBW_A2_INIT;
        FrobTables ft;
        for (int k = 0; k < 12; k++) for (int i = 0; i < NFP; i++) { ft.a1[k].l[i] = A1[k][i]; ft.b1[k].l[i] = B1[k][i]; ft.a2[k].l[i] = A2[k][i]; }
        if (hipMemcpy(s->frob.p, &ft, sizeof ft, hipMemcpyHostToDevice) != hipSuccess) return fail(KZG355_DEVICE_ERROR);
    }
    {
        static CoopInsn prog[COOP_PROGRAM_MAX];
        int hard_start = 0;
        const int n = build_pairing_program(prog, &hard_start);
        s->t.pairing_hard_start = hard_start;
        if (n > COOP_PROGRAM_MAX || s->prog.ensure(sizeof(CoopInsn) * n)) return fail(KZG355_INTERNAL);
        if (hipMemcpy(s->prog.p, prog, sizeof(CoopInsn) * n, hipMemcpyHostToDevice) != hipSuccess) return fail(KZG355_DEVICE_ERROR);
        s->t.pairing_prog = s->prog.as<CoopInsn>();
        s->t.pairing_prog_len = n;
        static CoopScheds sc;
        if (!build_coop_schedules(sc)) return fail(KZG355_INTERNAL);
        if (s->scheds.ensure(sizeof sc)) return fail(KZG355_NO_MEMORY);
        if (hipMemcpy(s->scheds.p, &sc, sizeof sc, hipMemcpyHostToDevice) != hipSuccess) return fail(KZG355_DEVICE_ERROR);
        s->t.coop_scheds = s->scheds.as<CoopScheds>();
    }
    {
        cpu_set_t cpus; CPU_ZERO(&cpus);
        int workers = sched_getaffinity(0, sizeof cpus, &cpus) == 0 ? CPU_COUNT(&cpus) / 2 : 1;     // cores this process may run on
        if (workers > 16) workers = 16;
        if (opt.host_threads > 0) workers = opt.host_threads;
        if (workers > 64) workers = 64;
        if (workers < 1) workers = 1;
        s->host_pool = new HostPool(workers - 1);                            // the calling thread is one of the workers
        s->host_hash = opt.host_hash;
        if (opt.host_hash_max_blobs > 0) s->host_hash_max = opt.host_hash_max_blobs;
        if (opt.host_hash_device_max_blobs != 0) s->host_hash_device_max = opt.host_hash_device_max_blobs > 0 ? opt.host_hash_device_max_blobs : 0;
        s->sha_impl = opt.host_sha;
        s->host_rhash = s->host_rhash_loaded = opt.host_rhash;
        if (opt.host_rhash_max_records > 0) s->host_rhash_max_records = opt.host_rhash_max_records;
        if (opt.chunk_mb > 0) s->chunk_bytes = (size_t)opt.chunk_mb << 20;
        s->pinned_ring = opt.staging_ring != 0;
        if (opt.chunks_in_flight >= 1 && opt.chunks_in_flight <= 8) s->chunks_in_flight = opt.chunks_in_flight;
    }
    {   // dispatch thresholds: measured on the 256-CU MI355X and kept as multiples of the CU count of the device at hand
        hipDeviceProp_t prop;
        int cus = 256;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        s->cu_count = cus;
        s->lc_chain_from = opt.lc_chain_from > 0 ? opt.lc_chain_from : 4 * cus;              // one Horner chain per class from 1024 batches on
        s->rhash_lanes_from = opt.rhash_lanes_from > 0 ? opt.rhash_lanes_from : 4 * cus;     // transcript hash with a lane per batch from one wave per SIMD on (1024)
        s->beside_max_blobs = opt.beside_max_blobs > 0 ? opt.beside_max_blobs : 64 * cus;    // point kernels beside the hash chain up to 16384 blobs
        s->pairing_two_wave_upto = opt.pairing_two_wave_upto < 0 ? 0 : opt.pairing_two_wave_upto > 0 ? opt.pairing_two_wave_upto : cus;     // two waves per pairing up to 256 batches
        if (const char *e = getenv("KZG355_QUOTIENT_FORM")) { const int v = atoi(e); if (v == 2 || v == 4 || v == 6) s->quotient_form = v; }
        if (const char *e = getenv("KZG355_MILLER_SEGMENTS")) { const int v = atoi(e); if (v >= 1 && v <= MILLER_SPLIT_MAX) s->miller_segments = v; }      // (tuning knob, not an option)
        s->pairing_hard12_from = opt.pairing_hard12_from < 0 ? 0 : opt.pairing_hard12_from > 0 ? opt.pairing_hard12_from : 16 * cus;   // hard part twelve lanes per check from 4096 batches on
        s->challenge_two_wave_upto = 2 * cus * 64;                                           // two-wave hash while every wave has a SIMD to itself (512 workgroups of 64 blobs)
    }
    s->lane_pairing = opt.pairing_lane != 0;
    if (opt.split_parts >= 1 && opt.split_parts <= 64) s->split_parts = opt.split_parts;
    if (opt.split_streams >= 1 && opt.split_streams <= 8) s->split_streams = opt.split_streams;
    s->challenge_form = opt.challenge_form;
    s->lincomb_mode = opt.lincomb_form;
    if (const char *e = getenv("KZG355_SIDE")) s->own_side_streams = strcmp(e, "shared") != 0;
    if (const char *e = getenv("GPU_MAX_HW_QUEUES")) { const int q = atoi(e); if (q >= 1 && q <= 128) s->hw_queues = q; }
    s->submit_mode = opt.submit_sets >= 0 && opt.submit_sets <= 2 ? opt.submit_sets : 0;
    if (hipMemcpy(g1b.p, g1_bytes, 48 * n1, hipMemcpyHostToDevice) != hipSuccess) return fail(KZG355_DEVICE_ERROR);
    if (hipMemcpy(g2b.p, g2_bytes, 96 * n2, hipMemcpyHostToDevice) != hipSuccess) return fail(KZG355_DEVICE_ERROR);
    if (hipMemset(err.p, 0, sizeof(int)) != hipSuccess) return fail(KZG355_DEVICE_ERROR);
    if (small) launch_setup_small(g1b.as<uint8_t>(), (int)n1, s->t, err.as<int>(), nullptr);
    if (launch_setup(g1b.as<uint8_t>(), g2b.as<uint8_t>(), s->t, err.as<int>(), nullptr)) return fail(KZG355_DEVICE_ERROR);
    launch_lines_to_w(s->t, nullptr);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return fail(KZG355_DEVICE_ERROR);
    int herr = 0;
    if (hipMemcpy(&herr, err.p, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return fail(KZG355_DEVICE_ERROR);
    if (herr) return fail(KZG355_BADARGS);                       // kzg.rs:863, 878, 823-826
    g1b.release(); g2b.release(); err.release();
    // the wide-window MSM table: built on first use (ensure_wide_table) unless asked for at load
    s->msm_bits_wanted = (small || opt.msm_bits == 8 || opt.verify_only) ? 8 : (opt.msm_bits >= 10 && opt.msm_bits <= 16 ? opt.msm_bits : 0);
    s->msm_glv = opt.msm_glv < 0 ? 0 : 1;
    if (!s->msm_glv && s->msm_bits_wanted == 16) s->msm_bits_wanted = 15;            // (the 256-bit form stops at 15-bit windows: 155 GB)
    s->msm_required = opt.msm_require_wide != 0 && s->msm_bits_wanted != 8;
    if (opt.self_test) {   // known-answer self-test of the freshly built handle (MSM through the bucket form: the wide table has its own check when it is built)
        tl_msm_inner = true;
        const int rc = device_self_test(s);
        tl_msm_inner = false;
        if (rc != KZG355_OK) { kzg355_free_trusted_setup(s); return rc; }
    }
    if (opt.msm_eager || s->msm_required) {
        const int rc = ensure_wide_table(s);
        if (rc != KZG355_OK && s->msm_required) { kzg355_free_trusted_setup(s); return rc; }
    }
    *out = s;
    return KZG355_OK;
}

// Known-answer self-test run once per handle (~10 ms).  It needs no fixture: for ANY Lagrange-form setup over the domain w_i,
//   sum_i L_i(tau) = 1        =>  the commitment of the all-ones blob is the G1 generator;
//   sum_i w_i L_i(tau) = tau  =>  the commitment C of the blob (w_0, .., w_{N-1}) is [tau]G1, the polynomial p(X) = X, and with
//                                 quotient (X - z)/(X - z) = 1 its proof at any z is the generator:  verify_kzg_proof(C, z, z, G) is
//                                 true, verify_kzg_proof(C, z, z + 1, G) is false.
// This runs the MSM (table build, kernel, finalize, compression), point validation, the r-power kernel, the linear combination (pre-
// shifted and bucket forms), the challenge and evaluation kernels and the cooperative pairing on the device they will run on, and turns a toolchain that miscompiles one of them (DESIGN.md section 4
// records such a case with hipcc 7.2 and four inlined G1 routines) into a load error instead of wrong verdicts.
static int device_self_test(kzg355_settings *s) {
    static const uint8_t G1_GEN[48] = {0x97, 0xf1, 0xd3, 0xa7, 0x31, 0x97, 0xd7, 0x94, 0x26, 0x95, 0x63, 0x8c, 0x4f, 0xa9, 0xac, 0x0f, 0xc3, 0x68, 0x8c, 0x4f, 0x97, 0x74, 0xb9, 0x05,
                                       0xa1, 0x4e, 0x3a, 0x3f, 0x17, 0x1b, 0xac, 0x58, 0x6c, 0x55, 0xe8, 0x3f, 0xf9, 0x7a, 0x1a, 0xef, 0xfb, 0x3a, 0xf0, 0x0a, 0xdb, 0x22, 0xc6, 0xbb};
    const size_t n = (size_t)s->t.n_fe, BB = 32 * n;
    DevBuf blobs;
    int rc = blobs.ensure(2 * BB);
    if (rc) return rc;
    auto fail = [&](const char *what) {
        fprintf(stderr, "kzg355: device self-test FAILED (%s): this build of the library does not compute correctly on this device\n", what);
        blobs.release();
        return KZG355_INTERNAL;
    };
    std::vector<uint8_t> ones(BB, 0);
    for (size_t i = 0; i < n; i++) ones[32 * i + 31] = 1;
    if (hipMemcpy(blobs.p, ones.data(), BB, hipMemcpyHostToDevice) != hipSuccess) { blobs.release(); return KZG355_DEVICE_ERROR; }
    launch_fr_to_bytes(s->t.roots, (int)n, blobs.as<uint8_t>() + BB, nullptr);            // the blob (w_0, .., w_{N-1}), big-endian canonical
    if (hipDeviceSynchronize() != hipSuccess) { blobs.release(); return KZG355_DEVICE_ERROR; }
    uint8_t c[96]; int st[2] = {0, 0};
    rc = msm_op_many_device_impl(c, st, blobs.as<uint8_t>(), nullptr, 2, s);
    if (rc == KZG355_NO_DEVICE || rc == KZG355_NO_MEMORY || rc == KZG355_DEVICE_ERROR) { blobs.release(); return rc; }
    if (rc != KZG355_OK) return fail("commitment kernels report an error on canonical blobs");
    if (memcmp(c, G1_GEN, 48) != 0) {
        // Either the arithmetic is wrong or the caller's points are not a Lagrange basis of this domain (the reference loads any
        // on-curve points that pass its pairing check, kzg.rs:833-899).  Tell the two apart with the other, independent MSM form:
        // if both forms agree the setup is merely unusual and the identities above do not apply -- nothing more can be checked.
        if (!s->wide_pub.load(std::memory_order_acquire)) { blobs.release(); return KZG355_OK; }
        uint8_t c2[96];
        tl_force_bucket = true;
        rc = msm_op_many_device_impl(c2, st, blobs.as<uint8_t>(), nullptr, 2, s);
        tl_force_bucket = false;
        if (rc != KZG355_OK || memcmp(c, c2, 96) != 0) return fail("the wide-table and the bucket form of the MSM disagree");
        blobs.release();
        return KZG355_OK;
    }
    uint8_t z[32] = {0}, y_bad[32] = {0};
    z[31] = 5; y_bad[31] = 6;
    bool ok = false;
    rc = kzg355_verify_kzg_proof(&ok, c + 48, z, z, G1_GEN, s);
    if (rc != KZG355_OK || !ok) return fail("verify_kzg_proof([tau]G1, z, z, G1) is not true");
    ok = true;
    rc = kzg355_verify_kzg_proof(&ok, c + 48, z, y_bad, G1_GEN, s);
    if (rc != KZG355_OK || ok) return fail("verify_kzg_proof([tau]G1, z, z + 1, G1) is not false");
    // The batch path, through the kernels only large launch sets take (bucket-form linear combination with the single-chain tail,
    // lane-per-batch transcript hash): two batches of 8 copies of the blob of p(X) = X with commitment [tau]G1 and proof G1 (true), the
    // second one with the proof of its last blob replaced by another valid point (false).
    {
        const size_t n = 8, G = 2;
        DevBuf bb, cc, pp;
        auto done = [&](int code) { bb.release(); cc.release(); pp.release(); blobs.release(); return code; };
        if ((rc = bb.ensure(BB * n * G)) || (rc = cc.ensure(48 * n * G)) || (rc = pp.ensure(48 * n * G))) return done(rc);
        std::vector<uint8_t> hc(48 * n * G), hp(48 * n * G);
        for (size_t i = 0; i < n * G; i++) { memcpy(&hc[48 * i], c + 48, 48); memcpy(&hp[48 * i], G1_GEN, 48); }
        memcpy(&hp[48 * (n * G - 1)], c + 48, 48);
        bool copy_ok = hipMemcpy(cc.p, hc.data(), hc.size(), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(pp.p, hp.data(), hp.size(), hipMemcpyHostToDevice) == hipSuccess;
        for (size_t i = 0; i < n * G && copy_ok; i++) copy_ok = hipMemcpy(bb.as<uint8_t>() + BB * i, blobs.as<uint8_t>() + BB, BB, hipMemcpyDeviceToDevice) == hipSuccess;
        if (!copy_ok) return done(KZG355_DEVICE_ERROR);
        const int keep_mode = s->lincomb_mode, keep_chain = s->lc_chain_from, keep_lanes = s->rhash_lanes_from;
        s->lincomb_mode = LC_FORM_BUCKET; s->lc_chain_from = 1; s->rhash_lanes_from = 1;
        bool oks[2] = {false, true}; int sts[2] = {0, 0};
        rc = verify_many_device_impl(oks, sts, bb.as<uint8_t>(), cc.as<uint8_t>(), pp.as<uint8_t>(), n, G, s);
        s->lincomb_mode = keep_mode; s->lc_chain_from = keep_chain; s->rhash_lanes_from = keep_lanes;
        if (rc == KZG355_NO_DEVICE || rc == KZG355_NO_MEMORY || rc == KZG355_DEVICE_ERROR) return done(rc);
        bb.release(); cc.release(); pp.release();
        if (rc != KZG355_OK || !oks[0] || oks[1]) return fail("verify_blob_kzg_proof_batch through the many-batch kernels: expected (true, false)");
    }
    blobs.release();
    return KZG355_OK;
}

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
typedef void *ncclComm_p;
struct RcclApi {
    void *lib = nullptr;
    int (*CommInitAll)(ncclComm_p *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_p) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_p, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    bool load() {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        return CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd;
    }
};
struct MultiDev {
    std::vector<kzg355_settings *> rep;        // rep[0] is the owner handle itself
    RcclApi rccl;
    std::vector<ncclComm_p> comms;             // one per replica when the RCCL exchange is usable (distinct devices)
    int exchange = 0;                          // 0 peer copies, 1 RCCL all-gather (KZG355_EXCHANGE=peer|rccl; default rccl when available)
    std::mutex ex_mu;                          // collectives on one communicator set are issued by one host thread at a time, in one order on every rank
    std::atomic<long> n_allgathers{0}, n_peer_exchanges{0};   // introspection for tests
};

static void free_single(kzg355_settings *s);
static std::vector<kzg355_settings *> replicas_of(kzg355_settings *s) { return s->multi ? s->multi->rep : std::vector<kzg355_settings *>{s}; }

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
    if (n_devices == 1 && !getenv("KZG355_FORCE_MULTI")) { *out = rep[0]; return KZG355_OK; }     // (test hook: a one-device "multi" handle)
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
    if (const char *e = getenv("KZG355_DEVICES")) {
        std::vector<int> devs;
        for (const char *p = e; *p;) {
            char *end = nullptr;
            const long v = strtol(p, &end, 10);
            if (end == p) break;
            devs.push_back((int)v);
            p = *end == ',' ? end + 1 : end;
        }
        if (!devs.empty()) return load_devices(g1_bytes, n1, g2_bytes, n2, devs.data(), devs.size(), o, out);
    }
    return load_on_device(g1_bytes, n1, g2_bytes, n2, -1, o, out);
}
int kzg355_settings_device_count(const kzg355_settings *s) { return !s ? 0 : s->multi ? (int)s->multi->rep.size() : 1; }
int kzg355_settings_exchange_stats(const kzg355_settings *s, long *allgathers, long *peer_exchanges) {
    if (!s || !allgathers || !peer_exchanges) return KZG355_BADARGS;
    *allgathers = s->multi ? s->multi->n_allgathers.load() : 0L;
    *peer_exchanges = s->multi ? s->multi->n_peer_exchanges.load() : 0L;
    return s->multi ? s->multi->exchange : -1;
}

int kzg355_load_trusted_setup_file(const char *path, kzg355_settings **out) {
    if (!path || !out) return KZG355_BADARGS;
    FILE *f = fopen(path, "r");
    if (!f) return KZG355_INVALID_TRUSTED_SETUP;                 // kzg.rs:907-909
    char line[1024];
    auto read_count = [&](size_t *v) {
        if (!fgets(line, sizeof line, f)) return false;
        char *end = nullptr;
        unsigned long x = strtoul(line, &end, 10);
        if (end == line) return false;
        while (*end == ' ' || *end == '\t' || *end == '\r' || *end == '\n') end++;
        if (*end) return false;
        *v = x;
        return true;
    };
    size_t n1 = 0, n2 = 0;
    if (!read_count(&n1) || (n1 != (size_t)N_FE && !(n1 >= (size_t)SMALL_N_MIN && n1 <= (size_t)SMALL_N_MAX && (n1 & (n1 - 1)) == 0))) { fclose(f); return KZG355_INVALID_TRUSTED_SETUP; }   // kzg.rs:916-932
    if (!read_count(&n2) || n2 != (size_t)N_G2) { fclose(f); return KZG355_INVALID_TRUSTED_SETUP; }   // kzg.rs:934-950
    std::vector<uint8_t> g1(48 * n1), g2(96 * n2);
    int rc = KZG355_OK;
    for (size_t i = 0; i < n1 + n2 && rc == KZG355_OK; i++) {
        const size_t want = i < n1 ? 48 : 96;
        uint8_t *dst = i < n1 ? &g1[48 * i] : &g2[96 * (i - n1)];
        if (!fgets(line, sizeof line, f)) { rc = KZG355_INVALID_TRUSTED_SETUP; break; }                // kzg.rs:957-959
        char *p = line; size_t len = strlen(p);
        while (len && (p[len - 1] == '\n' || p[len - 1] == '\r' || p[len - 1] == ' ' || p[len - 1] == '\t')) p[--len] = 0;
        if (len >= 2 && p[0] == '0' && p[1] == 'x') { p += 2; len -= 2; }                                // hex_to_bytes kzg.rs:82-86
        if (len % 2) { rc = KZG355_INVALID_HEX; break; }
        if (len != 2 * want) { rc = KZG355_BADARGS; break; }
        for (size_t k = 0; k < want; k++) {
            const int hi = hexval(p[2 * k]), lo = hexval(p[2 * k + 1]);
            if (hi < 0 || lo < 0) { rc = KZG355_INVALID_HEX; break; }
            dst[k] = (uint8_t)(hi * 16 + lo);
        }
    }
    fclose(f);
    if (rc != KZG355_OK) return rc;
    return kzg355_load_trusted_setup(g1.data(), n1, g2.data(), n2, out);
}

int kzg355_lagrange_setup_from_monomial(uint8_t *out, const uint8_t *monomial_g1, size_t n) {
    if (!out || !monomial_g1) return KZG355_BADARGS;
    if (n < (size_t)SMALL_N_MIN || n > (size_t)SMALL_N_MAX || (n & (n - 1))) return KZG355_BADARGS;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return KZG355_NO_DEVICE;
    DeviceScope scope;
    if (const char *e = getenv("KZG355_DEVICE")) { const int dev = atoi(e); if (dev < 0 || dev >= ndev || !scope.enter(dev)) return KZG355_NO_DEVICE; }
    DevBuf in, res, err;
    int rc = KZG355_OK;
    auto done = [&](int code) { in.release(); res.release(); err.release(); return code; };
    if ((rc = in.ensure(48 * n)) || (rc = res.ensure(48 * n)) || (rc = err.ensure(sizeof(int)))) return done(rc);
    if (hipMemcpy(in.p, monomial_g1, 48 * n, hipMemcpyHostToDevice) != hipSuccess || hipMemset(err.p, 0, sizeof(int)) != hipSuccess) return done(KZG355_DEVICE_ERROR);
    launch_lagrange_from_monomial(in.as<uint8_t>(), (int)n, res.as<uint8_t>(), err.as<int>(), nullptr);
    int herr = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&herr, err.p, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return done(KZG355_DEVICE_ERROR);
    if (herr) return done(KZG355_BADARGS);
    if (hipMemcpy(out, res.p, 48 * n, hipMemcpyDeviceToHost) != hipSuccess) return done(KZG355_DEVICE_ERROR);
    return done(KZG355_OK);
}

static void free_single(kzg355_settings *s) {
    if (!s) return;
    DeviceScope scope;
    (void)scope.enter(s->device);
    {
        // a caller's bug (kzg355.h: collect every ticket first).  The tickets hold this handle, its workspaces and streams: the free is DEFERRED to the
        // collect of the last of them (kzg355_verify_collect) -- nothing is leaked and no ticket dangles
        std::lock_guard<std::mutex> lk(s->pipe_mu);
        if (s->tickets_out.load() > 0) {
            fprintf(stderr, "kzg355: handle freed with %d submitted launch set(s) not collected: it is released when the last of them is collected\n", s->tickets_out.load());
            s->free_deferred = true;
            return;
        }
    }
    for (Workspace *w : s->pool) delete w;
    s->pool.clear();
    if (s->side_stream) { (void)hipStreamDestroy(s->side_stream); s->side_stream = nullptr; }
    if (s->side2_stream) { (void)hipStreamDestroy(s->side2_stream); s->side2_stream = nullptr; }
    if (s->pipe_main) { (void)hipStreamDestroy(s->pipe_main); s->pipe_main = nullptr; }
    if (s->pipe_tail) { (void)hipStreamDestroy(s->pipe_tail); s->pipe_tail = nullptr; }
    delete s->host_pool; s->host_pool = nullptr;
    s->roots.release(); s->eval_tab.release(); s->wide.release(); s->msm_table.release(); s->lines.release(); s->lines_inf.release(); s->g1_first2.release();
    s->lines_w.release(); s->frob.release(); s->prog.release(); s->scheds.release();
    delete s;
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

int kzg355_settings_device(const kzg355_settings *s) { return s ? s->device : -1; }
int kzg355_settings_field_elements_per_blob(const kzg355_settings *s) { return s ? s->t.n_fe : 0; }
int kzg355_settings_msm_form(const kzg355_settings *s) {
    if (!s) return 0;
    if (const kzg355_settings::WidePub *wp = s->wide_pub.load(std::memory_order_acquire)) return wp->shape.bits;
    if (s->wide_table_failed) return -8;
    return s->msm_bits_wanted == 8 ? 8 : s->msm_bits_wanted;     // not built yet: the width asked for, 0 = to be sized from the free HBM
}
int kzg355_settings_msm_shape(const kzg355_settings *s, int *bits, int *windows, int *glv, size_t *table_bytes) {
    if (!s) return KZG355_BADARGS;
    const kzg355_settings::WidePub *wp = s->wide_pub.load(std::memory_order_acquire);
    const bool built = wp != nullptr;
    if (bits) *bits = built ? wp->shape.bits : 0;
    if (windows) *windows = built ? wp->shape.windows : 0;
    if (glv) *glv = built ? wp->shape.glv : 0;
    if (table_bytes) *table_bytes = built ? wide_table_bytes(wp->shape) : 0;
    return KZG355_OK;
}
int kzg355_settings_build_msm_table(const kzg355_settings *cs) {
    if (!cs) return KZG355_BADARGS;
    kzg355_settings *s = const_cast<kzg355_settings *>(cs);
    for (kzg355_settings *r : replicas_of(s)) { const int rc = ensure_wide_table(r); if (rc) return rc; }
    return s->wide_pub.load(std::memory_order_acquire) ? KZG355_OK : (s->msm_bits_wanted == 8 ? KZG355_OK : s->wide_rc == KZG355_OK ? KZG355_NO_MEMORY : s->wide_rc);
}
void kzg355_set_kernel_timing(kzg355_settings *s, int enabled) { if (s) s->timing = enabled != 0; }
long kzg355_settings_host_hashed_calls(const kzg355_settings *s) { return s ? s->n_host_hashed.load() : 0L; }
int kzg355_settings_set_host_hash(kzg355_settings *s, int mode, int max_blobs) {
    if (!s || mode < -1 || mode > 1 || max_blobs < 0) return KZG355_BADARGS;
    // both Fiat-Shamir hashes follow the mode: -1 keeps the per-blob challenges AND the batch challenge r on the device
    for (kzg355_settings *r : s->multi ? replicas_of(s) : std::vector<kzg355_settings *>{s}) { r->host_hash = mode; r->host_rhash = mode < 0 ? -1 : r->host_rhash_loaded; if (max_blobs) r->host_hash_max = max_blobs; }
    return KZG355_OK;
}
int kzg355_host_sha256(uint8_t out[32], const uint8_t *msg, size_t len, int impl) {
    if (!out || (!msg && len) || impl < 0 || impl > 2) return KZG355_BADARGS;
    return kzg_host::sha256(out, msg, len, impl) ? KZG355_OK : KZG355_INTERNAL;      // INTERNAL: SHA extensions asked for, CPU has none
}
int kzg355_host_challenge_digests(uint8_t *out, const uint8_t *blobs, size_t blob_bytes, const uint8_t *commitments, size_t n, int impl) {
    if (!out || !blobs || !commitments || blob_bytes % 32 || blob_bytes == 0 || impl < 0 || impl > 2) return KZG355_BADARGS;
    if (impl == 2 && !kzg_host::sha256_have_shani()) return KZG355_INTERNAL;
    kzg_host::challenge_digests(out, blobs, blob_bytes, commitments, n, (uint64_t)(blob_bytes / 32), impl);
    return KZG355_OK;
}
double kzg355_last_kernel_ms(const kzg355_settings *cs, const char *family) {
    kzg355_settings *s = const_cast<kzg355_settings *>(cs);
    if (!s || !family) return -1.0;
    std::lock_guard<std::mutex> lk(s->mu);
    auto it = s->last_ms.find(family);
    return it == s->last_ms.end() ? -1.0 : it->second.last;
}
int kzg355_kernel_ms_stats(const kzg355_settings *cs, const char *family, double *total_ms, long *launches) {
    kzg355_settings *s = const_cast<kzg355_settings *>(cs);
    if (!s || !family || !total_ms || !launches) return KZG355_BADARGS;
    std::lock_guard<std::mutex> lk(s->mu);
    auto it = s->last_ms.find(family);
    if (it == s->last_ms.end()) { *total_ms = 0; *launches = 0; return KZG355_OK; }
    *total_ms = it->second.total; *launches = it->second.count;
    return KZG355_OK;
}
void kzg355_reset_kernel_stats(kzg355_settings *s) {
    if (!s) return;
    std::lock_guard<std::mutex> lk(s->mu);
    s->last_ms.clear();
}

// ---- asynchronous device-resident verification: submit / collect -----------------------------------------------------------------
// One host thread keeps several launch sets in flight: submit() takes a workspace (its own stream and scratch) from the handle's pool,
// queues the whole chain of a launch set on it without waiting and hands back a ticket; collect() waits for that set, writes its
// verdicts and returns the workspace.  Sets submitted back to back sit on different streams, so the narrow tail of set k (r powers,
// Horner chains, pairing: a few waves per SIMD at most) runs under the wide kernels of set k + 1 -- what a caller with mid-size sets
// (1024 batches = 8.6 GB of blobs) needs instead of one 69 GB set.  The batch challenge stays on the device here (no host round trip
// inside a chain that is one of several in flight).
struct kzg355_ticket {
    kzg355_settings *s = nullptr;
    Workspace *w = nullptr;
    std::unique_ptr<Timed> tm;
    size_t npg = 0, groups = 0;
    bool immediate = false;          // nothing was queued (groups == 0 or empty batches: kzg.rs:653-655)
    bool tail_queued = false;        // stage 2 of this set is on the tail stream (else it is the handle's pending_tail)
    int tail_rc = KZG355_OK;         // what queueing it returned
};

namespace {
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
}  // namespace

int kzg355_verify_blob_kzg_proof_batch_many_device_submit(kzg355_ticket **ticket, const uint8_t *d_blobs, const uint8_t *d_commitments,
                                                          const uint8_t *d_proofs, size_t n_per_group, size_t groups, const kzg355_settings *cs) {
    if (!ticket || !cs) return KZG355_BADARGS;
    *ticket = nullptr;
    if (n_per_group > (size_t)1 << 24 || groups > (size_t)1 << 24 || n_per_group * groups > (size_t)1 << 24) return KZG355_BADARGS;      // (each factor first: the product must not wrap)
    std::unique_ptr<kzg355_ticket> t(new kzg355_ticket());
    kzg355_settings *s = t->s = const_cast<kzg355_settings *>(cs);
    t->npg = n_per_group; t->groups = groups;
    if (groups == 0 || n_per_group == 0) { t->immediate = true; *ticket = t.release(); return KZG355_OK; }
    if (!d_blobs || !d_commitments || !d_proofs || ((uintptr_t)d_blobs & 15) || ((uintptr_t)d_commitments & 3) || ((uintptr_t)d_proofs & 3)) return KZG355_BADARGS;
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
        if (hipStreamCreateWithFlags(&s->pipe_main, hipStreamNonBlocking) != hipSuccess) { s->pipe_main = nullptr; (void)hipGetLastError(); return fail(KZG355_DEVICE_ERROR); }
        if (hipStreamCreateWithFlags(&s->pipe_tail, hipStreamNonBlocking) != hipSuccess) { s->pipe_tail = nullptr; (void)hipGetLastError(); return fail(KZG355_DEVICE_ERROR); }
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
        if (s->pending_tail == prev && prev) { s->pending_tail = nullptr; prev->tail_rc = queue_tail(prev, nullptr); }   // the earlier set must not wait for a set that never came
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
    if (!d_records || ((uintptr_t)d_records & 15) || ((uintptr_t)d_points & 3) || !d_blobs || ((uintptr_t)d_blobs & 15) || !d_commitments || !d_proofs) return refuse(KZG355_BADARGS);
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
    if ((rc = run_stage1(s, w, tm, d_blobs, d_commitments, d_proofs, (int)(n_local * groups), (int)n_local, d_records, reinterpret_cast<G1Affine *>(d_points), w->err.as<int>(), false,
                         via_host ? &hf : nullptr))) return rc;
    if ((rc = join_side(w))) return rc;
    if (d_words) {                                                // statuses stay on the device: no copy back, the caller reads them after its merge
        if ((uintptr_t)d_words & 3) return KZG355_BADARGS;
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

int kzg355_verify_shard_records_device(uint8_t *d_records, int *status, const uint8_t *d_blobs, const uint8_t *d_commitments,
                                       const uint8_t *d_proofs, size_t n_local, size_t groups, const kzg355_settings *cs) {
    return shard_records_impl(d_records, nullptr, status, d_blobs, d_commitments, d_proofs, n_local, groups, cs);
}
int kzg355_verify_shard_records_points_words_device(uint8_t *d_records, uint8_t *d_points, int32_t *d_status_words, const uint8_t *d_blobs,
                                                    const uint8_t *d_commitments, const uint8_t *d_proofs, size_t n_local, size_t groups, const kzg355_settings *cs) {
    if (!d_points || !d_status_words) return KZG355_BADARGS;
    return shard_records_impl(d_records, d_points, nullptr, d_blobs, d_commitments, d_proofs, n_local, groups, cs, d_status_words);
}
int kzg355_verify_shard_records_points_device(uint8_t *d_records, uint8_t *d_points, int *status, const uint8_t *d_blobs, const uint8_t *d_commitments,
                                              const uint8_t *d_proofs, size_t n_local, size_t groups, const kzg355_settings *cs) {
    if (!d_points) return KZG355_BADARGS;
    return shard_records_impl(d_records, d_points, status, d_blobs, d_commitments, d_proofs, n_local, groups, cs);
}

namespace {
// stage 2 over gathered records; `dump` (host, groups*128 bytes or null) receives r | proof_lincomb | rhs per batch; d_points (or
// null): the validated affine points of the records as stage 1 produced them ([batch][commitments, proofs]), sparing their decompression
static int verify_records_impl(bool *ok, int *status, uint8_t *dump, const uint8_t *d_records, size_t n, size_t groups, int validate, const kzg355_settings *cs,
                        const uint8_t *d_points = nullptr, int32_t *d_words = nullptr) {
    if (!cs || (!ok && !d_words)) return KZG355_BADARGS;
    if (groups == 0) return KZG355_OK;
    auto refuse = [&](int code) { if (status) for (size_t i = 0; i < groups; i++) status[i] = code; return code; };      // whole-call refusals mark every batch
    if (n == 0) return refuse(KZG355_BADARGS);                   // verify_kzg_proof_batch: n == 0 is an error (kzg.rs:588-592)
    if (n > (size_t)1 << 24 || groups > (size_t)1 << 24 || n * groups > (size_t)1 << 24) return refuse(KZG355_BADARGS);
    if (!d_records || ((uintptr_t)d_records & 15) || ((uintptr_t)d_points & 3)) return refuse(KZG355_BADARGS);      // the kernels read the records 16 bytes at a time
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
        tm.begin("points_from_records"); launch_points_from_records(d_records, (int)(n * groups), (int)n, w->pts.as<G1Affine>(), w->err.as<int>(), w->stream); tm.end();
    }
    if ((rc = run_stage2(s, w, tm, d_records, (int)n, G, validate, pts, w->err.as<int>(), w->ok.as<int>()))) return rc;
    if (dump) {
        launch_dump_intermediates(w->scal_a.as<uint32_t>(), w->pair_pts.as<PairPt>(), (int)n, G, w->out48.as<uint8_t>(), w->stream);
        HIPCHK(hipMemcpyAsync(w->h_out.p, w->out48.p, 128 * groups, hipMemcpyDeviceToHost, w->stream));
    }
    if (d_words) {                                                // 1 + ok + 256 * status per batch, left on the device (the sharded path's all-reduce takes them from there)
        if ((uintptr_t)d_words & 3) return KZG355_BADARGS;
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
}  // namespace

int kzg355_verify_records_device(bool *ok, int *status, const uint8_t *d_records, size_t n, size_t groups, const kzg355_settings *cs) {
    return verify_records_impl(ok, status, nullptr, d_records, n, groups, 0, cs);
}
int kzg355_verify_records_points_device(bool *ok, int *status, const uint8_t *d_records, const uint8_t *d_points, size_t n, size_t groups, const kzg355_settings *cs) {
    if (!d_points) return KZG355_BADARGS;
    return verify_records_impl(ok, status, nullptr, d_records, n, groups, 0, cs, d_points);
}
int kzg355_verify_records_points_words_device(int32_t *d_words, const uint8_t *d_records, const uint8_t *d_points, size_t n, size_t groups, const kzg355_settings *cs) {
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

// ---- multi-device execution of the host-buffer entry points ------------------------------------------------------------
namespace {

static int single_verify_many(bool *ok, int *status, const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs, size_t npg, size_t groups,
                       const kzg355_settings *cs) {
    HostCall hc{0, blobs, commitments, proofs, npg, ok, nullptr, status};
    return host_pipeline(hc, groups, cs);
}

// contiguous ranges of `units` over D devices; the work of device d is fn(d, first unit, count) on its own host thread
static int fan_out(size_t D, size_t units, const std::function<int(size_t, size_t, size_t)> &fn) {
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
static int multi_verify_sharded(bool *ok, int *status, const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs, size_t npg, size_t groups,
                         const kzg355_settings *cs, uint8_t *dump = nullptr) {
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
                                     w->records.as<uint8_t>(), w->pts.as<G1Affine>(), w->err.as<int>(), false))) return rc;
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
                if ((rc = run_stage2(rs, w, tm, gath.as<uint8_t>(), (int)npg, (int)G, 0, gpts.as<G1Affine>(), w->err.as<int>(), w->ok.as<int>()))) return rc;
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
                    for (size_t d = 0; d < D; d++) if (st == KZG355_OK) st = st1[d][g];    // an Err on any block is an Err of the batch (the `?`s of kzg.rs:673-682)
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

static int multi_verify_many(bool *ok, int *status, const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs, size_t npg, size_t groups,
                      const kzg355_settings *cs) {
    MultiDev *m = cs->multi;
    const size_t D = m->rep.size(), BB = blob_bytes_of(cs);
    const bool force_sharded = getenv("KZG355_FORCE_SHARDED") != nullptr && npg >= D;         // (test hook)
    if (!force_sharded && (groups >= D || npg < 2 * D)) {               // enough independent batches (or batches too small to cut): ranges of batches, no exchange
        return fan_out(D, groups, [&](size_t d, size_t g0, size_t n) -> int {
            if (hipSetDevice(m->rep[d]->device) != hipSuccess) return KZG355_NO_DEVICE;
            return single_verify_many(ok + g0, status ? status + g0 : nullptr, blobs + BB * npg * g0, commitments + 48 * npg * g0, proofs + 48 * npg * g0, npg, n, m->rep[d]);
        });
    }
    return multi_verify_sharded(ok, status, blobs, commitments, proofs, npg, groups, cs);
}

}  // namespace

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
    if (n_per_group > (size_t)1 << 24 || groups > (size_t)1 << 24 || n_per_group * groups > (size_t)1 << 24 || blob_bytes_of(cs) * n_per_group * groups > ((size_t)64 << 20)) return KZG355_BADARGS;      // one chunk
    HostCall hc{0, blobs, commitments, proofs, n_per_group, ok, nullptr, status};
    hc.records_out = records_out;
    return host_pipeline(hc, groups, cs);
}

int kzg355_debug_verify_sharded_intermediates(uint8_t *out, bool *ok, int *status, const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs,
                                              size_t n_per_group, size_t groups, const kzg355_settings *cs) {
    if (!cs || !ok || !out || !blobs || !commitments || !proofs || !cs->multi || groups == 0) return KZG355_BADARGS;
    if (n_per_group < cs->multi->rep.size() || n_per_group > (size_t)1 << 24 || groups > (size_t)1 << 24 || n_per_group * groups > (size_t)1 << 24) return KZG355_BADARGS;      // every device gets a block of every batch
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
        tm.begin("validate_points"); launch_validate_points(d_rec, d_rec + 112, 1, 1, w->pts.as<G1Affine>(), w->err.as<int>(), w->stream, RECORD_BYTES); tm.end();
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
        launch_small_proof(w->blobs.as<uint8_t>(), nullptr, w->z.as<Fr>(), 1, s->t, w->out48.as<uint8_t>(), w->records.as<uint8_t>(), w->err.as<int>(), w->stream);
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
