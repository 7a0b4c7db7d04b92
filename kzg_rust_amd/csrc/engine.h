// engine.h -- the host side of libkzg355.so behind its C ABI (include/kzg355.h): per-call workspaces, the settings handle, and the
// functions the translation units of the host code share.  Round 5 cut the single 2,200-line api.hip into
//   workspace.hip      workspace pool, side streams, dispatch helpers
//   verify_stages.hip  stage 1 / stage 2 of verification, launch sets, the device-resident call
//   msm_ops.hip        the fixed-base MSM table and the commitment / proof chains
//   host_pipeline.hip  host buffers -> chunks -> workspaces (H2D / kernels / results overlapped)
//   options.hip        kzg355_options, loading a handle on one device, the load-time self-test, getters / setters
//   multi_device.hip   handles over several devices: replicas, the RCCL binding, the record exchange
//   entry_points.hip   the C entry points: submit / collect, device-resident and host-buffer calls
// Mirrors the control flow of the reference's `impl Kzg` forwards and the functions behind them (src/kzg.rs:401-693, 833-979): argument
// checks and early exits happen on the host, all arithmetic on the device.  There is no CPU fallback: without a usable HIP device every
// entry point returns KZG355_NO_DEVICE (a HIP call that fails on a device that exists: KZG355_DEVICE_ERROR).
#pragma once
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

// KZG355_DEBUG / KZG355_DEBUG_PIPE (debug switches; read once, in options.hip like every other environment variable)
namespace kzg355_impl { bool debug_errors(); bool debug_pipe(); }

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            if (::kzg355_impl::debug_errors()) fprintf(stderr, "kzg355: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return _e == hipErrorOutOfMemory ? KZG355_NO_MEMORY                                       \
                   : (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice) ? KZG355_NO_DEVICE : KZG355_DEVICE_ERROR; \
        }                                                                                              \
    } while (0)

// The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue run one
// after the other.  A small call of this library is a chain on three streams, so the default lets one such call run at full speed and
// serialises the streams of concurrent ones (threads of n = 64 calls on one handle, median ms per call at 1 / 2 / 4 / 8 threads: 2.2 / 3.0 / 5.0 / 6.9
// with 4 queues, 2.2 / 2.3 / 2.9 / 3.3 with 24; profiles/r04/concurrent_small_calls.txt).  The variable belongs to the PROCESS: a host that serves
// concurrent small calls exports GPU_MAX_HW_QUEUES=24 before its first HIP call (kzg_rust_amd/_lib.py and bench.py do).  The library
// does not touch the environment (round 4 set it from a constructor: ADVICE r4); it reads the variable once per handle and otherwise assumes 4.


namespace kzg355_impl {

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
    // `stream` is this one except while a submitted set borrows the handle's pipeline streams
    hipStream_t own_stream = nullptr;
    // submit / collect: stage 1 queued on pipe_main is done; the whole set is done; this set's hash is done
    hipEvent_t ev_stage = nullptr, ev_done = nullptr, ev_fork2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_pts = nullptr, ev_shift = nullptr;
    // work on the side streams that the main stream has not waited for yet: everything (ev_join: the validation verdicts are in the error
    // words), the decoded points alone (ev_pts; recorded only when they are ready before the verdicts), the window shifts (ev_shift)
    bool side_pending = false, pts_pending = false, shift_pending = false;
    DevBuf blobs, commitments, proofs, records, z, y, pts, scal_a, scal_b, scal_c, pair_pts, pair_f, ok, err, digits, partials, q, out48, small, lc_partials,
            shifts, digests, zpow, qprep;
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
        for (DevBuf *b : {&blobs, &commitments, &proofs, &records, &z, &y, &pts, &scal_a, &scal_b, &scal_c, &pair_pts, &pair_f, &ok, &err, &digits, &partials,
                &q, &out48, &small, &lc_partials, &shifts, &digests, &zpow, &qprep}) b->release();
        h_ok.release(); h_err.release(); h_out.release(); h_stage.release(); h_stage_cp.release(); h_digests.release(); h_records.release(); h_rdig.release();
        if (ev_ok) for (auto &e : ev) (void)hipEventDestroy(e);
        for (hipEvent_t e : {ev_fork, ev_join, ev_pts, ev_shift, ev_stage, ev_done, ev_fork2}) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_d2h) if (e) (void)hipEventDestroy(e);
        if (own_stream) (void)hipStreamDestroy(own_stream);       // (side is the handle's shared stream unless owns_side)
        if (owns_side && side) (void)hipStreamDestroy(side);
        if (owns_side2 && side2) (void)hipStreamDestroy(side2);
    }
};

}  // namespace kzg355_impl
using namespace kzg355_impl;

struct MultiDev;
struct kzg355_ticket;
struct kzg355_settings {
    int device = 0;
    MultiDev *multi = nullptr;      // handles created over several devices: the replicas and the exchange (owner handle only)
    DeviceTables t{};
    DevBuf roots, eval_tab, wide, msm_table, lines, lines_inf, g1_first2, lines_w, frob, prog, scheds;
    bool lane_pairing = false;
    int split_parts = 1, split_streams = 2;   // KZG355_SPLIT=parts[,streams]: device-resident verify calls as several overlapped launch sets (default: one)
    int challenge_form = 0;   // 0 by size, 1 one-wave kernel, 2 two-wave kernel (KZG355_CHALLENGE=1w|2w)
    int quotient_form = 0;    // k_quotient_tree: 0 by size, 2 / 4 / 6 = 2^form leaves per lane (kzg355_options.quotient_form: tuning knob and test hook)
    // batches per launch set from which the bucket form ends in one Horner chain per class (KZG355_LC_CHAIN_FROM).  Round 4, with the chain walked by quads and
    // sets kept in flight (blobs/s, three sets in flight, 16 chains per class against one): 512 batches 2.81 M either way, 1024 3.99 -> 4.03 M, 2048 4.14 ->
    // 4.26 M, 4096 4.25 -> 4.36 M (profiles/r04/chain_from_sweep.txt); one set at a time it is within +-2 % from 512 to 4096
    int lc_chain_from = 1024;
    // batches per launch set from which the r-transcripts are hashed one lane per batch (KZG355_RHASH_LANES_FROM); measured: 1024 batches of 512 records 6.75
    // -> 3.47 ms, 8192 of 64: 2.44 -> 0.57 ms
    int rhash_lanes_from = 1024;
    int lincomb_mode = 0;     // 0 auto, 1 windowed per-term, 2 bucket method, 3 pre-shifted (KZG355_LINCOMB=window|bucket|preshift)
    int beside_max_blobs = 16384;  // blobs per launch set up to which the point kernels run on side streams beside the hash chain (64 per CU)
    int cu_count = 256;            // compute units of the device: the thresholds above and below are multiples of it (load_on_device)
    int pairing_two_wave_upto = 256;     // batches per launch set up to which a pairing runs its two Miller loops on two waves (1 per CU)
    // ... and on how many segments per loop (k_pairing_coop_split; 0: launch_pairing's default; kzg355_options.miller_segments = 1..4)
    int miller_segments = 0;
    // batches per launch set from which the final exponentiation's hard part runs twelve lanes per check (16 per CU; KZG355_PAIRING_HARD12_FROM, 0: never)
    int pairing_hard12_from = 4096;
    int challenge_two_wave_upto = 32768; // blobs per launch set up to which the Fiat-Shamir hash runs as producer / consumer wave pairs (2 workgroups per CU)
    std::mutex mu;
    // shared by the workspaces (point validation / window shifts of small calls next to the main chain)
    hipStream_t side_stream = nullptr, side2_stream = nullptr;
    // submit / collect: stage 1 of every submitted set in order on pipe_main, stage 2 on pipe_tail
    hipStream_t pipe_main = nullptr, pipe_tail = nullptr;
    std::mutex pipe_mu;                                           // orders the submits / collects that queue work on the two
    struct kzg355_ticket *pending_tail = nullptr;                 // the submitted set whose stage 2 is not queued yet (it goes out behind the next set's hash)
    std::atomic<int> tickets_out{0};                              // submitted and not yet collected
    // kzg355_free_trusted_setup came while tickets were out (the caller's bug): the handle stays alive until the last of them is collected (under pipe_mu)
    bool free_deferred = false;
    // side streams are per workspace (round 4; measured with 4 threads of n = 64 calls: median call 5.0-7.9 ms with one pair per handle, 3.6-5.2 ms own with 4
    // hardware queues, 2.5-3.9 ms with 8); the handle's shared pair is the fallback when a stream cannot be created
    bool force_sharded = false;      // test hook (kzg355_options.force_sharded): a multi-device handle shards every batch
    std::atomic<int> calls_in_flight{0};   // host-buffer calls inside host_pipeline right now
    int hw_queues = 4;                     // GPU_MAX_HW_QUEUES as the process has it when the handle is loaded (the runtime's default is 4)
    // 0 by size; 1: every submitted set on its workspace's own stream; 2: two-stage software pipeline over pipe_main / pipe_tail (KZG355_SUBMIT=sets|pipeline)
    int submit_mode = 0;
    // Fiat-Shamir hashing of host-buffer calls on host threads: 0 by size (<= host_hash_max blobs), 1 always, -1 never (KZG355_HOST_HASH=auto|on|off)
    int host_hash = 0;
    // blobs per call up to which the host hashes (KZG355_HOST_HASH_MAX): measured, profiles/r03/host_hash_crossover_v4.txt: host route ahead up to 4096 blobs
    // (17.1 against 18.4 ms), level at 8192
    int host_hash_max = 4096;
    // device-resident verify / blob-proof calls of up to this many blobs copy them BACK and hash on the host threads (0.16 ms of D2H per 64 blobs + ~35 us x
    // blobs / threads against the 3.7 ms device chain); KZG355_HOST_HASH_DEVICE_MAX, 0 in the options = 1024, -1 never.  Measured
    // (profiles/r05/device_host_hash_crossover.txt, ms per call, host route / device hash): 64 blobs 2.0 / 5.6, 512: 3.3 / 6.1, 1024: 4.8 / 6.3, 1536: 6.6 /
    // 6.5, 2048: 8.3 / 6.9
    int host_hash_device_max = 1024;
    // batch challenge r of lone small calls hashed on the host (records copied back): 0 by size, -1 never (KZG355_HOST_RHASH=off)
    int host_rhash = 0;
    int host_rhash_loaded = 0;       // ... as the handle was loaded: kzg355_settings_set_host_hash(-1) forces -1, any other mode puts this back
    // records per call up to which that is done (one 512-blob batch -- BASELINE config 5 -- is a 1281-compression chain = 2.6 ms on a device lane, ~0.1 ms this
    // way)
    int host_rhash_max_records = 1024;
    // host SHA-256 form: 0 auto (SHA extensions when the CPU has them), 1 portable, 2 SHA extensions (KZG355_HOST_SHA=portable|shani)
    int sha_impl = 0;
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

namespace kzg355_impl {

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
    // put the current device back whatever happens in between
    void hold() { if (hipGetDevice(&prev) == hipSuccess) changed = true; else (void)hipGetLastError(); }
    ~DeviceScope() { if (changed) (void)hipSetDevice(prev); }
};
Workspace *ws_acquire(kzg355_settings *s);
void ws_release(kzg355_settings *s, Workspace *w);
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

inline bool is_small(const kzg355_settings *s) { return s->t.n_fe != N_FE; }
inline size_t blob_bytes_of(const kzg355_settings *s) { return (size_t)32 * s->t.n_fe; }

// which form the batch linear combination takes (KZG355_LINCOMB pins it: 1 window, 2 bucket, 3 pre-shifted)
enum { LC_FORM_WINDOW = 1, LC_FORM_BUCKET = 2, LC_FORM_PRESHIFT = 3,
        LC_FORM_SINGLE = 4 /* one record per batch, many batches: never pinned, chosen by shape */ };

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
    std::shared_ptr<std::atomic<int>> failed;   // device-resident form: a worker could not wait for its chunk (the digests are then not to be used)
    void finish() { if (running) { running = false; pool->finish(job); job.reset(); } }
    bool ok() const { return !failed || failed->load() == 0; }
    ~HostFront() { finish(); }
};

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
    // 0 verify, 1 commit, 2 blob proof, 3 proof at a given z per blob (compute_kzg_proof: `commitments` is null, zs / ys_out set)
    int kind;
    const uint8_t *blobs, *commitments, *proofs;
    size_t npg;                      // blobs per unit
    bool *ok; uint8_t *out48; int *status;
    uint8_t *records_out = nullptr;  // verify, single-chunk calls only (kzg355_debug_verify_host_records): the stage-1 records, copied back after the chunk
    const uint8_t *zs = nullptr;     // kind 3: units x 32 bytes, the evaluation points (kzg.rs:446-457)
    uint8_t *ys_out = nullptr;       // kind 3: units x 32 bytes, y = p(z) of every unit whose status is OK
};

// ---- shared functions (definitions: see the file list at the top)
extern thread_local bool tl_msm_inner;      // this thread is inside the table build (or the load-time self-test): its MSM calls take the handle as it is
extern thread_local const kzg355_settings::WidePub *tl_wide_candidate;   // ... and, while the builder checks it, the table that is not published yet
extern thread_local bool tl_force_bucket;   // the load-time self-test's second opinion: the bucket form although a table is published
Workspace *ws_acquire(kzg355_settings *s);
void ws_release(kzg355_settings *s, Workspace *w);
bool ensure_side(kzg355_settings *s, Workspace *w);
bool ensure_side2(kzg355_settings *s, Workspace *w);
int lincomb_form(const kzg355_settings *s, int npg, int groups);
int status_from_err(int err);
int host_hash_from_device(kzg355_settings *s, Workspace *w, HostFront *hf, const uint8_t *d_blobs);
int enqueue_points_beside(kzg355_settings *s, Workspace *w, Timed &tm, const uint8_t *d_c, const uint8_t *d_p, int n_total, int npg, G1Affine *d_pts,
                          int *d_err, bool allow_preshift, int stride = 48 /* bytes between consecutive inputs: 48 packed, 160 inside records */);
int join_shifts(Workspace *w);
int join_points(Workspace *w, bool shifts_too = true);
int join_side(Workspace *w);
int run_stage1(kzg355_settings *s, Workspace *w, Timed &tm, const uint8_t *d_blobs, const uint8_t *d_c, const uint8_t *d_p, int n_total,
               int npg, uint8_t *d_records, G1Affine *d_pts, int *d_err, bool allow_preshift = true, HostFront *hf = nullptr,
               const std::function<int()> *after_challenge = nullptr);
int run_stage2(kzg355_settings *s, Workspace *w, Timed &tm, const uint8_t *d_records, int npg, int groups, int check_zy, const G1Affine *d_pts,
               int *d_err, int *d_ok, bool lone_call = false);
int verify_enqueue_stage1(kzg355_settings *s, Workspace *w, Timed &tm, const uint8_t *d_blobs, const uint8_t *d_c, const uint8_t *d_p, int npg, int G,
                          size_t res_cap = 0, HostFront *hf = nullptr, const std::function<int()> *after_challenge = nullptr);
int verify_enqueue_stage2(kzg355_settings *s, Workspace *w, Timed &tm, int npg, int G, size_t res_off = 0, bool lone_call = false);
int verify_enqueue(kzg355_settings *s, Workspace *w, Timed &tm, const uint8_t *d_blobs, const uint8_t *d_c, const uint8_t *d_p, int npg, int G,
                   size_t res_off = 0, size_t res_cap = 0, HostFront *hf = nullptr, bool lone_call = false);
int verify_collect(Workspace *w, Timed &tm, bool *ok, int *status, int G, size_t res_off = 0);
bool device_call_hashes_on_host(const kzg355_settings *s, size_t n_blobs);
int verify_many_device_impl(bool *ok, int *status, const uint8_t *d_blobs, const uint8_t *d_c, const uint8_t *d_p, size_t npg, size_t groups,
                            const kzg355_settings *cs);
int ensure_wide_table(kzg355_settings *s);
int msm_to_host(kzg355_settings *s, Workspace *w, Timed &tm, int n, const uint8_t *d_blobs, const Fr *d_scalars);
int prove_common(kzg355_settings *s, Workspace *w, Timed &tm, const uint8_t *d_blobs, int n);
// d_zs (device, n x 32 bytes, or null): proofs at these points instead of each blob's Fiat-Shamir challenge (compute_kzg_proof; d_c is then null), and
// the n values y = p(z) go back next to the proofs (msm_op_collect's ys_out)
int msm_op_enqueue(kzg355_settings *s, Workspace *w, Timed &tm, const uint8_t *d_blobs, const uint8_t *d_c, size_t n, HostFront *hf = nullptr,
                   const uint8_t *d_zs = nullptr);
int msm_op_collect(Workspace *w, Timed &tm, uint8_t *out, int *status, size_t n, uint8_t *ys_out = nullptr);
int msm_op_many_device_impl(uint8_t *out, int *status, const uint8_t *d_blobs, const uint8_t *d_c, size_t n, const kzg355_settings *cs,
                            const uint8_t *d_zs = nullptr, uint8_t *ys_out = nullptr);
int stage_to_device(Workspace *w, DevBuf &dst, const uint8_t *src, size_t bytes);
int stage_via_pinned(kzg355_settings *s, Workspace *w, PinBuf &pin, size_t pin_off, DevBuf &dst, const uint8_t *src, size_t bytes);
int host_pipeline(const HostCall &hc, size_t units, const kzg355_settings *cs);

}  // namespace kzg355_impl

// RCCL bound at run time, and the state of a handle over several devices (multi_device.hip)
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

namespace kzg355_impl {
kzg355_options options_of(const kzg355_options *opt);
std::vector<int> env_device_list();      // KZG355_DEVICES=0,1,...: the device list of kzg355_load_trusted_setup (options.hip)
int load_on_device(const uint8_t *g1_bytes, size_t n1, const uint8_t *g2_bytes, size_t n2, int dev_or_minus1, const kzg355_options &opt, kzg355_settings **out);
void free_single(kzg355_settings *s);
std::vector<kzg355_settings *> replicas_of(kzg355_settings *s);
int single_verify_many(bool *ok, int *status, const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs, size_t npg, size_t groups,
                       const kzg355_settings *cs);
int fan_out(size_t D, size_t units, const std::function<int(size_t, size_t, size_t)> &fn);
int multi_verify_sharded(bool *ok, int *status, const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs, size_t npg, size_t groups,
                         const kzg355_settings *cs, uint8_t *dump = nullptr);
int multi_verify_many(bool *ok, int *status, const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs, size_t npg, size_t groups,
                      const kzg355_settings *cs);

}  // namespace kzg355_impl
