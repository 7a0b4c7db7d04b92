// options.hip -- kzg355_options, loading a handle on one device, the load-time self-test, the file loader, getters and setters (host side of libkzg355.so; see
// engine.h).
#include "engine.h"

namespace kzg355_impl {

// the caller's struct may be older (smaller) than this library's: fields beyond its struct_size keep their defaults
kzg355_options options_of(const kzg355_options *opt) {
    kzg355_options o;
    kzg355_options_default(&o);
    if (opt && opt->struct_size >= sizeof(size_t)) {
        const size_t n = opt->struct_size < sizeof o ? opt->struct_size : sizeof o;
        memcpy(&o, opt, n);
        o.struct_size = sizeof o;
    }
    return o;
}

static int device_self_test(kzg355_settings *s);
int load_on_device(const uint8_t *g1_bytes, size_t n1, const uint8_t *g2_bytes, size_t n2, int dev_or_minus1, const kzg355_options &opt,
        kzg355_settings **out) {
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
    if (!small && (rc = s->eval_tab.ensure(sizeof(EvalPiece) * EVAL_TAB_PIECES))) return fail(rc);
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
    s->t.eval_tab = s->eval_tab.as<EvalPiece>();
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
        // transcript hash with a lane per batch from one wave per SIMD on (1024)
        s->rhash_lanes_from = opt.rhash_lanes_from > 0 ? opt.rhash_lanes_from : 4 * cus;
        s->beside_max_blobs = opt.beside_max_blobs > 0 ? opt.beside_max_blobs : 64 * cus;    // point kernels beside the hash chain up to 16384 blobs
        // two waves per pairing up to 256 batches
        s->pairing_two_wave_upto = opt.pairing_two_wave_upto < 0 ? 0 : opt.pairing_two_wave_upto > 0 ? opt.pairing_two_wave_upto : cus;
        if (opt.quotient_form == 2 || opt.quotient_form == 4 || opt.quotient_form == 6) s->quotient_form = opt.quotient_form;
        if (opt.miller_segments >= 1 && opt.miller_segments <= MILLER_SPLIT_MAX) s->miller_segments = opt.miller_segments;
        // hard part twelve lanes per check from 4096 batches on
        s->pairing_hard12_from = opt.pairing_hard12_from < 0 ? 0 : opt.pairing_hard12_from > 0 ? opt.pairing_hard12_from : 16 * cus;
        // two-wave hash while every wave has a SIMD to itself (512 workgroups of 64 blobs)
        s->challenge_two_wave_upto = 2 * cus * 64;
    }
    s->lane_pairing = opt.pairing_lane != 0;
    if (opt.split_parts >= 1 && opt.split_parts <= 64) s->split_parts = opt.split_parts;
    if (opt.split_streams >= 1 && opt.split_streams <= 8) s->split_streams = opt.split_streams;
    s->challenge_form = opt.challenge_form;
    s->lincomb_mode = opt.lincomb_form;
    s->force_sharded = opt.force_sharded != 0;
    // (the HIP runtime's own variable, not one of this library's: how many hardware queues the process's streams are multiplexed onto, engine.h)
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
    // known-answer self-test of the freshly built handle (MSM through the bucket form: the wide table has its own check when it is built)
    if (opt.self_test) {
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
// shifted and bucket forms), the challenge and evaluation kernels and the cooperative pairing on the device they will run on, and turns a toolchain that
// miscompiles one of them (DESIGN.md section 4
// records such a case with hipcc 7.2 and four inlined G1 routines) into a load error instead of wrong verdicts.
static int device_self_test(kzg355_settings *s) {
    static const uint8_t G1_GEN[48] = {0x97, 0xf1, 0xd3, 0xa7, 0x31, 0x97, 0xd7, 0x94, 0x26, 0x95, 0x63, 0x8c, 0x4f, 0xa9, 0xac, 0x0f, 0xc3, 0x68, 0x8c, 0x4f,
            0x97, 0x74, 0xb9, 0x05,
                                       0xa1, 0x4e, 0x3a, 0x3f, 0x17, 0x1b, 0xac, 0x58, 0x6c, 0x55, 0xe8, 0x3f, 0xf9, 0x7a, 0x1a, 0xef, 0xfb, 0x3a, 0xf0, 0x0a,
                                               0xdb, 0x22, 0xc6, 0xbb};
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
        bool copy_ok = hipMemcpy(cc.p, hc.data(), hc.size(), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(pp.p, hp.data(), hp.size(),
                hipMemcpyHostToDevice) == hipSuccess;
        for (size_t i = 0; i < n * G && copy_ok; i++) copy_ok = hipMemcpy(bb.as<uint8_t>() + BB * i, blobs.as<uint8_t>() + BB, BB,
                hipMemcpyDeviceToDevice) == hipSuccess;
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

// The debug switches and the device list of the plain load function: with kzg355_options_from_env above, every read of the environment the library makes.
bool debug_errors() { static const bool on = getenv("KZG355_DEBUG") != nullptr; return on; }
bool debug_pipe() { static const bool on = getenv("KZG355_DEBUG_PIPE") != nullptr; return on; }
std::vector<int> env_device_list() {
    std::vector<int> devs;
    if (const char *e = getenv("KZG355_DEVICES")) {
        for (const char *p = e; *p;) {
            char *end = nullptr;
            const long v = strtol(p, &end, 10);
            if (end == p) break;
            devs.push_back((int)v);
            p = *end == ',' ? end + 1 : end;
        }
    }
    return devs;
}

static int hexval(int ch) { return ch >= '0' && ch <= '9' ? ch - '0' : ch >= 'a' && ch <= 'f' ? ch - 'a' + 10 : ch >= 'A' && ch <= 'F' ? ch - 'A' + 10 : -1; }

void free_single(kzg355_settings *s) {
    if (!s) return;
    DeviceScope scope;
    (void)scope.enter(s->device);
    {
        // a caller's bug (kzg355.h: collect every ticket first).  The tickets hold this handle, its workspaces and streams: the free is DEFERRED to the
        // collect of the last of them (kzg355_verify_collect) -- nothing is leaked and no ticket dangles
        std::lock_guard<std::mutex> lk(s->pipe_mu);
        if (s->tickets_out.load() > 0) {
            fprintf(stderr, "kzg355: handle freed with %d submitted launch set(s) not collected: it is released when the last of them is collected\n",
                    s->tickets_out.load());
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

}  // namespace kzg355_impl

extern "C" {
#pragma GCC visibility push(default)

const char *kzg355_version(void) { return "kzg355 0.1 (gfx950, 29-bit-limb Montgomery, fixed-base Pippenger, precomputed-line pairing)"; }

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
    auto num = [](const char *name, long lo, long hi, int *dst) { if (const char *e = getenv(name)) { const long v = atol(e); if (v >= lo &&
            v <= hi) *dst = (int)v; } };
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
    if (const char *e = getenv("KZG355_LINCOMB")) o->lincomb_form = strcmp(e, "bucket") == 0 ? 2 : strcmp(e, "window") == 0 ? 1 : strcmp(e,
            "preshift") == 0 ? 3 : 0;
    if (const char *e = getenv("KZG355_EXCHANGE")) o->exchange = strcmp(e, "peer") == 0 ? 1 : strcmp(e, "rccl") == 0 ? 2 : 0;
    num("KZG355_VERIFY_ONLY", 0, 1, &o->verify_only);
    if (const char *e = getenv("KZG355_SUBMIT")) o->submit_sets = strcmp(e, "sets") == 0 ? 1 : strcmp(e, "pipeline") == 0 ? 2 : 0;
    num("KZG355_QUOTIENT_FORM", 2, 6, &o->quotient_form);
    num("KZG355_MILLER_SEGMENTS", 1, 8, &o->miller_segments);
    num("KZG355_FORCE_MULTI", 0, 1, &o->force_multi);
    num("KZG355_FORCE_SHARDED", 0, 1, &o->force_sharded);
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
    // kzg.rs:916-932
    if (!read_count(&n1) || (n1 != (size_t)N_FE && !(n1 >= (size_t)SMALL_N_MIN && n1 <= (size_t)SMALL_N_MAX && (n1 & (n1 - 1)) == 0))) { fclose(f);
            return KZG355_INVALID_TRUSTED_SETUP; }
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
    if (hipMemcpy(in.p, monomial_g1, 48 * n, hipMemcpyHostToDevice) != hipSuccess || hipMemset(err.p, 0,
            sizeof(int)) != hipSuccess) return done(KZG355_DEVICE_ERROR);
    launch_lagrange_from_monomial(in.as<uint8_t>(), (int)n, res.as<uint8_t>(), err.as<int>(), nullptr);
    int herr = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&herr, err.p, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return done(KZG355_DEVICE_ERROR);
    if (herr) return done(KZG355_BADARGS);
    if (hipMemcpy(out, res.p, 48 * n, hipMemcpyDeviceToHost) != hipSuccess) return done(KZG355_DEVICE_ERROR);
    return done(KZG355_OK);
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
    return s->wide_pub.load(std::memory_order_acquire) ? KZG355_OK : (s->msm_bits_wanted == 8 ? KZG355_OK : s->wide_rc == KZG355_OK ? KZG355_NO_MEMORY :
            s->wide_rc);
}
void kzg355_set_kernel_timing(kzg355_settings *s, int enabled) { if (s) s->timing = enabled != 0; }
long kzg355_settings_host_hashed_calls(const kzg355_settings *s) { return s ? s->n_host_hashed.load() : 0L; }
int kzg355_settings_host_threads(const kzg355_settings *s) { return !s ? 0 : s->host_pool ? s->host_pool->workers() + 1 : 1; }
int kzg355_settings_set_host_hash(kzg355_settings *s, int mode, int max_blobs) {
    if (!s || mode < -1 || mode > 1 || max_blobs < 0) return KZG355_BADARGS;
    // both Fiat-Shamir hashes follow the mode: -1 keeps the per-blob challenges AND the batch challenge r on the device
    for (kzg355_settings *r : s->multi ? replicas_of(s) : std::vector<kzg355_settings *>{s}) { r->host_hash = mode;
            r->host_rhash = mode < 0 ? -1 : r->host_rhash_loaded; if (max_blobs) r->host_hash_max = max_blobs; }
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

#pragma GCC visibility pop
}  // extern "C"
