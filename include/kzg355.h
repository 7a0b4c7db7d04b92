/*
 * kzg355.h -- C ABI of libkzg355.so, the MI355X-native (HIP / gfx950) KZG-4844 engine.
 *
 * This is the drop-in boundary for the hot path of pawanjay176/kzg_rust: every entry point below
 * replaces one associated function of `pub struct Kzg` (reference src/kzg.rs:983-1079) -- i.e. it sits
 * one level ABOVE the 35 blst `extern "C"` symbols the reference binds today,
 * because per-field-op FFI is too fine-grained for a GPU.  Bytes in, bytes out: no blst limb layouts,
 * no torch types, plain pointers and sizes.  INTEGRATION.md shows the Rust `extern "C"` block and the
 * `impl Kzg` forwards a maintainer would add.
 *
 * Status codes mirror `enum Error` (src/kzg.rs:10-22).  The reference's tests only distinguish
 * Ok / Err (src/lib.rs:47-50), so Ok-vs-Err and the returned bytes / bool are exact; the particular
 * non-zero code is best effort.  Outputs are written only on KZG355_OK.
 *
 * Ownership: the caller owns every buffer; the library copies in and retains nothing after return.
 * kzg355_settings is created by a load function, destroyed by kzg355_free_trusted_setup, immutable in
 * between, and may be used from several host threads at once (each call takes a private workspace +
 * HIP stream from a pool inside the handle) -- the reference takes `&KzgSettings` everywhere.
 *
 * There is NO CPU fallback: every function fails with KZG355_NO_DEVICE if no HIP device is usable.
 */
#ifndef KZG355_H
#define KZG355_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KZG355_BYTES_PER_FIELD_ELEMENT 32   /* src/consts.rs:5  */
#define KZG355_BYTES_PER_COMMITMENT 48      /* src/consts.rs:8  */
#define KZG355_BYTES_PER_PROOF 48           /* src/consts.rs:11 */
#define KZG355_FIELD_ELEMENTS_PER_BLOB 4096 /* src/consts.rs:13 */
#define KZG355_BYTES_PER_BLOB 131072        /* src/consts.rs:16 */
#define KZG355_BYTES_PER_G1 48              /* src/consts.rs:31 */
#define KZG355_BYTES_PER_G2 96              /* src/consts.rs:34 */
#define KZG355_NUM_G2_POINTS 65             /* src/consts.rs:37 */
#define KZG355_BYTES_PER_RECORD 160         /* C(48) | z(32) | y(32) | proof(48): one r-transcript record, utils.rs:454-463 */

enum {
    KZG355_OK = 0,
    KZG355_BADARGS = 1,               /* Error::BadArgs            kzg.rs:13 */
    KZG355_INTERNAL = 2,              /* Error::InternalError      kzg.rs:15 */
    KZG355_INVALID_BYTES_LENGTH = 3,  /* Error::InvalidBytesLength kzg.rs:17 */
    KZG355_INVALID_HEX = 4,           /* Error::InvalidHexFormat   kzg.rs:19 */
    KZG355_INVALID_TRUSTED_SETUP = 5, /* Error::InvalidTrustedSetup kzg.rs:21 */
    KZG355_NO_DEVICE = 6,             /* no usable HIP device (no reference counterpart; there is no CPU fallback) */
    KZG355_NO_MEMORY = 7,             /* a device or pinned-host allocation failed (no reference counterpart) */
    KZG355_DEVICE_ERROR = 8           /* a HIP runtime / RCCL call failed on a device that exists: a launch, copy, stream or collective error (no reference counterpart) */
};

typedef struct kzg355_settings kzg355_settings; /* opaque; replaces `KzgSettings` (kzg.rs:28-40) */

/* ---- trusted setup ---------------------------------------------------------------------------- */
/* Kzg::load_trusted_setup (kzg.rs:1005 -> 45-78 -> 833-899).  g1: n1*48 bytes, g2: n2*96 bytes, compressed,
 * Lagrange form, file order.  n1 not in {4096} + {4, 8, .., 64} or n2 != 65 -> INVALID_TRUSTED_SETUP; bad point / monomial form -> BADARGS.
 * Builds the device-resident tables (roots of unity, bit-reversed G1 table and its per-window multiples,
 * Miller-loop line tables of the two G2 points the verify path uses).  The wide-window MSM table of commitments / proofs (every multiple
 * 1..2^(c-1) of 2^(c w) * g1[i] over 128 bits: 10.9 .. 143.5 GB, kzg355_options.msm_bits) is built by the FIRST commitment / proof call,
 * sized from the HBM free at that moment -- a handle that only verifies never allocates it; if the allocation fails, or with
 * KZG355_MSM=bucket, only the 15 MB 8-bit table is kept and commitments / proofs take the bucket path (same results).  Takes the defaults of
 * kzg355_options below with the KZG355_* environment overrides applied; kzg355_load_trusted_setup_ex is the explicit form. */
int kzg355_load_trusted_setup(const uint8_t *g1_bytes, size_t n1, const uint8_t *g2_bytes, size_t n2, kzg355_settings **out);
/* The same over several GPUs of one node: a full replica of the tables per device, and the host-buffer entry points below spread
 * their work over the devices INSIDE the call, invisibly to the caller (the reference has no notion of devices): independent
 * batches / blobs go to the devices in contiguous ranges with no exchange; a call with fewer batches than devices (one 512-blob
 * batch on 8 GPUs) cuts every batch into per-device blocks of blobs, runs stage 1 per block, exchanges the 160-byte records with
 * ONE all-gather (RCCL ncclAllGather over xGMI on a communicator set kept in the handle; peer copies if RCCL is not available,
 * the device list has duplicates, the blocks are ragged, or KZG355_EXCHANGE=peer) and runs stage 2 of each batch on one device.
 * kzg355_load_trusted_setup does the same when KZG355_DEVICES=0,1,... is set.  The *_device entry points (device pointers) and
 * kzg355_settings_device refer to the first device of the list. */
int kzg355_load_trusted_setup_devices(const uint8_t *g1_bytes, size_t n1, const uint8_t *g2_bytes, size_t n2, const int *devices, size_t n_devices,
                                      kzg355_settings **out);
/* Everything a deployment may want to pin when a handle is created.  kzg355_options_default fills in the defaults (and struct_size, which
 * lets a library newer than the caller's header keep its own defaults for fields the caller does not know); 0 means "default" for every
 * numeric field unless stated otherwise.  The dispatch thresholds default to multiples of the device's compute-unit count -- the
 * measured crossovers of the 256-CU MI355X scaled to the device at hand.  The KZG355_* environment variables named
 * below are TEST / EXPERIMENT overrides: only kzg355_load_trusted_setup, _devices and _file read them (through
 * kzg355_options_from_env, the library's one reader of the environment beside the KZG355_DEBUG* switches and KZG355_DEVICES, all in csrc/options.hip);
 * kzg355_load_trusted_setup_ex takes what it is given and reads no environment. */
typedef struct kzg355_options {
    size_t struct_size;        /* sizeof(kzg355_options) as the caller compiled it */
    int device;                /* device ordinal; -1: the calling thread's current device                                   KZG355_DEVICE */
    int msm_bits;              /* fixed-base MSM table (built on the first commitment / proof call): 0 = the widest form whose table fits half of
                                  the HBM free at that moment; 12 / 13 / 15 / 16 = 10.9 / 20.1 / 68.9 / 143.5 GB at 22 / 20 / 18 / 16 table rows
                                  per scalar; 8 = the 15 MB 8-bit bucket form only                                KZG355_MSM_BITS, KZG355_MSM=bucket */
    int msm_require_wide;      /* 1: failing to allocate / build the wide table fails the load (else: bucket form, noted on stderr)  KZG355_MSM=wide */
    int self_test;             /* 1 (default): known-answer self-test of the new handle (~10 ms); 0: skip                   KZG355_SELFTEST */
    int host_threads;          /* host worker threads of the handle (Fiat-Shamir hashing of small host-buffer calls, staging copies);
                                  0 = min(16, cpus the process may run on / 2)                                             KZG355_HOST_THREADS */
    int host_hash;             /* challenges of host-buffer verify / blob-proof calls hashed on the host: 0 by size, 1 always, -1 never  KZG355_HOST_HASH=auto|on|off */
    int host_hash_max_blobs;   /* ... up to this many blobs per call (0 = 4096: measured crossover, profiles/r03/host_hash_crossover_v4.txt)          KZG355_HOST_HASH_MAX */
    int host_sha;              /* host SHA-256 form: 0 SHA extensions when the CPU has them, 1 portable C, 2 SHA extensions      KZG355_HOST_SHA=portable|shani */
    int host_rhash;            /* batch challenge r of lone small calls hashed on the host (records copied back, ~60 us instead of a 0.33 ms
                                  device chain): 0 by size, -1 never                                                       KZG355_HOST_RHASH=off */
    int host_rhash_max_records;/* ... up to this many records per call (0 = 1024: one 512-blob batch is a 2.6 ms chain on the device)  KZG355_HOST_RHASH_MAX */
    int challenge_form;        /* device Fiat-Shamir kernel: 0 by size (two-wave form up to 2 workgroups per CU), 1 one wave, 2 two waves   KZG355_CHALLENGE=1w|2w */
    int lincomb_form;          /* batch linear combination: 0 by size, 1 per-term windows, 2 buckets, 3 pre-shifted        KZG355_LINCOMB=window|bucket|preshift */
    int pairing_lane;          /* 1: one-lane pairing kernel (A/B and tests)                                               KZG355_PAIRING=lane */
    int pairing_two_wave_upto; /* batches per launch set up to which a pairing runs its two Miller loops on two waves; 0 = 1 per CU; -1 never  KZG355_PAIRING_2W_UPTO */
    int lc_chain_from;         /* batches from which the bucket form ends in one Horner chain per class; 0 = 4 per CU       KZG355_LC_CHAIN_FROM */
    int rhash_lanes_from;      /* batches from which the r-transcripts are hashed one lane per batch; 0 = 4 per CU          KZG355_RHASH_LANES_FROM */
    int beside_max_blobs;      /* blobs per launch set up to which the point kernels run on side streams; 0 = 64 per CU */
    int split_parts;           /* device-resident verify calls as this many overlapped launch sets (0 / 1: one set)         KZG355_SPLIT=parts[,streams] */
    int split_streams;         /* ... over this many streams (0 = 2) */
    int chunk_mb;              /* MiB of blobs per chunk of a streamed host-buffer call (0 = 1024)                          KZG355_CHUNK_MB */
    int chunks_in_flight;      /* workspaces such a call rotates over (0 = 3)                                               KZG355_CHUNKS_IN_FLIGHT */
    int staging_ring;          /* 1: stage caller memory through pinned slots instead of letting the runtime DMA from it    KZG355_STAGING=ring */
    int exchange;              /* handles over several devices: 0 RCCL all-gather when available, 1 peer copies, 2 RCCL or fail   KZG355_EXCHANGE=peer|rccl */
    int verify_only;           /* 1: the handle serves verification: no wide-window MSM table is built or kept in HBM (the verify path never reads it;
                                  commitments / proofs still work, through the 15 MB bucket form)                          KZG355_VERIFY_ONLY=1 */
    int msm_glv;               /* 0 (default): scalars are split k = a + b x^2 and the table spans 128 bits (half the memory per window width);
                                  -1: round 3's form, windows over all 256 bits (12 .. 15 bits: 23.6 / 42.9 / 81.6 / 154.6 GB)     KZG355_MSM_GLV=off */
    int msm_eager;             /* 1: build the table inside the load call instead of on first use                        KZG355_MSM_EAGER=1 */
    int pairing_hard12_from;   /* batches per launch set from which the hard part of the final exponentiation runs twelve lanes per check, five checks per
                                  wave (throughput) instead of one wave per check (latency); 0 = 16 per CU; -1 never                KZG355_PAIRING_HARD12_FROM */
    int submit_sets;           /* submit / collect: 0 by size (sets of <= 128 blobs per CU as 1, larger ones as 2); 1 = every submitted set on a stream of its
                                  own; 2 = two-stage pipeline: stage 1 of the submitted sets in order on one stream, stage 2 of a set on a second
                                  one, queued behind the NEXT set's Fiat-Shamir kernel                                   KZG355_SUBMIT=sets|pipeline */
    int host_hash_device_max_blobs; /* DEVICE-RESIDENT verify / blob-proof calls (the *_device entry points) of up to this many blobs copy their blobs back
                                  to the host (8 MiB = 0.16 ms per 64) and hash the challenges on the host threads instead of the 3.7 ms device
                                  chain: 0 = 1024 (measured crossover ~1500: profiles/r05/device_host_hash_crossover.txt), -1 never; host_hash = -1 turns it off as well                          KZG355_HOST_HASH_DEVICE_MAX */
    int quotient_form;         /* k_quotient_tree's leaves per lane as a power of two: 0 by size (2 below 512 blobs, else 4); 2 / 4 / 6 pin one (tuning knob
                                  and test hook)                                                                          KZG355_QUOTIENT_FORM */
    int miller_segments;       /* few batches: segments per Miller loop of the multi-wave pairing form, two waves each: 0 = 2; 1 .. 4   KZG355_MILLER_SEGMENTS */
    int force_multi;           /* test hook: a device list of ONE device still builds the multi-device handle (replica list, exchange)    KZG355_FORCE_MULTI=1 */
    int force_sharded;         /* test hook: a multi-device handle cuts EVERY batch into per-device blocks, whatever the batch count     KZG355_FORCE_SHARDED=1 */
} kzg355_options;
void kzg355_options_default(kzg355_options *options);
void kzg355_options_from_env(kzg355_options *options);     /* defaults, then the KZG355_* overrides listed above */
/* kzg355_load_trusted_setup with explicit options, on options->device or -- devices != NULL -- over the listed devices (as
 * kzg355_load_trusted_setup_devices).  options == NULL: all defaults.  Reads no environment variable. */
int kzg355_load_trusted_setup_ex(const uint8_t *g1_bytes, size_t n1, const uint8_t *g2_bytes, size_t n2, const int *devices, size_t n_devices,
                                 const kzg355_options *options, kzg355_settings **out);
/* Kzg::load_trusted_setup_file (kzg.rs:995 -> 906-979): "4096\n65\n" + hex lines. */
int kzg355_load_trusted_setup_file(const char *path, kzg355_settings **out);
/* Minimal preset helper: n compressed MONOMIAL points [tau^k]G1 (n a power of two in [4, 64], e.g. the first four `setup_G1`
 * entries of a ceremony JSON) -> the n compressed points of the size-n LAGRANGE setup, L_j = (1/n) sum_k w^(-jk) [tau^k]G1, in
 * file (natural) order, ready for kzg355_load_trusted_setup.  (src/trusted_setup.rs:144-151 truncates the 4096-point Lagrange
 * setup instead; that loads but is not a basis of the smaller domain.)  Bad point -> BADARGS. */
int kzg355_lagrange_setup_from_monomial(uint8_t *out /* n*48 */, const uint8_t *monomial_g1 /* n*48 */, size_t n);
/* Drop for KzgSettings. */
void kzg355_free_trusted_setup(kzg355_settings *s);

/* ---- the reference's seven operations, host buffers in / out ---------------------------------- */
/* Kzg::blob_to_kzg_commitment (kzg.rs:1013 -> 401-406). blob: 131072 bytes. */
int kzg355_blob_to_kzg_commitment(uint8_t out[48], const uint8_t *blob, const kzg355_settings *s);
/* Kzg::compute_kzg_proof (kzg.rs:1021 -> 446-457): returns proof and y = p(z). */
int kzg355_compute_kzg_proof(uint8_t proof_out[48], uint8_t y_out[32], const uint8_t *blob, const uint8_t z_bytes[32], const kzg355_settings *s);
/* Kzg::compute_blob_kzg_proof (kzg.rs:1030 -> 533-544). */
int kzg355_compute_blob_kzg_proof(uint8_t proof_out[48], const uint8_t *blob, const uint8_t commitment[48], const kzg355_settings *s);
/* Kzg::verify_kzg_proof (kzg.rs:1039 -> 429-443). */
int kzg355_verify_kzg_proof(bool *ok, const uint8_t commitment[48], const uint8_t z_bytes[32], const uint8_t y_bytes[32], const uint8_t proof[48], const kzg355_settings *s);
/* Kzg::verify_blob_kzg_proof (kzg.rs:1050 -> 547-569). */
int kzg355_verify_blob_kzg_proof(bool *ok, const uint8_t *blob, const uint8_t commitment[48], const uint8_t proof[48], const kzg355_settings *s);
/* Kzg::verify_blob_kzg_proof_batch (kzg.rs:1066 -> 637-693).  The Rust slices `&[Blob]`, `&[KzgCommitment]`,
 * `&[KzgProof]` carry their own lengths; the shim passes all three so that the reference's length check
 * (kzg.rs:644-651 -> BadArgs) is made on this side of the boundary too.  blobs: n_blobs*131072 contiguous bytes
 * (the shim gathers the `Box`ed blobs into one staging buffer).  n == 0 -> ok = true (kzg.rs:653-655). */
int kzg355_verify_blob_kzg_proof_batch(bool *ok, const uint8_t *blobs, size_t n_blobs, const uint8_t *commitments, size_t n_commitments,
                                       const uint8_t *proofs, size_t n_proofs, const kzg355_settings *s);

/* ---- throughput extensions (same semantics, many independent units per call) ------------------ */
/* n independent blob_to_kzg_commitment calls.  status[i] per blob (may be NULL); returns first non-OK status. */
int kzg355_blob_to_kzg_commitment_many(uint8_t *out /* n*48 */, int *status /* n or NULL */, const uint8_t *blobs, size_t n, const kzg355_settings *s);
/* n independent compute_blob_kzg_proof calls. */
int kzg355_compute_blob_kzg_proof_many(uint8_t *out /* n*48 */, int *status, const uint8_t *blobs, const uint8_t *commitments, size_t n, const kzg355_settings *s);
/* `groups` independent verify_blob_kzg_proof_batch calls of n_per_group blobs each, executed by ONE set of kernel
 * launches.  ok[g] / status[g] are per group.  Inputs are group-major contiguous. */
int kzg355_verify_blob_kzg_proof_batch_many(bool *ok /* groups */, int *status /* groups */, const uint8_t *blobs, const uint8_t *commitments,
                                            const uint8_t *proofs, size_t n_per_group, size_t groups, const kzg355_settings *s);

/* n independent verify_kzg_proof calls (kzg.rs:1039 -> 429-443; benches/kzg_benches.rs:70-81 times one per call): ok[i] / status[i] per check
 * (status may be NULL), the inputs of check i at commitments + 48 i, zs + 32 i, ys + 32 i, proofs + 48 i.  A bad point or a non-canonical z / y is
 * that check's KZG355_BADARGS; the others are unaffected.  Returns the first non-OK status.  One set of kernel launches per 2^17 checks: four
 * scalar-multiplication ladder lanes and twelve pairing lanes per check instead of a 1.9 ms latency-bound chain per call. */
int kzg355_verify_kzg_proof_many(bool *ok /* n */, int *status /* n or NULL */, const uint8_t *commitments /* n*48 */, const uint8_t *zs /* n*32 */,
                                 const uint8_t *ys /* n*32 */, const uint8_t *proofs /* n*48 */, size_t n, const kzg355_settings *s);
/* n independent verify_blob_kzg_proof calls (kzg.rs:1050 -> 547-569): = kzg355_verify_blob_kzg_proof_batch_many with one blob per batch (the
 * batch equation with r^0 = 1 is the single check, kzg.rs:658-660). */
int kzg355_verify_blob_kzg_proof_many(bool *ok /* n */, int *status /* n or NULL */, const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs,
                                      size_t n, const kzg355_settings *s);
/* n independent compute_kzg_proof calls (kzg.rs:1021 -> 446-457): proof i and y_i = p_i(z_i) for blob i at the caller's point zs + 32 i.  A
 * non-canonical z_i or blob is that unit's KZG355_BADARGS; outputs of a unit are written only on its KZG355_OK. */
int kzg355_compute_kzg_proof_many(uint8_t *proofs_out /* n*48 */, uint8_t *ys_out /* n*32 */, int *status /* n or NULL */, const uint8_t *blobs,
                                  const uint8_t *zs /* n*32 */, size_t n, const kzg355_settings *s);

/* ---- device-resident inputs (what bench.py times: blobs already in HBM) ----------------------- */
/* As above, but d_* are DEVICE pointers (hipMalloc / torch tensors .data_ptr()) on the settings' device.
 * ok / status are host pointers.  Synchronous: returns when the verdicts are on the host. */
int kzg355_verify_blob_kzg_proof_batch_many_device(bool *ok, int *status, const uint8_t *d_blobs, const uint8_t *d_commitments,
                                                   const uint8_t *d_proofs, size_t n_per_group, size_t groups, const kzg355_settings *s);
/* The same call in two halves, for a caller that keeps several launch sets in flight from ONE host thread (a set of 1024 batches is 8.6 GB of
 * blobs; the synchronous call wants ~8192 batches = 69 GB per call to reach the same rate): _submit queues the whole set on a private stream of
 * the handle and returns at once with a ticket; kzg355_verify_collect waits for that set, writes ok[g] / status[g] (host, `groups` entries as
 * submitted) and CONSUMES the ticket, whatever it returns.  Stage 2 of a submitted set (r powers, linear combination, pairing) runs beside the
 * evaluation and point kernels of the set submitted after it; keep THREE sets in flight (submit k + 2 before collecting k) and 1024-batch sets
 * run at 4.0 M blobs/s against 3.4 M one at a time (profiles/r04/pipeline_sweep.txt).  Collect in any order; every ticket must be collected before the handle is freed; the
 * device buffers of a set must stay untouched until its collect returns.  Returns as the synchronous call (first non-OK status at collect). */
typedef struct kzg355_ticket kzg355_ticket;
int kzg355_verify_blob_kzg_proof_batch_many_device_submit(kzg355_ticket **ticket, const uint8_t *d_blobs, const uint8_t *d_commitments,
                                                          const uint8_t *d_proofs, size_t n_per_group, size_t groups, const kzg355_settings *s);
int kzg355_verify_collect(kzg355_ticket *ticket, bool *ok /* groups */, int *status /* groups or NULL */);
int kzg355_blob_to_kzg_commitment_many_device(uint8_t *out /* host n*48 */, int *status /* host n or NULL */, const uint8_t *d_blobs, size_t n,
                                              const kzg355_settings *s);
int kzg355_compute_blob_kzg_proof_many_device(uint8_t *out /* host n*48 */, int *status, const uint8_t *d_blobs, const uint8_t *d_commitments,
                                              size_t n, const kzg355_settings *s);

/* The *_many forms of the single-proof functions on device memory: d_records = n records C_i | z_i | y_i | proof_i of KZG355_BYTES_PER_RECORD bytes
 * (16-byte aligned; = kzg355_verify_records_checked_device with one record per batch); d_blobs / d_zs as the host forms lay them out.  (n independent
 * verify_blob_kzg_proof checks on device memory are kzg355_verify_blob_kzg_proof_batch_many_device with n_per_group = 1.) */
int kzg355_verify_kzg_proof_many_device(bool *ok /* host n */, int *status /* host n or NULL */, const uint8_t *d_records, size_t n, const kzg355_settings *s);
int kzg355_compute_kzg_proof_many_device(uint8_t *proofs_out /* host n*48 */, uint8_t *ys_out /* host n*32 */, int *status, const uint8_t *d_blobs,
                                         const uint8_t *d_zs /* n*32 */, size_t n, const kzg355_settings *s);

/* ---- sharded verification (one process per GPU; the exchange between the two stages is the caller's
 *      exchange of the 160-byte records, e.g. torch.distributed over RCCL) --------------------------- */
/* Stage 1, per rank, over its contiguous shard of every batch: `groups` batches, n_local blobs of each (group-major:
 * blob (g, i) at index g*n_local + i).  Validates C_i / proof_i, blob -> field elements, Fiat-Shamir challenge z_i
 * (kzg.rs:298-339), y_i = p_i(z_i) (kzg.rs:346-389).  Writes the records C_i|z_i|y_i|proof_i -- byte for byte the body of
 * the r-transcript (utils.rs:454-463) -- to d_records (device, same indexing), and per-batch status (KZG355_OK or
 * KZG355_BADARGS) to status[g] (host).  Returns the first non-OK status.  Device pointers: d_blobs and d_records 16-byte aligned
 * (hipMalloc and torch allocations are), else KZG355_BADARGS. */
int kzg355_verify_shard_records_device(uint8_t *d_records /* groups*n_local*160, device */, int *status /* groups, host */,
                                       const uint8_t *d_blobs, const uint8_t *d_commitments, const uint8_t *d_proofs, size_t n_local,
                                       size_t groups, const kzg355_settings *s);
/* Stage 2 over gathered records (device; every rank its share of the batches, or replicated): `groups` batches of n records each
 * (group-major).  r-powers (utils.rs:426-474), the three linear combinations and the pairing check (kzg.rs:579-627).
 * n == 1 reproduces the single-blob path (kzg.rs:658); n == 0 is an error like kzg.rs:588-592. */
/* The same two stages with the decoded points travelling next to the records: stage 1 also writes the validated affine points of its
 * shard (d_points: groups x [n_local commitments, n_local proofs] x KZG355_BYTES_PER_POINT opaque bytes), and stage 2 takes the points
 * of the gathered batches (groups x [n commitments, n proofs]) instead of decompressing C_i / proof_i again (a 381-bit square root
 * per point).  The caller exchanges both buffers (kzg_rust_amd/sharded.py: one all-to-all, every rank receives the batches of its share). */
#define KZG355_BYTES_PER_POINT 112
int kzg355_verify_shard_records_points_device(uint8_t *d_records, uint8_t *d_points, int *status, const uint8_t *d_blobs, const uint8_t *d_commitments,
                                              const uint8_t *d_proofs, size_t n_local, size_t groups, const kzg355_settings *s);
int kzg355_verify_records_points_device(bool *ok, int *status, const uint8_t *d_records, const uint8_t *d_points, size_t n, size_t groups,
                                        const kzg355_settings *s);
/* The same two calls with the per-batch results left ON THE DEVICE (int32 words, device memory, 4-byte aligned), for a caller that merges them
 * across ranks with a collective and reads them back once (kzg_rust_amd/sharded.py): stage 1 writes d_status_words[g] = the KZG355 status of
 * batch g on this rank's shard; stage 2 writes d_words[g] = 1 + ok + 256 * status.  Both return after their stream has been synchronised (the
 * words may be read from any stream), and return only whole-call failures (KZG355_BADARGS for bad pointers, device errors). */
int kzg355_verify_shard_records_points_words_device(uint8_t *d_records, uint8_t *d_points, int32_t *d_status_words, const uint8_t *d_blobs,
                                                    const uint8_t *d_commitments, const uint8_t *d_proofs, size_t n_local, size_t groups, const kzg355_settings *s);
int kzg355_verify_records_points_words_device(int32_t *d_words, const uint8_t *d_records, const uint8_t *d_points, size_t n, size_t groups,
                                              const kzg355_settings *s);
/* PRECONDITION: the records come from kzg355_verify_shard_records_device (here or on another rank) and every rank's stage-1
 * status has been merged into the verdict by the caller (kzg_rust_amd/sharded.py does): this entry point decompresses C_i and
 * proof_i WITHOUT the subgroup test and does not re-check that z_i, y_i are canonical -- stage 1 already did both.  A caller
 * holding records of unknown origin must use kzg355_verify_records_checked_device instead. */
int kzg355_verify_records_device(bool *ok /* groups */, int *status /* groups */, const uint8_t *d_records /* groups*n*160, device */,
                                 size_t n, size_t groups, const kzg355_settings *s);
/* The same stage 2 with full input validation: validate_kzg_g1 (utils.rs:282-310, incl. subgroup) on every C_i / proof_i and
 * bytes_to_bls_field (utils.rs:262-275) on every z_i / y_i; = verify_kzg_proof_batch (kzg.rs:579-627) on untrusted bytes. */
int kzg355_verify_records_checked_device(bool *ok /* groups */, int *status /* groups */, const uint8_t *d_records, size_t n, size_t groups,
                                         const kzg355_settings *s);
/* Test / audit readback of the stage-2 intermediates of `groups` batches of n records: out[128 g ..] = r (32 bytes big-endian, the
 * Fiat-Shamir batch challenge of utils.rs:426-474) | proof_lincomb (48, kzg.rs:601) | rhs (48, kzg.rs:618-622), the two points
 * ZCash-compressed.  For n == 1 the r field reads 1 (r^0: the single-blob path takes no challenge).  Also returns the verdicts.
 * Not on the hot path; tests/test_gpu_parity.py diffs these against the oracle and the committed n = 64 / 512 fixtures. */
int kzg355_debug_batch_intermediates(uint8_t *out /* groups*128, host */, bool *ok /* groups */, int *status /* groups or NULL */,
                                     const uint8_t *d_records /* device */, size_t n, size_t groups, const kzg355_settings *s);

/* ---- introspection ---------------------------------------------------------------------------- */
/* Device ordinal the handle lives on, and average duration in milliseconds of the most recent launch of a named
 * kernel family on that handle ("verify_eval", "msm_bucket", ...), measured with HIP events on the launch stream;
 * returns a negative number if that kernel has not run.  Used by bench.py for the roofline line. */
int kzg355_settings_device(const kzg355_settings *s);
/* Number of devices the handle spans (1 for a plain handle). */
int kzg355_settings_device_count(const kzg355_settings *s);
/* Multi-device handles: how many record exchanges ran as an RCCL all-gather / as peer copies so far.  Returns the exchange the
 * handle is set up for (1 RCCL, 0 peer copies; -1 for a plain handle). */
int kzg355_settings_exchange_stats(const kzg355_settings *s, long *allgathers, long *peer_exchanges);
/* FIELD_ELEMENTS_PER_BLOB of the handle (consts.rs:13 is a compile-time 4096; the reference's README also names a minimal preset
 * with 4).  It is fixed by the number of G1 points given to the load function: 4096 -> the mainnet kernels; a power of two in
 * [4, 64] -> the small-domain path (one lane per blob, naive lincomb as utils.rs:369-371 takes below 8 points).  Every `blob`
 * argument of this header is 32 * FIELD_ELEMENTS_PER_BLOB bytes for the handle it is passed with. */
int kzg355_settings_field_elements_per_blob(const kzg355_settings *s);
/* Which MSM form commitments / proofs take on this handle: 10 .. 16 = wide-window table of that digit width; 8 = the 8-bit
 * bucket form because msm_bits = 8 / verify_only / KZG355_MSM=bucket asked for it; -8 = the bucket form because the wide table could NOT be
 * allocated or built (also reported once on stderr; msm_require_wide / KZG355_MSM=wide makes that an error instead).  Before the table
 * exists (it is built by the first commitment / proof call unless msm_eager): the width asked for, 0 = to be sized from the free HBM. */
int kzg355_settings_msm_form(const kzg355_settings *s);
/* The table as built: digit width, windows per (half-)scalar, 1 if the scalars are GLV-split, bytes of HBM; all 0 while it does not exist.
 * Any pointer may be NULL. */
int kzg355_settings_msm_shape(const kzg355_settings *s, int *bits, int *windows, int *glv, size_t *table_bytes);
/* Build the table now (every device of the handle) instead of inside the first commitment / proof call.  OK if it exists afterwards or
 * the handle is set to the bucket form. */
int kzg355_settings_build_msm_table(const kzg355_settings *s);
double kzg355_last_kernel_ms(const kzg355_settings *s, const char *kernel_family);
/* Accumulated HIP-event time and launch count of a kernel family since timing was enabled / last reset. 0 on success. */
int kzg355_kernel_ms_stats(const kzg355_settings *s, const char *kernel_family, double *total_ms, long *launches);
void kzg355_reset_kernel_stats(kzg355_settings *s);
/* Enable (1) / disable (0) per-kernel HIP-event timing on the handle (off by default: it adds event records). */
void kzg355_set_kernel_timing(kzg355_settings *s, int enabled);
const char *kzg355_version(void);

/* ---- host-side Fiat-Shamir hashing ------------------------------------------------ */
/* compute_challenge (kzg.rs:298-339) hashes 131,152 bytes per blob: on the device a 3.7 ms dependent chain for a small call, on a
 * host core with the SHA extensions ~60 us.  Host-buffer verify / blob-proof calls of at most `max_blobs` blobs (one chunk) therefore
 * hash their transcripts on the handle's host threads WHILE the H2D copy and the point kernels run and upload 32-byte digests; larger
 * and device-resident calls keep the device kernels.  mode: 0 by size (default), 1 always, -1 never (the batch challenge r of lone small
 * calls -- kzg355_options.host_rhash -- follows the same mode: -1 keeps every hash on the device); max_blobs 0 keeps the current
 * crossover (default 4096; KZG355_HOST_HASH=auto|on|off and KZG355_HOST_HASH_MAX in the environment set the same at load). */
int kzg355_settings_set_host_hash(kzg355_settings *s, int mode, int max_blobs);
/* How many host-buffer calls on this handle had their challenges hashed on the host so far. */
long kzg355_settings_host_hashed_calls(const kzg355_settings *s);
/* Host threads that hash for one call on this handle: the handle's workers plus the calling thread (kzg355_options.host_threads; the default is
 * min(16, cpus / 2)).  A CPU figure quoted next to a host-hashed call should use as many threads (bench.py: cpu_baseline.threads_matched_value). */
int kzg355_settings_host_threads(const kzg355_settings *s);
/* The host hash itself, exported for the tests (tests/test_host_sha256.py checks it against hashlib): impl 0 auto, 1 portable C,
 * 2 SHA extensions (KZG355_INTERNAL if the CPU has none).  kzg355_host_challenge_digests writes the n digests of
 * "FSBLOBVERIFY_V1_" | u64be(0) | u64be(blob_bytes / 32) | blob_i | commitment_i. */
/* Test / audit form of kzg355_verify_blob_kzg_proof_batch_many for calls of at most 64 MiB of blobs on a single-device handle: also
 * returns the stage-1 records C_i | z_i | y_i | proof_i of every blob (the body of the r-transcript, utils.rs:454-463), so that the z_i
 * of the host-hashed and the device-hashed route can each be diffed against the oracle (tests/test_gpu_host_hash.py). */
int kzg355_debug_verify_host_records(uint8_t *records_out /* groups*n_per_group*160, host */, bool *ok /* groups */, int *status /* groups or NULL */,
                                     const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs, size_t n_per_group, size_t groups,
                                     const kzg355_settings *s);
/* Test / audit form of the SHARDED execution of kzg355_verify_blob_kzg_proof_batch_many on a handle over several devices (BASELINE config 5: one
 * 512-blob batch cut into per-device blocks): every batch goes through stage 1 per block -> the record exchange -> stage 2 on one
 * device, whatever the batch count, and out[128 g ..] receives r | proof_lincomb | rhs of batch g as kzg355_debug_batch_intermediates lays them
 * out.  n_per_group >= the handle's device count; plain handles -> KZG355_BADARGS.  tests/test_gpu_multi_device.py diffs these against
 * tests/golden/batch512.json. */
int kzg355_debug_verify_sharded_intermediates(uint8_t *out /* groups*128, host */, bool *ok /* groups */, int *status /* groups or NULL */,
                                              const uint8_t *blobs, const uint8_t *commitments, const uint8_t *proofs, size_t n_per_group, size_t groups,
                                              const kzg355_settings *s);
int kzg355_host_sha256(uint8_t out[32], const uint8_t *msg, size_t len, int impl);
int kzg355_host_challenge_digests(uint8_t *out /* n*32 */, const uint8_t *blobs, size_t blob_bytes, const uint8_t *commitments /* n*48 */, size_t n, int impl);

#ifdef __cplusplus
}
#endif
#endif /* KZG355_H */
