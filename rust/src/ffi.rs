//! `extern "C"` block for include/kzg355.h -- the seam that replaces the 35 blst symbols the reference binds
//!.  One declaration per entry point of the header; bytes in, bytes out.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_long};

#[repr(C)]
pub struct kzg355_settings {
    _private: [u8; 0],
}

pub const KZG355_OK: c_int = 0;
pub const KZG355_BADARGS: c_int = 1;
pub const KZG355_INTERNAL: c_int = 2;
pub const KZG355_INVALID_BYTES_LENGTH: c_int = 3;
pub const KZG355_INVALID_HEX: c_int = 4;
pub const KZG355_INVALID_TRUSTED_SETUP: c_int = 5;
pub const KZG355_NO_DEVICE: c_int = 6;
pub const KZG355_NO_MEMORY: c_int = 7;
pub const KZG355_DEVICE_ERROR: c_int = 8;

/// `struct kzg355_options` (include/kzg355.h): same fields, same order.  Fill with `kzg355_options_default` first.
#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct kzg355_options {
    pub struct_size: usize,
    pub device: c_int,
    pub msm_bits: c_int,
    pub msm_require_wide: c_int,
    pub self_test: c_int,
    pub host_threads: c_int,
    pub host_hash: c_int,
    pub host_hash_max_blobs: c_int,
    pub host_sha: c_int,
    pub host_rhash: c_int,
    pub host_rhash_max_records: c_int,
    pub challenge_form: c_int,
    pub lincomb_form: c_int,
    pub pairing_lane: c_int,
    pub pairing_two_wave_upto: c_int,
    pub lc_chain_from: c_int,
    pub rhash_lanes_from: c_int,
    pub beside_max_blobs: c_int,
    pub split_parts: c_int,
    pub split_streams: c_int,
    pub chunk_mb: c_int,
    pub chunks_in_flight: c_int,
    pub staging_ring: c_int,
    pub exchange: c_int,
    pub verify_only: c_int,
    pub msm_glv: c_int,
    pub msm_eager: c_int,
    pub pairing_hard12_from: c_int,
    pub submit_sets: c_int,
    pub host_hash_device_max_blobs: c_int,
    pub quotient_form: c_int,
    pub miller_segments: c_int,
    pub force_multi: c_int,
    pub force_sharded: c_int,
}

extern "C" {
    pub fn kzg355_load_trusted_setup(g1: *const u8, n1: usize, g2: *const u8, n2: usize, out: *mut *mut kzg355_settings) -> c_int;
    pub fn kzg355_load_trusted_setup_devices(g1: *const u8, n1: usize, g2: *const u8, n2: usize, devices: *const c_int, n_devices: usize,
                                             out: *mut *mut kzg355_settings) -> c_int;
    pub fn kzg355_options_default(options: *mut kzg355_options);
    pub fn kzg355_options_from_env(options: *mut kzg355_options);
    pub fn kzg355_load_trusted_setup_ex(g1: *const u8, n1: usize, g2: *const u8, n2: usize, devices: *const c_int, n_devices: usize,
                                        options: *const kzg355_options, out: *mut *mut kzg355_settings) -> c_int;
    pub fn kzg355_load_trusted_setup_file(path: *const c_char, out: *mut *mut kzg355_settings) -> c_int;
    pub fn kzg355_lagrange_setup_from_monomial(out: *mut u8, monomial_g1: *const u8, n: usize) -> c_int;
    pub fn kzg355_free_trusted_setup(s: *mut kzg355_settings);

    pub fn kzg355_blob_to_kzg_commitment(out: *mut u8, blob: *const u8, s: *const kzg355_settings) -> c_int;
    pub fn kzg355_compute_kzg_proof(proof: *mut u8, y: *mut u8, blob: *const u8, z: *const u8, s: *const kzg355_settings) -> c_int;
    pub fn kzg355_compute_blob_kzg_proof(proof: *mut u8, blob: *const u8, commitment: *const u8, s: *const kzg355_settings) -> c_int;
    pub fn kzg355_verify_kzg_proof(ok: *mut bool, commitment: *const u8, z: *const u8, y: *const u8, proof: *const u8, s: *const kzg355_settings) -> c_int;
    pub fn kzg355_verify_blob_kzg_proof(ok: *mut bool, blob: *const u8, commitment: *const u8, proof: *const u8, s: *const kzg355_settings) -> c_int;
    pub fn kzg355_verify_blob_kzg_proof_batch(ok: *mut bool, blobs: *const u8, n_blobs: usize, commitments: *const u8, n_commitments: usize,
                                              proofs: *const u8, n_proofs: usize, s: *const kzg355_settings) -> c_int;

    pub fn kzg355_blob_to_kzg_commitment_many(out: *mut u8, status: *mut c_int, blobs: *const u8, n: usize, s: *const kzg355_settings) -> c_int;
    pub fn kzg355_compute_blob_kzg_proof_many(out: *mut u8, status: *mut c_int, blobs: *const u8, commitments: *const u8, n: usize,
                                              s: *const kzg355_settings) -> c_int;
    pub fn kzg355_verify_blob_kzg_proof_batch_many(ok: *mut bool, status: *mut c_int, blobs: *const u8, commitments: *const u8, proofs: *const u8,
                                                   n_per_group: usize, groups: usize, s: *const kzg355_settings) -> c_int;
    pub fn kzg355_verify_kzg_proof_many(ok: *mut bool, status: *mut c_int, commitments: *const u8, zs: *const u8, ys: *const u8, proofs: *const u8, n: usize,
                                        s: *const kzg355_settings) -> c_int;
    pub fn kzg355_verify_blob_kzg_proof_many(ok: *mut bool, status: *mut c_int, blobs: *const u8, commitments: *const u8, proofs: *const u8, n: usize,
                                             s: *const kzg355_settings) -> c_int;
    pub fn kzg355_compute_kzg_proof_many(proofs_out: *mut u8, ys_out: *mut u8, status: *mut c_int, blobs: *const u8, zs: *const u8, n: usize,
                                         s: *const kzg355_settings) -> c_int;

    pub fn kzg355_settings_device(s: *const kzg355_settings) -> c_int;
    pub fn kzg355_settings_device_count(s: *const kzg355_settings) -> c_int;
    pub fn kzg355_settings_field_elements_per_blob(s: *const kzg355_settings) -> c_int;
    pub fn kzg355_settings_msm_form(s: *const kzg355_settings) -> c_int;
    pub fn kzg355_settings_msm_shape(s: *const kzg355_settings, bits: *mut c_int, windows: *mut c_int, glv: *mut c_int, table_bytes: *mut usize) -> c_int;
    pub fn kzg355_settings_build_msm_table(s: *const kzg355_settings) -> c_int;
    pub fn kzg355_settings_exchange_stats(s: *const kzg355_settings, allgathers: *mut c_long, peer_exchanges: *mut c_long) -> c_int;
    pub fn kzg355_settings_set_host_hash(s: *mut kzg355_settings, mode: c_int, max_blobs: c_int) -> c_int;
    pub fn kzg355_settings_host_hashed_calls(s: *const kzg355_settings) -> c_long;
    pub fn kzg355_settings_host_threads(s: *const kzg355_settings) -> c_int;
    pub fn kzg355_version() -> *const c_char;
}
