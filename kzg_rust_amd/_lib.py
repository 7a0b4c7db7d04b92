"""Loader for libkzg355.so (the HIP engine).  There is deliberately no fallback: if the shared library
is missing or cannot be loaded, importing fails loudly -- the product path never routes through a CPU
implementation."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (KZG355_LIBRARY: another build of the same library, for A/B measurements of compiler flags on one box -- tools only; there is still no fallback)
LIB_PATH = os.environ.get("KZG355_LIBRARY") or os.path.join(_HERE, "libkzg355.so")


class KzgLibraryMissing(ImportError):
    pass


class Options(C.Structure):
    """struct kzg355_options (include/kzg355.h): the same fields in the same order."""
    _fields_ = [("struct_size", C.c_size_t)] + [(name, C.c_int) for name in (
        "device", "msm_bits", "msm_require_wide", "self_test", "host_threads", "host_hash", "host_hash_max_blobs", "host_sha", "host_rhash", "host_rhash_max_records", "challenge_form",
        "lincomb_form", "pairing_lane", "pairing_two_wave_upto", "lc_chain_from", "rhash_lanes_from", "beside_max_blobs", "split_parts",
        "split_streams", "chunk_mb", "chunks_in_flight", "staging_ring", "exchange", "verify_only", "msm_glv", "msm_eager", "pairing_hard12_from", "submit_sets", "host_hash_device_max_blobs", "quotient_form",
        "miller_segments", "force_multi", "force_sharded")]


def load():
    if not os.path.exists(LIB_PATH):
        raise KzgLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C kzg_rust_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 (same SONAMEs as
    # /opt/rocm).  If torch is going to be used in this process (device tensors handed to the *_device entry points,
    # torch.distributed for the sharded path) it must be imported BEFORE libkzg355.so so that both bind to the same
    # runtime; loading ours first leaves torch unable to see the GPU.  Pure C / Rust consumers simply get /opt/rocm.
    # (24 hardware queues instead of the HIP runtime's 4: concurrent small calls are chains on several streams each, csrc/engine.h.  The variable
    # belongs to the process, so it is set HERE, before the first HIP call -- the library itself no longer touches the environment)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
    if os.environ.get("KZG355_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise KzgLibraryMissing(f"cannot load {LIB_PATH}: {e}") from e
    vp, sz, u8p, ip = C.c_void_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_int)
    bp = C.POINTER(C.c_bool)
    sigs = {
        "kzg355_load_trusted_setup": [u8p, sz, u8p, sz, C.POINTER(vp)],
        "kzg355_load_trusted_setup_file": [u8p, C.POINTER(vp)],
        "kzg355_blob_to_kzg_commitment": [u8p, u8p, vp],
        "kzg355_compute_kzg_proof": [u8p, u8p, u8p, u8p, vp],
        "kzg355_compute_blob_kzg_proof": [u8p, u8p, u8p, vp],
        "kzg355_verify_kzg_proof": [bp, u8p, u8p, u8p, u8p, vp],
        "kzg355_verify_blob_kzg_proof": [bp, u8p, u8p, u8p, vp],
        "kzg355_verify_blob_kzg_proof_batch": [bp, u8p, sz, u8p, sz, u8p, sz, vp],
        "kzg355_blob_to_kzg_commitment_many": [u8p, ip, u8p, sz, vp],
        "kzg355_compute_blob_kzg_proof_many": [u8p, ip, u8p, u8p, sz, vp],
        "kzg355_verify_blob_kzg_proof_batch_many": [bp, ip, u8p, u8p, u8p, sz, sz, vp],
        "kzg355_verify_kzg_proof_many": [bp, ip, u8p, u8p, u8p, u8p, sz, vp],
        "kzg355_verify_blob_kzg_proof_many": [bp, ip, u8p, u8p, u8p, sz, vp],
        "kzg355_compute_kzg_proof_many": [u8p, u8p, ip, u8p, u8p, sz, vp],
        "kzg355_verify_kzg_proof_many_device": [bp, ip, vp, sz, vp],
        "kzg355_compute_kzg_proof_many_device": [u8p, u8p, ip, vp, vp, sz, vp],
        "kzg355_verify_blob_kzg_proof_batch_many_device": [bp, ip, vp, vp, vp, sz, sz, vp],
        "kzg355_blob_to_kzg_commitment_many_device": [u8p, ip, vp, sz, vp],
        "kzg355_compute_blob_kzg_proof_many_device": [u8p, ip, vp, vp, sz, vp],
        "kzg355_verify_shard_records_device": [vp, ip, vp, vp, vp, sz, sz, vp],
        "kzg355_verify_records_device": [bp, ip, vp, sz, sz, vp],
        "kzg355_verify_shard_records_points_device": [vp, vp, ip, vp, vp, vp, sz, sz, vp],
        "kzg355_verify_records_points_device": [bp, ip, vp, vp, sz, sz, vp],
        "kzg355_verify_records_checked_device": [bp, ip, vp, sz, sz, vp],
        "kzg355_debug_batch_intermediates": [u8p, bp, ip, vp, sz, sz, vp],
        "kzg355_kernel_ms_stats": [vp, u8p, C.POINTER(C.c_double), C.POINTER(C.c_long)],
        "kzg355_settings_device": [vp],
        "kzg355_settings_host_threads": [vp],
        "kzg355_settings_msm_form": [vp],
        "kzg355_load_trusted_setup_devices": [u8p, sz, u8p, sz, C.POINTER(C.c_int), sz, C.POINTER(vp)],
        "kzg355_settings_device_count": [vp],
        "kzg355_settings_exchange_stats": [vp, C.POINTER(C.c_long), C.POINTER(C.c_long)],
        "kzg355_lagrange_setup_from_monomial": [u8p, u8p, sz],
        "kzg355_settings_field_elements_per_blob": [vp],
        "kzg355_set_kernel_timing": [vp, C.c_int],
        "kzg355_settings_set_host_hash": [vp, C.c_int, C.c_int],
        "kzg355_host_sha256": [u8p, u8p, sz, C.c_int],
        "kzg355_debug_verify_host_records": [u8p, bp, ip, u8p, u8p, u8p, sz, sz, vp],
        "kzg355_host_challenge_digests": [u8p, u8p, sz, u8p, sz, C.c_int],
        "kzg355_debug_verify_sharded_intermediates": [u8p, bp, ip, u8p, u8p, u8p, sz, sz, vp],
        "kzg355_verify_blob_kzg_proof_batch_many_device_submit": [C.POINTER(vp), vp, vp, vp, sz, sz, vp],
        "kzg355_verify_collect": [vp, bp, ip],
        "kzg355_settings_msm_shape": [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)],
        "kzg355_settings_build_msm_table": [vp],
        "kzg355_verify_shard_records_points_words_device": [vp, vp, vp, vp, vp, vp, sz, sz, vp],
        "kzg355_verify_records_points_words_device": [vp, vp, vp, sz, sz, vp],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.kzg355_free_trusted_setup.argtypes = [vp]
    lib.kzg355_free_trusted_setup.restype = None
    lib.kzg355_set_kernel_timing.restype = None
    lib.kzg355_reset_kernel_stats.argtypes = [vp]
    lib.kzg355_reset_kernel_stats.restype = None
    lib.kzg355_last_kernel_ms.argtypes = [vp, u8p]
    lib.kzg355_last_kernel_ms.restype = C.c_double
    lib.kzg355_version.restype = C.c_char_p
    lib.kzg355_load_trusted_setup_ex.argtypes = [u8p, sz, u8p, sz, C.POINTER(C.c_int), sz, C.POINTER(Options), C.POINTER(vp)]
    lib.kzg355_load_trusted_setup_ex.restype = C.c_int
    for name in ("kzg355_options_default", "kzg355_options_from_env"):
        getattr(lib, name).argtypes = [C.POINTER(Options)]
        getattr(lib, name).restype = None
    lib.kzg355_settings_host_hashed_calls.argtypes = [vp]
    lib.kzg355_settings_host_hashed_calls.restype = C.c_long
    return lib


EXPORTED_SYMBOLS = [
    "kzg355_load_trusted_setup", "kzg355_load_trusted_setup_file", "kzg355_free_trusted_setup",
    "kzg355_blob_to_kzg_commitment", "kzg355_compute_kzg_proof", "kzg355_compute_blob_kzg_proof",
    "kzg355_verify_kzg_proof", "kzg355_verify_blob_kzg_proof", "kzg355_verify_blob_kzg_proof_batch",
    "kzg355_blob_to_kzg_commitment_many", "kzg355_compute_blob_kzg_proof_many", "kzg355_verify_blob_kzg_proof_batch_many",
    "kzg355_verify_blob_kzg_proof_batch_many_device", "kzg355_blob_to_kzg_commitment_many_device",
    "kzg355_compute_blob_kzg_proof_many_device", "kzg355_verify_shard_records_device", "kzg355_verify_records_device",
    "kzg355_settings_device", "kzg355_last_kernel_ms", "kzg355_set_kernel_timing", "kzg355_version",
    "kzg355_kernel_ms_stats", "kzg355_reset_kernel_stats",
    "kzg355_verify_records_checked_device", "kzg355_debug_batch_intermediates", "kzg355_settings_msm_form", "kzg355_verify_shard_records_points_device", "kzg355_verify_records_points_device", "kzg355_load_trusted_setup_devices", "kzg355_settings_device_count", "kzg355_settings_exchange_stats", "kzg355_lagrange_setup_from_monomial", "kzg355_settings_field_elements_per_blob",
    "kzg355_settings_set_host_hash", "kzg355_settings_host_hashed_calls", "kzg355_host_sha256", "kzg355_host_challenge_digests", "kzg355_debug_verify_host_records",
    "kzg355_options_default", "kzg355_options_from_env", "kzg355_load_trusted_setup_ex", "kzg355_debug_verify_sharded_intermediates",
    "kzg355_verify_blob_kzg_proof_batch_many_device_submit", "kzg355_verify_collect", "kzg355_settings_msm_shape", "kzg355_settings_build_msm_table",
    "kzg355_verify_shard_records_points_words_device", "kzg355_verify_records_points_words_device",
    "kzg355_verify_kzg_proof_many", "kzg355_verify_blob_kzg_proof_many", "kzg355_compute_kzg_proof_many", "kzg355_verify_kzg_proof_many_device",
    "kzg355_compute_kzg_proof_many_device", "kzg355_settings_host_threads",
]
