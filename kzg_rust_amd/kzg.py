"""Host-side mirror of the reference's public surface (pawanjay176/kzg_rust src/kzg.rs:10-22, 88-279,
983-1079) over the C ABI of libkzg355.so.  Same names, same argument meaning, same error behaviour, so the
parity tests read like the reference's own (src/lib.rs:30-203).  All arithmetic happens in the HIP library;
this module only checks lengths / hex (as the Rust newtypes do before any FFI) and forwards.

The reference's host language is Rust; no Rust toolchain exists in this image, so the shim a maintainer
would compile is in rust/ (source only, see rust/README.md) -- this Python mirror is what the test-suite drives.
"""
import ctypes as C
import os

from . import _lib

BYTES_PER_FIELD_ELEMENT = 32      # consts.rs:5
BYTES_PER_COMMITMENT = 48         # consts.rs:8
BYTES_PER_PROOF = 48              # consts.rs:11
FIELD_ELEMENTS_PER_BLOB = 4096    # consts.rs:13
BYTES_PER_BLOB = 131072           # consts.rs:16
BYTES_PER_G1 = 48                 # consts.rs:31
BYTES_PER_G2 = 96                 # consts.rs:34
TRUSTED_SETUP_NUM_G2_POINTS = 65  # consts.rs:37


class Error(Exception):
    """enum Error (kzg.rs:10-22)."""
    code = None


class BadArgs(Error):
    code = 1


class InternalError(Error):
    code = 2


class InvalidBytesLength(Error):
    code = 3


class InvalidHexFormat(Error):
    code = 4


class InvalidTrustedSetup(Error):
    code = 5


class NoDevice(Error):
    """No usable HIP device (no reference counterpart; there is no CPU fallback)."""
    code = 6


class NoMemory(Error):
    """A device or pinned-host allocation failed (no reference counterpart)."""
    code = 7


class DeviceError(Error):
    """A HIP runtime / RCCL call failed on a device that exists (no reference counterpart)."""
    code = 8


_ERRORS = {c.code: c for c in (BadArgs, InternalError, InvalidBytesLength, InvalidHexFormat, InvalidTrustedSetup, NoDevice, NoMemory, DeviceError)}
# statuses that can only describe the call as a whole (a device, allocation or library failure): a *_many call that returns one of
# them may have left units untouched, so it is raised even when earlier units carry a per-unit status
_WHOLE_CALL = (InternalError.code, NoDevice.code, NoMemory.code, DeviceError.code)


def _whole_call_failed(rc, st, n):
    return rc in _WHOLE_CALL or (rc != 0 and not any(st[i] for i in range(n)))


def _check(rc, what):
    if rc != 0:
        raise _ERRORS.get(rc, InternalError)(f"{what}: status {rc}")


def hex_to_bytes(hex_str):
    """kzg.rs:82-86: hex with or without the 0x prefix."""
    s = hex_str[2:] if hex_str.startswith("0x") else hex_str
    try:
        return bytes.fromhex(s)
    except ValueError as e:
        raise InvalidHexFormat(f"Failed to decode hex: {e}")


class _Fixed:
    SIZE = 0
    LENGTH_ERROR = InvalidBytesLength

    def __init__(self, b):
        b = bytes(b)
        if len(b) != self.SIZE:
            raise self.LENGTH_ERROR(f"Invalid byte length. Expected {self.SIZE} got {len(b)}")
        self.bytes = b

    @classmethod
    def from_bytes(cls, b):
        return cls(b)

    @classmethod
    def from_hex(cls, s):
        return cls(hex_to_bytes(s))

    def to_bytes(self):
        return self.bytes

    def __bytes__(self):
        return self.bytes

    def __eq__(self, other):
        return isinstance(other, _Fixed) and self.bytes == other.bytes

    def __hash__(self):
        return hash(self.bytes)

    def __repr__(self):
        return f"{type(self).__name__}(0x{self.bytes[:8].hex()}..)"


class Bytes32(_Fixed):
    """kzg.rs:101-122 (length error is BadArgs for this type)."""
    SIZE = 32
    LENGTH_ERROR = BadArgs


class Bytes48(_Fixed):
    """kzg.rs:124-152."""
    SIZE = 48


class Blob(_Fixed):
    """kzg.rs:154-178 (mainnet preset: 4096 field elements; kzg_rust_amd.kzg_minimal.Blob is the 4-element one)."""
    SIZE = BYTES_PER_BLOB


class KzgCommitment(Bytes48):
    """kzg.rs:180-191."""


class KzgProof(Bytes48):
    """kzg.rs:193-204."""


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _lib.load()
    return _LIB


class KzgSettings:
    """Opaque handle to the device-resident trusted setup (replaces struct KzgSettings, kzg.rs:28-40)."""

    def __init__(self, handle):
        self._h = handle

    @property
    def handle(self):
        if self._h is None:
            raise BadArgs("KzgSettings already freed")
        return self._h

    @staticmethod
    def load_trusted_setup(g1_bytes, g2_bytes, devices=None):
        """kzg.rs:45-78: lists of 48-byte / 96-byte strings.  devices (extension): list of GPU ordinals the handle spans; the
        host-buffer calls then spread their work over them inside the library."""
        g1_bytes, g2_bytes = list(g1_bytes), list(g2_bytes)
        if any(len(x) != BYTES_PER_G1 for x in g1_bytes) or any(len(x) != BYTES_PER_G2 for x in g2_bytes):
            raise InvalidBytesLength("trusted setup point length")
        h = C.c_void_p()
        if devices is None:
            rc = lib().kzg355_load_trusted_setup(b"".join(g1_bytes), len(g1_bytes), b"".join(g2_bytes), len(g2_bytes), C.byref(h))
        else:
            devs = (C.c_int * len(devices))(*devices)
            rc = lib().kzg355_load_trusted_setup_devices(b"".join(g1_bytes), len(g1_bytes), b"".join(g2_bytes), len(g2_bytes), devs, len(devices), C.byref(h))
        _check(rc, "load_trusted_setup")
        return KzgSettings(h)

    @staticmethod
    def load_trusted_setup_ex(g1_bytes, g2_bytes, devices=None, **options):
        """kzg355_load_trusted_setup_ex: explicit options (field names of struct kzg355_options, e.g. msm_bits=14, host_hash=-1); reads no
        KZG355_* environment variable."""
        g1_bytes, g2_bytes = list(g1_bytes), list(g2_bytes)
        if any(len(x) != BYTES_PER_G1 for x in g1_bytes) or any(len(x) != BYTES_PER_G2 for x in g2_bytes):
            raise InvalidBytesLength("trusted setup point length")
        o = _lib.Options()
        lib().kzg355_options_default(C.byref(o))
        for k, v in options.items():
            if k == "struct_size" or not hasattr(o, k):
                raise BadArgs(f"unknown option {k}")
            setattr(o, k, v)
        devs = (C.c_int * len(devices))(*devices) if devices else None
        h = C.c_void_p()
        rc = lib().kzg355_load_trusted_setup_ex(b"".join(g1_bytes), len(g1_bytes), b"".join(g2_bytes), len(g2_bytes), devs, len(devices) if devices else 0,
                                                C.byref(o), C.byref(h))
        _check(rc, "load_trusted_setup_ex")
        return KzgSettings(h)

    @staticmethod
    def load_trusted_setup_file(path):
        h = C.c_void_p()
        rc = lib().kzg355_load_trusted_setup_file(os.fsencode(path), C.byref(h))
        _check(rc, "load_trusted_setup_file")
        return KzgSettings(h)

    @property
    def device(self):
        return lib().kzg355_settings_device(self.handle)

    @property
    def device_count(self):
        return lib().kzg355_settings_device_count(self.handle)

    def exchange_stats(self):
        """(exchange kind: 1 RCCL all-gather, 0 peer copies, -1 plain handle; all-gathers so far; peer exchanges so far)."""
        a, p = C.c_long(), C.c_long()
        kind = lib().kzg355_settings_exchange_stats(self.handle, C.byref(a), C.byref(p))
        return kind, a.value, p.value

    @property
    def field_elements_per_blob(self):
        """4096 (mainnet) or 4 (minimal preset): fixed by the number of G1 points the handle was loaded from."""
        return lib().kzg355_settings_field_elements_per_blob(self.handle)

    def msm_shape(self):
        """(digit width, windows per half-scalar, GLV split 0 / 1, table bytes) of the fixed-base MSM table; all 0 while it has not been built."""
        b, w, g, n = C.c_int(), C.c_int(), C.c_int(), C.c_size_t()
        _check(lib().kzg355_settings_msm_shape(self.handle, C.byref(b), C.byref(w), C.byref(g), C.byref(n)), "msm_shape")
        return b.value, w.value, g.value, n.value

    def build_msm_table(self):
        """Build the fixed-base MSM table now instead of inside the first commitment / proof call."""
        _check(lib().kzg355_settings_build_msm_table(self.handle), "build_msm_table")

    @property
    def msm_form(self):
        """12 / 13 / 14: wide-window table of that digit width; 8: bucket form by request; -8: bucket form because the table
        could not be allocated."""
        return lib().kzg355_settings_msm_form(self.handle)

    def set_host_hash(self, mode, max_blobs=0):
        """Fiat-Shamir hashing of small host-buffer calls on host threads: mode 0 by size, 1 always, -1 never (kzg355.h)."""
        _check(lib().kzg355_settings_set_host_hash(self.handle, mode, max_blobs), "set_host_hash")

    @property
    def host_hashed_calls(self):
        return lib().kzg355_settings_host_hashed_calls(self.handle)

    @property
    def host_threads(self):
        """host threads that hash for one call on this handle (its workers + the calling thread)"""
        return lib().kzg355_settings_host_threads(self.handle)

    def set_kernel_timing(self, enabled=True):
        lib().kzg355_set_kernel_timing(self.handle, 1 if enabled else 0)

    def last_kernel_ms(self, family):
        return lib().kzg355_last_kernel_ms(self.handle, family.encode())

    def free(self):
        if self._h is not None:
            lib().kzg355_free_trusted_setup(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _b(x, cls):
    return x.bytes if isinstance(x, _Fixed) else cls(x).bytes


def _blob(x, s):
    """Blob bytes for the handle's preset: the newtype's length rule (kzg.rs:160-173) with BYTES_PER_BLOB = 32 * FIELD_ELEMENTS_PER_BLOB."""
    b = x.bytes if isinstance(x, _Fixed) else bytes(x)
    want = 32 * s.field_elements_per_blob
    if len(b) != want:
        raise InvalidBytesLength(f"Invalid byte length. Expected {want} got {len(b)}")
    return b


class Kzg:
    """pub struct Kzg (kzg.rs:983-1079): the seven associated functions, forwarded to the HIP engine."""

    @staticmethod
    def load_trusted_setup_file(path):  # kzg.rs:995
        return KzgSettings.load_trusted_setup_file(path)

    @staticmethod
    def load_trusted_setup(g1_bytes, g2_bytes, devices=None):  # kzg.rs:1005
        return KzgSettings.load_trusted_setup(g1_bytes, g2_bytes, devices)

    @staticmethod
    def blob_to_kzg_commitment(blob, s):  # kzg.rs:1013
        out = C.create_string_buffer(48)
        _check(lib().kzg355_blob_to_kzg_commitment(out, _blob(blob, s), s.handle), "blob_to_kzg_commitment")
        return KzgCommitment(out.raw)

    @staticmethod
    def compute_kzg_proof(blob, z_bytes, s):  # kzg.rs:1021
        pr, y = C.create_string_buffer(48), C.create_string_buffer(32)
        _check(lib().kzg355_compute_kzg_proof(pr, y, _blob(blob, s), _b(z_bytes, Bytes32), s.handle), "compute_kzg_proof")
        return KzgProof(pr.raw), Bytes32(y.raw)

    @staticmethod
    def compute_blob_kzg_proof(blob, commitment, s):  # kzg.rs:1030
        pr = C.create_string_buffer(48)
        _check(lib().kzg355_compute_blob_kzg_proof(pr, _blob(blob, s), _b(commitment, KzgCommitment), s.handle), "compute_blob_kzg_proof")
        return KzgProof(pr.raw)

    @staticmethod
    def verify_kzg_proof(commitment, z_bytes, y_bytes, proof, s):  # kzg.rs:1039
        ok = C.c_bool()
        _check(lib().kzg355_verify_kzg_proof(C.byref(ok), _b(commitment, KzgCommitment), _b(z_bytes, Bytes32), _b(y_bytes, Bytes32),
                                             _b(proof, KzgProof), s.handle), "verify_kzg_proof")
        return bool(ok.value)

    @staticmethod
    def verify_blob_kzg_proof(blob, commitment, proof, s):  # kzg.rs:1050
        ok = C.c_bool()
        _check(lib().kzg355_verify_blob_kzg_proof(C.byref(ok), _blob(blob, s), _b(commitment, KzgCommitment), _b(proof, KzgProof), s.handle),
               "verify_blob_kzg_proof")
        return bool(ok.value)

    @staticmethod
    def verify_blob_kzg_proof_batch(blobs, commitments, proofs, s):  # kzg.rs:1066
        bl = [_blob(x, s) for x in blobs]
        cs = [_b(x, KzgCommitment) for x in commitments]
        ps = [_b(x, KzgProof) for x in proofs]
        ok = C.c_bool()
        _check(lib().kzg355_verify_blob_kzg_proof_batch(C.byref(ok), b"".join(bl), len(bl), b"".join(cs), len(cs), b"".join(ps), len(ps),
                                                        s.handle), "verify_blob_kzg_proof_batch")
        return bool(ok.value)

    # ---- throughput extensions (no reference counterpart; same semantics per unit) ----
    @staticmethod
    def blob_to_kzg_commitment_many(blobs, s):
        bl = [_blob(x, s) for x in blobs]
        n = len(bl)
        out = C.create_string_buffer(48 * max(n, 1))
        st = (C.c_int * max(n, 1))()
        rc = lib().kzg355_blob_to_kzg_commitment_many(out, st, b"".join(bl), n, s.handle)
        if _whole_call_failed(rc, st, n):                         # no device, out of memory, n too large: nothing usable came back
            _check(rc, "blob_to_kzg_commitment_many")
        return [KzgCommitment(out.raw[48 * i:48 * i + 48]) if st[i] == 0 else _ERRORS.get(st[i], InternalError)("commit") for i in range(n)]

    @staticmethod
    def compute_blob_kzg_proof_many(blobs, commitments, s):
        bl = [_blob(x, s) for x in blobs]
        cs = [_b(x, KzgCommitment) for x in commitments]
        if len(bl) != len(cs):
            raise BadArgs("length mismatch")
        n = len(bl)
        out = C.create_string_buffer(48 * max(n, 1))
        st = (C.c_int * max(n, 1))()
        rc = lib().kzg355_compute_blob_kzg_proof_many(out, st, b"".join(bl), b"".join(cs), n, s.handle)
        if _whole_call_failed(rc, st, n):
            _check(rc, "compute_blob_kzg_proof_many")
        return [KzgProof(out.raw[48 * i:48 * i + 48]) if st[i] == 0 else _ERRORS.get(st[i], InternalError)("proof") for i in range(n)]

    @staticmethod
    def verify_blob_kzg_proof_batch_many(groups, s):
        """groups: list of (blobs, commitments, proofs) with equal group sizes.  Returns a list of bool / Error."""
        if not groups:
            return []
        npg = len(groups[0][0])
        flat_b, flat_c, flat_p = [], [], []
        for bl, cs, ps in groups:
            if not (len(bl) == len(cs) == len(ps) == npg):
                raise BadArgs("all groups must have the same size")
            flat_b += [_blob(x, s) for x in bl]
            flat_c += [_b(x, KzgCommitment) for x in cs]
            flat_p += [_b(x, KzgProof) for x in ps]
        G = len(groups)
        ok = (C.c_bool * G)()
        st = (C.c_int * G)()
        rc = lib().kzg355_verify_blob_kzg_proof_batch_many(ok, st, b"".join(flat_b), b"".join(flat_c), b"".join(flat_p), npg, G, s.handle)
        if _whole_call_failed(rc, st, G):                         # whole-call failure: the per-batch statuses are not to be trusted
            _check(rc, "verify_blob_kzg_proof_batch_many")
        return [bool(ok[i]) if st[i] == 0 else _ERRORS.get(st[i], InternalError)("verify") for i in range(G)]

    @staticmethod
    def verify_kzg_proof_many(commitments, zs, ys, proofs, s):
        """n independent verify_kzg_proof checks (kzg.rs:1039) in one call.  Returns a list of bool / Error."""
        cs = [_b(x, KzgCommitment) for x in commitments]
        zz = [_b(x, Bytes32) for x in zs]
        yy = [_b(x, Bytes32) for x in ys]
        ps = [_b(x, KzgProof) for x in proofs]
        n = len(cs)
        if not (len(zz) == len(yy) == len(ps) == n):
            raise BadArgs("length mismatch")
        ok = (C.c_bool * max(n, 1))()
        st = (C.c_int * max(n, 1))()
        rc = lib().kzg355_verify_kzg_proof_many(ok, st, b"".join(cs), b"".join(zz), b"".join(yy), b"".join(ps), n, s.handle)
        if _whole_call_failed(rc, st, n):
            _check(rc, "verify_kzg_proof_many")
        return [bool(ok[i]) if st[i] == 0 else _ERRORS.get(st[i], InternalError)("verify") for i in range(n)]

    @staticmethod
    def verify_blob_kzg_proof_many(blobs, commitments, proofs, s):
        """n independent verify_blob_kzg_proof checks (kzg.rs:1050) in one call.  Returns a list of bool / Error."""
        bl = [_blob(x, s) for x in blobs]
        cs = [_b(x, KzgCommitment) for x in commitments]
        ps = [_b(x, KzgProof) for x in proofs]
        n = len(bl)
        if not (len(cs) == len(ps) == n):
            raise BadArgs("length mismatch")
        ok = (C.c_bool * max(n, 1))()
        st = (C.c_int * max(n, 1))()
        rc = lib().kzg355_verify_blob_kzg_proof_many(ok, st, b"".join(bl), b"".join(cs), b"".join(ps), n, s.handle)
        if _whole_call_failed(rc, st, n):
            _check(rc, "verify_blob_kzg_proof_many")
        return [bool(ok[i]) if st[i] == 0 else _ERRORS.get(st[i], InternalError)("verify") for i in range(n)]

    @staticmethod
    def compute_kzg_proof_many(blobs, zs, s):
        """n independent compute_kzg_proof calls (kzg.rs:1021) in one call.  Returns a list of (KzgProof, Bytes32) / Error."""
        bl = [_blob(x, s) for x in blobs]
        zz = [_b(x, Bytes32) for x in zs]
        n = len(bl)
        if len(zz) != n:
            raise BadArgs("length mismatch")
        out = C.create_string_buffer(48 * max(n, 1))
        ys = C.create_string_buffer(32 * max(n, 1))
        st = (C.c_int * max(n, 1))()
        rc = lib().kzg355_compute_kzg_proof_many(out, ys, st, b"".join(bl), b"".join(zz), n, s.handle)
        if _whole_call_failed(rc, st, n):
            _check(rc, "compute_kzg_proof_many")
        return [(KzgProof(out.raw[48 * i:48 * i + 48]), Bytes32(ys.raw[32 * i:32 * i + 32])) if st[i] == 0 else _ERRORS.get(st[i], InternalError)("proof")
                for i in range(n)]
