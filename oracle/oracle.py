"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The API mirrors the reference's `Kzg` associated functions (src/kzg.rs:983-1079); errors surface
as OracleError carrying the status code (null <=> Err convention of src/lib.rs:47-50).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
BYTES_PER_BLOB = 131072


class OracleError(Exception):
    def __init__(self, code, what=""):
        super().__init__(f"oracle status {code} {what}")
        self.code = code


def build(native=False):
    target = "native" if native else "all"
    subprocess.run(["make", "-C", _HERE, target], check=True, stdout=subprocess.DEVNULL)


def _load(native=False, preset="mainnet"):
    name = "liboracle_minimal.so" if preset == "minimal" else "liboracle_native.so" if native else "liboracle.so"
    path = os.path.join(_HERE, name)
    # tests/test_sanitizers.py: the AddressSanitizer / UndefinedBehaviorSanitizer build of the same two sources (`make asan`), loaded into a Python
    # that was started with the sanitizer runtime preloaded
    if os.environ.get("KZG355_ORACLE_ASAN") == "1" and preset != "minimal":
        path = os.path.join(_HERE, "liboracle_asan.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", _HERE, "asan"], check=True, stdout=subprocess.DEVNULL)
    if not os.path.exists(path):
        build(native and preset != "minimal")
    lib = C.CDLL(path)
    lib.okzg_init()
    return lib


class Oracle:
    @property
    def has_fast_primitives(self):
        """the loaded build carries the mulx / adcx / adox field products and SHA-extension hash (liboracle_native.so on a CPU that has them)"""
        return bool(self.lib.okzg_have_fast_primitives())

    def set_fast_primitives(self, on):
        self.lib.okzg_set_fast_primitives(1 if on else 0)

    def __init__(self, native=False, preset="mainnet"):
        """preset "mainnet": FIELD_ELEMENTS_PER_BLOB = 4096; "minimal": 4 (the same source built with -DN_FE=4)."""
        self.lib = _load(native, preset)
        L = self.lib
        self.field_elements_per_blob = L.okzg_field_elements_per_blob()
        self.bytes_per_blob = 32 * self.field_elements_per_blob
        vp, sz, u8p = C.c_void_p, C.c_size_t, C.c_char_p
        L.okzg_load_trusted_setup.argtypes = [u8p, sz, u8p, sz, C.POINTER(vp)]
        L.okzg_load_trusted_setup_file.argtypes = [u8p, C.POINTER(vp)]
        L.okzg_free_trusted_setup.argtypes = [vp]
        L.okzg_blob_to_kzg_commitment.argtypes = [u8p, u8p, vp]
        L.okzg_compute_kzg_proof.argtypes = [u8p, u8p, u8p, u8p, vp]
        L.okzg_compute_blob_kzg_proof.argtypes = [u8p, u8p, u8p, vp]
        L.okzg_verify_kzg_proof.argtypes = [C.POINTER(C.c_bool), u8p, u8p, u8p, u8p, vp]
        L.okzg_verify_blob_kzg_proof.argtypes = [C.POINTER(C.c_bool), u8p, u8p, u8p, vp]
        L.okzg_verify_blob_kzg_proof_batch.argtypes = [C.POINTER(C.c_bool), u8p, sz, u8p, sz, u8p, sz, vp]
        L.okzg_compute_challenge.argtypes = [u8p, u8p, u8p]
        L.okzg_evaluate_polynomial.argtypes = [u8p, u8p, u8p, vp]
        L.okzg_verify_batch_intermediates.argtypes = [C.POINTER(C.c_bool), u8p, u8p, u8p, u8p, u8p, sz, vp]
        L.okzg_get_roots_of_unity.argtypes = [u8p, vp]
        L.okzg_get_g1_values.argtypes = [u8p, vp]
        L.okzg_sha256.argtypes = [u8p, u8p, sz]
        L.okzg_fp_op.argtypes = [C.c_int, u8p, u8p, u8p]
        L.okzg_fr_op.argtypes = [C.c_int, u8p, u8p, u8p]
        L.okzg_g1_validate.argtypes = [u8p]
        L.okzg_g1_uncompress_only.argtypes = [u8p]
        L.okzg_g1_mul_add.argtypes = [u8p, u8p, u8p, u8p]
        L.okzg_g1_lincomb.argtypes = [u8p, u8p, u8p, sz, C.c_int]
        L.okzg_g2_uncompress_check.argtypes = [u8p]
        L.okzg_pairings_verify.argtypes = [C.POINTER(C.c_bool), u8p, u8p, u8p, u8p]
        L.okzg_g2_gen_mul.argtypes = [u8p, u8p]
        L.okzg_shard_records.argtypes = [u8p, u8p, u8p, u8p, sz, vp]
        L.okzg_verify_records.argtypes = [C.POINTER(C.c_bool), u8p, sz, vp]

    # ---- settings
    def load_trusted_setup(self, g1_bytes: bytes, g2_bytes: bytes, n1=None, n2=None):
        h = C.c_void_p()
        n1 = len(g1_bytes) // 48 if n1 is None else n1
        n2 = len(g2_bytes) // 96 if n2 is None else n2
        rc = self.lib.okzg_load_trusted_setup(g1_bytes, n1, g2_bytes, n2, C.byref(h))
        if rc:
            raise OracleError(rc, "load_trusted_setup")
        return h

    def load_trusted_setup_file(self, path):
        h = C.c_void_p()
        rc = self.lib.okzg_load_trusted_setup_file(os.fsencode(path), C.byref(h))
        if rc:
            raise OracleError(rc, "load_trusted_setup_file")
        return h

    def free_trusted_setup(self, h):
        self.lib.okzg_free_trusted_setup(h)

    # ---- the seven entry points
    def blob_to_kzg_commitment(self, blob, s):
        assert len(blob) == self.bytes_per_blob
        out = C.create_string_buffer(48)
        rc = self.lib.okzg_blob_to_kzg_commitment(out, bytes(blob), s)
        if rc:
            raise OracleError(rc)
        return out.raw

    def compute_kzg_proof(self, blob, z, s):
        assert len(blob) == self.bytes_per_blob and len(z) == 32
        pr, y = C.create_string_buffer(48), C.create_string_buffer(32)
        rc = self.lib.okzg_compute_kzg_proof(pr, y, bytes(blob), bytes(z), s)
        if rc:
            raise OracleError(rc)
        return pr.raw, y.raw

    def compute_blob_kzg_proof(self, blob, c, s):
        assert len(blob) == self.bytes_per_blob and len(c) == 48
        pr = C.create_string_buffer(48)
        rc = self.lib.okzg_compute_blob_kzg_proof(pr, bytes(blob), bytes(c), s)
        if rc:
            raise OracleError(rc)
        return pr.raw

    def verify_kzg_proof(self, c, z, y, proof, s):
        assert len(c) == 48 and len(z) == 32 and len(y) == 32 and len(proof) == 48
        ok = C.c_bool()
        rc = self.lib.okzg_verify_kzg_proof(C.byref(ok), bytes(c), bytes(z), bytes(y), bytes(proof), s)
        if rc:
            raise OracleError(rc)
        return bool(ok.value)

    def verify_blob_kzg_proof(self, blob, c, proof, s):
        assert len(blob) == self.bytes_per_blob and len(c) == 48 and len(proof) == 48
        ok = C.c_bool()
        rc = self.lib.okzg_verify_blob_kzg_proof(C.byref(ok), bytes(blob), bytes(c), bytes(proof), s)
        if rc:
            raise OracleError(rc)
        return bool(ok.value)

    def verify_blob_kzg_proof_batch(self, blobs, cs, proofs, s):
        assert all(len(b) == self.bytes_per_blob for b in blobs)
        assert all(len(c) == 48 for c in cs) and all(len(p) == 48 for p in proofs)
        ok = C.c_bool()
        rc = self.lib.okzg_verify_blob_kzg_proof_batch(C.byref(ok), b"".join(blobs), len(blobs), b"".join(cs), len(cs),
                                                       b"".join(proofs), len(proofs), s)
        if rc:
            raise OracleError(rc)
        return bool(ok.value)

    # ---- intermediates
    def compute_challenge(self, blob, c):
        z = C.create_string_buffer(32)
        rc = self.lib.okzg_compute_challenge(z, bytes(blob), bytes(c))
        if rc:
            raise OracleError(rc)
        return z.raw

    def evaluate_polynomial(self, blob, z, s):
        y = C.create_string_buffer(32)
        rc = self.lib.okzg_evaluate_polynomial(y, bytes(blob), bytes(z), s)
        if rc:
            raise OracleError(rc)
        return y.raw

    def verify_batch_intermediates(self, blobs, cs, proofs, s):
        n = len(blobs)
        ok = C.c_bool()
        dump, zy = C.create_string_buffer(128), C.create_string_buffer(64 * n)
        rc = self.lib.okzg_verify_batch_intermediates(C.byref(ok), dump, zy, b"".join(blobs), b"".join(cs), b"".join(proofs), n, s)
        if rc:
            raise OracleError(rc)
        d = dump.raw
        return {"ok": bool(ok.value), "r": d[:32], "proof_lincomb": d[32:80], "rhs": d[80:128],
                "z": [zy.raw[64 * i:64 * i + 32] for i in range(n)], "y": [zy.raw[64 * i + 32:64 * i + 64] for i in range(n)]}

    def roots_of_unity(self, s):
        out = C.create_string_buffer(self.field_elements_per_blob * 32)
        self.lib.okzg_get_roots_of_unity(out, s)
        return out.raw

    def g1_values(self, s):
        out = C.create_string_buffer(self.field_elements_per_blob * 48)
        self.lib.okzg_get_g1_values(out, s)
        return out.raw

    # ---- primitives
    def sha256(self, msg):
        out = C.create_string_buffer(32)
        self.lib.okzg_sha256(out, bytes(msg), len(msg))
        return out.raw

    def fp_op(self, op, a, b=bytes(48)):
        out = C.create_string_buffer(48)
        rc = self.lib.okzg_fp_op({"add": 0, "sub": 1, "mul": 2, "inv": 3, "sqrt": 4}[op], out, a, b)
        if rc:
            raise OracleError(rc)
        return out.raw

    def fr_op(self, op, a, b=bytes(32)):
        out = C.create_string_buffer(32)
        rc = self.lib.okzg_fr_op({"add": 0, "sub": 1, "mul": 2, "inv": 3}[op], out, a, b)
        if rc:
            raise OracleError(rc)
        return out.raw

    def g1_validate(self, b):
        return self.lib.okzg_g1_validate(bytes(b))

    def g1_uncompress_only(self, b):
        return self.lib.okzg_g1_uncompress_only(bytes(b))

    def g1_mul_add(self, p, k_be, q=None):
        out = C.create_string_buffer(48)
        rc = self.lib.okzg_g1_mul_add(out, bytes(p), bytes(k_be), None if q is None else bytes(q))
        if rc:
            raise OracleError(rc)
        return out.raw

    def g1_lincomb(self, points, scalars_be, fast=True):
        out = C.create_string_buffer(48)
        rc = self.lib.okzg_g1_lincomb(out, b"".join(points), b"".join(scalars_be), len(points), int(fast))
        if rc:
            raise OracleError(rc)
        return out.raw

    def g2_uncompress_check(self, b):
        return self.lib.okzg_g2_uncompress_check(bytes(b))

    def pairings_verify(self, p1, q1, p2, q2):
        ok = C.c_bool()
        rc = self.lib.okzg_pairings_verify(C.byref(ok), bytes(p1), bytes(q1), bytes(p2), bytes(q2))
        if rc:
            raise OracleError(rc)
        return bool(ok.value)

    def g2_gen_mul(self, k_be):
        out = C.create_string_buffer(192)
        self.lib.okzg_g2_gen_mul(out, bytes(k_be))
        return out.raw

    # ---- record-level stages (sharded path tests)
    def shard_records(self, blobs, cs, proofs, s):
        n = len(blobs)
        out = C.create_string_buffer(160 * max(n, 1))
        rc = self.lib.okzg_shard_records(out, b"".join(blobs), b"".join(cs), b"".join(proofs), n, s)
        if rc:
            raise OracleError(rc)
        return out.raw[:160 * n]

    def verify_records(self, records, s):
        ok = C.c_bool()
        rc = self.lib.okzg_verify_records(C.byref(ok), bytes(records), len(records) // 160, s)
        if rc:
            raise OracleError(rc)
        return bool(ok.value)
