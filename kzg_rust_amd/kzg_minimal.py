"""`kzg_minimal` of the reference's README (README.md:8-9): FIELD_ELEMENTS_PER_BLOB = 4, blobs of 128 bytes.

The reference snapshot carries no code or vectors for this preset (only the truncation in src/trusted_setup.rs:144-151, which
does not yield a Lagrange basis of the size-4 domain).  Here the handle is loaded from a genuine size-4 Lagrange setup --
`lagrange_setup_from_monomial` derives it from the first four monomial points [tau^k]G1 of a ceremony file such as the
reference's testing_trusted_setups.json (`setup_G1`) -- and every `Kzg` function then works on 128-byte blobs."""
import ctypes as C

from . import kzg as _k
from .kzg import Bytes32, Bytes48, Error, Kzg, KzgCommitment, KzgProof, KzgSettings  # noqa: F401

FIELD_ELEMENTS_PER_BLOB = 4
BYTES_PER_BLOB = 32 * FIELD_ELEMENTS_PER_BLOB
BYTES_PER_FIELD_ELEMENT = _k.BYTES_PER_FIELD_ELEMENT
BYTES_PER_COMMITMENT = _k.BYTES_PER_COMMITMENT
BYTES_PER_PROOF = _k.BYTES_PER_PROOF


class Blob(_k._Fixed):
    """kzg.rs:154-178 with the minimal preset's BYTES_PER_BLOB."""
    SIZE = BYTES_PER_BLOB


def lagrange_setup_from_monomial(monomial_g1, device_settings=None):
    """n compressed monomial points [tau^k]G1 (n a power of two, 4 <= n <= 64) -> the n compressed Lagrange points
    L_j(tau) G1 = (1/n) sum_k w^(-jk) [tau^k]G1 in file (natural) order, computed on the GPU."""
    pts = [bytes(x) for x in monomial_g1]
    if any(len(x) != 48 for x in pts):
        raise _k.InvalidBytesLength("monomial point length")
    out = C.create_string_buffer(48 * len(pts))
    _k._check(_k.lib().kzg355_lagrange_setup_from_monomial(out, b"".join(pts), len(pts)), "lagrange_setup_from_monomial")
    return [out.raw[48 * i:48 * i + 48] for i in range(len(pts))]
