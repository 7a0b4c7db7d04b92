"""CPU-side checks of the host SHA-256 (kzg_rust_amd/csrc/host_sha256.cpp) that hashes the Fiat-Shamir transcripts of small
host-buffer calls (reference src/kzg.rs:298-339, consts.rs:19-22).  The checker is Python's hashlib -- neither the oracle nor the
device code.  Both forms (portable C, SHA extensions) and the interleaved pair form are covered."""
import ctypes as C
import hashlib
import os
import random

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from kzg_rust_amd import _lib
    return _lib.load()


def _impls(lib):
    out = [1]
    buf = C.create_string_buffer(32)
    if lib.kzg355_host_sha256(buf, b"", 0, 2) == 0:
        out.append(2)
    return out


def test_sha256_known_answers_and_lengths(lib):
    rnd = random.Random(4844)
    msgs = [b"", b"abc", b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq"]
    msgs += [bytes(rnd.getrandbits(8) for _ in range(n)) for n in (1, 55, 56, 57, 63, 64, 65, 119, 120, 127, 128, 129, 1000, 4096 + 31)]
    for impl in [0] + _impls(lib):
        for m in msgs:
            buf = C.create_string_buffer(32)
            assert lib.kzg355_host_sha256(buf, m, len(m), impl) == 0
            assert buf.raw == hashlib.sha256(m).digest(), (impl, len(m))


@pytest.mark.parametrize("n_fe", [4096, 4])
def test_challenge_digests_match_the_transcript(lib, n_fe):
    rnd = random.Random(n_fe)
    bb = 32 * n_fe
    for count in (1, 2, 3, 8):
        blobs = bytes(rnd.getrandbits(8) for _ in range(bb * count)) if n_fe == 4 else os.urandom(bb * count)
        cms = os.urandom(48 * count)
        want = b"".join(
            hashlib.sha256(b"FSBLOBVERIFY_V1_" + (0).to_bytes(8, "big") + n_fe.to_bytes(8, "big") + blobs[bb * i:bb * (i + 1)] + cms[48 * i:48 * (i + 1)]).digest()
            for i in range(count))
        for impl in [0] + _impls(lib):
            out = C.create_string_buffer(32 * count)
            assert lib.kzg355_host_challenge_digests(out, blobs, bb, cms, count, impl) == 0
            assert out.raw == want, (impl, count)


def test_bad_arguments(lib):
    out = C.create_string_buffer(32)
    assert lib.kzg355_host_sha256(out, b"x", 1, 3) == 1
    assert lib.kzg355_host_challenge_digests(out, b"x" * 33, 33, bytes(48), 1, 0) == 1
