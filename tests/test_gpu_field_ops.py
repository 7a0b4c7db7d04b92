"""-m gpu: the field layer of kzg_rust_amd/csrc (field.h: 29-bit-limb Montgomery Fp / Fr, lazy products; modinv.h: divstep inversion)
run ON THE DEVICE, one element per lane, against Python big integers -- SURVEY 8 row a5 (fr_batch_inv / fr_div / fr_pow and the blst
field calls under them, reference src/utils.rs:35-140).  The same operation switch runs on the host in tests/test_device_math_host.py;
this is the gfx950 build of it (tests/native/gpu_probe.hip, built by __graft_entry__.build())."""
import ctypes as C
import os
import random

import pytest

from oracle.pyref import P, R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def probe():
    so = os.path.join(ROOT, "tests", "native", "libgpu_probe.so")
    assert os.path.exists(so), "run __graft_entry__.build() first"
    return C.CDLL(so)


def _run(fn, op, a_vals, b_vals, width):
    n = len(a_vals)
    a = b"".join(v.to_bytes(width, "big") for v in a_vals); b = b"".join(v.to_bytes(width, "big") for v in b_vals)
    out = C.create_string_buffer(width * n); rc = (C.c_int * n)()
    assert fn(op, n, a, b, out, rc) == 0
    return [int.from_bytes(out.raw[width * i:width * (i + 1)], "big") for i in range(n)], list(rc)


def test_fp_layer_on_device(probe):
    rnd = random.Random(2101)
    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 1 << 380, (1 << 29) - 1, 1 << 29, (1 << 377) - 1]
    a = edge + [rnd.randrange(P) for _ in range(1024 - len(edge))]
    b = [rnd.choice(edge) if i % 7 == 0 else rnd.randrange(P) for i in range(len(a))]
    for op, f in ((0, lambda x, y: (x + y) % P), (1, lambda x, y: (x - y) % P), (2, lambda x, y: x * y % P), (5, lambda x, y: -x % P),
                  (6, lambda x, y: 2 * x % P), (9, lambda x, y: x * x % P), (8, lambda x, y: pow(x, -1, P) if x else 0),
                  (10, lambda x, y: (x * y + x) * y % P)):
        got, rc = _run(probe.gpu_fp_ops, op, a, b, 48)
        assert rc == [0] * len(a)
        assert got == [f(x, y) for x, y in zip(a, b)], op
    got, rc = _run(probe.gpu_fp_ops, 3, a[:128], b[:128], 48)                 # Fermat inversion: the cross-check of the divstep form
    assert got == [pow(x, -1, P) if x else 0 for x in a[:128]]
    sq = [x * x % P for x in a[:256]] + [x for x in a[256:512]]               # squares and arbitrary values
    got, rc = _run(probe.gpu_fp_ops, 4, sq, sq, 48)
    for x, s, c in zip(sq, got, rc):
        is_sq = x == 0 or pow(x, (P - 1) // 2, P) == 1
        assert c == (0 if is_sq else 2)
        if is_sq:
            assert s * s % P == x
    _, rc = _run(probe.gpu_fp_ops, 7, a, b, 48)                               # y > (p - 1) / 2: the sign bit of the compressed form
    assert rc == [101 if x > (P - 1) // 2 else 100 for x in a]
    _, rc = _run(probe.gpu_fp_ops, 0, [P, P + 5], [0, 0], 48)                 # the byte decoder refuses >= p
    assert rc == [1, 1]


def test_fr_layer_on_device(probe):
    rnd = random.Random(2102)
    edge = [0, 1, R - 1, R - 2, R, R + 1, (1 << 256) - 1, 1 << 255, 2 * R, 2 * R + 1]
    a = edge + [rnd.randrange(1 << 256) for _ in range(1024 - len(edge))]
    b = [rnd.choice(edge) if i % 5 == 0 else rnd.randrange(1 << 256) for i in range(len(a))]
    inv = lambda x: pow(x % R, -1, R) if x % R else 0
    for op, f in ((0, lambda x, y: (x + y) % R), (1, lambda x, y: (x - y) % R), (2, lambda x, y: x * y % R), (5, lambda x, y: inv(x)),
                  (6, lambda x, y: x * inv(y) % R), (7, lambda x, y: pow(x % R, y & 0xffffffff, R)), (8, lambda x, y: 2 * x * y * y % R)):
        got, rc = _run(probe.gpu_fr_ops, op, a, b, 32)
        assert rc == [0] * len(a)
        assert got == [f(x, y) for x, y in zip(a, b)], op
    got, _ = _run(probe.gpu_fr_ops, 3, a[:128], b[:128], 32)
    assert got == [inv(x) for x in a[:128]]
    _, rc = _run(probe.gpu_fr_ops, 4, a, b, 32)
    assert rc == [100 if x < R else 101 for x in a]


def test_fr_batch_inversion_on_device(probe):
    rnd = random.Random(2103)
    vals = [1, R - 1, 2] + [rnd.randrange(1, R) for _ in range(4093)]        # 4096: the length kzg.rs:368 inverts
    out = C.create_string_buffer(32 * len(vals)); rc = (C.c_int * 1)()
    assert probe.gpu_fr_batch_inv(len(vals), b"".join(v.to_bytes(32, "big") for v in vals), out, rc) == 0 and rc[0] == 0
    got = [int.from_bytes(out.raw[32 * i:32 * i + 32], "big") for i in range(len(vals))]
    assert got == [pow(v, -1, R) for v in vals]
    vals[77] = 0                                                            # a zero input is an error (utils.rs:70-72)
    assert probe.gpu_fr_batch_inv(len(vals), b"".join(v.to_bytes(32, "big") for v in vals), out, rc) == 0 and rc[0] == 3
