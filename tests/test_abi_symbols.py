"""CPU-side (-m "not gpu") checks of the drop-in boundary: libkzg355.so builds, loads without a GPU, exports every
entry point include/kzg355.h declares, fails loudly (KZG355_NO_DEVICE, never a CPU fallback) when no HIP device
exists, and the host-side mirror enforces the reference's length / hex rules before any FFI call."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(ROOT, "kzg_rust_amd", "libkzg355.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "kzg_rust_amd", "csrc"), "-j4"], check=True, stdout=subprocess.DEVNULL)
    from kzg_rust_amd import _lib
    return _lib.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "kzg355.h")).read()
    names = sorted(set(re.findall(r"\b(kzg355_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    from kzg_rust_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == names


def test_product_does_not_link_the_oracle():
    so = os.path.join(ROOT, "kzg_rust_amd", "libkzg355.so")
    out = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
    assert "liboracle" not in out
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    assert "okzg_" not in syms
    for root, _, files in os.walk(os.path.join(ROOT, "kzg_rust_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f


def test_no_device_is_an_error_not_a_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = lib.kzg355_load_trusted_setup(bytes(48 * 4096), 4096, bytes(96 * 65), 65, C.byref(h))
    assert rc == 6  # KZG355_NO_DEVICE
    rc = lib.kzg355_load_trusted_setup(bytes(48 * 4095), 4095, bytes(96 * 65), 65, C.byref(h))
    assert rc == 5  # count check precedes any device work (kzg.rs:49-62)


def test_host_mirror_type_rules():
    import kzg_rust_amd as kz
    with pytest.raises(kz.InvalidBytesLength):
        kz.Blob(bytes(131071))                       # kzg.rs:160-167
    with pytest.raises(kz.InvalidBytesLength):
        kz.Bytes48(bytes(47))                        # kzg.rs:130-137
    with pytest.raises(kz.BadArgs):
        kz.Bytes32(bytes(33))                        # kzg.rs:107-113
    with pytest.raises(kz.InvalidHexFormat):
        kz.KzgProof.from_hex("0xzz")                 # kzg.rs:82-86
    assert kz.Bytes32.from_hex("0x" + "11" * 32) == kz.Bytes32.from_hex("11" * 32)
    assert kz.KzgCommitment.from_hex("c0" + "00" * 47).to_bytes()[0] == 0xC0


def test_trusted_setup_json_helper(setup_bytes):
    """TrustedSetup (src/trusted_setup.rs): hex with/without 0x, truncation of G1 to 4096, round trip, length errors."""
    import json
    import kzg_rust_amd as kz
    g1, g2 = setup_bytes
    g1l = [g1[48 * i:48 * i + 48] for i in range(4096)]
    g2l = [g2[96 * i:96 * i + 96] for i in range(65)]
    text = json.dumps({"setup_G1": ["ignored"], "setup_G1_lagrange": ["0x" + p.hex() for p in g1l] + [g1l[0].hex()] * 3,
                       "setup_G2": [p.hex() for p in g2l], "roots_of_unity": []})
    ts = kz.TrustedSetup.from_json(text)
    assert ts.g1_len() == 4096 and ts.g2_len() == 65          # truncated (trusted_setup.rs:151)
    assert ts.g1_points() == g1l and ts.g2_points() == g2l
    assert kz.TrustedSetup.from_json(ts.to_json()).g1_points() == g1l
    with pytest.raises(kz.InvalidBytesLength):
        kz.TrustedSetup.from_json(json.dumps({"setup_G1_lagrange": ["0x00"], "setup_G2": []}))
    with pytest.raises(kz.InvalidHexFormat):
        kz.TrustedSetup.from_json(json.dumps({"setup_G1_lagrange": ["0xzz"], "setup_G2": []}))
