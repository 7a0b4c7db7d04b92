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


def test_library_exports_the_c_abi_and_nothing_else(lib):
    """Every dynamic symbol libkzg355.so defines is an entry point of include/kzg355.h (csrc/exports.map): no unprefixed helpers, no C++
    standard-library instantiations, no kernel stubs -- nothing that could collide inside the program it is linked into."""
    so = os.path.join(ROOT, "kzg_rust_amd", "libkzg355.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    assert names and not [n for n in names if not n.startswith("kzg355_")]
    from kzg_rust_amd import _lib
    assert sorted(names) == sorted(_lib.EXPORTED_SYMBOLS)


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


def test_options_struct_defaults_env_overrides_and_layout(lib, monkeypatch):
    """kzg355_options (include/kzg355.h): defaults, the KZG355_* overrides folded in by kzg355_options_from_env only, and the ctypes
    mirror's layout (struct_size is sizeof as the C side sees it)."""
    from kzg_rust_amd import _lib
    o = _lib.Options()
    lib.kzg355_options_default(C.byref(o))
    assert o.struct_size == C.sizeof(_lib.Options)
    assert (o.device, o.self_test, o.msm_bits, o.host_hash, o.host_hash_max_blobs, o.lc_chain_from) == (-1, 1, 0, 0, 0, 0)
    for k, v in {"KZG355_MSM_BITS": "14", "KZG355_HOST_HASH": "off", "KZG355_HOST_HASH_MAX": "77", "KZG355_LINCOMB": "bucket", "KZG355_SPLIT": "4,3",
                 "KZG355_SELFTEST": "0", "KZG355_DEVICE": "2", "KZG355_EXCHANGE": "peer", "KZG355_PAIRING_2W_UPTO": "0"}.items():
        monkeypatch.setenv(k, v)
    lib.kzg355_options_from_env(C.byref(o))
    assert (o.msm_bits, o.host_hash, o.host_hash_max_blobs, o.lincomb_form, o.split_parts, o.split_streams, o.self_test, o.device, o.exchange,
            o.pairing_two_wave_upto) == (14, -1, 77, 2, 4, 3, 0, 2, 1, -1)
    monkeypatch.setenv("KZG355_MSM", "bucket")
    lib.kzg355_options_from_env(C.byref(o))
    assert o.msm_bits == 8
    lib.kzg355_options_default(C.byref(o))                       # the explicit form never reads the environment
    assert (o.msm_bits, o.host_hash, o.self_test) == (0, 0, 1)
    # every KZG355_* override kzg355_options_from_env knows, name by name -> the field it sets (the library's one reader of the environment: options.hip)
    table = {"KZG355_MSM_EAGER": ("1", "msm_eager", 1), "KZG355_MSM_GLV": ("off", "msm_glv", -1), "KZG355_PAIRING": ("lane", "pairing_lane", 1),
             "KZG355_HOST_THREADS": ("5", "host_threads", 5), "KZG355_HOST_SHA": ("portable", "host_sha", 1), "KZG355_HOST_RHASH": ("off", "host_rhash", -1),
             "KZG355_HOST_RHASH_MAX": ("33", "host_rhash_max_records", 33), "KZG355_HOST_HASH_DEVICE_MAX": ("0", "host_hash_device_max_blobs", -1),
             "KZG355_CHUNK_MB": ("7", "chunk_mb", 7), "KZG355_CHUNKS_IN_FLIGHT": ("2", "chunks_in_flight", 2), "KZG355_STAGING": ("ring", "staging_ring", 1),
             "KZG355_PAIRING_HARD12_FROM": ("0", "pairing_hard12_from", -1), "KZG355_LC_CHAIN_FROM": ("9", "lc_chain_from", 9),
             "KZG355_RHASH_LANES_FROM": ("11", "rhash_lanes_from", 11), "KZG355_CHALLENGE": ("2w", "challenge_form", 2), "KZG355_VERIFY_ONLY": ("1", "verify_only", 1),
             "KZG355_SUBMIT": ("pipeline", "submit_sets", 2), "KZG355_QUOTIENT_FORM": ("6", "quotient_form", 6), "KZG355_MILLER_SEGMENTS": ("3", "miller_segments", 3),
             "KZG355_FORCE_MULTI": ("1", "force_multi", 1), "KZG355_FORCE_SHARDED": ("1", "force_sharded", 1), "KZG355_MSM": ("wide", "msm_require_wide", 1)}
    for k in list(os.environ):
        if k.startswith("KZG355_"):
            monkeypatch.delenv(k)
    for name, (value, field, want) in table.items():
        monkeypatch.setenv(name, value)
        lib.kzg355_options_from_env(C.byref(o))
        assert getattr(o, field) == want, (name, field, getattr(o, field))
        monkeypatch.delenv(name)
        lib.kzg355_options_from_env(C.byref(o))
        assert getattr(o, field) == (1 if field == "self_test" else -1 if field == "device" else 0), name
    # ... and nothing in the library reads the environment outside options.hip
    csrc = os.path.join(ROOT, "kzg_rust_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h", ".cpp", ".inc")) and f != "options.hip":
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f


def test_load_ex_checks_counts_before_any_device_work(lib):
    import torch
    from kzg_rust_amd import _lib
    h = C.c_void_p()
    o = _lib.Options()
    lib.kzg355_options_default(C.byref(o))
    assert lib.kzg355_load_trusted_setup_ex(bytes(48 * 4095), 4095, bytes(96 * 65), 65, None, 0, C.byref(o), C.byref(h)) == 5
    if not torch.cuda.is_available():
        assert lib.kzg355_load_trusted_setup_ex(bytes(48 * 4096), 4096, bytes(96 * 65), 65, None, 0, None, C.byref(h)) == 6


def test_header_is_plain_c_and_a_c_program_links_against_the_library(lib, tmp_path):
    """The boundary is a C ABI: include/kzg355.h compiles as C11 with -pedantic (what a cgo / bindgen / ctypes consumer sees), and a C program that calls
    through it links against libkzg355.so and -- without a GPU -- gets KZG355_NO_DEVICE from the load, never a fallback."""
    import torch
    src = tmp_path / "consumer.c"
    src.write_text('#include "kzg355.h"\n#include <stdio.h>\n'
                   'int main(void) {\n'
                   '    kzg355_options o; kzg355_options_default(&o);\n'
                   '    if (o.struct_size != sizeof o || o.device != -1 || o.self_test != 1) return 10;\n'
                   '    static unsigned char g1[48 * 4096], g2[96 * 65];\n'
                   '    kzg355_settings *s = 0;\n'
                   '    int rc = kzg355_load_trusted_setup(g1, 4095, g2, 65, &s);      /* wrong count: BADARGS / INVALID_TRUSTED_SETUP before any device work */\n'
                   '    if (rc == KZG355_OK || s) return 11;\n'
                   '    rc = kzg355_load_trusted_setup(g1, 4096, g2, 65, &s);          /* all-zero bytes are no valid points; without a GPU: NO_DEVICE */\n'
                   '    printf("%s|%d\\n", kzg355_version(), rc);\n'
                   '    return rc == KZG355_OK ? 12 : 0;\n'
                   '}\n')
    exe = tmp_path / "consumer"
    so_dir = os.path.join(ROOT, "kzg_rust_amd")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L" + so_dir, "-lkzg355", "-Wl,-rpath," + so_dir], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    ver, rc = r.stdout.strip().rsplit("|", 1)
    assert ver.startswith("kzg355")
    if not torch.cuda.is_available():
        assert int(rc) == 6, rc                                   # KZG355_NO_DEVICE
