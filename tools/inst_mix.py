#!/usr/bin/env python3
"""Static VALU instruction mix of every kernel of libkzg355.so, from the gfx950 disassembly of the built objects, priced with the MEASURED
issue rates of the two instruction classes (profiles/r02/valu_issue_rates.txt, tools/ubench/valu_rates.hip on MI355X):

  full rate  1.12 ns per wave-instruction per SIMD   v_mov / v_add_u32 / v_sub / v_xor / v_and / v_or / v_bitop3 / v_fma_f32 / 32-bit shifts,
                                                      compares, v_cndmask, DPP moves ...: every 32-bit VOP1 / VOP2 / VOPC form
  half rate  1.78 ns                                  MEASURED: v_add3_u32, v_alignbit_b32, v_perm_b32, v_and_or_b32, carry adds (v_add_co / v_addc_co),
                                                      v_lshl_add_u64, v_lshrrev_b64, v_mad_u64_u32, v_mul_lo_u32, v_mul_hi_u32, v_fma_f64;
                                                      INFERRED from those (same encoding class, not timed one by one): every other 64-bit operation,
                                                      v_sub_co / v_subb_co, v_mad_i64_i32, v_mad_u32_u24 and the three-operand integer VOP3 forms
                                                      (v_lshl_or, v_lshl_add_u32, v_add_lshl, v_xad, v_or3, v_bfe, v_bfi, v_min3 / v_max3 / v_med3)

The "mix floor" of a kernel is what its OWN instruction stream would cost if every instruction issued at its class's measured plateau rate with
no stall at all:  floor_ns = f_full * 1.12 + f_half * 1.78  per wave-instruction per SIMD.  bench.py divides it by the achieved figure
(launch time x 1024 SIMDs / wave-instructions executed, the latter from the committed SQ_INSTS_VALU pass) -> roofline.alu.<kernel>.frac_of_mix_floor.
The mix is STATIC (every instruction of the kernel's code weighted once, whatever its trip count); the kernels of this path are one or two loop
bodies that make up nearly all of their code, so the static and the executed mix are close -- but it is an approximation and labelled as such.

usage: tools/inst_mix.py [out.json]     (reads kzg_rust_amd/csrc/k_*.o; needs /opt/rocm/lib/llvm/bin/llvm-objdump)
"""
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
FULL_NS, HALF_NS = 1.12, 1.78

HALF_MEASURED = ("v_add3_u32", "v_alignbit_b32", "v_perm_b32", "v_and_or_b32", "v_add_co_u32", "v_addc_co_u32", "v_lshl_add_u64", "v_lshrrev_b64",
                 "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_fma_f64")
HALF_INFERRED = ("v_sub_co_u32", "v_subb_co_u32", "v_subrev_co_u32", "v_subbrev_co_u32", "v_mad_i64_i32", "v_mad_u32_u24", "v_mad_i32_i24", "v_mul_hi_i32",
                 "v_lshl_or_b32", "v_lshl_add_u32", "v_add_lshl_u32", "v_xad_u32", "v_or3_b32", "v_bfe_u32", "v_bfe_i32", "v_bfi_b32", "v_min3_u32", "v_max3_u32",
                 "v_med3_u32", "v_min3_i32", "v_max3_i32", "v_med3_i32", "v_mul_u32_u24", "v_mul_i32_i24", "v_alignbyte_b32", "v_mbcnt_hi_u32_b32", "v_mbcnt_lo_u32_b32")


def classify(mn):
    base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", mn)
    if base in HALF_MEASURED or base in HALF_INFERRED:
        return "half"
    if re.search(r"_(b64|u64|i64|f64)$", base) and not base.startswith("v_cmp") and not base.startswith("v_cmpx"):
        return "half"
    if re.match(r"v_cmpx?_\w+_(u64|i64|f64)$", base):
        return "half"
    return "full"


def device_object(obj):
    """the gfx950 code object bundled in a hipcc .o (llvm-objdump --offloading writes it next to the .o)"""
    out = obj + ".0.hipv4-amdgcn-amd-amdhsa--gfx950"
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(obj):
        subprocess.run([OBJDUMP, "--offloading", obj], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def mix_of(obj):
    co = device_object(obj)
    txt = subprocess.run([OBJDUMP, "-d", "--demangle", co], check=True, capture_output=True, text=True).stdout
    per, cur = {}, None
    for ln in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", ln)
        if m:
            cur = m.group(1)
            per[cur] = {"full": 0, "half": 0, "salu": 0, "lds": 0, "vmem": 0, "half_top": {}}
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+(\w+)", ln)
        if not m:
            continue
        mn = m.group(1)
        d = per[cur]
        if mn.startswith("v_"):
            c = classify(mn)
            d[c] += 1
            if c == "half":
                d["half_top"][mn] = d["half_top"].get(mn, 0) + 1
        elif mn.startswith("s_"):
            d["salu"] += 1
        elif mn.startswith("ds_"):
            d["lds"] += 1
        elif mn.startswith(("global_", "buffer_", "scratch_", "flat_")):
            d["vmem"] += 1
    return per


def short_name(sym):
    m = re.match(r"(?:void )?(?:kzg::)?(k_\w+(?:<[^>]*>)?)", sym)
    return m.group(1) if m else None


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    result = {"note": __doc__.split("usage:")[0].strip(), "rates_ns": {"full": FULL_NS, "half": HALF_NS}, "per_kernel": {}, "per_object_all_symbols": {}}
    for obj in sorted(glob.glob(os.path.join(ROOT, "kzg_rust_amd", "csrc", "k_*.o"))):
        per = mix_of(obj)
        tot = {"full": 0, "half": 0}
        for sym, d in per.items():
            tot["full"] += d["full"]; tot["half"] += d["half"]
            k = short_name(sym)
            if not k or d["full"] + d["half"] == 0:
                continue
            v = d["full"] + d["half"]
            top = sorted(d["half_top"].items(), key=lambda kv: -kv[1])[:4]
            result["per_kernel"][k] = {"valu_static": v, "full": d["full"], "half": d["half"], "half_frac": round(d["half"] / v, 4),
                                       "mix_floor_ns": round((d["full"] * FULL_NS + d["half"] * HALF_NS) / v, 4), "salu_static": d["salu"], "lds_static": d["lds"],
                                       "vmem_static": d["vmem"], "half_top": dict(top), "object": os.path.basename(obj)}
        v = tot["full"] + tot["half"]
        if v:
            # objects whose kernels call out-of-line device routines (k_pairing.o: the tower routines): the mix over all of the object's code
            result["per_object_all_symbols"][os.path.basename(obj)] = {"valu_static": v, "half_frac": round(tot["half"] / v, 4),
                                                                       "mix_floor_ns": round((tot["full"] * FULL_NS + tot["half"] * HALF_NS) / v, 4)}
    txt = json.dumps(result, indent=1)
    if out:
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        open(out, "w").write(txt + "\n")
    for k, d in sorted(result["per_kernel"].items(), key=lambda kv: -kv[1]["valu_static"]):
        print(f"{k[:48]:48s} valu {d['valu_static']:7d}  half {d['half_frac']:.3f}  floor {d['mix_floor_ns']:.3f} ns  salu {d['salu_static']:6d}  lds {d['lds_static']:5d}  vmem {d['vmem_static']:5d}")


if __name__ == "__main__":
    main()
