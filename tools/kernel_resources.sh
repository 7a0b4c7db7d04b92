#!/bin/bash
# Register / scratch / LDS figures of every kernel of a .hip file, from the gfx950 assembly hipcc emits (device pass only).
# usage: tools/kernel_resources.sh kzg_rust_amd/csrc/k_pairing.hip [extra flags]   (keeps the assembly in /tmp/kres/<name>.s)
SRC=$1; shift
mkdir -p /tmp/kres
OUT=/tmp/kres/$(basename ${SRC%.hip}).s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result --cuda-device-only -S "$@" $SRC -o $OUT || exit 1
python3 - "$OUT" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, flags=re.S):
    name, body = m.group(1), m.group(2)
    get = lambda k: (re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body) or [None, "?"])[1]
    print(name[:60].ljust(60), "vgpr", get("next_free_vgpr"), "accum_off", get("accum_offset"), "sgpr", get("next_free_sgpr"), "scratch(private)", get("private_segment_fixed_size"), "lds", get("group_segment_fixed_size"))
for m in re.finditer(r"; Function info:|\.vgpr_spill_count:\s+(\d+)", txt):
    pass
for m in re.finditer(r"- \.agpr_count:.*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", txt, flags=re.S):
    print("meta", m.group(1)[:60].ljust(60), "private", m.group(2), "sgpr_spill", m.group(3), "vgpr", m.group(4), "vgpr_spill", m.group(5))
PY
