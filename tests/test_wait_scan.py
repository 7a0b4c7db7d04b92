"""tools/wait_scan.py reads the gfx950 assembly of a kernel and lists every `s_waitcnt vmcnt(N)` that sits a few instructions behind the load it has to wait
for (how round 5 found the request under a branch that cost k_eval a memory latency per step).  Checked here on a synthetic listing: a function label in
front of the kernel must not swallow it, an immediate wait is reported with its distance and its load, a wait that leaves the young loads in flight is not."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "wait_scan.py")

LISTING = """\t.text
_ZN3kzg6helperERNS_2FpE:                ; @_ZN3kzg6helperERNS_2FpE
\tv_add_u32_e32 v0, v1, v2
\ts_setpc_b64 s[30:31]
_ZN3kzg8k_sampleEPKhPi:                 ; @_ZN3kzg8k_sampleEPKhPi
; %bb.0:
\tglobal_load_dwordx4 v[4:7], v[0:1], off
\tglobal_load_dwordx4 v[8:11], v[0:1], off offset:16
\tv_mov_b32_e32 v20, v21
\ts_waitcnt vmcnt(1)
\tv_add_u32_e32 v4, v4, v5
""" + "".join(f"\tv_add_u32_e32 v{30 + (i % 8)}, v4, v5\n" for i in range(100)) + """\ts_waitcnt vmcnt(0)
\tv_add_u32_e32 v8, v8, v9
\tglobal_load_dword v12, v[2:3], off
\ts_waitcnt vmcnt(0)
\tv_add_u32_e32 v12, v12, v8
\ts_endpgm
"""


def test_wait_scan_reports_the_waits_right_behind_their_loads(tmp_path):
    src = tmp_path / "k.s"
    src.write_text(LISTING)
    out = subprocess.run([sys.executable, TOOL, str(src), "k_sample", "40"], capture_output=True, text=True, check=True).stdout
    lines = [l for l in out.splitlines() if l.strip()]
    assert lines[0].startswith("_ZN3kzg8k_sample") and "2 wait(s) within 40" in lines[0], out
    # the first wait (vmcnt(1)) is three instructions behind the FIRST load; the vmcnt(0) a hundred instructions later is not reported; the last one is
    assert "vmcnt(1) 3 instructions behind" in lines[1] and "global_load_dwordx4 v[4:7]" in lines[1], out
    assert "vmcnt(0) 1 instructions behind" in lines[2] and "global_load_dword v12" in lines[2], out
    assert "helper" not in out
