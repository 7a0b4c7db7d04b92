#!/usr/bin/env python3
"""Where a kernel waits for memory right behind the request: for every `s_waitcnt vmcnt(N)` of a kernel's gfx950 assembly (tools/kernel_resources.sh keeps it in
/tmp/kres/<file>.s), the number of instructions since the youngest load it has to wait for.  A wait a few instructions behind its load is a full memory latency
that nothing hides (k_eval's roots table under a branch was one: DESIGN.md section 4d).  usage: wait_scan.py /tmp/kres/k_prove.s [kernel-substring] [max-distance]"""
import re
import sys


def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    near = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    parts = re.split(r"^(_Z\w+):[^\n]*\n", txt, flags=re.M)          # [preamble, name, body, name, body, ...]; functions called by kernels have labels too
    for name, text in zip(parts[1::2], parts[2::2]):
        if want not in name or "s_endpgm" not in text:
            continue
        body = text[:text.index("s_endpgm")].split("\n")
        insts = [l.strip() for l in body if l.strip() and not l.strip().startswith((";", ".")) or re.match(r"^\.LBB", l.strip())]
        loads = []                                   # instruction index of every outstanding-counter load, in issue order
        hits = []
        for i, l in enumerate(insts):
            op = l.split()[0]
            if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
                loads.append(i)
            w = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", l)
            if w and loads:
                n = int(w.group(1))
                if n < len(loads):
                    youngest = loads[len(loads) - 1 - n]      # the load that must have landed
                    d = i - youngest
                    if d <= near:
                        hits.append((i, n, d, insts[youngest][:60]))
                    loads = loads[len(loads) - n:] if n else []
        print(f"{name[:70]}: {len(insts)} lines, {len(hits)} wait(s) within {near} instructions of their load")
        for i, n, d, ld in hits:
            print(f"    line {i}: vmcnt({n}) {d} instructions behind  {ld}")


if __name__ == "__main__":
    main()
