#!/usr/bin/env python3
"""Static check of the inline-assembly LDS reads (ds_read_b64_tr_b16, mlp_bwd.hip): the compiler does not know that their
destination registers are written LATER, so nothing may touch those registers -- read OR write -- between a read and the
next `s_waitcnt lgkmcnt(0)`.  (A register the compiler believes dead gets reused; the late return then clobbers the new
value.  Found once: the read-ahead of the last step of wgrad_steps_h, fixed by a wait the fragments pass through.)
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/bwd.s nerf_amd/csrc/mlp_bwd.hip
    python scripts/check_tr_reads.py /tmp/bwd.s"""
import re
import sys


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


txt = open(sys.argv[1]).read()
total = 0
for m in re.finditer(r"^(_ZN3nrf\w+):", txt, re.M):
    start, end = m.end(), txt.index("s_endpgm", m.end())
    pending, bad, nreads = set(), 0, 0
    for line in txt[start:end].split("\n"):
        line = line.strip()
        if not line or line.startswith((".", ";")) or line.endswith(":"):
            continue
        op = line.split()[0]
        toks = [t.strip().rstrip(",") for t in re.split(r"[ ,]+", line.split(";")[0])[1:]]
        if op == "ds_read_b64_tr_b16":
            pending |= regs(toks[0])
            nreads += 1
        elif op == "s_waitcnt" and "lgkmcnt(0)" in line:
            pending = set()
        elif pending and any(regs(t) & pending for t in toks):
            bad += 1
            if bad <= 3:
                print("  touched before the wait:", line[:110])
    if nreads:
        print(f"{m.group(1)[:60]}: {nreads} transposing reads, {bad} violations")
        total += bad
sys.exit(1 if total else 0)
