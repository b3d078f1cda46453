"""Per-kernel register / scratch usage from `hipcc -Rpass-analysis=kernel-resource-usage` output files (build check:
a spilled kernel shows ScratchSize > 0).  usage: python scripts/kernel_resources.py /tmp/mlp_fused.res ..."""
import re
import subprocess
import sys

for path in sys.argv[1:]:
    txt = open(path).read()
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split("\n")[0].strip().split()[0]

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return m.group(1) if m else "?"
        try:
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
        scratch, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
        print(f"{name[:120]:120s} VGPR {g('VGPRs'):>4} AGPR {g('AGPRs'):>4} scratch {scratch:>5} occ {occ}")
