#!/usr/bin/env python3
"""Copy the judged summaries of a scripts/profile_gpu.sh run from gpurun_out/prof_<tag>/ into profiles/ and
build profiles/<round>_pmc_traffic.json (read by bench.py for roofline.traffic).
    python scripts/collect_profiles.py <tag> <round-prefix>      e.g.  r1c r1
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section), so fetch bytes are reported raw and doubled ("corrected")."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag, rnd):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    for w in ("render", "train", "dnerf", "dnerf_bf16", "dnerf_bf16x3", "render_bf16", "render_bf16x3"):
        p = os.path.join(src, f"stats_{w}", f"{w}_kernel_stats.csv")
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, f"{rnd}_{w}_kernel_stats.csv"))
    traffic = {"_note": "per-launch means from separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of "
                        "`bench.py --steps 1` (scripts/profile_gpu.sh); KiB * 1024; fetch_corrected = 2 * fetch_raw"}
    for w in ("render", "train", "dnerf", "render_bf16", "render_bf16x3"):
        per = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            p = os.path.join(src, f"pmc_{w}_{c}.summary.json")
            if not os.path.exists(p):
                continue
            shutil.copy(p, os.path.join(dst, f"{rnd}_pmc_{w}_{c}.summary.json"))
            for k, v in json.load(open(p)).items():
                if "at::" in k or "rocclr" in k or not k.endswith("_kernel") and "_kernel<" not in k:
                    continue
                name = k.replace("nrf::", "").strip()   # template arguments stay: instantiations are different kernels
                e = per.setdefault(name, {})
                e["launches"] = v[c]["launches"]
                if c == "FETCH_SIZE":
                    e["fetch_bytes_raw"] = v[c]["per_launch"] * 1024
                    e["fetch_bytes_corrected"] = 2 * v[c]["per_launch"] * 1024
                else:
                    e["write_bytes"] = v[c]["per_launch"] * 1024
        for e in per.values():
            if "fetch_bytes_corrected" in e and "write_bytes" in e:
                e["hbm_bytes"] = e["fetch_bytes_corrected"] + e["write_bytes"]
        if per:
            # steps executed by the profiled command (scripts/profile_gpu.sh: render 0+1, train 1+1, dnerf 0+1)
            per["_steps_profiled"] = {"render": 1, "train": 2, "dnerf": 1, "render_bf16": 1, "render_bf16x3": 1}[w]
            traffic[w] = per
    out_path = os.path.join(dst, f"{rnd}_pmc_traffic.json")
    if os.path.exists(out_path):
        old = json.load(open(out_path))
        for k, v in old.items():
            traffic.setdefault(k, v)
    with open(out_path, "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
    print(json.dumps(traffic, indent=1, sort_keys=True)[:3000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
