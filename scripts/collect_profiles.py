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


def mfma_util(src, dst, rnd):
    """profiles/<round>_mfma_util.json from the `sq_all` passes of scripts/profile_gpu.sh: per MFMA kernel
      mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES)   (MFMA_BUSY counts cycles, WAVE_CYCLES quad-cycles;
                       all of these kernels run one wave per SIMD, so wave time == SIMD time)
      clock_ghz      = GRBM_GUI_ACTIVE / 8 / duration                     (the counter sums the 8 XCDs; MI355X_MICROARCH.md)
      mfma_mops      = SQ_INSTS_VALU_MFMA_MOPS_{F32,BF16} per launch."""
    util = {"_note": "one rocprofv3 --pmc pass per workload (7 SQ counters + GRBM_GUI_ACTIVE), scripts/profile_gpu.sh sq_all; "
                     "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES); clock_ghz = GRBM_GUI_ACTIVE / 8 / duration; "
                     "profiled passes run slower than un-profiled ones (counter collection serialises launches)"}
    for w in ("render", "render_bf16", "render_bf16x3", "train", "train_bf16x3", "train_bf16x3_a16", "train_bf16x3_w16", "dnerf", "dnerf_bf16"):
        p = os.path.join(src, f"pmc_{w}_sq.summary.json")
        if not os.path.exists(p):
            continue
        shutil.copy(p, os.path.join(dst, f"{rnd}_pmc_{w}_sq.summary.json"))
        per = {}
        for k, v in json.load(open(p)).items():
            if "SQ_VALU_MFMA_BUSY_CYCLES" not in v or v["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] <= 0:
                continue
            g = lambda c: v[c]["per_launch"] if c in v else None  # noqa: E731
            e = {"launches": v["SQ_WAVE_CYCLES"]["launches"],
                 "mfma_busy_frac": g("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * g("SQ_WAVE_CYCLES")),
                 "mfma_mops_f32": g("SQ_INSTS_VALU_MFMA_MOPS_F32"), "mfma_mops_bf16": g("SQ_INSTS_VALU_MFMA_MOPS_BF16")}
            if "_duration_ns" in v and g("GRBM_GUI_ACTIVE"):
                e["duration_ms"] = v["_duration_ns"]["per_launch"] * 1e-6
                e["clock_ghz"] = g("GRBM_GUI_ACTIVE") / 8.0 / v["_duration_ns"]["per_launch"]
            per[k.replace("nrf::", "").strip()] = e
        if per:
            util[w] = per
    if len(util) > 1:
        out_path = os.path.join(dst, f"{rnd}_mfma_util.json")
        if os.path.exists(out_path):        # a partial run (one workload re-profiled) keeps the round's other workloads
            for k, v in json.load(open(out_path)).items():
                util.setdefault(k, v)
        with open(out_path, "w") as f:
            json.dump(util, f, indent=1, sort_keys=True)
        print(json.dumps(util, indent=1, sort_keys=True)[:4000])


def main(tag, rnd):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    for w in ("render", "train", "train_bf16x3", "train_bf16x3_a16", "train_bf16x3_w16", "dnerf", "dnerf_bf16", "dnerf_bf16x3", "render_bf16", "render_bf16x3",
              "dnerf_straight", "fern"):
        p = os.path.join(src, f"stats_{w}", f"{w}_kernel_stats.csv")
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, f"{rnd}_{w}_kernel_stats.csv"))
    traffic = {"_note": "per-launch means from separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of "
                        "`bench.py --steps 1` (scripts/profile_gpu.sh); KiB * 1024; fetch_corrected = 2 * fetch_raw"}
    for w in ("render", "train", "train_bf16x3", "train_bf16x3_a16", "train_bf16x3_w16", "dnerf", "render_bf16", "render_bf16x3"):
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
            per["_steps_profiled"] = {"render": 1, "train": 2, "train_bf16x3": 2, "train_bf16x3_a16": 2, "train_bf16x3_w16": 2, "dnerf": 1, "render_bf16": 1, "render_bf16x3": 1}[w]
            traffic[w] = per
    mfma_util(src, dst, rnd)
    if len(traffic) == 1:        # an SQ-only run: leave the traffic file of the round alone
        return
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
