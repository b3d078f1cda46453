#!/bin/bash
# A/B of library variants on ONE box (run through gpurun): build variants into scratch/<name>.so, then
#   bash scripts/ab_variants.sh a.so b.so      -> per-kernel ms per training step from rocprofv3 stats, two rounds
#   AB_ARGS='--train-precision bf16x3_w16' picks the step's arithmetic; 'orig' = the library in the tree
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/nerf_amd/libnrf.so /tmp/libnrf_orig.so
trap 'cp /tmp/libnrf_orig.so $R/nerf_amd/libnrf.so' EXIT     # leave the tree's own library in place
for round in 1 2; do
for v in "$@"; do
  if [ $v = orig ]; then cp /tmp/libnrf_orig.so $R/nerf_amd/libnrf.so; else cp $R/scratch/$v $R/nerf_amd/libnrf.so; fi
  rm -rf /tmp/ab_$v
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -o t -- python3 $R/bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline --no-extras $AB_ARGS > /tmp/ab_$v.log 2>&1
  python3 - "$v" /tmp/ab_$v/t_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    n = r["Name"]
    if any(k in n for k in ("wgrad", "query_fwd", "dgrad")):
        print(sys.argv[1], n.split("(")[0][-40:], "%.3f ms/step" % (float(r["TotalDurationNs"]) / 7 / 1e6))
PY
done; done
