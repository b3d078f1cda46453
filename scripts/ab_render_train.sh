#!/bin/bash
# same-box A/B of library variants: bf16x3 render and the w16 training step, two rounds
R=$GRAFT_REPO_ROOT
cp $R/nerf_amd/libnrf.so /tmp/libnrf_orig.so
trap 'cp /tmp/libnrf_orig.so $R/nerf_amd/libnrf.so' EXIT     # leave the tree's own library in place
for round in 1 2; do
for v in orig "$@"; do
  if [ $v = orig ]; then cp /tmp/libnrf_orig.so $R/nerf_amd/libnrf.so; else cp $R/scratch/$v $R/nerf_amd/libnrf.so; fi
  timeout -k 10 120 python3 $R/bench.py --precision bf16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v render_x3 ms', round(d['ms_per_step'],2))" || exit 1
  timeout -k 10 120 python3 $R/bench.py --workload train --train-precision bf16x3_w16 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v train_w16 ms', round(d['ms_per_step'],3))" || exit 1
done; done
