#!/bin/bash
# N = 2 rehearsal of every bench workload on ONE GPU (both ranks on cuda:0, gloo): exercises the multi-rank code paths only.
# bench.py starts its own ranks: the command is the driver's (`python bench.py --gpus N ...`), no external launcher.
set -e
cd $GRAFT_REPO_ROOT
export NRF_BENCH_BACKEND=gloo
for w in render train fern dnerf; do
  echo "== $w"
  timeout -k 10 300 python bench.py --gpus 2 --steps 2 --warmup 1 --workload $w 2>&1 | grep -E '^\{|Error|error|Traceback|bench.py:' | cut -c1-400
done
