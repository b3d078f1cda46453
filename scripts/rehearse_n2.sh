#!/bin/bash
# N = 2 rehearsal of every bench workload on ONE GPU (both ranks on cuda:0, gloo): exercises the multi-rank code paths only
set -e
cd $GRAFT_REPO_ROOT
export NRF_BENCH_BACKEND=gloo
for w in render train fern dnerf; do
  echo "== $w"
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --workload $w 2>&1 | grep -E '^\{|Error|error|Traceback' | cut -c1-400
done
