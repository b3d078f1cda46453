#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats + separate PMC passes for the render and train
# benches.  Outputs land under gpurun_out/prof_<tag>/; summaries are copied into profiles/ by hand.
#   bash scripts/profile_gpu.sh <tag> [stats|pmc|sq|list] ...
set -e -o pipefail
TAG=${1:?tag}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
RENDER="$ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
TRAIN="$ROOT/bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline"
for what in "$@"; do
  case $what in
    list)
      timeout -s KILL 60 rocprofv3 -L > "$OUT/counters.txt" 2>&1 || true ;;
    stats)
      timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_render" -o render -- python3 $RENDER > "$OUT/render_stats.log" 2>&1
      timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_train" -o train -- python3 $TRAIN > "$OUT/train_stats.log" 2>&1 ;;
    train_x3)   # the training step on the bf16x3 arithmetic: kernel stats, HBM traffic (two PMC passes), SQ pass
      TX3="$ROOT/bench.py --workload train --train-precision bf16x3"
      timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_train_bf16x3" -o train_bf16x3 -- python3 $TX3 --steps 5 --warmup 2 > "$OUT/train_bf16x3_stats.log" 2>&1
      for c in FETCH_SIZE WRITE_SIZE; do
        timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_train_bf16x3_$c" -o train_bf16x3 -- python3 $TX3 --steps 1 --warmup 1 > "$OUT/pmc_train_bf16x3_$c.log" 2>&1
      done
      timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_train_bf16x3_sq" -o train_bf16x3 -- python3 $TX3 --steps 2 --warmup 1 > "$OUT/pmc_train_bf16x3_sq.log" 2>&1 ;;
    train_a16|train_w16)   # bf16x3 training with f16 workspaces (a16: activations; w16: gradient blocks too): stats, HBM traffic, SQ
      M=${what#train_}
      TA="$ROOT/bench.py --workload train --train-precision bf16x3_$M"
      timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_train_bf16x3_$M" -o train_bf16x3_$M -- python3 $TA --steps 5 --warmup 2 > "$OUT/train_bf16x3_${M}_stats.log" 2>&1
      for c in FETCH_SIZE WRITE_SIZE; do
        timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_train_bf16x3_${M}_$c" -o train_bf16x3_$M -- python3 $TA --steps 1 --warmup 1 > "$OUT/pmc_train_bf16x3_${M}_$c.log" 2>&1
      done
      timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_train_bf16x3_${M}_sq" -o train_bf16x3_$M -- python3 $TA --steps 2 --warmup 1 > "$OUT/pmc_train_bf16x3_${M}_sq.log" 2>&1 ;;
    stats_train)
      timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_train" -o train -- python3 $TRAIN --no-extras > "$OUT/train_stats.log" 2>&1
      head -6 "$OUT"/stats_train/train_kernel_stats.csv | cut -c1-150 ;;
    pmc)
      for c in FETCH_SIZE WRITE_SIZE; do
        timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_render_$c" -o render -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_render_$c.log" 2>&1
        timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_train_$c" -o train -- python3 $ROOT/bench.py --workload train --steps 1 --warmup 1 > "$OUT/pmc_train_$c.log" 2>&1
      done ;;
    sq)   # SQ_COUNTERS="A B C" (<= 8 SQ counters), EXTRA="--precision bf16" etc.
      timeout -s KILL 240 rocprofv3 --pmc $SQ_COUNTERS --output-format csv -d "$OUT/pmc_render_sq" -o render -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline $EXTRA > "$OUT/pmc_render_sq.log" 2>&1 ;;
    sq_all)   # MFMA-utilisation evidence: ONE pass (7 SQ + 1 GRBM counters) per workload; scripts/collect_profiles.py
              # turns them into profiles/<round>_mfma_util.json (matrix-pipe busy fraction, held shader clock)
      SQC="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
      for P in fp32 bf16 bf16x3; do
        W=render_$P; [ $P = fp32 ] && W=render
        timeout -s KILL 240 rocprofv3 --pmc $SQC --output-format csv -d "$OUT/pmc_${W}_sq" -o $W -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --precision $P > "$OUT/pmc_${W}_sq.log" 2>&1
      done
      timeout -s KILL 240 rocprofv3 --pmc $SQC --output-format csv -d "$OUT/pmc_train_sq" -o train -- python3 $ROOT/bench.py --workload train --steps 2 --warmup 1 > "$OUT/pmc_train_sq.log" 2>&1
      for P in fp32 bf16; do
        W=dnerf_$P; [ $P = fp32 ] && W=dnerf
        timeout -s KILL 240 rocprofv3 --pmc $SQC --output-format csv -d "$OUT/pmc_${W}_sq" -o $W -- python3 $ROOT/bench.py --workload dnerf --steps 1 --warmup 0 --precision $P > "$OUT/pmc_${W}_sq.log" 2>&1
      done ;;
    stats_bf16|stats_bf16x3)
      P=${what#stats_}
      timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_render_$P" -o render_$P -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision $P > "$OUT/render_${P}_stats.log" 2>&1 ;;
    pmc_bf16|pmc_bf16x3)
      P=${what#pmc_}
      for c in FETCH_SIZE WRITE_SIZE; do
        timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_render_${P}_$c" -o render_$P -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --precision $P > "$OUT/pmc_render_${P}_$c.log" 2>&1
      done ;;
    stats_straight)   # network_type='straightforward' (render fp32 / bf16x3, training step) and the LayerNorm kernel
      timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_dnerf_straight" -o dnerf_straight -- python3 $ROOT/scripts/bench_straight.py > "$OUT/dnerf_straight_stats.log" 2>&1 ;;
    stats_fern)    # BASELINE configs[3]: the 1008x756 NDC view (configs/fern.txt), 64+128 samples, f32
      timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_fern" -o fern -- python3 $ROOT/bench.py --workload fern --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/fern_stats.log" 2>&1 ;;
    stats_dnerf)   # the deformation chain at the three precisions, one kernel-trace run each
      for P in fp32 bf16 bf16x3; do
        W=dnerf_$P; [ $P = fp32 ] && W=dnerf
        timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$W" -o $W -- python3 $ROOT/bench.py --workload dnerf --steps 2 --warmup 1 --precision $P > "$OUT/${W}_stats.log" 2>&1
      done ;;
  esac
done
for d in "$OUT"/pmc_*/; do [ -d "$d" ] && python3 $ROOT/scripts/pmc_summary.py "$d" > "${d%/}.summary.json"; done
find "$OUT" \( -name '*.db' -o -name '*_trace.csv' -o -name '*counter_collection.csv' \) -size +2M -delete   # keep the merge-back small
ls -R "$OUT" | head -60
