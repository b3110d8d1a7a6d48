#!/bin/bash
# Regenerates the committed profile artifacts of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r05      -> gpurun_out/prof_r04/{bench_kernel_stats.csv,pmc_summary.json,bench_default.json,...}
# Copy the files you want judged into profiles/ afterwards (gpurun_out/ is scratch).
# Every PMC pass is its own run with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one TCC pass).
set -u
R=${1:-r06}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$R
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-split-extra --no-cells --no-live-pmc"
SQSET="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
# kernel-trace statistics: the default workload (B = 4096: the row-owner launch), the same batch on the per-layer kernels, and the other cells
# (200 steps: behind every idle gap the chip needs ~10 launches / 30 ms to come back to its clock - tools/launch_timing_check.py -
# and a run has three such gaps; with 200 steps they stay a percent of the row-owner launch's average)
for V in "default:" "per_layer:--gemm-variant 180" "b512:--batch 512" "b128:--batch 128" "f16x3:--precision f16x3" "fetcharm8192:--model fetch_arm__large__mh186_9.25m --batch 8192"; do
  TAG=${V%%:*}; FLAGS=${V#*:}
  STEPS=200; [ "$TAG" = default ] && STEPS=1000   # (the headline: 1000 launches, so that the clock ramp behind the run's idle gaps is 1 % of the raw average)
  rocprofv3 --kernel-trace --stats -d /tmp/kt_${R}_$TAG -o kt --output-format csv -- python $REPO/bench.py --steps $STEPS --warmup 10 $COMMON $FLAGS > "$OUT/kt_$TAG.log" 2>&1
  NAME=bench_${TAG}_kernel_stats.csv; [ "$TAG" = default ] && NAME=bench_kernel_stats.csv
  cp "$(find /tmp/kt_${R}_$TAG -name '*kernel_stats.csv' | head -1)" "$OUT/$NAME"
  # launch by launch: median and steady-state mean of the dominant kernel (tools/kernel_trace_steady.py)
  KERN=k_flow_cluster; case "$TAG" in default|fetcharm8192) KERN=k_flow_rowowner;; per_layer) KERN="k_flow_gemm<";; f16x3) KERN=k_split_gemm;; esac
  SNAME=bench_${TAG}_kernel_steady.json; [ "$TAG" = default ] && SNAME=bench_kernel_steady.json
  python "$REPO/tools/kernel_trace_steady.py" "$(find /tmp/kt_${R}_$TAG -name '*kernel_trace.csv' | head -1)" "$KERN" "$OUT/$SNAME" > /dev/null 2>&1
done
# counters: FETCH_SIZE | WRITE_SIZE | SQ set, per regime, each pass a separate run
for V in "default::k_flow_rowowner" "per_layer:--gemm-variant 180:k_flow_gemm" "b512:--batch 512:k_flow_cluster" "b128:--batch 128:k_flow_cluster" "b16:--batch 16:k_flow_cluster" "f16x3:--precision f16x3:k_split_gemm" "fetcharm8192:--model fetch_arm__large__mh186_9.25m --batch 8192:k_flow_rowowner"; do
  TAG=${V%%:*}; REST=${V#*:}; FLAGS=${REST%%:*}; KERN=${REST#*:}
  i=0; FILES=""
  for C in "FETCH_SIZE" "WRITE_SIZE" "$SQSET"; do
    i=$((i+1))
    rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_${R}_${TAG}_$i -o p --output-format csv -- python $REPO/bench.py --steps 3 --warmup 1 $COMMON $FLAGS > "$OUT/pmc_${TAG}_$i.log" 2>&1
    F="$(find /tmp/pmc_${R}_${TAG}_$i -name '*counter_collection.csv' | head -1)"
    [ -n "$F" ] && FILES="$FILES $F"
  done
  NAME=pmc_summary_$TAG.json; [ "$TAG" = default ] && NAME=pmc_summary.json
  python "$REPO/tools/pmc_summarize.py" "$OUT/$NAME" $KERN $FILES
done
cd "$REPO"
mkdir -p profiles
for F in "$OUT"/pmc_summary*.json "$OUT"/bench_*kernel_steady.json "$OUT"/bench_*kernel_stats.csv; do cp "$F" "profiles/${R}_$(basename "$F")"; done   # so the bench below reports THIS run's figures
python bench.py > "$OUT/bench_default.log" 2>&1; tail -1 "$OUT/bench_default.log" > "$OUT/bench_default.json"
python bench.py --precision f16x3 $COMMON 2>&1 | tail -1 > "$OUT/bench_f16x3.json"
echo "profile_round: done -> $OUT"; ls -la "$OUT"
