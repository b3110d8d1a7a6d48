#!/bin/bash
# Regenerates the committed profile artifacts of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01      -> gpurun_out/prof_r01/{bench_kernel_stats.csv,pmc_summary.json,bench_default.json,...}
# Copy the files you want judged into profiles/ afterwards (gpurun_out/ is scratch).
set -u
R=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$R
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-extra --no-cells --no-live-pmc"
rocprofv3 --kernel-trace --stats -d /tmp/kt_$R -o kt --output-format csv -- $BENCH > "$OUT/kt.log" 2>&1
cp "$(find /tmp/kt_$R -name '*kernel_stats.csv' | head -1)" "$OUT/bench_kernel_stats.csv"
for V in "b512:--batch 512" "f16x3:--precision f16x3" "b128:--batch 128"; do
  TAG=${V%%:*}; FLAGS=${V#*:}
  rocprofv3 --kernel-trace --stats -d /tmp/kt_${R}_$TAG -o kt --output-format csv -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-extra --no-cells --no-live-pmc $FLAGS > "$OUT/kt_$TAG.log" 2>&1
  cp "$(find /tmp/kt_${R}_$TAG -name '*kernel_stats.csv' | head -1)" "$OUT/bench_${TAG}_kernel_stats.csv"
done
SHORT="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-split-extra --no-cells --no-live-pmc"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_${R}_$i -o p --output-format csv -- $SHORT > "$OUT/pmc_$i.log" 2>&1
  cp "$(find /tmp/pmc_${R}_$i -name '*counter_collection.csv' | head -1)" "/tmp/pmc_${R}_$i.csv"
done
python "$REPO/tools/pmc_summarize.py" "$OUT/pmc_summary.json" k_flow_gemm /tmp/pmc_${R}_1.csv /tmp/pmc_${R}_2.csv /tmp/pmc_${R}_3.csv
# SQ / LDS counters of the other regimes' kernels (f16x3 LDS-DMA kernel, small-batch kernels)
for V in "f16x3:--precision f16x3:k_split_gemm" "b128:--batch 128:k_flow_gemm_skinny"; do
  TAG=${V%%:*}; REST=${V#*:}; FLAGS=${REST%%:*}; KERN=${REST#*:}
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_${R}_$TAG -o p --output-format csv -- $SHORT $FLAGS > "$OUT/pmc_$TAG.log" 2>&1
  python "$REPO/tools/pmc_summarize.py" "$OUT/pmc_summary_$TAG.json" $KERN "$(find /tmp/pmc_${R}_$TAG -name '*counter_collection.csv' | head -1)"
done
cd "$REPO"
mkdir -p profiles && cp "$OUT/pmc_summary.json" "profiles/${R}_pmc_summary.json"   # so the bench below reports this traffic
python bench.py > "$OUT/bench_default.log" 2>&1; tail -1 "$OUT/bench_default.log" > "$OUT/bench_default.json"
python bench.py --precision f16x3 --no-cpu-baseline --no-split-extra --no-cells --no-live-pmc 2>&1 | tail -1 > "$OUT/bench_f16x3.json"
echo "profile_round: done -> $OUT"; ls -la "$OUT"
