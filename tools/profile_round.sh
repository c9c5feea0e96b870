#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the bench step for both models, and three separate PMC passes per model
#   (FETCH_SIZE | WRITE_SIZE | MFMA busy + clocks) as MI355X_MICROARCH.md prescribes.  Output: gpurun_out/prof_<tag>/.
set -e
trap 'rm -rf $OUT/stats_* $OUT/pmc_* 2>/dev/null' EXIT
R=$PWD
TAG=${1:-r01_d}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
COMMON="--stream-requests 0 --mix 0 --no-cpu-baseline --no-roofline --no-sd3 --no-stages --no-parity --no-cached-mix --no-two-model"
for M in sdxl sd3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$M -o $M -- python3 $R/bench.py --model $M --steps 5 --warmup 2 $COMMON > $OUT/stats_$M.log 2>&1
  echo "stats $M done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$M -- python3 $R/bench.py --model $M --steps 1 --warmup 1 $COMMON > $OUT/pmc_fetch_$M.log 2>&1
  echo "fetch $M done"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$M -- python3 $R/bench.py --model $M --steps 1 --warmup 1 $COMMON > $OUT/pmc_write_$M.log 2>&1
  echo "write $M done"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma_$M -- python3 $R/bench.py --model $M --steps 1 --warmup 1 $COMMON > $OUT/pmc_mfma_$M.log 2>&1
  echo "mfma $M done"
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2_$M -- python3 $R/bench.py --model $M --steps 1 --warmup 1 $COMMON > $OUT/pmc_l2_$M.log 2>&1
  echo "l2 $M done"
done
cd $R
mkdir -p $OUT/summary
for M in sdxl sd3; do
  KS=$(find $OUT/stats_$M -name "*kernel_stats*.csv" | head -1)
  if [ -z "$KS" ]; then echo "no kernel_stats csv for $M:"; find $OUT/stats_$M | head -20; tail -5 $OUT/stats_$M.log; else
    python3 tools/trim_rocprof.py $KS $OUT/summary/${TAG}_kernel_stats_$M.txt; fi
  python3 tools/pmc_traffic.py $OUT/pmc_fetch_$M $OUT/pmc_write_$M $OUT/summary/${TAG}_pmc_traffic_${M}_step.txt
  python3 tools/pmc_mfma.py $OUT/pmc_mfma_$M $OUT/summary/${TAG}_pmc_mfma_util_$M.txt
  python3 tools/pmc_l2.py $OUT/pmc_l2_$M $OUT/summary/${TAG}_pmc_l2_hit_$M.txt
done
# raw counter CSVs are large: keep only the summaries for the merge back
rm -rf $OUT/stats_* $OUT/pmc_* 2>/dev/null || true
ls -la $OUT/summary
