# kernel-trace statistics of the SDXL bench step (7 steps): usage  bash tools/exp/stats_only.sh <tag> [ENV=VALUE ...]
set -e
TAG=${1:-r05_a}; shift || true
for kv in "$@"; do export "$kv"; done
R=$PWD; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_sdxl -o sdxl -- python3 $R/bench.py --model sdxl --steps 5 --warmup 2 --stream-requests 0 --mix 0 --no-cpu-baseline --no-roofline --no-sd3 --no-stages --no-parity --no-cached-mix --no-two-model > $OUT/stats_sdxl.log 2>&1
cd $R
KS=$(find $OUT/stats_sdxl -name "*kernel_stats*.csv" | head -1)
python3 tools/trim_rocprof.py $KS $OUT/${TAG}_kernel_stats_sdxl.txt
rm -rf $OUT/stats_sdxl
cat $OUT/${TAG}_kernel_stats_sdxl.txt
