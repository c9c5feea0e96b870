# kernel statistics of the patch-unit block cache against the exact path (tools/exp/cache_breakdown.py): usage  bash tools/exp/cache_breakdown.sh <out.txt>
set -e
R=$PWD; OUTF=${1:-$R/gpurun_out/r05_e_cache_breakdown.txt}; D=$R/gpurun_out/prof_cache; mkdir -p $D; cd /tmp; export TMPDIR=/tmp
: > $OUTF
for MODE in exact f1.0 f0.5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/$MODE -o s -- python3 $R/tools/exp/cache_breakdown.py $MODE 6 > $D/$MODE.log 2>&1
  KS=$(find $D/$MODE -name "*kernel_stats*.csv" | head -1)
  echo "==== $MODE ====" >> $OUTF
  grep WALL $D/$MODE.log >> $OUTF
  python3 $R/tools/trim_rocprof.py $KS $D/$MODE.txt
  cut -c1-140 $D/$MODE.txt >> $OUTF
  rm -rf $D/$MODE
done
cat $OUTF
