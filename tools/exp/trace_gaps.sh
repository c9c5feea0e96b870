# one rocprofv3 --kernel-trace pass of the SDXL step and the idle time between its kernels (tools/gap_analysis.py)
set -e
R=$PWD; OUT=$R/gpurun_out/prof_gaps; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o sdxl -- python3 $R/bench.py --model sdxl --steps 5 --warmup 2 --stream-requests 0 --mix 0 --no-cpu-baseline --no-roofline --no-sd3 --no-stages --no-parity > $OUT/trace.log 2>&1
cd $R
KT=$(find $OUT/trace -name "*kernel_trace*.csv" | head -1)
head -1 $KT > $OUT/header.txt
python3 tools/gap_analysis.py $KT $OUT/r04_gaps_sdxl.txt
rm -rf $OUT/trace
cat $OUT/r04_gaps_sdxl.txt
