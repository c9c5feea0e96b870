# kernel-trace statistics of tools/exp/small_m_bench.py in both forms (kernel time without the Python call overhead)
set -e
R=$PWD; OUT=$R/gpurun_out/prof_small_m; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in new old; do
  if [ $v = old ]; then export MX_SMALL_M=0; fi
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$v -o t -- python3 $R/tools/exp/small_m_bench.py > $OUT/$v.log 2>&1
  F=$(find $OUT/$v -name "*kernel_trace.csv" | head -1)
  python3 - "$F" "$v" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# launches in order: 11 shapes x 55 launches of the GEMM kernel (the other kernels are torch's)
g = [r for r in rows if "gemm" in r["Kernel_Name"]]
per = len(g) // 11
print(sys.argv[2], "gemm launches", len(g))
for i in range(11):
    chunk = g[i * per + 5:(i + 1) * per]
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in chunk)
    print(f"  shape {i}: {chunk[0]['Kernel_Name'][:60]:60s} median {d[len(d)//2]:7.1f} us  grid {chunk[0]['Grid_Size_X'] if 'Grid_Size_X' in chunk[0] else chunk[0].get('Grid_Size','?')}")
PY
done
