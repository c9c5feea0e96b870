"""Mean of rocprofv3 --pmc counters per kernel symbol.  Usage: python tools/pmc_kernel.py <dir> [substring]"""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        a = acc[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for k, cs in acc.items():
    print(k)
    for c, (n, v) in sorted(cs.items()):
        print(f"   {c:32s} n={n:5d} mean={v / n:16.1f}")
