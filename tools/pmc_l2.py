"""Per-kernel L2 hit rate from one rocprofv3 --pmc pass (TCC_HIT_sum TCC_MISS_sum): hit / (hit + miss), summed over the eight
XCD L2s (MI355X_MICROARCH.md section L2).  Usage: python tools/pmc_l2.py <dir> [out.txt]"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void\s+", "", name)
    return name if len(name) <= 60 else name[:57] + "..."


def main():
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "TCC_HIT_sum":
                cnt[k] += 1
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    out.write(f"{'kernel':60s} {'launches':>8s} {'hit_M/launch':>13s} {'miss_M/launch':>14s} {'l2_hit_rate':>12s}\n")
    rows = []
    for k, c in acc.items():
        h, m = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
        if h + m > 0:
            rows.append((h + m, k, cnt[k], h, m))
    for _t, k, n, h, m in sorted(rows, reverse=True)[:24]:
        out.write(f"{k:60s} {n:8d} {h / max(n, 1) / 1e6:13.2f} {m / max(n, 1) / 1e6:14.2f} {h / (h + m):12.3f}\n")


if __name__ == "__main__":
    main()
