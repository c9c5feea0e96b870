"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short table (kernel names cut to 70 chars).
Usage: python tools/trim_rocprof.py <kernel_stats.csv> [out.txt]"""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    if len(name) > 70:
        name = name[:67] + "..."
    return name


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    out.write(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'%':>7s}\n")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        t = float(r["TotalDurationNs"])
        if t / total < 0.0005:
            continue
        out.write(f"{short(r['Name']):70s} {int(r['Calls']):7d} {t / 1e6:10.3f} {float(r['AverageNs']) / 1e3:10.2f} {100 * t / total:7.2f}\n")
    out.write(f"total kernel time {total / 1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} dispatches\n")


if __name__ == "__main__":
    main()
