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
    # the denoising steps launch only the library's kernels (namespace mx::); everything else in the trace is model load and request setup
    # (torch RNG / casts / copies of the 5 GB of synthetic weights), outside the timed steps: reported as one line, not mixed into the table
    hot = [r for r in rows if "mx::" in r["Name"]]
    other = [r for r in rows if "mx::" not in r["Name"]]
    total = sum(float(r["TotalDurationNs"]) for r in hot) or 1.0
    out.write(f"{'kernel (hot path: mx:: symbols only)':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'%':>7s}\n")
    for r in sorted(hot, key=lambda r: -float(r["TotalDurationNs"])):
        t = float(r["TotalDurationNs"])
        if t / total < 0.0005:
            continue
        out.write(f"{short(r['Name']):70s} {int(r['Calls']):7d} {t / 1e6:10.3f} {float(r['AverageNs']) / 1e3:10.2f} {100 * t / total:7.2f}\n")
    out.write(f"total hot-path kernel time {total / 1e6:.3f} ms over {sum(int(r['Calls']) for r in hot)} dispatches\n")
    out.write(f"excluded (model load / request setup, not part of any step): {sum(int(r['Calls']) for r in other)} dispatches of {len(other)} torch / runtime "
              f"kernels, {sum(float(r['TotalDurationNs']) for r in other) / 1e6:.3f} ms\n")


if __name__ == "__main__":
    main()
