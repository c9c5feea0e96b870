"""Idle time between the kernels of a denoising step, from a rocprofv3 --kernel-trace CSV (start / end timestamps per dispatch).
The hot-path dispatches (mx:: symbols) are sorted by start time and cut into steps at the step's first kernel; per step: wall time from the first start to the last end, the sum of
the kernel durations, and the gaps between consecutive dispatches (next start - previous end), with a histogram and the per-predecessor-kernel mean gap.
Usage: python tools/gap_analysis.py <kernel_trace.csv> [out.txt]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void\s+", "", name)
    m = re.match(r"(mx::\w+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if "mx::" in r["Kernel_Name"]), key=lambda t: t[0])
    if not ks:
        out.write("no mx:: dispatches in the trace\n"); return
    # a step starts with the pipeline's input-scaling kernel (steps run back to back: no pause to cut at)
    first = short(ks[0][2])
    steps, cur = [], []
    for k in ks:
        if cur and short(k[2]) == first:
            steps.append(cur); cur = []
        cur.append(k)
    steps.append(cur)
    steps = [s for s in steps if len(s) > 100]
    out.write(f"{len(ks)} hot-path dispatches, {len(steps)} steps of {len(steps[-1])} dispatches (cut at every '{first}')\n")
    out.write(f"{'step':>4s} {'wall_ms':>9s} {'kernels_ms':>11s} {'gaps_ms':>8s} {'overlap_ms':>10s} {'mean_gap_us':>11s}\n")
    allgaps, by_prev = [], defaultdict(list)
    for i, s in enumerate(steps):
        wall = (s[-1][1] - s[0][0]) / 1e6
        dur = sum(e - b for b, e, _ in s) / 1e6
        gaps = [(s[j + 1][0] - s[j][1]) / 1e3 for j in range(len(s) - 1)]
        pos = [g for g in gaps if g > 0]
        neg = [-g for g in gaps if g < 0]
        out.write(f"{i:4d} {wall:9.3f} {dur:11.3f} {sum(pos) / 1e3:8.3f} {sum(neg) / 1e3:10.3f} {sum(pos) / max(len(pos), 1):11.2f}\n")
        if i >= 2:                                         # skip the warm-up steps
            allgaps += gaps
            for j, g in enumerate(gaps):
                by_prev[short(s[j][2])].append(g)
    if allgaps:
        allgaps.sort()
        n = len(allgaps)
        out.write(f"gaps over the timed steps (us): n {n}  p10 {allgaps[n // 10]:.2f}  p50 {allgaps[n // 2]:.2f}  p90 {allgaps[9 * n // 10]:.2f}  max {allgaps[-1]:.2f}  "
                  f"sum per step {sum(g for g in allgaps if g > 0) / 1e3 / max(len(steps) - 2, 1):.3f} ms\n")
        out.write("mean gap AFTER a kernel of each kind (us), count per trace:\n")
        for k, v in sorted(by_prev.items(), key=lambda kv: -len(kv[1])):
            out.write(f"  {k[:90]:90s} {len(v):6d} {sum(v) / len(v):8.2f}\n")


if __name__ == "__main__":
    main()
