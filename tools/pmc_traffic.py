"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as MI355X_MICROARCH.md
prescribes: counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced streaming reads, so it is
doubled; WRITE_SIZE is exact for 16-byte streaming stores.
Usage: python tools/pmc_traffic.py <dir_with_fetch_pass> <dir_with_write_pass> [out.txt]"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void\s+", "", name)
    return name if len(name) <= 60 else name[:57] + "..."


def load(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            a = acc[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
    out.write(f"{'kernel':60s} {'launches':>8s} {'fetch_MB(x2)':>13s} {'write_MB':>10s} {'traffic_MB/launch':>18s}\n")
    rows = []
    for k in set(fetch) | set(write):
        nf, f = fetch.get(k, [0, 0.0])
        nw, w = write.get(k, [0, 0.0])
        n = max(nf, nw, 1)
        fb = 2.0 * f * 1024 / max(nf, 1)
        wb = w * 1024 / max(nw, 1)
        rows.append((fb * nf + wb * nw, k, n, fb, wb))
    for _t, k, n, fb, wb in sorted(rows, reverse=True)[:25]:
        out.write(f"{k:60s} {n:8d} {fb / 1e6:13.2f} {wb / 1e6:10.2f} {(fb + wb) / 1e6:18.2f}\n")


if __name__ == "__main__":
    main()
