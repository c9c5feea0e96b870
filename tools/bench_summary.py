"""Prints ms/step, images/s and the per-kernel table of bench.py JSON lines (files given on the command line)."""
import json
import sys

for f in sys.argv[1:]:
    line = [l for l in open(f).read().splitlines() if l.startswith("{")]
    if not line:
        print(f, "no JSON line")
        continue
    d = json.loads(line[-1])
    print(f"{f}: {d['ms_per_step']:.2f} ms/step  {d['value']:.3f} {d['unit']}")
    for k in d.get("kernels", []):
        print(f"    {k['kernel']:36s} n={k['launches']:4d} {k['ms_total']:8.2f} ms  {k['tflops'] or 0:7.1f} TF  {k['gbps']:7.1f} GB/s")
