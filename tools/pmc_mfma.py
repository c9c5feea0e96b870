"""Matrix-core utilisation per kernel from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE).

  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 256 CUs * 4 SIMDs)

SQ_VALU_MFMA_BUSY_CYCLES is summed over every SIMD of the chip (32 per v_mfma_f32_32x32x16_bf16, 16 per 16x16x32);
rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (MI355X_MICROARCH.md 'DVFS give-back'), so /8 is the
kernel's duration in shader clocks.  This is the fraction of matrix-core issue slots that were busy at the clock
the chip actually held -- not a fraction of the 2.5 PFLOP/s nameplate, which assumes 2.4 GHz.
Usage: python tools/pmc_mfma.py <dir> [out.txt]"""
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
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cnt[k] += 1
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    out.write(f"{'kernel':60s} {'launches':>8s} {'mfma_busy_Mcyc':>15s} {'kernel_Mclk':>12s} {'mfma_util':>10s} {'cu_busy':>8s}\n")
    rows = []
    for k, c in acc.items():
        clk = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if clk <= 0:
            continue
        util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (clk * 256 * 4)
        cub = c.get("SQ_BUSY_CU_CYCLES", 0.0) / (clk * 256) if "SQ_BUSY_CU_CYCLES" in c else float("nan")
        rows.append((clk, k, cnt[k], c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), util, cub))
    for clk, k, n, busy, util, cub in sorted(rows, reverse=True)[:24]:
        out.write(f"{k:60s} {n:8d} {busy / 1e6:15.2f} {clk / 1e6:12.3f} {util:10.3f} {cub:8.2f}\n")


if __name__ == "__main__":
    main()
