#!/usr/bin/env python3
"""Matrix-core utilisation per kernel from one rocprofv3 pass
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py ...
over ALL dispatches of the run (prefill included).  SQ_VALU_MFMA_BUSY_CYCLES sums, over
the chip's SIMDs, the cycles a matrix instruction occupies its SIMD; GRBM_GUI_ACTIVE sums
the busy clock cycles of the 8 XCDs.  util = MFMA_BUSY / (SIMDs * GUI_ACTIVE / 8).

    python tools/mfma_util.py <..._counter_collection.csv> [--simds 1024]
"""
import argparse
import csv
import re
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("nvllm::", "")
    return re.sub(r"\(.*", "", name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--simds", type=int, default=1024)
    args = ap.parse_args()
    disp = defaultdict(lambda: defaultdict(float))
    names = {}
    for r in csv.DictReader(open(args.csv)):
        d = int(r["Dispatch_Id"])
        disp[d][r["Counter_Name"]] += float(r["Counter_Value"])
        names[d] = short(r["Kernel_Name"])
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for d, c in disp.items():
        a = agg[names[d]]
        a[0] += 1
        a[1] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        a[2] += c.get("GRBM_GUI_ACTIVE", 0.0)
    print(f"{'kernel':46s} {'launches':>8s} {'mfma_busy/launch':>17s} {'clk/launch':>11s} {'mfma_util':>9s}")
    for k, (n, busy, gui) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        clk = gui / 8.0
        util = busy / (args.simds * clk) if clk else 0.0
        print(f"{k[:46]:46s} {n:8d} {busy / n:17.0f} {clk / n:11.0f} {util:9.3f}")


if __name__ == "__main__":
    main()
