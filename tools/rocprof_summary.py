#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per kernel for the decode steps only.

A decode step of the benchmark ends with one k_argmax launch over `batch` rows and
contains no launch whose grid says "more tokens than sequences"; the last
`--steps` such steps are summarised: calls per step, average/min/max duration,
idle gap in front of each launch, VGPRs, LDS, scratch.

    python tools/rocprof_summary.py gpurun_out/prof/x_kernel_trace.csv --steps 8
"""
import argparse
import csv
import re
from collections import defaultdict


def short(name):
    m = re.match(r"(?:void )?(?:nvllm::)?([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=8)
    args = ap.parse_args()
    rows = list(csv.DictReader(open(args.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "k_argmax" in r["Kernel_Name"]]
    if len(ends) < args.steps + 1:
        raise SystemExit(f"only {len(ends)} steps in the trace")
    first = ends[-args.steps - 1] + 1
    sel = rows[first:ends[-1] + 1]
    t_begin, t_end = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
    stat = defaultdict(lambda: dict(n=0, tot=0, mn=1 << 60, mx=0, gap=0, meta=None))
    prev_end = None
    for r in sel:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        d = stat[short(r["Kernel_Name"])]
        d["n"] += 1
        d["tot"] += e - s
        d["mn"] = min(d["mn"], e - s)
        d["mx"] = max(d["mx"], e - s)
        if prev_end is not None:
            d["gap"] += max(0, s - prev_end)
        prev_end = e
        d["meta"] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"],
                     r["Workgroup_Size_X"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])),
                     r["Grid_Size_Y"], r["Grid_Size_Z"])
    busy = sum(d["tot"] for d in stat.values())
    wall = t_end - t_begin
    print(f"# {args.steps} decode steps: wall {wall / args.steps / 1e3:.1f} us/step, kernels busy "
          f"{busy / args.steps / 1e3:.1f} us/step ({100.0 * busy / wall:.1f}% of wall)")
    print(f"{'kernel':34s} {'calls/step':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'us/step':>9s} "
          f"{'gap_us':>7s}  vgpr agpr sgpr lds scratch wg grid")
    for k, d in sorted(stat.items(), key=lambda kv: -kv[1]["tot"]):
        print(f"{k:34s} {d['n'] / args.steps:10.1f} {d['tot'] / d['n'] / 1e3:9.2f} {d['mn'] / 1e3:9.2f} "
              f"{d['mx'] / 1e3:9.2f} {d['tot'] / args.steps / 1e3:9.1f} {d['gap'] / d['n'] / 1e3:7.2f}  "
              + " ".join(str(x) for x in d["meta"]))


if __name__ == "__main__":
    main()
