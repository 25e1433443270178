#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc CSV over the decode steps of the
benchmark (the dispatches between the last `--steps`+1 k_argmax launches).
FETCH_SIZE / WRITE_SIZE are printed in bytes per launch (counter unit: KiB)."""
import argparse
import csv
import re
from collections import defaultdict


def short(name):
    m = re.match(r"(?:void )?(?:nvllm::)?([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--steps", type=int, default=2)
    args = ap.parse_args()
    rows = list(csv.DictReader(open(args.csv)))
    disp = defaultdict(dict)
    names = {}
    grid = {}
    for r in rows:
        d = int(r["Dispatch_Id"])
        disp[d][r["Counter_Name"]] = disp[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        names[d] = r["Kernel_Name"]
        grid[d] = (r["Grid_Size"], r["Workgroup_Size"])
    ids = sorted(disp)
    ends = [i for i, d in enumerate(ids) if "k_argmax" in names[d]]
    sel = ids[ends[-args.steps - 1] + 1: ends[-1] + 1]
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for d in sel:
        k = short(names[d]) + " g" + grid[d][0]
        cnt[k] += 1
        for c, v in disp[d].items():
            agg[k][c] += v
    counters = sorted({c for k in agg for c in agg[k]})
    print("kernel".ljust(40), "n".rjust(4), " ".join(c[:16].rjust(16) for c in counters))
    for k in sorted(agg, key=lambda k: -cnt[k]):
        vals = []
        for c in counters:
            v = agg[k][c] / cnt[k]
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                v *= 1024
            vals.append(f"{v:16.0f}")
        print(k[:40].ljust(40), str(cnt[k]).rjust(4), " ".join(vals))


if __name__ == "__main__":
    main()
