#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc CSV over the PREFILL passes of bench.py: the launches of the
dense-tile projection kernel and of attention in passes of more than 64 tokens, grouped by kernel and grid.
FETCH_SIZE / WRITE_SIZE are printed in bytes per launch (counter unit: KiB; FETCH_SIZE as counted: on gfx950
it reports half of wide streaming reads)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.match(r"(?:void )?(?:nvllm::)?(?:\(anonymous namespace\)::)?([A-Za-z0-9_]+)(<[^>]*>)?", name.replace("nvllm::(anonymous namespace)::", ""))
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def main(path):
    disp, names, grid = defaultdict(dict), {}, {}
    for r in csv.DictReader(open(path)):
        d = int(r["Dispatch_Id"])
        disp[d][r["Counter_Name"]] = disp[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        names[d] = r["Kernel_Name"]
        grid[d] = r["Grid_Size"]
    agg, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(int)
    for d in disp:
        if "k_gemm_prefill" not in names[d]:
            continue
        k = short(names[d]) + " g" + grid[d]
        cnt[k] += 1
        for c, v in disp[d].items():
            agg[k][c] += v
    counters = sorted({c for k in agg for c in agg[k]})
    print("kernel".ljust(58), "n".rjust(5), " ".join(c[:16].rjust(16) for c in counters))
    for k in sorted(agg, key=lambda k: -cnt[k]):
        vals = [f"{agg[k][c] / cnt[k] * (1024 if c in ('FETCH_SIZE', 'WRITE_SIZE') else 1):16.0f}" for c in counters]
        print(k[:58].ljust(58), str(cnt[k]).rjust(5), " ".join(vals))


if __name__ == "__main__":
    main(sys.argv[1])
