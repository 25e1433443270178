#!/usr/bin/env python3
"""Per-kernel time of the PREFILL forward passes in a rocprofv3 kernel trace of bench.py
(passes are delimited by the embedding kernel; a pass with more than 64 tokens carries
prefill chunks).

    python tools/prefill_summary.py <..._kernel_trace.csv>
"""
import collections
import csv
import re
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "k_embed_prescale" in r["Kernel_Name"]]
    tokens_of = lambda r: int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    agg = collections.defaultdict(lambda: [0, 0.0])
    npass = toks = 0
    wall = 0.0
    for a, b in zip(starts, starts[1:] + [len(rows)]):
        T = tokens_of(rows[a])
        if T <= 64:
            continue
        npass += 1
        toks += T
        for r in rows[a:b]:
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("nvllm::", "")
            name = re.sub(r"\(.*", "", name)
            agg[name][0] += 1
            agg[name][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        wall += (int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
    if not npass:
        print("no prefill passes in the trace")
        return
    busy = sum(v[1] for v in agg.values())
    print(f"# {npass} prefill passes, {toks / npass:.0f} tokens/pass, wall {wall / npass:.1f} us/pass, "
          f"kernels busy {busy / npass:.1f} us/pass, {toks / wall * 1e6:.0f} tokens/s on the GPU clock")
    print(f"{'kernel':58s} {'calls/pass':>10s} {'avg_us':>9s} {'us/pass':>10s} {'share':>6s}")
    for name, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name[:58]:58s} {c / npass:10.1f} {d / c:9.1f} {d / npass:10.1f} {100 * d / busy:5.1f}%")


if __name__ == "__main__":
    main(sys.argv[1])
