#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per run)
over the benchmark's decode steps into per-kernel HBM bytes per launch.

FETCH_SIZE / WRITE_SIZE count KiB.  On gfx950 FETCH_SIZE tallies the 128-byte
requests of wide coalesced reads at 64 bytes, so it is doubled; WRITE_SIZE is taken
as is.  Output: JSON keyed by kernel symbol (template arguments dropped).

    python tools/pmc_to_json.py fetch.csv write.csv --steps 2 > profiles/rNN_pmc_traffic.json
"""
import argparse
import csv
import json
import re
from collections import defaultdict


def symbol(name):
    m = re.match(r"(?:void )?(?:nvllm::)?([A-Za-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


def per_kernel(path, counter, steps):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    disp, names = defaultdict(float), {}
    for r in rows:
        d = int(r["Dispatch_Id"])
        disp[d] += float(r["Counter_Value"])
        names[d] = r["Kernel_Name"]
    ids = sorted(disp)
    ends = [i for i, d in enumerate(ids) if "k_argmax" in names[d]]
    sel = ids[ends[-steps - 1] + 1: ends[-1] + 1]
    tot, cnt = defaultdict(float), defaultdict(int)
    for d in sel:
        k = symbol(names[d])
        tot[k] += disp[d] * 1024.0
        cnt[k] += 1
    return tot, cnt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--steps", type=int, default=2)
    a = ap.parse_args()
    f, fc = per_kernel(a.fetch_csv, "FETCH_SIZE", a.steps)
    w, wc = per_kernel(a.write_csv, "WRITE_SIZE", a.steps)
    out = {}
    for k in sorted(f):
        n = fc[k]
        out[k] = dict(launches_per_step=n / a.steps, fetch_bytes_per_launch=round(2.0 * f[k] / n),
                      write_bytes_per_launch=round(w.get(k, 0.0) / max(wc.get(k, 1), 1)))
        out[k]["hbm_bytes_per_launch"] = out[k]["fetch_bytes_per_launch"] + out[k]["write_bytes_per_launch"]
    print(json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate runs) over decode steps of "
                                "bench.py's default workload; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B)",
                      "steps": a.steps, "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
