#!/usr/bin/env python3
"""Compare the per-kernel register / LDS / scratch use of two hipcc builds of one source file.

    hipcc ... -c file.hip -Rpass-analysis=kernel-resource-usage 2> usage.txt
    python tools/kernel_usage.py old_usage.txt new_usage.txt     # kernels whose numbers changed
    python tools/kernel_usage.py usage.txt                       # all kernels of one build
"""
import re
import sys

KEYS = ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+)", line)
        if m:
            cur = m.group(1)
            out[cur] = {}
        for key in KEYS:
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and cur:
                out[cur][key.split(" ")[0]] = int(m.group(1))
    return out


def main():
    if len(sys.argv) == 2:
        for k, v in sorted(parse(sys.argv[1]).items()):
            print(k, v)
        return
    old, new = parse(sys.argv[1]), parse(sys.argv[2])
    for k in sorted(set(old) | set(new)):
        if old.get(k) != new.get(k):
            print(k, old.get(k), "->", new.get(k))


if __name__ == "__main__":
    main()
