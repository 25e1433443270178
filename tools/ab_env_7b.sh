#!/bin/bash
# Usage (GPU box, repo root): tools/ab_env_7b.sh "<ENV=VAL ...>" ...   -- the 7B-shape line (bench.py --tp, one rank) per setting
i=0
for setting in "$@"; do
  i=$((i+1))
  ( export $setting; timeout -k 10 500 python3 bench.py --tp --steps 16 --warmup 4 --cpu-steps 0 --no-variants > gpurun_out/ab7_$i.json 2> gpurun_out/ab7_$i.err ) || { echo "$setting failed"; tail -3 gpurun_out/ab7_$i.err; exit 1; }
  python3 - "$i" "$setting" <<PY
import json, sys
d = json.load(open("gpurun_out/ab7_%s.json" % sys.argv[1]))
print("%-40s %9.1f tok/s %7.4f ms  " % (sys.argv[2], d["value"], d["ms_per_step"]) + "  ".join("%s %.1f" % (k.replace("gemm_", ""), v["avg_us"]) for k, v in d["kernels"].items()))
PY
done
