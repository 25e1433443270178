#!/bin/bash
# Usage (GPU box, repo root): tools/ab_libs.sh <lib.so> [<lib.so> ...] -- [bench args]
# Runs the default bench workload once per build of the library (NVLLM_HIP_LIB) and prints
# tokens/s and the per-class kernel times.  "base" names the in-tree library.
libs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do libs+=("$1"); shift; done
shift
for lib in "${libs[@]}"; do
  tag=$(basename "$lib" .so)
  if [ "$lib" = base ]; then unset NVLLM_HIP_LIB; else export NVLLM_HIP_LIB="$PWD/$lib"; fi
  timeout -k 10 300 python3 bench.py --steps 32 --warmup 4 --cpu-steps 0 --no-variants "$@" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err || { echo "$tag failed"; tail -3 gpurun_out/ab_$tag.err; exit 1; }
  python3 - "$tag" <<PY
import json, sys
d = json.load(open("gpurun_out/ab_%s.json" % sys.argv[1]))
print("%-16s %9.1f tok/s %7.4f ms  " % (sys.argv[1], d["value"], d["ms_per_step"]) + "  ".join("%s %.2f" % (k.replace("gemm_", ""), v["avg_us"]) for k, v in d["kernels"].items()))
PY
done
