#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/gpu_ab.sh <tag> [ENV=VAL ...] -- [bench args]
# Runs bench.py under rocprofv3 --kernel-trace with the given environment and leaves
# the trace under gpurun_out/ab_<tag>/ plus a decode-step summary in gpurun_out/ab_<tag>.txt
tag=$1; shift
envs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for e in "${envs[@]}"; do export "$e"; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ab_$tag -- python3 bench.py --steps 16 --warmup 2 --cpu-steps 0 --profile-steps 0 "$@" > gpurun_out/ab_$tag.log 2>&1 || { echo "run $tag failed"; tail -5 gpurun_out/ab_$tag.log; exit 1; }
python3 tools/rocprof_summary.py gpurun_out/ab_$tag/*/*_kernel_trace.csv --steps 8 > gpurun_out/ab_$tag.txt
grep -o '"value": [0-9.]*' gpurun_out/ab_$tag.log | head -1
head -12 gpurun_out/ab_$tag.txt
