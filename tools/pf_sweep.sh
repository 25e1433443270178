#!/bin/bash
# Usage (GPU box, repo root): tools/pf_sweep.sh <tag> <which> <tokens> [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; which=$2; tokens=$3; shift 3
for kv in "$@"; do export "$kv"; done
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pfsw_$tag -- python3 tools/pf_sweep.py --which $which --tokens $tokens > gpurun_out/pfsw_$tag.log 2>&1 || { echo "run failed"; tail -5 gpurun_out/pfsw_$tag.log; exit 1; }
echo "== $tag $which $*"
python3 tools/pf_sweep.py --summarise gpurun_out/pfsw_$tag/*/*_kernel_trace.csv | tee gpurun_out/pfsw_$tag.txt
