#!/bin/bash
# Usage (GPU box, repo root): tools/prefill_ab.sh "<ENV=VAL ...>" ["<ENV=VAL ...>" ...]
# Per setting: the per-kernel summary of the prefill passes of bench.py (rocprofv3 kernel trace).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for setting in "$@"; do
  i=$((i+1))
  echo "== $setting"
  ( export $setting; timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pfab_$i -- python3 bench.py --steps 2 --warmup 1 --cpu-steps 0 --profile-steps 0 --no-variants > gpurun_out/pfab_$i.log 2>&1 ) || { echo "run failed"; tail -5 gpurun_out/pfab_$i.log; exit 1; }
  python3 tools/prefill_summary.py gpurun_out/pfab_$i/*/*_kernel_trace.csv > gpurun_out/pfab_$i.txt || exit 1
  sed -n 1,6p gpurun_out/pfab_$i.txt
done
