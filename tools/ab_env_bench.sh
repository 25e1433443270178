#!/bin/bash
# Usage (GPU box, repo root): tools/ab_env_bench.sh "<ENV=VAL ...>" "<ENV=VAL ...>" ...   (settings alternate, two rounds)
# Prints tokens/s, ms/step and the per-class kernel averages of bench.py's default workload per setting.
for round in 1 2; do
  for setting in "$@"; do
    ( export $setting; timeout -k 10 200 python bench.py --no-variants --cpu-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$setting', d['value'], d['ms_per_step'], {n:(v['avg_us']) for n,v in k.items()})" ) || exit 1
  done
done
