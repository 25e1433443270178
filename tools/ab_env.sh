#!/bin/bash
# Usage (GPU box, repo root): tools/ab_env.sh "<tag> [ENV=VAL ...]" ... -- [bench args]
# Runs the default bench workload once per environment setting and prints tokens/s and kernel times.
cfgs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do cfgs+=("$1"); shift; done
shift
for cfg in "${cfgs[@]}"; do
  set -- $cfg "$@"
  tag=$1; shift
  envs=()
  while [[ "$1" == *=* ]]; do envs+=("$1"); shift; done
  env "${envs[@]}" timeout -k 10 300 python3 bench.py --steps 64 --warmup 8 --cpu-steps 0 --no-variants --profile-steps 0 > gpurun_out/abe_$tag.json 2> gpurun_out/abe_$tag.err || { echo "$tag failed"; tail -3 gpurun_out/abe_$tag.err; exit 1; }
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/abe_$tag.json')); print('%-14s %9.1f tok/s %7.4f ms/step  forward %7.4f' % ('$tag', d['value'], d['ms_per_step'], d['forward_ms_per_step']))"
done
