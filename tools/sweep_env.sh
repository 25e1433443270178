#!/bin/bash
# usage: tools/sweep_env.sh "<bench args>" VAR1=a,b VAR2=c,d ...  -- runs bench.py for the cross product (GPU box)
args="$1"; shift
combos=("")
for spec in "$@"; do
  var="${spec%%=*}"; vals="${spec#*=}"; next=()
  IFS=',' read -ra arr <<< "$vals"
  for c in "${combos[@]}"; do for v in "${arr[@]}"; do next+=("$c $var=$v"); done; done
  combos=("${next[@]}")
done
for c in "${combos[@]}"; do
  out=$(env $c timeout -k 10 200 python bench.py $args --cpu-steps 0 2>/dev/null) || { echo "FAILED: $c"; exit 1; }
  echo "$out" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$c', 'ms/step', d['ms_per_step'], ' '.join('%s=%.1f' % (n, v['avg_us']) for n, v in k.items()))"
done
