set -e
for M in 1 0 1 0; do
  NVLLM_ATTN_MERGE=$M timeout -k 10 200 python bench.py --kv-bytes 4 --steps 32 --no-variants --cpu-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('merge=$M', d['value'], d['ms_per_step'], {n:(v['avg_us']) for n,v in k.items() if 'attn' in n})"
done
