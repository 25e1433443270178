#!/bin/bash
# Usage (GPU box, repo root): tools/chain_cycle.sh <tag> [notests]
# One measurement cycle for the one-launch decode step: its GPU tests, the in-kernel timeline from the
# stamps build (nano-vllm_amd/variants/libnvllm_stamps.so) and a short benchmark; everything lands in
# gpurun_out/ under the tag.
tag=$1
if [ "$2" != notests ]; then
  timeout -k 10 400 python -u -m pytest tests/test_gpu_chain.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -12 || exit 1
fi
NVLLM_HIP_LIB=$PWD/nano-vllm_amd/variants/libnvllm_stamps.so timeout -k 10 300 python tools/chain_stamps.py > gpurun_out/chain_stamps_$tag.txt 2>&1 || { tail -5 gpurun_out/chain_stamps_$tag.txt; exit 1; }
grep -v "^Model\|^  \|^File\|^Expected\|^Detected\|^Loading\|^DEBUG" gpurun_out/chain_stamps_$tag.txt
NVLLM_DECODE_CHAIN=1 timeout -k 10 300 python bench.py --steps 32 --warmup 8 --no-variants --cpu-steps 0 > gpurun_out/bench_chain_$tag.json 2> gpurun_out/bench_chain_$tag.err || { tail -5 gpurun_out/bench_chain_$tag.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/bench_chain_$tag.json')); print(d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
