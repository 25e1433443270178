#!/bin/bash
# Usage (GPU box, repo root): tools/bench_lines.sh <tag>
# The side lines that go with a round's headline bench line: other workloads, the one-launch decode step
# switched on, one rank's share of the tensor-parallel 7B configuration, and the in-kernel timelines.
tag=$1
run() { name=$1; shift; echo "== $name"; "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err || { tail -5 gpurun_out/${tag}_$name.err; exit 1; }
        python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print(d['value'], d['ms_per_step'], d['step_hbm']['frac_of_peak'])" gpurun_out/${tag}_$name.json; }
S="--cpu-steps 0 --no-variants"
run stories42m_b32 timeout -k 10 300 python bench.py --model stories42M --batch 32 --prompt-len 192 --max-seq-len 1024 --chunk 64 $S
run ctx4k timeout -k 10 400 python bench.py --prompt-len 3840 --max-seq-len 4096 $S
run chain_on env NVLLM_DECODE_CHAIN=1 timeout -k 10 300 python bench.py $S
run stories42m_b32_chain_on env NVLLM_DECODE_CHAIN=1 timeout -k 10 300 python bench.py --model stories42M --batch 32 --prompt-len 192 --max-seq-len 1024 --chunk 64 $S
run llama7b_tp1 timeout -k 10 900 python bench.py --tp --steps 32 $S
run llama7b_tp8_rank0_slice timeout -k 10 600 python bench.py --tp --tp-slice 0/8 --steps 32 $S
export NVLLM_HIP_LIB=$PWD/nano-vllm_amd/variants/libnvllm_stamps.so
timeout -k 10 300 python tools/chain_stamps.py > gpurun_out/${tag}_chain_timeline_tinyllama.txt 2>&1 || exit 1
timeout -k 10 300 python tools/chain_stamps.py --model stories42M --batch 32 --context 236 --layers 8 > gpurun_out/${tag}_chain_timeline_stories42m.txt 2>&1 || exit 1
timeout -k 10 300 python tools/gemm_stamps.py --layers 3 > gpurun_out/${tag}_projection_timeline.txt 2>&1 || exit 1
echo timelines done
