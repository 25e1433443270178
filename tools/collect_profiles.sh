#!/bin/bash
# Usage (here, after tools/gpu_evidence.sh <tag> ran on the GPU box and its output was merged back):
#   tools/collect_profiles.sh <tag> <round>     e.g.  tools/collect_profiles.sh r02a r02
# Copies the summaries that the bench line refers to from gpurun_out/ (scratch) to profiles/ (tracked).
tag=$1; rnd=$2
set -e
cp gpurun_out/${tag}_kernels.txt profiles/${rnd}_decode_kernels.txt
cat gpurun_out/${tag}_fetch.txt gpurun_out/${tag}_write.txt > profiles/${rnd}_pmc_fetch_write.txt
cp gpurun_out/${tag}_l2.txt profiles/${rnd}_pmc_l2_requests.txt
python3 tools/pmc_to_json.py gpurun_out/${tag}_fetch/*/*_counter_collection.csv gpurun_out/${tag}_write/*/*_counter_collection.csv --steps 2 > profiles/${rnd}_pmc_traffic.json
cp gpurun_out/${tag}_stats/*/*_kernel_stats.csv profiles/${rnd}_rocprofv3_kernel_stats.csv 2>/dev/null || true
[ -f gpurun_out/${tag}_bench.json ] && cp gpurun_out/${tag}_bench.json profiles/${rnd}_bench_line.json
ls -la profiles/${rnd}_*
