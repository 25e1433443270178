#!/bin/bash
# Usage (GPU box, repo root): tools/gpu_evidence.sh <tag>
# Collects what the bench line's roofline refers to, for the default bench workload:
#   gpurun_out/<tag>_stats/      rocprofv3 --kernel-trace --stats of bench.py
#   gpurun_out/<tag>_fetch/      --pmc FETCH_SIZE pass   (own run: TCC slots)
#   gpurun_out/<tag>_write/      --pmc WRITE_SIZE pass
#   gpurun_out/<tag>_l2/         --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum pass:
#                                requests the CUs' vector caches send to L2 (bytes INTO the CUs) and L2 hits
# and prints the per-kernel summaries.
tag=$1
cd "$GRAFT_REPO_ROOT" && rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_l2   # one run per directory: the summaries glob
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="bench.py --steps 16 --warmup 2 --cpu-steps 0 --profile-steps 0 --no-variants"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 $B > gpurun_out/${tag}_stats.log 2>&1 || { echo "stats run failed"; tail -5 gpurun_out/${tag}_stats.log; exit 1; }
python3 tools/rocprof_summary.py gpurun_out/${tag}_stats/*/*_kernel_trace.csv --steps 8 > gpurun_out/${tag}_kernels.txt || exit 1
cat gpurun_out/${tag}_kernels.txt
P="bench.py --steps 3 --warmup 1 --cpu-steps 0 --profile-steps 0 --no-variants"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- python3 $P > gpurun_out/${tag}_fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 gpurun_out/${tag}_fetch.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/${tag}_fetch/*/*_counter_collection.csv --steps 2 > gpurun_out/${tag}_fetch.txt || exit 1
cat gpurun_out/${tag}_fetch.txt
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- python3 $P > gpurun_out/${tag}_write.log 2>&1 || { echo "write pass failed"; tail -5 gpurun_out/${tag}_write.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/${tag}_write/*/*_counter_collection.csv --steps 2 > gpurun_out/${tag}_write.txt || exit 1
cat gpurun_out/${tag}_write.txt
timeout -s KILL 240 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/${tag}_l2 -- python3 $P > gpurun_out/${tag}_l2.log 2>&1 || { echo "l2 pass failed"; tail -5 gpurun_out/${tag}_l2.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/${tag}_l2/*/*_counter_collection.csv --steps 2 > gpurun_out/${tag}_l2.txt || exit 1
cat gpurun_out/${tag}_l2.txt
