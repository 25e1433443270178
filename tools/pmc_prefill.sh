#!/bin/bash
# Usage (GPU box, repo root): tools/pmc_prefill.sh <tag>
# L2 requests / hits / misses and FETCH_SIZE of the dense-tile projection kernel over bench.py's prefill passes.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/${tag}_pfl2 gpurun_out/${tag}_pffetch
P="bench.py --steps 2 --warmup 1 --cpu-steps 0 --profile-steps 0 --no-variants"
timeout -s KILL 240 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/${tag}_pfl2 -- python3 $P > gpurun_out/${tag}_pfl2.log 2>&1 || { echo "l2 pass failed"; tail -5 gpurun_out/${tag}_pfl2.log; exit 1; }
python3 tools/pmc_prefill.py gpurun_out/${tag}_pfl2/*/*_counter_collection.csv | tee gpurun_out/${tag}_pfl2.txt
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pffetch -- python3 $P > gpurun_out/${tag}_pffetch.log 2>&1 || { echo "fetch pass failed"; tail -5 gpurun_out/${tag}_pffetch.log; exit 1; }
python3 tools/pmc_prefill.py gpurun_out/${tag}_pffetch/*/*_counter_collection.csv | tee gpurun_out/${tag}_pffetch.txt
