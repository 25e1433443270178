#!/bin/bash
# Usage (GPU box, repo root): tools/prefill_evidence.sh <tag>
# The prefill side of the evidence: per-kernel time of bench.py's prefill passes (kernel trace) and the matrix cores'
# utilisation per kernel (one --pmc pass of its own).
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/${tag}_pftrace gpurun_out/${tag}_pfmfma
P="bench.py --steps 2 --warmup 1 --cpu-steps 0 --profile-steps 0 --no-variants"
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_pftrace -- python3 $P > gpurun_out/${tag}_pftrace.log 2>&1 || { echo "trace failed"; tail -5 gpurun_out/${tag}_pftrace.log; exit 1; }
python3 tools/prefill_summary.py gpurun_out/${tag}_pftrace/*/*_kernel_trace.csv | tee gpurun_out/${tag}_prefill_kernels.txt
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_pfmfma -- python3 $P > gpurun_out/${tag}_pfmfma.log 2>&1 || { echo "mfma pass failed"; tail -5 gpurun_out/${tag}_pfmfma.log; exit 1; }
python3 tools/mfma_util.py gpurun_out/${tag}_pfmfma/*/*_counter_collection.csv | tee gpurun_out/${tag}_prefill_mfma_util.txt
