#!/bin/bash
# Usage (GPU box, repo root): tools/ab_lib_decode_prefill.sh <variant.so>
# The in-tree library against a variant build, twice each, alternating: decode step (bench.py) and prefill passes (kernel trace).
V=$1
bash tools/ab_libs.sh base $V base $V -- || exit 1
bash tools/prefill_ab.sh "NVLLM_X=0" "NVLLM_HIP_LIB=$PWD/$V" "NVLLM_X=0" "NVLLM_HIP_LIB=$PWD/$V" | grep -E "^==|^#|attn"
