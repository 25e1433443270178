#!/bin/bash
# Usage (GPU box, repo root): tools/ab_attn_rowswap.sh  -- row-swap maxima (in-tree build) against shuffle maxima
# (variants/libnvllm_shfl.so, -DNVLLM_ATTN_SHFL_MAX): decode step and prefill passes.
bash tools/ab_libs.sh base nano-vllm_amd/variants/libnvllm_shfl.so base nano-vllm_amd/variants/libnvllm_shfl.so -- || exit 1
bash tools/prefill_ab.sh "NVLLM_X=0" "NVLLM_HIP_LIB=$PWD/nano-vllm_amd/variants/libnvllm_shfl.so" "NVLLM_X=0" "NVLLM_HIP_LIB=$PWD/nano-vllm_amd/variants/libnvllm_shfl.so" | grep -E "^==|^#|attn"
