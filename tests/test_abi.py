"""The drop-in boundary: record layouts equal the reference's, the HIP library
loads without a GPU and exports every entry point that include/*.h declares."""
import ctypes as C
import os
import re
import subprocess

import pytest

from nvllm_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
REF_INC = "/root/reference/include"

PROBE = r"""
#include <stdio.h>
#include <stddef.h>
#include "structs.h"
#define S(t) printf(#t " %zu\n", sizeof(t))
#define O(t, f) printf(#t "." #f " %zu\n", offsetof(t, f))
int main(void) {
  S(Config); O(Config, dim); O(Config, hidden_dim); O(Config, n_layers); O(Config, n_heads); O(Config, n_kv_heads);
  O(Config, vocab_size); O(Config, max_seq_len); O(Config, head_dim); O(Config, rope_theta);
  S(LayerWeights); O(LayerWeights, rms_att_weight); O(LayerWeights, wq); O(LayerWeights, wk); O(LayerWeights, wv);
  O(LayerWeights, wo); O(LayerWeights, rms_ffn_weight); O(LayerWeights, w_gate); O(LayerWeights, w_up); O(LayerWeights, w_down);
  S(Weights); O(Weights, token_embedding_table); O(Weights, layers); O(Weights, rms_final_weight); O(Weights, lm_head); O(Weights, _data_buffer);
  S(RunState); O(RunState, x); O(RunState, xb); O(RunState, xb2); O(RunState, hb); O(RunState, hb2); O(RunState, q);
  O(RunState, k); O(RunState, v); O(RunState, att); O(RunState, logits); O(RunState, key_cache); O(RunState, value_cache);
  S(KVCacheManager); O(KVCacheManager, block_size); O(KVCacheManager, num_blocks); O(KVCacheManager, free_blocks_count);
  O(KVCacheManager, free_block_indices); O(KVCacheManager, pool_k); O(KVCacheManager, pool_v);
  S(BlockTable); O(BlockTable, block_indices); O(BlockTable, num_blocks);
  S(Sequence); O(Sequence, id); O(Sequence, active); O(Sequence, pos); O(Sequence, seq_len); O(Sequence, current_token);
  O(Sequence, prompt_tokens); O(Sequence, num_prompt_tokens); O(Sequence, output_history); O(Sequence, state);
  O(Sequence, table); O(Sequence, arrival_step); O(Sequence, status); O(Sequence, first_token_latency);
  S(ProbIndex); O(ProbIndex, prob); O(ProbIndex, index);
  printf("enum %d %d %d %d\n", SEQ_WAITING, SEQ_PREFILLING, SEQ_DECODING, SEQ_FINISHED);
  return 0;
}
"""

# layout of reference include/structs.h on x86-64 (LP64), recorded from the probe above
EXPECTED = {"Config": 36, "LayerWeights": 72, "Weights": 40, "RunState": 96, "KVCacheManager": 40,
            "BlockTable": 16, "Sequence": 88, "ProbIndex": 8}


def _probe(inc, tmp_path, tag):
    src = tmp_path / f"probe_{tag}.c"
    src.write_text(PROBE)
    exe = tmp_path / f"probe_{tag}"
    subprocess.check_call(["gcc", "-I", inc, "-o", str(exe), str(src)])
    return subprocess.check_output([str(exe)], text=True)


def test_struct_layout(tmp_path):
    ours = _probe(INC, tmp_path, "ours")
    sizes = dict(line.split() for line in ours.splitlines() if "." not in line.split()[0] and not line.startswith("enum"))
    assert {k: int(v) for k, v in sizes.items()} == EXPECTED
    assert "enum 0 1 2 3" in ours
    if os.path.isdir(REF_INC):   # same probe against the reference's own header
        assert _probe(REF_INC, tmp_path, "ref") == ours
    # the ctypes mirrors used by the Python helpers agree too
    for name in ("Config", "LayerWeights", "Weights", "RunState", "KVCacheManager", "BlockTable", "ProbIndex"):
        assert C.sizeof(getattr(capi, name)) == EXPECTED[name], name


def _declared_functions():
    names = set()
    for h in ("backend.h", "ops.h", "model.h", "sampler.h", "nvllm.h"):
        text = open(os.path.join(INC, h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"^typedef[^;{]*\(\s*\*[^;]*;", "", text, flags=re.M)   # function-pointer types are not entry points
        for m in re.finditer(r"^[A-Za-z_][A-Za-z0-9_ \*]*?\b([a-z_][a-z0-9_]*)\s*\(", text, flags=re.M):
            names.add(m.group(1))
    # ASCII dashboards: declared for the driver's benefit, provided by its own visualizer.c
    return names - {"visualize_attention", "visualize_kv_cache_usage"}


def test_library_loads_and_exports_every_declared_symbol():
    assert os.path.exists(capi.HIP_LIB), "build the HIP library first (__graft_entry__.build())"
    dll = C.CDLL(capi.HIP_LIB, mode=C.RTLD_LOCAL)   # dlopen only: no GPU call is made
    declared = _declared_functions()
    assert {"device_malloc", "transformer_batch", "paged_attention_batch", "nvllm_engine_step", "sample"} <= declared
    missing = [n for n in sorted(declared) if not hasattr(dll, n)]
    assert not missing, missing
    # and the ctypes table covers the same set
    assert declared <= set(capi.declared_symbols(hip=True)) | {"nvllm_free_model"}, declared - set(capi.declared_symbols())


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="needs the reference sources")
def test_unmodified_reference_driver_links_against_the_hip_library(tmp_path):
    """reference src/main.c + its tokenizer/visualizer/log/sampler, compiled against
    THIS repository's headers and linked with libnvllm_hip.so instead of the
    reference's model.c / memory.c / kernels (no GPU needed to link)."""
    ref = "/root/reference/src"
    exe = tmp_path / "nano_vllm_hip"
    cmd = ["gcc", "-O2", "-w", "-I", INC, "-I", "/root/reference/include", "-o", str(exe)]
    cmd += [f"{ref}/main.c", f"{ref}/tokenizer.c", f"{ref}/visualizer.c", f"{ref}/log.c", f"{ref}/sampler.c"]
    cmd += ["-L", os.path.dirname(capi.HIP_LIB), "-lnvllm_hip", f"-Wl,-rpath,{os.path.dirname(capi.HIP_LIB)}", "-lm"]
    subprocess.check_call(cmd)
    syms = subprocess.check_output(["nm", "-D", "--undefined-only", str(exe)], text=True)
    for needed in ("transformer_batch", "load_model", "malloc_run_state", "init_kv_cache_manager", "alloc_block", "init_ops"):
        assert needed in syms, needed
