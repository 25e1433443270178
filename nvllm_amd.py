"""Import alias: the package directory is named ``nano-vllm_amd`` (not a valid
identifier), so ``import nvllm_amd`` loads it and re-exports its modules."""
import importlib
import sys

_pkg = importlib.import_module("nano-vllm_amd")
capi = _pkg.capi
synth = _pkg.synth
sys.modules[__name__ + ".capi"] = capi
sys.modules[__name__ + ".synth"] = synth
