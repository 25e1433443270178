"""MI355X-native backend for the forward pass of nano-vllm (host-side Python helpers).

The product is ``libnvllm_hip.so`` (HIP kernels + C ABI, see include/*.h); this
package only holds ctypes bindings (``capi``) and synthetic-input generators
(``synth``).  The directory name contains a hyphen, so import it with
``importlib.import_module("nano-vllm_amd")`` or through ``nvllm_amd.py`` at the
repository root.
"""
from . import capi, synth  # noqa: F401
