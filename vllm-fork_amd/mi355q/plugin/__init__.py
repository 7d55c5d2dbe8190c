"""Host-side mirror of the reference's plug-in surface for the quantized decode path.

`register()` is what the `vllm.general_plugins` entry point calls in every process
(vllm/plugins/__init__.py:16-90): it is idempotent.  Inside a vLLM installation it hooks the
classes of this package into vLLM's own registries (quantization configs, mixed-precision
linear kernels, attention backend); without vLLM the package's own registries (same functions,
same semantics) are used, which is what the tests and the bundled Llama runner do.
"""
from .awq import AutoAWQConfig, AutoAWQLinearMethod
from .compressed_tensors import (CompressedTensorsConfig, CompressedTensorsLinearMethod, CompressedTensorsW8A16Fp8,
                                 CompressedTensorsWNA16, CompressedTensorsWNA16MoEMethod)
from .fp8 import Fp8Config, Fp8LinearMethod
from .gptq import AutoGPTQConfig, AutoGPTQLinearMethod
from .inc import INCConfig, INCLayerConfig, INCLinearMethod, INCWna16Scheme, resolve_scheme
from .moe import FusedMoE, MoeWNA16Config, MoeWNA16Method, fused_experts, fused_topk
from .mp_kernel import (MI355XW4A16LinearKernel, MI355XW8A16LinearKernel, MPLinearKernel, MPLinearLayerConfig, choose_mp_linear_kernel,
                        register_linear_kernel)
from .registry import (QUANTIZATION_METHODS, QuantizationConfig, QuantizeMethodBase, get_quantization_config,
                       register_quantization_config, resolve_quantization)

_REGISTERED = False


def register() -> None:
    global _REGISTERED
    if _REGISTERED:
        return
    _REGISTERED = True
    from .torch_ops import register_torch_ops
    register_torch_ops()
    try:
        import vllm  # noqa: F401
    except Exception:
        return
    from .vllm_hooks import hook_into_vllm
    hook_into_vllm()
