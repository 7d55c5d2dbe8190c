"""KV-cache scale handling for fp8 caches.

Mirrors vllm/model_executor/layers/quantization/kv_cache.py:42-195 (BaseKVCacheMethod) and the
scale buffers the Attention layer owns (vllm/model_executor/layers/attention/attention.py:124-215):
`_k_scale`, `_v_scale`, `_q_scale`, `_prob_scale` are fp32 device scalars the kernels read
through pointers; `_*_scale_float` are host copies.  MI355X uses OCP e4m3fn, so the
"fnuz x2" adjustment of gfx94x never applies.
"""
import torch

from .registry import QuantizationConfig, QuantizeMethodBase


def is_quantized_kv_cache(kv_cache_dtype: str) -> bool:
    return kv_cache_dtype.startswith("fp8")


class KVCacheScaleParameter(torch.nn.Parameter):
    """Scalar scale, -1.0 until a checkpoint provides it."""

    def __new__(cls):
        return super().__new__(cls, torch.tensor(-1.0), requires_grad=False)

    @staticmethod
    def weight_loader(param: torch.nn.Parameter, loaded_weight: torch.Tensor) -> None:
        if loaded_weight.numel() != 1:
            raise ValueError(f"KV-cache scale expects a scalar weight, got shape {tuple(loaded_weight.shape)}")
        param.data.copy_(loaded_weight.reshape(()))


def set_default_quant_scales(layer: torch.nn.Module, register_buffer: bool = False) -> None:
    """Scales of 1.0 (attention.py `set_default_quant_scales`)."""
    for name in ("_k_scale", "_v_scale", "_q_scale", "_prob_scale"):
        t = torch.tensor(1.0, dtype=torch.float32)
        if register_buffer:
            layer.register_buffer(name, t)
        else:
            getattr(layer, name).fill_(1.0)
    layer._q_scale_float = layer._k_scale_float = layer._v_scale_float = 1.0


class BaseKVCacheMethod(QuantizeMethodBase):
    def __init__(self, quant_config: QuantizationConfig | None):
        self.quant_config = quant_config

    def create_weights(self, layer: torch.nn.Module):
        layer.q_scale = KVCacheScaleParameter()
        layer.k_scale = KVCacheScaleParameter()
        layer.v_scale = KVCacheScaleParameter()
        layer.prob_scale = KVCacheScaleParameter()

    def apply(self, layer: torch.nn.Module) -> torch.Tensor:
        raise RuntimeError(f"{type(self).__name__}.apply should not be called.")

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        if not hasattr(layer, "q_scale"):
            return
        k, v = float(layer.k_scale), float(layer.v_scale)
        if is_quantized_kv_cache(layer.kv_cache_dtype):
            if k > 0.0 and v > 0.0:
                pass
            elif k < 0.0 and v < 0.0:
                k = v = 1.0                      # nothing in the checkpoint
            else:
                k = v = max(k, v)                # one fused kv_scale, remapped onto k_scale by the loader
            layer._k_scale.fill_(k)
            layer._v_scale.fill_(v)
            layer._k_scale_float, layer._v_scale_float = k, v
            if float(layer.q_scale) < 0.0:
                layer._q_scale.fill_(k)
                layer._q_scale_float = k
        q = float(layer.q_scale) if float(layer.q_scale) > 0.0 else 1.0
        p = float(layer.prob_scale) if float(layer.prob_scale) > 0.0 else 1.0
        layer._q_scale.fill_(q)
        layer._q_scale_float = q
        layer._prob_scale.fill_(p)
        del layer.k_scale, layer.v_scale, layer.q_scale, layer.prob_scale
