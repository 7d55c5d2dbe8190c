"""fp8 (float8_e4m3fn) checkpoints, run weight-only: fp8 weights x 16-bit activations.

Mirrors vllm/model_executor/layers/quantization/fp8.py:92-231 (Fp8Config), :234-456
(Fp8LinearMethod: weight [N, K] fp8; `weight_scale` one fp32 per logical matrix, or
`weight_scale_inv` [ceil(N/bn), ceil(K/bk)] for block-quantised checkpoints) and :819-…
(Fp8KVCacheMethod).  On ROCm the reference runs these checkpoints as w8a8 scaled-mm and has no
weight-only kernel (kernels/linear/__init__.py:405-407); this method keeps activations in
16 bits: out = x @ (fp8(W) * scale)^T, the arithmetic of the reference's dequantise-then-linear
fallback (:433-454).  `input_scale` tensors of static-activation checkpoints are accepted and
ignored.  16-bit checkpoints are quantised per tensor at load time
(quantization/online/fp8.py `Fp8PerTensorOnlineLinearMethod`).
"""
from typing import Any

import torch

from .. import ops
from .awq import is_layer_skipped
from .kv_cache import BaseKVCacheMethod
from .linear import LinearBase, LinearMethodBase, UnquantizedLinearMethod
from .parameter import BlockQuantScaleParameter, ModelWeightParameter, PerTensorScaleParameter
from .registry import QuantizationConfig, register_quantization_config

ACTIVATION_SCHEMES = ["static", "dynamic"]
FP8_MAX = 448.0


class Fp8KVCacheMethod(BaseKVCacheMethod):
    pass


@register_quantization_config("fp8")
class Fp8Config(QuantizationConfig):
    def __init__(self, is_checkpoint_fp8_serialized: bool = False, activation_scheme: str = "dynamic",
                 ignored_layers: list[str] | None = None, weight_block_size: list[int] | None = None,
                 store_dtype: str | None = None) -> None:
        super().__init__()
        if activation_scheme not in ACTIVATION_SCHEMES:
            raise ValueError(f"Unsupported activation scheme {activation_scheme}")
        if weight_block_size is not None:
            if not is_checkpoint_fp8_serialized:
                raise ValueError("The block-wise quantization only supports fp8-serialized checkpoint for now.")
            if len(weight_block_size) != 2:
                raise ValueError("The quantization block size of weight must have 2 dimensions, but got "
                                 f"{len(weight_block_size)} dimensions")
            if activation_scheme != "dynamic":
                raise ValueError("The block-wise quantization only supports dynamic activation scheme for now, "
                                 f"but got {activation_scheme} activation scheme.")
        self.is_checkpoint_fp8_serialized = is_checkpoint_fp8_serialized
        self.activation_scheme = activation_scheme
        self.ignored_layers = ignored_layers or []
        self.weight_block_size = weight_block_size
        self.store_dtype = store_dtype

    @classmethod
    def get_name(cls) -> str:
        return "fp8"

    @classmethod
    def get_supported_act_dtypes(cls) -> list[torch.dtype]:
        return [torch.bfloat16, torch.half]

    @classmethod
    def get_min_capability(cls) -> int:
        return 75

    @classmethod
    def get_config_filenames(cls) -> list[str]:
        return []

    @classmethod
    def from_config(cls, config: dict[str, Any]) -> "Fp8Config":
        method = cls.get_from_keys(config, ["quant_method"])
        ignored = cls.get_from_keys_or(config, ["ignored_layers"], None) or \
            cls.get_from_keys_or(config, ["modules_to_not_convert"], None)
        return cls(is_checkpoint_fp8_serialized="fp8" in method,
                   activation_scheme=cls.get_from_keys(config, ["activation_scheme"]), ignored_layers=ignored,
                   weight_block_size=cls.get_from_keys_or(config, ["weight_block_size"], None),
                   store_dtype=cls.get_from_keys_or(config, ["store_dtype"], None))

    def get_quant_method(self, layer: torch.nn.Module, prefix: str):
        from .attention import Attention
        if isinstance(layer, LinearBase):
            if is_layer_skipped(prefix, self.ignored_layers, self.packed_modules_mapping):
                return UnquantizedLinearMethod()
            return Fp8LinearMethod(self) if self.is_checkpoint_fp8_serialized else Fp8PerTensorOnlineLinearMethod()
        if isinstance(layer, Attention):
            return Fp8KVCacheMethod(self)
        return None

    _CACHE_SCALE_SUFFIXES = {".k_proj.output_scale": ".attn.k_scale", ".v_proj.output_scale": ".attn.v_scale",
                             ".q_proj.output_scale": ".attn.q_scale",
                             ".self_attn.prob_output_scale": ".self_attn.attn.prob_scale"}

    @classmethod
    def map_cache_scale_name(cls, name: str) -> str:
        """compressed-tensors style `*_proj.output_scale` names first, then the common rules
        (fp8.py:219-231)."""
        for old, new in cls._CACHE_SCALE_SUFFIXES.items():
            if name.endswith(old):
                return name[: -len(old)] + new
        return super().map_cache_scale_name(name)


def _set(layer: torch.nn.Module, name: str, t: torch.Tensor | None) -> None:
    if name in layer._parameters:
        del layer._parameters[name]
    layer.register_parameter(name, None if t is None else torch.nn.Parameter(t, requires_grad=False))


class Fp8LinearMethod(LinearMethodBase):
    def __init__(self, quant_config: Fp8Config):
        self.quant_config = quant_config
        self.weight_block_size = quant_config.weight_block_size
        self.block_quant = self.weight_block_size is not None

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs):
        loader = extra_weight_attrs.get("weight_loader")
        n_part = sum(output_partition_sizes)
        layer.logical_widths = list(output_partition_sizes)
        layer.input_size_per_partition, layer.output_size_per_partition = input_size_per_partition, n_part
        layer.orig_dtype = params_dtype
        layer.weight_block_size = self.weight_block_size
        if input_size_per_partition % 16:
            raise ValueError(f"fp8 weight-only linear needs the input size per rank ({input_size_per_partition}) "
                             "to be a multiple of 16")
        layer.register_parameter("weight", ModelWeightParameter(
            data=torch.empty(n_part, input_size_per_partition, dtype=torch.float8_e4m3fn),
            input_dim=1, output_dim=0, weight_loader=loader))
        if not self.block_quant:
            s = PerTensorScaleParameter(data=torch.full((len(output_partition_sizes),), torch.finfo(torch.float32).min,
                                                        dtype=torch.float32), weight_loader=loader)
            layer.register_parameter("weight_scale", s)
        else:
            bn, bk = self.weight_block_size
            tp_sharded_k = input_size != input_size_per_partition
            if bk % 16 or (tp_sharded_k and input_size_per_partition % bk):
                raise ValueError(f"Weight input_size_per_partition = {input_size_per_partition} is not divisible by "
                                 f"weight quantization block_k = {bk}.")
            for w in output_partition_sizes[:-1] if len(output_partition_sizes) > 1 else []:
                if w % bn:
                    raise ValueError(f"Weight output_partition_size = {w} is not divisible by weight quantization "
                                     f"block_n = {bn}.")
            s = BlockQuantScaleParameter(
                data=torch.empty((n_part + bn - 1) // bn, (input_size_per_partition + bk - 1) // bk, dtype=torch.float32),
                input_dim=1, output_dim=0, weight_loader=loader)
            layer.register_parameter("weight_scale_inv", s)
        if self.quant_config.activation_scheme == "static":
            layer.register_parameter("input_scale", PerTensorScaleParameter(
                data=torch.full((len(output_partition_sizes),), torch.finfo(torch.float32).min, dtype=torch.float32),
                weight_loader=loader))

    def process_weights_after_loading(self, layer) -> None:
        _set(layer, "weight", layer.weight.data)
        if self.block_quant:
            _set(layer, "weight_scale_inv", layer.weight_scale_inv.data.contiguous())
            layer.mi355q_scale_block = tuple(self.weight_block_size)
        else:
            # one scale per logical matrix -> one per output channel (no requantisation to a common
            # maximum is needed, unlike the reference's w8a8 route, fp8.py:384-404)
            s = layer.weight_scale.data
            widths = torch.tensor(layer.logical_widths, device=s.device)
            if s.numel() == 1:
                s = s.reshape(1).expand(len(layer.logical_widths))
            _set(layer, "weight_scale", torch.repeat_interleave(s.to(torch.float32), widths).contiguous())
            layer.mi355q_scale_block = None
        if hasattr(layer, "input_scale"):
            _set(layer, "input_scale", None)

    def apply(self, layer, x, bias=None):
        scales = layer.weight_scale_inv if self.block_quant else layer.weight_scale
        return ops.w8a16_gemm(x, layer.weight.data.view(torch.uint8), scales.data, layer.mi355q_scale_block, bias)


class Fp8PerTensorOnlineLinearMethod(Fp8LinearMethod):
    """16-bit checkpoint, quantised at load: scale = max|W| / 448, W8 = fp8(W / scale)."""

    def __init__(self):
        self.quant_config, self.weight_block_size, self.block_quant = None, None, False

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs):
        layer.logical_widths = list(output_partition_sizes)
        layer.register_parameter("weight", ModelWeightParameter(
            data=torch.empty(sum(output_partition_sizes), input_size_per_partition, dtype=params_dtype),
            input_dim=1, output_dim=0, weight_loader=extra_weight_attrs.get("weight_loader")))

    def process_weights_after_loading(self, layer) -> None:
        w = layer.weight.data.to(torch.float32)
        scale = (w.abs().amax() / FP8_MAX).clamp(min=1e-12)
        _set(layer, "weight", (w / scale).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn))
        _set(layer, "weight_scale", scale.reshape(1).expand(w.shape[0]).contiguous())
        layer.mi355q_scale_block = None
