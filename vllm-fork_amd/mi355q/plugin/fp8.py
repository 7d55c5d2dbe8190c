"""fp8 (float8_e4m3fn) checkpoints.

Mirrors vllm/model_executor/layers/quantization/fp8.py:92-231 (Fp8Config), :234-456
(Fp8LinearMethod: weight [N, K] fp8; `weight_scale` one fp32 per logical matrix, or
`weight_scale_inv` [ceil(N/bn), ceil(K/bk)] for block-quantised checkpoints) and :819-…
(Fp8KVCacheMethod).  Two ways to run a linear layer (`Fp8Config.linear_mode`, default from
MI355Q_FP8_LINEAR, else "w8a8"):

  w8a8   what the reference does on ROCm for per-tensor checkpoints: the logical matrices of a fused
         layer are requantised to their largest scale (w8a8_utils.py:76-107), activations are quantised
         to fp8 -- with the checkpoint's `input_scale` (static scheme, the largest over the fused
         shards, fp8.py:397-399) or with a per-tensor scale found at run time (dynamic scheme on
         ROCm, fp8.py:282-287) -- and the product runs as a scaled fp8 x fp8 matrix product
         (kernels/linear/scaled_mm/rocm.py:19-52, ScaledMMLinearKernel.py:135-170).
  w8a16  weight-only: out = x @ (fp8(W) * scale)^T with 16-bit activations, the arithmetic of the
         reference's dequantise-then-linear fallback (:433-454); the reference has no such kernel
         on ROCm (kernels/linear/__init__.py:405-407).  `input_scale` tensors are accepted and ignored.

Block-quantised checkpoints (`weight_block_size` [block_n, block_k], scales in `weight_scale_inv`) in w8a8 mode
run the reference's W8A8BlockFp8LinearOp (fp8_utils.py: per_token_group_quant_fp8 with groups of block_k, then
w8a8_triton_block_scaled_mm): activations get one fp32 scale per (token, block_k-wide group) and every k block of
the product is scaled on its own.  That path is built for block_k = 128 and block_n a multiple of 32; any other
block shape, and w8a16 mode, run weight-only.

16-bit checkpoints are quantised per tensor at load time (quantization/online/fp8.py
`Fp8PerTensorOnlineLinearMethod`).
"""
import os
from typing import Any

import torch

from .. import ops
from .awq import is_layer_skipped
from .kv_cache import BaseKVCacheMethod
from .linear import LinearBase, LinearMethodBase, UnquantizedLinearMethod
from .parameter import BlockQuantScaleParameter, ModelWeightParameter, PerTensorScaleParameter
from .registry import QuantizationConfig, register_quantization_config

ACTIVATION_SCHEMES = ["static", "dynamic"]
LINEAR_MODES = ["w8a8", "w8a16"]
FP8_MAX = 448.0


class Fp8KVCacheMethod(BaseKVCacheMethod):
    pass


@register_quantization_config("fp8")
class Fp8Config(QuantizationConfig):
    def __init__(self, is_checkpoint_fp8_serialized: bool = False, activation_scheme: str = "dynamic",
                 ignored_layers: list[str] | None = None, weight_block_size: list[int] | None = None,
                 store_dtype: str | None = None, linear_mode: str | None = None) -> None:
        super().__init__()
        if activation_scheme not in ACTIVATION_SCHEMES:
            raise ValueError(f"Unsupported activation scheme {activation_scheme}")
        linear_mode = linear_mode or os.environ.get("MI355Q_FP8_LINEAR") or "w8a8"
        if linear_mode not in LINEAR_MODES:
            raise ValueError(f"Unsupported fp8 linear mode {linear_mode!r}; expected one of {LINEAR_MODES}")
        self.linear_mode = linear_mode
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
            return Fp8LinearMethod(self) if self.is_checkpoint_fp8_serialized else Fp8PerTensorOnlineLinearMethod(self)
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
        self.w8a8 = quant_config.linear_mode == "w8a8" and not self.block_quant
        self.w8a8_block = (quant_config.linear_mode == "w8a8" and self.block_quant and self.weight_block_size[1] == 128
                           and self.weight_block_size[0] % 32 == 0)
        self.act_q_static = quant_config.activation_scheme == "static"

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

    def _process_w8a8(self, layer) -> None:
        """process_fp8_weight_tensor_strategy (fp8_utils.py:1321-1346) without the fnuz step: one scale for
        the whole fused weight; shards whose scale differs are dequantised (through fp16, as
        per_tensor_dequantize does) and requantised with the largest."""
        w, ws = layer.weight.data, layer.weight_scale.data.to(torch.float32).reshape(-1)
        max_s = ws.max()
        loaded_per_shard = ws[-1] > torch.finfo(torch.float8_e4m3fn).min   # else: a fused checkpoint, one scale loaded
        if loaded_per_shard and ws.numel() > 1:
            start = 0
            for idx, width in enumerate(layer.logical_widths):
                if width == 0:
                    continue
                dq = w[start:start + width].to(torch.float16) * ws[idx]
                if w.is_cuda:
                    q, _ = ops.scaled_fp8_quant(dq.contiguous(), scale=max_s.reshape(1))
                else:   # host-side loading: the same arithmetic in torch
                    q = (dq.float() * (1.0 / max_s)).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
                w[start:start + width] = q
                start += width
        _set(layer, "weight", w)
        _set(layer, "weight_scale", max_s.reshape(1).contiguous())
        if self.act_q_static:
            if getattr(layer, "input_scale", None) is None:
                raise ValueError("static activation scheme needs an input_scale in the checkpoint")
            _set(layer, "input_scale", layer.input_scale.data.to(torch.float32).max().reshape(1).contiguous())
        else:
            _set(layer, "input_scale", None)   # fp8.py:406-409
        layer.mi355q_scale_block = None

    def process_weights_after_loading(self, layer) -> None:
        if self.w8a8:
            return self._process_w8a8(layer)
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
        if self.w8a8:
            # FP8ScaledMMLinearKernel.apply_weights (ScaledMMLinearKernel.py:135-170): quantise, multiply
            x2 = x.reshape(-1, x.shape[-1])
            xq, xs = ops.scaled_fp8_quant(x2, scale=layer.input_scale.data if self.act_q_static else None)
            out = ops.fp8_scaled_mm(xq, layer.weight.data, x.dtype, xs, layer.weight_scale.data, bias)
            return out.reshape(x.shape[:-1] + (out.shape[-1],))
        if self.w8a8_block:
            # W8A8BlockFp8LinearOp.apply: per-token-group activation scales, block-scaled product
            x2 = x.reshape(-1, x.shape[-1])
            xq, xs = ops.per_token_group_quant_fp8(x2, self.weight_block_size[1])
            out = ops.w8a8_block_scaled_mm(xq, layer.weight.data, xs, layer.weight_scale_inv.data,
                                           list(self.weight_block_size), x.dtype, bias)
            return out.reshape(x.shape[:-1] + (out.shape[-1],))
        scales = layer.weight_scale_inv if self.block_quant else layer.weight_scale
        return ops.w8a16_gemm(x, layer.weight.data.view(torch.uint8), scales.data, layer.mi355q_scale_block, bias)


class Fp8PerTensorOnlineLinearMethod(Fp8LinearMethod):
    """16-bit checkpoint, quantised at load: scale = max|W| / 448, W8 = fp8(W / scale)."""

    def __init__(self, quant_config: Fp8Config | None = None):
        self.quant_config, self.weight_block_size, self.block_quant, self.w8a8_block = quant_config, None, False, False
        self.w8a8 = quant_config is not None and quant_config.linear_mode == "w8a8"
        self.act_q_static = False   # no calibrated input scale exists for a 16-bit checkpoint

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
        _set(layer, "weight_scale", scale.reshape(1).contiguous() if self.w8a8
             else scale.reshape(1).expand(w.shape[0]).contiguous())
        layer.mi355q_scale_block = None
