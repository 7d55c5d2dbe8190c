"""compressed-tensors checkpoints (`quant_method: "compressed-tensors"`): weight-only schemes and fp8 W8A8.

Mirrors vllm/model_executor/layers/quantization/compressed_tensors/ of the reference:
`CompressedTensorsConfig` (compressed_tensors.py:82-968: config groups -> target scheme map, ignore
list, per-layer scheme resolution), `CompressedTensorsLinearMethod` (:971-1019),
`CompressedTensorsKVCacheMethod` (:1022-1072), the target / ignore matching helpers (utils.py:40-239)
and the schemes that reach this package's kernels:

  CompressedTensorsWNA16     (schemes/compressed_tensors_wNa16.py:51-257)  pack-quantized INT4 / INT8,
                              group or channel scales, optional zero points and activation ordering
                              -> MPLinearKernel chooser -> MI355X W4A16 / W8A16 kernels
  CompressedTensorsW8A16Fp8  (schemes/compressed_tensors_w8a16_fp8.py:38-160)  fp8 e4m3 weights with
                              tensor / channel / block scales -> fp8 weight-only GEMM
  CompressedTensorsW8A8Fp8   (schemes/compressed_tensors_w8a8_fp8.py:60-207)  fp8 weights with tensor / channel
                              scales, activations quantised per tensor (static) or per token (dynamic)
                              -> scaled fp8 x fp8 product

The `compressed_tensors` library itself (pydantic models `QuantizationArgs`, enums) is a
third-party dependency that is absent from the reference tree; the fields the reference reads from it
are restated in `QuantizationArgs` below.  Pack-quantized storage (the library's `pack_to_int32`):
signed values are offset by 2^(bits-1) to unsigned and packed least-significant-first, weights along
the input dimension (`weight_packed` int32 [N, K*bits/32]), zero points along the output dimension
(`weight_zero_point` int32 [N*bits/32, K/G]); scales are `weight_scale` [N, K/G].

Other schemes that quantize activations (W8A8 int8, W4A8, nvfp4 / mxfp4 / mxfp8) are a different arithmetic
path and are not built: asking for their scheme raises NotImplementedError naming the scheme.  A
block-quantised fp8 W8A8 checkpoint with blocks other than 128 x 128 (and any fp8 W8A8 checkpoint under
MI355Q_FP8_LINEAR=w8a16) is served weight-only (its activation scale is loaded and dropped), which is what
the reference does on devices without fp8 matrix support (compressed_tensors.py:831-837).
"""
import math
import os
import re
from dataclasses import dataclass, field
from fractions import Fraction
from types import MappingProxyType
from typing import Any, Iterable, Mapping

import torch

from .. import ops
from .kv_cache import BaseKVCacheMethod
from .linear import LinearBase, LinearMethodBase, UnquantizedLinearMethod
from .mp_kernel import MPLinearLayerConfig, choose_mp_linear_kernel
from .parameter import (BasevLLMParameter, BlockQuantScaleParameter, ChannelQuantScaleParameter,
                        GroupQuantScaleParameter, ModelWeightParameter, PackedColumnParameter, PackedvLLMParameter,
                        PerTensorScaleParameter, RowvLLMParameter)
from .registry import QuantizationConfig, register_quantization_config
from .scalar_type import scalar_types

# CompressionFormat values of the compressed-tensors library that the reference tests for
PACK_QUANTIZED = "pack-quantized"
_ACTIVATION_QUANTIZATION_FORMATS = ("naive-quantized", "int-quantized", "float-quantized", "nvfp4-pack-quantized")


def is_activation_quantization_format(fmt: str | None) -> bool:
    """utils.py:40-47."""
    return fmt in _ACTIVATION_QUANTIZATION_FORMATS


@dataclass
class QuantizationArgs:
    """The fields of compressed_tensors.quantization.QuantizationArgs that the reference reads."""
    num_bits: int = 8
    type: str = "int"                      # "int" | "float"
    symmetric: bool = True
    group_size: int | None = None
    strategy: str | None = None            # "tensor" | "channel" | "group" | "block" | "token" | "tensor_group" | "attn_head"
    block_structure: list[int] | None = None
    dynamic: bool | str = False            # True / False / "local"
    actorder: str | None = None            # "group" | "weight" | None
    scale_dtype: str | None = None
    extra: dict = field(default_factory=dict)

    @classmethod
    def model_validate(cls, d: "dict[str, Any] | QuantizationArgs | None") -> "QuantizationArgs | None":
        if d is None or isinstance(d, QuantizationArgs):
            return d
        known = {"num_bits", "type", "symmetric", "group_size", "strategy", "block_structure", "dynamic",
                 "actorder", "scale_dtype"}
        a = cls(**{k: v for k, v in d.items() if k in known}, extra={k: v for k, v in d.items() if k not in known})
        a.type = str(a.type).lower()
        # the library's validators: strategy from group_size when it is not given; "static" / "dynamic"
        # and booleans as aliases of the two activation orderings; "NxM" strings as block shapes
        if a.strategy is None:
            if a.group_size is None:
                a.strategy = "tensor"
            elif a.group_size > 0:
                a.strategy = "group"
            elif a.group_size == -1:
                a.strategy = "channel"
            else:
                raise ValueError(f"group_size={a.group_size} is neither positive nor -1 (channel-wise)")
        a.strategy = str(a.strategy).lower()
        if a.strategy == "group" and (a.group_size is None or a.group_size <= 0):
            raise ValueError("strategy 'group' requires a positive group_size")
        if isinstance(a.actorder, bool):
            a.actorder = "group" if a.actorder else None
        elif isinstance(a.actorder, str):
            a.actorder = {"dynamic": "group", "static": "weight"}.get(a.actorder.lower(), a.actorder.lower())
            if a.actorder not in ("group", "weight"):
                raise ValueError(f"unknown actorder {a.actorder!r}")
        if isinstance(a.block_structure, str):
            a.block_structure = [int(v) for v in a.block_structure.lower().split("x")]
        if a.block_structure is not None and len(a.block_structure) != 2:
            raise ValueError(f"block_structure must have two entries, got {a.block_structure}")
        return a


# -- target / ignore matching (utils.py:50-239) ---------------------------------------------------
def _is_equal_or_regex_match(value: str, target: str, check_contains: bool = False) -> bool:
    if target.startswith("re:"):
        return re.match(target[3:], value) is not None
    if check_contains:
        return target.lower() in value.lower()
    return target == value


def check_equal_or_regex_match(layer_name: str, targets: Iterable[str]) -> bool:
    return any(_is_equal_or_regex_match(layer_name, t) for t in targets)


def should_ignore_layer(layer_name: str | None, ignore: Iterable[str] = (),
                        fused_mapping: Mapping[str, list[str]] = MappingProxyType({})) -> bool:
    """A fused layer (qkv_proj, gate_up_proj) is ignored when every checkpoint shard of it is; shards
    that disagree are an error (utils.py:50-102)."""
    if layer_name is None:
        return False
    ignore = list(ignore)
    proj_name = layer_name.split(".")[-1]
    if proj_name in fused_mapping and layer_name not in ignore:
        verdicts = {check_equal_or_regex_match(layer_name.replace(proj_name, shard), ignore)
                    for shard in fused_mapping[proj_name]}
        if len(verdicts) != 1:
            raise ValueError(f"Found a different quantization schemes for {fused_mapping[proj_name]} in "
                             f"{layer_name}. vLLM requires all to use the same scheme.")
        return verdicts.pop()
    return check_equal_or_regex_match(layer_name, ignore)


def _find_first_match(value: str, targets: Iterable[str], check_contains: bool = False) -> str | None:
    return next((t for t in targets if _is_equal_or_regex_match(value, t, check_contains)), None)


def _match_fused_layer(layer_name: str, targets: Iterable[str], fused_mapping: Mapping[str, list[str]]) -> str | None:
    """Every shard of the fused layer must match a target; the first shard's target is returned
    (utils.py:196-239)."""
    fused = next((key for key in fused_mapping if layer_name.endswith(key)), None)
    if fused is None:
        return None
    targets = list(targets)
    matches = [next((t for t in targets if _is_equal_or_regex_match(layer_name.replace(fused, shard), t)), None)
               for shard in fused_mapping[fused]]
    return matches[0] if matches and all(matches) else None


def find_matched_target(layer_name: str | None, module: torch.nn.Module, targets: Iterable[str],
                        fused_mapping: Mapping[str, list[str]] = MappingProxyType({})) -> str | None:
    """Layer name, then the fused-layer expansion, then the module's class name (substring match,
    case-insensitive: "Linear" matches QKVParallelLinear) -- utils.py:113-152."""
    layer_name = layer_name or ""
    targets = list(targets)
    return (_find_first_match(layer_name, targets)
            or _match_fused_layer(layer_name, targets, fused_mapping)
            or _find_first_match(module.__class__.__name__, targets, True))


# -- schemes ----------------------------------------------------------------------------------------
class CompressedTensorsScheme:
    """schemes/compressed_tensors_scheme.py: what CompressedTensorsLinearMethod delegates to."""

    @classmethod
    def get_min_capability(cls) -> int:
        raise NotImplementedError

    def create_weights(self, *args, **kwargs):
        raise NotImplementedError

    def apply_weights(self, layer, x, bias):
        raise NotImplementedError

    def process_weights_after_loading(self, layer) -> None:
        raise NotImplementedError


WNA16_SUPPORTED_TYPES_MAP = {4: scalar_types.uint4b8, 8: scalar_types.uint8b128}
WNA16_ZP_SUPPORTED_TYPES_MAP = {4: scalar_types.uint4}
# widths the format defines (compressed_tensors_wNa16.py:38-47) whose kernels are not built here
_WNA16_FORMAT_BITS = (2, 3, 4, 5, 6, 7, 8)


def repeat_scales_on_all_ranks(act_order: bool, group_size: int, is_row_parallel: bool) -> bool:
    """marlin_utils.py:440-446."""
    return act_order or (group_size == -1 and is_row_parallel)


class CompressedTensorsWNA16(CompressedTensorsScheme):
    """schemes/compressed_tensors_wNa16.py:51-257."""
    _kernel_backends_being_used: set[str] = set()

    def __init__(self, strategy: str, num_bits: int, group_size: int | None = None, symmetric: bool | None = True,
                 actorder: str | None = None, layer_name: str | None = None):
        self.num_bits = num_bits
        self.pack_factor = Fraction(32, num_bits)
        self.strategy = strategy
        self.symmetric = symmetric
        self.group_size = -1 if group_size is None else group_size
        self.has_g_idx = actorder == "group"
        self.layer_name = layer_name
        if self.group_size == -1 and self.strategy != "channel":
            raise ValueError("Pack-quantized format requires group quantization or channelwise quantization, but "
                             "found no group size and strategy is not channelwise.")
        if num_bits not in _WNA16_FORMAT_BITS:
            raise ValueError(f"Unsupported num_bits = {num_bits}. Supported num_bits = {list(_WNA16_FORMAT_BITS)}")
        table = WNA16_SUPPORTED_TYPES_MAP if self.symmetric else WNA16_ZP_SUPPORTED_TYPES_MAP
        if num_bits not in table:
            kind = "Symmetric" if self.symmetric else "Asymmetric"
            raise ValueError(f"{kind} quantization not supported for num_bits = {num_bits}. Supported: {list(table)}")
        self.quant_type = table[num_bits]

    @classmethod
    def get_min_capability(cls) -> int:
        return 75

    def create_weights(self, layer, output_size, input_size, output_partition_sizes, input_size_per_partition,
                       params_dtype, weight_loader, **kwargs):
        n_part = sum(output_partition_sizes)
        layer.input_size_per_partition, layer.output_size_per_partition = input_size_per_partition, n_part
        layer.output_partition_sizes, layer.params_dtype = output_partition_sizes, params_dtype
        kcfg = MPLinearLayerConfig(full_weight_shape=(input_size, output_size),
                                   partition_weight_shape=(input_size_per_partition, n_part),
                                   weight_type=self.quant_type, act_type=params_dtype, group_size=self.group_size,
                                   zero_points=not self.symmetric, has_g_idx=self.has_g_idx)
        kernel_type = choose_mp_linear_kernel(kcfg)
        self._kernel_backends_being_used.add(kernel_type.__name__)

        G = self.group_size if self.group_size != -1 else input_size
        row_parallel = input_size != input_size_per_partition
        partition_scales = not repeat_scales_on_all_ranks(self.has_g_idx, self.group_size, row_parallel)
        groups = input_size // G
        if partition_scales:
            if input_size_per_partition % G:
                raise ValueError(f"{self.layer_name}: input size per rank {input_size_per_partition} is not a "
                                 f"multiple of the group size {G}")
            groups = input_size_per_partition // G

        pf = self.pack_factor
        layer.register_parameter("weight_packed", PackedvLLMParameter(
            data=torch.empty(n_part, math.ceil(input_size_per_partition * self.num_bits / 32), dtype=torch.int32),
            input_dim=1, output_dim=0, packed_dim=1, packed_factor=pf, weight_loader=weight_loader))
        s = torch.empty(n_part, groups, dtype=params_dtype)
        z = torch.zeros(math.ceil(n_part * self.num_bits / 32), groups, dtype=torch.int32)
        if not partition_scales:
            layer.register_parameter("weight_scale", ChannelQuantScaleParameter(data=s, output_dim=0,
                                                                                weight_loader=weight_loader))
            if not self.symmetric:
                layer.register_parameter("weight_zero_point", PackedColumnParameter(
                    data=z, output_dim=0, packed_dim=0, packed_factor=pf, weight_loader=weight_loader))
        else:
            layer.register_parameter("weight_scale", GroupQuantScaleParameter(data=s, output_dim=0, input_dim=1,
                                                                              weight_loader=weight_loader))
            if not self.symmetric:
                layer.register_parameter("weight_zero_point", PackedvLLMParameter(
                    data=z, input_dim=1, output_dim=0, packed_dim=0, packed_factor=pf, weight_loader=weight_loader))
        # the unpacked [N, K] shape the checkpoint records beside the packed words
        layer.register_parameter("weight_shape", BasevLLMParameter(data=torch.empty(2, dtype=torch.int64),
                                                                   weight_loader=weight_loader))
        if self.has_g_idx:
            layer.register_parameter("weight_g_idx", RowvLLMParameter(
                data=torch.empty(input_size_per_partition, dtype=torch.int32), input_dim=0,
                weight_loader=weight_loader))
        self.kernel = kernel_type(kcfg, w_q_param_name="weight_packed", w_s_param_name="weight_scale",
                                  w_zp_param_name="weight_zero_point", w_gidx_param_name="weight_g_idx")

    def process_weights_after_loading(self, layer) -> None:
        self.kernel.process_weights_after_loading(layer)

    def apply_weights(self, layer, x, bias=None):
        return self.kernel.apply_weights(layer, x, bias)


def _set(layer: torch.nn.Module, name: str, t: torch.Tensor | None) -> None:
    if name in layer._parameters:
        del layer._parameters[name]
    layer.register_parameter(name, None if t is None else torch.nn.Parameter(t, requires_grad=False))


class CompressedTensorsW8A16Fp8(CompressedTensorsScheme):
    """schemes/compressed_tensors_w8a16_fp8.py:38-160.  weight float8_e4m3fn [N, K]; weight_scale fp32,
    one per logical matrix (tensor), [N, 1] (channel) or [N/bn, K/bk] (block)."""

    def __init__(self, weight_quant: QuantizationArgs, is_static_input_scheme: bool):
        self.weight_quant = weight_quant
        self.strategy = weight_quant.strategy
        self.is_static_input_scheme = bool(is_static_input_scheme)
        self.weight_block_size = weight_quant.block_structure
        if self.strategy not in ("tensor", "channel", "block"):
            raise ValueError(f"fp8 weight-only scheme: unsupported weight strategy {self.strategy!r}")
        if self.strategy == "block" and self.weight_block_size is None:
            raise ValueError("strategy 'block' needs block_structure")

    @classmethod
    def get_min_capability(cls) -> int:
        return 75

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, weight_loader, **kwargs):
        n_part = sum(output_partition_sizes)
        layer.logical_widths = list(output_partition_sizes)
        layer.input_size_per_partition, layer.output_size_per_partition = input_size_per_partition, n_part
        layer.orig_dtype = params_dtype
        layer.weight_block_size = None
        if input_size_per_partition % 16:
            raise ValueError(f"fp8 weight-only linear needs the input size per rank ({input_size_per_partition}) "
                             "to be a multiple of 16")
        layer.register_parameter("weight", ModelWeightParameter(
            data=torch.empty(n_part, input_size_per_partition, dtype=torch.float8_e4m3fn),
            input_dim=1, output_dim=0, weight_loader=weight_loader))
        if self.strategy == "block":
            bn, bk = self.weight_block_size
            layer.weight_block_size = [bn, bk]
            # fp8_utils.validate_fp8_block_shape
            if bk % 16 or (input_size != input_size_per_partition and input_size_per_partition % bk):
                raise ValueError(f"Weight input_size_per_partition = {input_size_per_partition} is not divisible by "
                                 f"weight quantization block_k = {bk}.")
            for w in output_partition_sizes[:-1] if len(output_partition_sizes) > 1 else []:
                if w % bn:
                    raise ValueError(f"Weight output_partition_size = {w} is not divisible by weight quantization "
                                     f"block_n = {bn}.")
            scale = BlockQuantScaleParameter(
                data=torch.empty((n_part + bn - 1) // bn, (input_size_per_partition + bk - 1) // bk,
                                 dtype=torch.float32), input_dim=1, output_dim=0, weight_loader=weight_loader)
        elif self.strategy == "channel":
            scale = ChannelQuantScaleParameter(data=torch.empty(n_part, 1, dtype=torch.float32), output_dim=0,
                                               weight_loader=weight_loader)
        else:
            scale = PerTensorScaleParameter(
                data=torch.full((len(output_partition_sizes),), torch.finfo(torch.float32).min, dtype=torch.float32),
                weight_loader=weight_loader)
        layer.register_parameter("weight_scale", scale)
        if self.is_static_input_scheme:
            layer.register_parameter("input_scale", PerTensorScaleParameter(
                data=torch.empty(len(output_partition_sizes), dtype=torch.float32), weight_loader=weight_loader))

    def process_weights_after_loading(self, layer) -> None:
        _set(layer, "weight", layer.weight.data)
        s = layer.weight_scale.data.to(torch.float32)
        if self.strategy == "block":
            _set(layer, "weight_scale", s.contiguous())
            layer.mi355q_scale_block = tuple(self.weight_block_size)
        else:
            if self.strategy == "tensor":
                # one scale per logical matrix -> one per output channel (w8a8_utils.convert_to_channelwise)
                widths = torch.tensor(layer.logical_widths, device=s.device)
                if s.numel() == 1:
                    s = s.reshape(1).expand(len(layer.logical_widths))
                s = torch.repeat_interleave(s, widths)
            _set(layer, "weight_scale", s.reshape(-1).contiguous())
            layer.mi355q_scale_block = None
        if hasattr(layer, "input_scale"):
            _set(layer, "input_scale", None)      # weight-only: the activation scale is not used

    def apply_weights(self, layer, x, bias=None):
        return ops.w8a16_gemm(x, layer.weight.data.view(torch.uint8), layer.weight_scale.data,
                              layer.mi355q_scale_block, bias)


class CompressedTensorsW8A8Fp8(CompressedTensorsW8A16Fp8):
    """schemes/compressed_tensors_w8a8_fp8.py:60-207: activations are quantised to fp8 -- with the checkpoint's
    input_scale (static: the largest over fused shards, :193-194), with one run-time scale per token (dynamic:
    kFp8DynamicTokenSym, :47-50), or, for 128 x 128 block-quantised weights, with one run-time scale per token
    and 128-wide group (:69-79) -- and the product runs as a scaled fp8 x fp8 matrix product.  Per-tensor weights of a fused layer are requantised to their largest scale
    (process_fp8_weight_tensor_strategy, fp8_utils.py:1321-1346); per-channel weights keep their scales
    (process_fp8_weight_channel_strategy, :1349-1364).  Checkpoint tensors are those of the weight-only
    scheme."""

    @staticmethod
    def block_shape_served(block_structure) -> bool:
        """Weight blocks the block-scaled product takes: 128 x 128 (the activation groups are as wide as a block
        is tall, :73, and one group must span exactly one block of K)."""
        return block_structure is not None and list(block_structure) == [128, 128]

    def __init__(self, weight_quant: QuantizationArgs, is_static_input_scheme: bool):
        super().__init__(weight_quant, is_static_input_scheme)
        if self.strategy == "block":
            if self.is_static_input_scheme:      # :72
                raise ValueError("block-quantised fp8 weights take dynamic activation scales only")
            if not self.block_shape_served(self.weight_block_size):
                raise ValueError(f"block-quantised fp8 W8A8 serves 128 x 128 blocks, not {self.weight_block_size}; "
                                 "other shapes run weight-only (CompressedTensorsW8A16Fp8)")

    def process_weights_after_loading(self, layer) -> None:
        from .fp8 import FP8_MAX
        w, ws = layer.weight.data, layer.weight_scale.data.to(torch.float32)
        if self.strategy == "block":              # :167-174: weights and [N/bn, K/bk] scales stay as loaded
            _set(layer, "weight", w)
            _set(layer, "weight_scale", ws.contiguous())
            if hasattr(layer, "input_scale"):
                _set(layer, "input_scale", None)
            layer.mi355q_scale_block = tuple(self.weight_block_size)
            return
        if self.strategy == "tensor":
            ws = ws.reshape(-1)
            max_s = ws.max()
            if ws.numel() > 1 and ws[-1] > torch.finfo(torch.float8_e4m3fn).min:   # one scale per shard was loaded
                start = 0
                for idx, width in enumerate(layer.logical_widths):
                    if width:
                        dq = w[start:start + width].to(torch.float16) * ws[idx]
                        if w.is_cuda:
                            q, _ = ops.scaled_fp8_quant(dq.contiguous(), scale=max_s.reshape(1))
                        else:
                            q = (dq.float() * (1.0 / max_s)).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
                        w[start:start + width] = q
                    start += width
            ws = max_s.reshape(1)
        else:
            ws = ws.reshape(-1)
        _set(layer, "weight", w)
        _set(layer, "weight_scale", ws.contiguous())
        if self.is_static_input_scheme:
            _set(layer, "input_scale", layer.input_scale.data.to(torch.float32).max().reshape(1).contiguous())
        else:
            _set(layer, "input_scale", None)
        layer.mi355q_scale_block = None

    def apply_weights(self, layer, x, bias=None):
        x2 = x.reshape(-1, x.shape[-1])
        if self.strategy == "block":
            # one run-time scale per token and 128-wide group, then the block-scaled product
            # (W8A8BlockFp8LinearOp, fp8_utils.py:533-659)
            bn, bk = self.weight_block_size
            xq, xs = ops.per_token_group_quant_fp8(x2.contiguous(), bk)
            out = ops.w8a8_block_scaled_mm(xq, layer.weight.data, xs, layer.weight_scale.data, [bn, bk], x.dtype, bias)
            return out.reshape(x.shape[:-1] + (out.shape[-1],))
        if self.is_static_input_scheme:
            xq, xs = ops.scaled_fp8_quant(x2, scale=layer.input_scale.data)
        else:
            xq, xs = ops.scaled_fp8_quant(x2, use_per_token_if_dynamic=True)
        out = ops.fp8_scaled_mm(xq, layer.weight.data, x.dtype, xs, layer.weight_scale.data, bias)
        return out.reshape(x.shape[:-1] + (out.shape[-1],))


# -- config ----------------------------------------------------------------------------------------
@register_quantization_config("compressed-tensors")
class CompressedTensorsConfig(QuantizationConfig):
    def __init__(self, target_scheme_map: dict[str, Any], ignore: list[str], quant_format: str,
                 kv_cache_scheme: dict[str, Any] | None = None, config: dict[str, Any] | None = None,
                 transform_config: dict[str, Any] | None = None, total_num_heads: int | None = None,
                 total_num_kv_heads: int | None = None):
        super().__init__()
        self.ignore = ignore
        self.quant_format = quant_format
        self.target_scheme_map = target_scheme_map
        self.kv_cache_scheme = kv_cache_scheme
        self.config = config
        self.total_num_heads, self.total_num_kv_heads = total_num_heads, total_num_kv_heads
        if transform_config:
            raise NotImplementedError("compressed-tensors transform_config (rotations applied around the linear "
                                      "layers) is not supported")
        self.transform_config = None

    def get_linear_method(self) -> "CompressedTensorsLinearMethod":
        return CompressedTensorsLinearMethod(self)

    @classmethod
    def get_supported_act_dtypes(cls) -> list[torch.dtype]:
        return [torch.float32, torch.float16, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 70

    @classmethod
    def get_name(cls) -> str:
        return "compressed-tensors"

    @classmethod
    def get_config_filenames(cls) -> list[str]:
        return []

    def apply_vllm_mapper(self, hf_to_vllm_mapper) -> None:
        """Layer paths among the targets follow the model's renaming; class names ("Linear") and
        "re:" patterns are kept (compressed_tensors.py:122-151)."""
        def _map(target: str):
            if "." in target and not target.startswith("re:"):
                return hf_to_vllm_mapper._map_name(target)
            return target

        self.target_scheme_map = {k: v for t, v in self.target_scheme_map.items() if (k := _map(t)) is not None}
        self.ignore = [t for x in self.ignore if (t := _map(x)) is not None]
        if self.kv_cache_scheme is not None:
            self.kv_cache_scheme = {k: v for t, v in self.kv_cache_scheme.items() if (k := _map(t)) is not None}

    @classmethod
    def from_config(cls, config: dict[str, Any]) -> "CompressedTensorsConfig":
        # groups that target only an attention module are dropped: attention quantization is tied to
        # the KV-cache scheme (compressed_tensors.py:235-247)
        if "config_groups" in config:
            config = dict(config)
            config["config_groups"] = {
                k: v for k, v in config["config_groups"].items()
                if not (len(v["targets"]) == 1 and v["targets"][0].endswith("Attention"))}
        sparsity = config.get("sparsity_config")
        if sparsity and sparsity.get("targets"):
            raise DeprecationWarning("Sparsity support has been removed from compressed-tensors. Please use a model "
                                     "without sparsity configuration.")
        return cls(target_scheme_map=cls._quantization_scheme_map_from_config(config),
                   ignore=list(config.get("ignore", []) or []), quant_format=config.get("format"), config=config,
                   transform_config=config.get("transform_config"), kv_cache_scheme=config.get("kv_cache_scheme"),
                   total_num_heads=config.get("total_num_heads"), total_num_kv_heads=config.get("total_num_kv_heads"))

    @classmethod
    def _quantization_scheme_map_from_config(cls, config: dict[str, Any]) -> dict[str, Any]:
        """target -> {"weights", "input_activations", "output_activations", "format"}
        (compressed_tensors.py:299-369)."""
        out: dict[str, Any] = {}
        quant_format = config.get("format")
        for group in (config.get("config_groups") or {}).values():
            for target in group.get("targets"):
                entry = {"weights": QuantizationArgs.model_validate(group.get("weights")), "input_activations": None,
                         "format": group.get("format"), "output_activations": None}
                fmt = entry["format"] if entry["format"] is not None else quant_format
                input_activations = group.get("input_activations")
                if is_activation_quantization_format(fmt) or input_activations:
                    if not input_activations:
                        # the one activation-quantization format without activation arguments is fp8 W8A16
                        if entry["weights"].type != "float":
                            raise ValueError(f"target {target}: format {fmt} without input_activations is only "
                                             "defined for float weights")
                    else:
                        entry["input_activations"] = QuantizationArgs.model_validate(input_activations)
                if group.get("output_activations"):
                    entry["output_activations"] = QuantizationArgs.model_validate(group["output_activations"])
                out[target] = entry
        return out

    # -- scheme predicates (compressed_tensors.py:404-720, the ones this package acts on) --------
    @staticmethod
    def _is_fp4_or_mx(weight_quant: QuantizationArgs) -> str | None:
        if weight_quant.type == "float" and weight_quant.num_bits == 4:
            return "nvfp4 / mxfp4 (4-bit float weights)"
        if weight_quant.type == "float" and weight_quant.num_bits == 8 and weight_quant.strategy in ("group", "tensor_group"):
            return "mxfp8 (8-bit float weights with group scales)"
        return None

    @staticmethod
    def _is_fp8_w8a8(weight_quant: QuantizationArgs, input_quant: QuantizationArgs | None) -> bool:
        """compressed_tensors.py:526-561."""
        if weight_quant is None or input_quant is None:
            return False
        if weight_quant.type != "float" or input_quant.type != "float":
            return False
        if not (weight_quant.symmetric and not weight_quant.dynamic
                and weight_quant.strategy in ("tensor", "channel", "block")):
            return False
        if input_quant.dynamic:
            return True
        return bool(input_quant.symmetric) and input_quant.strategy == "tensor"

    @staticmethod
    def _is_fp8_w8a16(weight_quant: QuantizationArgs, input_quant: QuantizationArgs | None) -> bool:
        """compressed_tensors.py:610-634."""
        if weight_quant is None or weight_quant.type != "float":
            return False
        return bool(weight_quant.symmetric) and not weight_quant.dynamic and \
            weight_quant.strategy in ("tensor", "channel", "block")

    @staticmethod
    def _is_wNa16_group_channel(weight_quant: QuantizationArgs, input_quant: QuantizationArgs | None) -> bool:
        """compressed_tensors.py:636-647."""
        return weight_quant.strategy in ("channel", "group") and input_quant is None and not weight_quant.dynamic

    def _get_scheme_from_parts(self, weight_quant: QuantizationArgs, input_quant: QuantizationArgs | None,
                               output_quant: QuantizationArgs | None = None, format: str | None = None,
                               layer_name: str | None = None) -> CompressedTensorsScheme:
        """Same order of tests as compressed_tensors.py:722-874, reduced to the weight-only schemes."""
        format = format if format is not None else self.quant_format
        unsupported = self._is_fp4_or_mx(weight_quant)
        if unsupported:
            raise NotImplementedError(f"compressed-tensors scheme {unsupported} is not supported ({layer_name=})")
        if weight_quant.type == "int" and (input_quant is not None or output_quant is not None):
            raise NotImplementedError(
                f"compressed-tensors integer weights with quantized activations (W{weight_quant.num_bits}A"
                f"{input_quant.num_bits if input_quant else 16}) are not supported ({layer_name=})")
        if self._is_wNa16_group_channel(weight_quant, input_quant) and format == PACK_QUANTIZED:
            return CompressedTensorsWNA16(num_bits=weight_quant.num_bits, strategy=weight_quant.strategy,
                                          symmetric=weight_quant.symmetric, group_size=weight_quant.group_size,
                                          actorder=weight_quant.actorder, layer_name=layer_name)
        if is_activation_quantization_format(format):
            if self._is_fp8_w8a8(weight_quant, input_quant):
                # block shapes other than 128 x 128 and MI355Q_FP8_LINEAR=w8a16 are served weight-only: the
                # activation arguments are then ignored
                blocked = weight_quant.strategy == "block"
                served = not blocked or (input_quant.dynamic and
                                         CompressedTensorsW8A8Fp8.block_shape_served(weight_quant.block_structure))
                if served and os.environ.get("MI355Q_FP8_LINEAR", "w8a8") == "w8a8":
                    return CompressedTensorsW8A8Fp8(weight_quant=weight_quant,
                                                    is_static_input_scheme=not input_quant.dynamic)
                return CompressedTensorsW8A16Fp8(weight_quant=weight_quant,
                                                 is_static_input_scheme=not input_quant.dynamic)
            if self._is_fp8_w8a16(weight_quant, input_quant):
                return CompressedTensorsW8A16Fp8(
                    weight_quant=weight_quant,
                    is_static_input_scheme=bool(input_quant and not input_quant.dynamic))
        raise NotImplementedError(f"No compressed-tensors compatible scheme was found for {layer_name=}, "
                                  f"{weight_quant=}, {input_quant=}, {output_quant=}, {format=}")

    def get_scheme_dict(self, layer: torch.nn.Module, layer_name: str | None = None) -> dict[str, Any] | None:
        """compressed_tensors.py:925-958."""
        if should_ignore_layer(layer_name, ignore=self.ignore, fused_mapping=self.packed_modules_mapping):
            return None
        if self.target_scheme_map:
            matched = find_matched_target(layer_name=layer_name, module=layer, targets=self.target_scheme_map.keys(),
                                          fused_mapping=self.packed_modules_mapping)
            if matched is not None:
                scheme_dict = self.target_scheme_map[matched]
                if scheme_dict.get("format") is None:
                    scheme_dict["format"] = self.quant_format
                return scheme_dict
        return None

    def get_scheme(self, layer: torch.nn.Module, layer_name: str | None = None) -> CompressedTensorsScheme | None:
        """None = the layer stays unquantized (compressed_tensors.py:876-923)."""
        d = self.get_scheme_dict(layer, layer_name)
        if not d or d.get("weights") is None:
            return None
        return self._get_scheme_from_parts(weight_quant=d["weights"], input_quant=d.get("input_activations"),
                                           output_quant=d.get("output_activations"), format=d.get("format"),
                                           layer_name=layer_name)

    def get_quant_method(self, layer: torch.nn.Module, prefix: str):
        from .attention import Attention
        if isinstance(layer, LinearBase):
            scheme = self.get_scheme(layer=layer, layer_name=prefix)
            if scheme is None:
                return UnquantizedLinearMethod()
            layer.scheme = scheme
            return CompressedTensorsLinearMethod(self)
        if isinstance(layer, Attention):
            return CompressedTensorsKVCacheMethod(self)
        if "fusedmoe" in type(layer).__name__.lower() or "routedexperts" in type(layer).__name__.lower():
            return get_moe_method(self, layer, prefix)
        return None

    def _add_fused_moe_to_target_scheme_map(self) -> None:
        """A scheme that targets "Linear" also covers the routed experts those linears were fused into
        (compressed_tensors.py:215-227)."""
        if "Linear" not in self.target_scheme_map:
            return
        for name in ("RoutedExperts", "FusedMoE"):
            self.target_scheme_map.setdefault(name, self.target_scheme_map["Linear"])

    def has_blocked_weights(self) -> bool:
        return any(s.get("weights") is not None and s["weights"].strategy == "block"
                   for s in self.target_scheme_map.values())


def get_moe_method(quant_config: "CompressedTensorsConfig", layer: torch.nn.Module, layer_name: str):
    """The expert method of a routed-experts layer, resolved from the scheme of expert 0's three projections,
    which must agree (compressed_tensors_moe.py:28-76).  Weight-only 4-bit pack-quantized experts are served;
    an ignored layer (no scheme) has no unquantized expert path here and is refused, as every other scheme is."""
    quant_config._add_fused_moe_to_target_scheme_map()
    dicts = [quant_config.get_scheme_dict(layer, layer_name + proj)
             for proj in (".0.gate_proj", ".0.up_proj", ".0.down_proj")]
    scheme = dicts.pop()
    if not all(d == scheme for d in dicts):
        raise ValueError("All MoE projections need to have same quantization scheme but found multiple")
    if scheme is None:
        raise NotImplementedError(f"compressed-tensors: routed experts '{layer_name}' are not quantized; only "
                                  "quantized experts are served on this path")
    weight_quant, input_quant, fmt = scheme.get("weights"), scheme.get("input_activations"), scheme.get("format")
    if quant_config._is_wNa16_group_channel(weight_quant, input_quant):
        if weight_quant.num_bits != 4 or fmt != PACK_QUANTIZED:
            raise ValueError(f"For Fused MoE layers, only format {PACK_QUANTIZED} with 4-bit weights is supported, "
                             f"got format {fmt} and {weight_quant.num_bits} bits")
        return CompressedTensorsWNA16MoEMethod(weight_quant, input_quant, layer_name)
    raise NotImplementedError(f"compressed-tensors: no routed-experts method for {weight_quant=}, {input_quant=}")


class CompressedTensorsWNA16MoEMethod:
    """Weight-only int4 routed experts of a compressed-tensors checkpoint
    (compressed_tensors_moe/compressed_tensors_moe_wna16.py:52-130, the checks of the branch taken without
    Marlin -- group strategy, no activation reordering by group -- and zero points as the Marlin branch holds them).  Parameters carry the reference's names and
    its transposed shapes (:136-214, :236-330): `w13_weight_packed [E, H/8, 2I]`, `w13_weight_scale [E, H/G, 2I]`,
    `w2_weight_packed [E, I/8, H]`, `w2_weight_scale [E, I/G, H]` (int32 words hold 8 consecutive input channels,
    value + 8 for symmetric checkpoints; asymmetric ones add `w13_weight_zero_point [E, H/G, 2I/8]`,
    `w2_weight_zero_point [E, I/G, H/8]`), plus the `*_weight_shape` pairs the checkpoint stores.  After loading they become the strip layout of
    the W4A16 kernels, gate and up channels paired, and the three routes of `fused_experts` serve them."""

    def __init__(self, weight_quant: "QuantizationArgs", input_quant: "QuantizationArgs | None",
                 layer_name: str | None = None):
        self.weight_quant, self.input_quant = weight_quant, input_quant
        self.num_bits, self.packed_factor = weight_quant.num_bits, 32 // weight_quant.num_bits
        self.strategy, self.group_size, self.actorder = weight_quant.strategy, weight_quant.group_size, weight_quant.actorder
        if self.num_bits != 4:
            raise ValueError("CompressedTensorsWNA16MoEMethod on MI355X supports int4 experts only")
        if weight_quant.strategy != "group" or not self.group_size or self.group_size <= 0:
            raise ValueError("channelwise is not supported by this kernel")
        if weight_quant.actorder in ("group", "dynamic"):
            raise ValueError("grouped actorder isn't supported by this kernel")
        self.symmetric = weight_quant.symmetric   # zero points as the reference's Marlin branch takes them (:357-387)

    def create_weights(self, layer, num_experts: int, hidden_size: int, intermediate_size_per_partition: int,
                       params_dtype: torch.dtype, **extra_weight_attrs):
        E, H, I, G, pf = num_experts, hidden_size, intermediate_size_per_partition, self.group_size, self.packed_factor
        if H % G or I % G:
            raise ValueError(f"group size {G} must divide the hidden size {H} and this rank's intermediate size {I}")
        layer.group_size = G

        def reg(name, t):
            p = torch.nn.Parameter(t, requires_grad=False)
            p.weight_loader = self._loader(layer)
            layer.register_parameter(name, p)

        reg("w13_weight_packed", torch.empty(E, H // pf, 2 * I, dtype=torch.int32))
        reg("w2_weight_packed", torch.empty(E, I // pf, H, dtype=torch.int32))
        reg("w13_weight_scale", torch.ones(E, H // G, 2 * I, dtype=params_dtype))
        reg("w2_weight_scale", torch.ones(E, I // G, H, dtype=params_dtype))
        reg("w13_weight_shape", torch.empty(E, 2, dtype=torch.int64))
        reg("w2_weight_shape", torch.empty(E, 2, dtype=torch.int64))
        if not self.symmetric:                # eight consecutive output channels per word
            reg("w13_weight_zero_point", torch.zeros(E, H // G, 2 * I // pf, dtype=torch.int32))
            reg("w2_weight_zero_point", torch.zeros(E, I // G, H // pf, dtype=torch.int32))

    @staticmethod
    def _loader(layer):
        def loader(param, loaded_weight: torch.Tensor, weight_name: str, shard_id: str, expert_id: int,
                   return_success: bool = False):
            """Checkpoint tensors are output-major (`weight_packed [N, K/8]`, `weight_scale [N, K/G]`,
            `weight_zero_point [N/8, K/G]`); they are
            transposed into the parameter and sharded along the transposed dims: gate (w1) / up (w3) over the
            output channels, down (w2) over the input channels (fused_moe/layer.py weight_loader, `is_transposed`)."""
            if "g_idx" in weight_name:        # static reordering was applied to the weights when they were quantized
                return False if return_success else None
            dst = param.data[expert_id]
            if "weight_shape" in weight_name:
                dst.copy_(loaded_weight.reshape(-1)[:2])
                return True if return_success else None
            t = loaded_weight.t()              # [K/8 or K/G, N]
            rank, size = layer.tp_rank, layer.tp_size
            if shard_id in ("w1", "w3"):
                cols = dst.shape[1] // 2
                src = t.reshape(t.shape[0], size, -1)[:, rank]
                dst[:, :cols].copy_(src) if shard_id == "w1" else dst[:, cols:].copy_(src)
            elif shard_id == "w2":
                dst.copy_(t.reshape(size, -1, t.shape[1])[rank])
            else:
                raise ValueError(f"unknown expert shard id {shard_id}")
            return True if return_success else None

        return loader

    def process_weights_after_loading(self, layer) -> None:
        def packed(w: torch.Tensor, s: torch.Tensor, z: "torch.Tensor | None", pair: bool):
            # output-major bytes, two input channels per byte, low nibble first: the int32 words seen as bytes
            qw = w.data.transpose(1, 2).contiguous().view(torch.uint8)
            # zero points: the words seen as bytes hold two output channels each, [E, K/G, N/2] -> [E, N/2, K/G]
            qz = None if z is None else z.data.contiguous().view(torch.uint8).transpose(1, 2).contiguous()
            return ops.moe_w4a16_repack(qw, s.data.transpose(1, 2).contiguous(), qz, layer.group_size, pair_halves=pair,
                                        decode_slots=int(getattr(layer, "top_k", 1)))

        z13 = None if self.symmetric else layer.w13_weight_zero_point
        z2 = None if self.symmetric else layer.w2_weight_zero_point
        layer.w13_packed = packed(layer.w13_weight_packed, layer.w13_weight_scale, z13, True)
        layer.w2_packed = packed(layer.w2_weight_packed, layer.w2_weight_scale, z2, False)
        for name in ("w13_weight_packed", "w2_weight_packed", "w13_weight_scale", "w2_weight_scale",
                     "w13_weight_shape", "w2_weight_shape", "w13_weight_zero_point", "w2_weight_zero_point"):
            layer._parameters.pop(name, None)

    def apply(self, layer, x: torch.Tensor, topk_weights: torch.Tensor, topk_ids: torch.Tensor) -> torch.Tensor:
        from .moe import fused_experts
        return fused_experts(x, layer.w13_packed, layer.w2_packed, topk_weights, topk_ids, layer.expert_map)


class CompressedTensorsLinearMethod(LinearMethodBase):
    """compressed_tensors.py:971-1019: everything is delegated to the scheme stored on the layer."""

    def __init__(self, quantization_config: CompressedTensorsConfig):
        self.quantization_config = quantization_config

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs):
        layer.scheme.create_weights(layer=layer, input_size=input_size,
                                    input_size_per_partition=input_size_per_partition,
                                    output_partition_sizes=output_partition_sizes, output_size=output_size,
                                    params_dtype=params_dtype, weight_loader=extra_weight_attrs.get("weight_loader"))

    def process_weights_after_loading(self, layer) -> None:
        layer.scheme.process_weights_after_loading(layer)

    def apply(self, layer, x, bias=None):
        scheme = getattr(layer, "scheme", None)
        if scheme is None:
            raise ValueError("A scheme must be defined for each layer")
        return scheme.apply_weights(layer, x, bias=bias)


class CompressedTensorsKVCacheMethod(BaseKVCacheMethod):
    """compressed_tensors.py:1022-1072.  One scale per tensor; per-head scales (strategy "attn_head")
    are refused: the attention kernels of this package read one k and one v scale per layer."""

    def __init__(self, quant_config: CompressedTensorsConfig):
        self.validate_kv_cache_scheme(quant_config.kv_cache_scheme)
        super().__init__(quant_config)

    @staticmethod
    def validate_kv_cache_scheme(kv_cache_scheme: dict[str, Any] | None) -> None:
        if kv_cache_scheme is None:
            return
        type_, num_bits = kv_cache_scheme.get("type"), kv_cache_scheme.get("num_bits")
        if type_ != "float" or num_bits != 8:
            raise NotImplementedError("Currently supported kv cache quantization is num_bits=8, type=float, however "
                                      f"received num_bits={num_bits}, type={type_}")
        strategy = kv_cache_scheme.get("strategy")
        if strategy != "tensor":
            raise NotImplementedError("Invalid strategy for compressed-tensors KV cache. Expected strategies: "
                                      f"('tensor',), found strategy: {strategy}")
        if not kv_cache_scheme.get("symmetric"):
            raise NotImplementedError("Only support symmetric scaling factor for compressed-tensors KV cache. "
                                      f"However found symmetric: {kv_cache_scheme.get('symmetric')}")
