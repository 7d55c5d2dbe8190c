"""GPTQ checkpoints (AutoGPTQ / GPTQModel `quantize_config.json`).

Mirrors vllm/model_executor/layers/quantization/auto_gptq.py:97-303 (AutoGPTQConfig),
:306-464 (AutoGPTQLinearMethod) and utils/gptq_utils.py:25-163 (per-module `dynamic`
overrides, `modules_in_block_to_quantize`).
"""
import copy
import re
from typing import Any

import torch

from .. import _lib
from .linear import (LinearBase, LinearMethodBase, ParallelLMHead, UnquantizedEmbeddingMethod,
                     UnquantizedLinearMethod)
from .mp_kernel import MPLinearLayerConfig, choose_mp_linear_kernel
from .parameter import (ChannelQuantScaleParameter, GroupQuantScaleParameter, PackedColumnParameter,
                        PackedvLLMParameter, RowvLLMParameter)
from .registry import QuantizationConfig, register_quantization_config
from .scalar_type import scalar_types


def get_dynamic_override(config: "AutoGPTQConfig", layer_name: str, key: str | None = None, default_value=None):
    """First matching `dynamic` rule wins.  "-:regex" excludes the module (returns False);
    "+:regex" or a bare regex overrides fields (gptq_utils.py:51-69)."""
    for pattern, fields in config.dynamic.items():
        if pattern.startswith("-:"):
            if re.match(pattern[2:], layer_name):
                return False
        elif re.match(pattern[2:] if pattern.startswith("+:") else pattern, layer_name):
            return fields if key is None else fields.get(key, default_value)
    return default_value


def override_config(config: "AutoGPTQConfig", prefix: str) -> None:
    """Apply the `dynamic` overrides of module `prefix` in place (gptq_utils.py:25-48)."""
    for attr, key, typ in (("weight_bits", "bits", int), ("group_size", "group_size", int),
                           ("desc_act", "desc_act", bool), ("is_sym", "sym", bool)):
        v = get_dynamic_override(config, prefix, key, getattr(config, attr))
        if isinstance(v, typ) and not (typ is int and isinstance(v, bool)):
            setattr(config, attr, v)
    config.pack_factor = 32 // config.weight_bits
    if (config.weight_bits, config.is_sym) not in config.TYPE_MAP:
        raise ValueError(f"Unsupported quantization config: bits={config.weight_bits}, sym={config.is_sym}")
    config.quant_type = config.TYPE_MAP[(config.weight_bits, config.is_sym)]


def _flatten(x):
    for i in x:
        if isinstance(i, (list, tuple)):
            yield from _flatten(i)
        else:
            yield i


def is_layer_gptq_quantized(prefix: str, quantized_layers: list, fused_mapping: dict[str, list[str]] | None = None) -> bool:
    """Is `prefix` named (by substring) in `modules_in_block_to_quantize`?  For a fused module
    every checkpoint shard must agree (gptq_utils.py:86-130)."""
    names = list(_flatten(quantized_layers))
    proj = prefix.split(".")[-1]
    shards = (fused_mapping or {}).get(proj)
    if not shards:
        return any(n in prefix for n in names)
    verdicts = {any(n in prefix.replace(proj, s) for n in names) for s in shards}
    if len(verdicts) != 1:
        raise ValueError(f"Detected some but not all shards of {prefix} are quantized. "
                         "All shards of fused layers to have the same precision.")
    return verdicts.pop()


def get_linear_quant_method(config: "AutoGPTQConfig", layer: torch.nn.Module, prefix: str, linear_method_cls: type):
    """Per-layer method: unquantized when excluded, otherwise `linear_method_cls` on a private
    copy of the config with this module's overrides applied (gptq_utils.py:133-163)."""
    cfg = copy.deepcopy(config)
    lm_head = isinstance(layer, ParallelLMHead) and cfg.lm_head_quantized
    if not (isinstance(layer, LinearBase) or lm_head):
        return None
    listed = (not cfg.modules_in_block_to_quantize or
              is_layer_gptq_quantized(prefix, cfg.modules_in_block_to_quantize, cfg.packed_modules_mapping))
    if get_dynamic_override(cfg, prefix) is False or not listed:
        return UnquantizedEmbeddingMethod() if lm_head else UnquantizedLinearMethod()
    if prefix:
        override_config(cfg, prefix)
    return linear_method_cls(cfg)


@register_quantization_config("auto_gptq")
class AutoGPTQConfig(QuantizationConfig):
    TYPE_MAP = {(4, True): scalar_types.uint4b8, (8, True): scalar_types.uint8b128}

    def __init__(self, weight_bits: int, group_size: int, desc_act: bool, is_sym: bool, lm_head_quantized: bool,
                 dynamic: dict[str, dict[str, int | bool]], full_config: dict[str, Any],
                 modules_in_block_to_quantize: list[str] | None = None) -> None:
        super().__init__()
        # one group per channel: the activation order cannot matter
        self.desc_act = desc_act and group_size != -1
        self.dynamic = dynamic
        self.weight_bits, self.is_sym = weight_bits, is_sym
        self.pack_factor = 32 // weight_bits
        self.group_size = group_size
        self.lm_head_quantized = lm_head_quantized
        self.full_config = full_config
        if (weight_bits, is_sym) not in self.TYPE_MAP:
            raise ValueError(f"Unsupported quantization config: bits={weight_bits}, sym={is_sym}")
        self.quant_type = self.TYPE_MAP[(weight_bits, is_sym)]
        self.modules_in_block_to_quantize = modules_in_block_to_quantize or []
        self.autoround_version = full_config.get("autoround_version", "")

    def __repr__(self) -> str:
        return (f"AutoGPTQConfig(quant_type={self.quant_type}, group_size={self.group_size}, "
                f"desc_act={self.desc_act}, lm_head_quantized={self.lm_head_quantized}, dynamic={self.dynamic}, "
                f"modules_in_block_to_quantize={self.modules_in_block_to_quantize})")

    @classmethod
    def get_name(cls) -> str:
        return "auto_gptq"

    @classmethod
    def get_supported_act_dtypes(cls) -> list[torch.dtype]:
        return [torch.half, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 60

    @classmethod
    def get_config_filenames(cls) -> list[str]:
        return ["quantize_config.json"]

    @classmethod
    def from_config(cls, config: dict[str, Any]) -> "AutoGPTQConfig":
        dynamic = cls.get_from_keys_or(config, ["dynamic"], default={}) or {}
        return cls(cls.get_from_keys(config, ["bits"]), cls.get_from_keys(config, ["group_size"]),
                   cls.get_from_keys(config, ["desc_act"]), cls.get_from_keys(config, ["sym"]),
                   cls.get_from_keys_or(config, ["lm_head"], default=False), dynamic, config,
                   cls.get_from_keys_or(config, ["modules_in_block_to_quantize"], default=None))

    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant, hf_config=None) -> str | None:
        if str(hf_quant_cfg.get("quant_method", "")).lower() != "gptq":
            return None
        if user_quant is None or user_quant in ("gptq", "gptq_marlin", "auto_gptq", "marlin"):
            return cls.get_name()
        return None

    def get_quant_method(self, layer: torch.nn.Module, prefix: str):
        return get_linear_quant_method(self, layer, prefix, AutoGPTQLinearMethod)

    def maybe_update_config(self, model_name: str, hf_config=None, revision: str | None = None,
                            safetensors_metadata: dict[str, dict] | None = None) -> None:
        """Without an explicit module list, every module whose checkpoint tensors are not
        f16/bf16/f32 counts as quantized (auto_gptq.py:278-303).  The reference fetches the
        safetensors header itself; here the caller passes it ({tensor name: {"dtype": "I32"}})."""
        if self.modules_in_block_to_quantize:
            self.modules_in_block_to_quantize = list(_flatten(self.modules_in_block_to_quantize))
            return
        if safetensors_metadata is None:
            return
        plain = {"F16", "BF16", "F32"}
        self.modules_in_block_to_quantize = sorted(
            {name.rsplit(".", 1)[0] for name, info in safetensors_metadata.items()
             if info.get("dtype") and info["dtype"] not in plain})


def repeat_scales_on_all_ranks(act_order: bool, group_size: int, is_row_parallel: bool) -> bool:
    """Scales cannot be sharded along K when groups are visited in activation order, or when a
    row-parallel layer has a single group (marlin_utils.py:440-446)."""
    return act_order or (group_size == -1 and is_row_parallel)


class AutoGPTQLinearMethod(LinearMethodBase):
    _kernel_backends_being_used: set[str] = set()

    def __init__(self, quant_config: AutoGPTQConfig) -> None:
        self.quant_config = quant_config
        self.quant_type = quant_config.quant_type
        self.input_dtype = None

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs) -> None:
        qc = self.quant_config
        n_part = sum(output_partition_sizes)
        row_parallel = input_size != input_size_per_partition
        loader = extra_weight_attrs.get("weight_loader")
        kcfg = MPLinearLayerConfig(full_weight_shape=(input_size, output_size),
                                   partition_weight_shape=(input_size_per_partition, n_part),
                                   weight_type=qc.quant_type, act_type=self.input_dtype or params_dtype,
                                   group_size=qc.group_size, zero_points=False, has_g_idx=qc.desc_act)
        kernel_type = choose_mp_linear_kernel(kcfg)
        self._kernel_backends_being_used.add(kernel_type.__name__)

        G = qc.group_size if qc.group_size != -1 else input_size
        replicate = repeat_scales_on_all_ranks(qc.desc_act, qc.group_size, row_parallel)
        groups = (input_size if replicate else input_size_per_partition) // G
        pf = qc.pack_factor

        layer.register_parameter("qweight", PackedvLLMParameter(
            data=torch.empty(input_size_per_partition // pf, n_part, dtype=torch.int32),
            input_dim=0, output_dim=1, packed_dim=0, packed_factor=pf, weight_loader=loader))
        layer.register_parameter("g_idx", RowvLLMParameter(
            data=torch.empty(input_size_per_partition, dtype=torch.int32), input_dim=0, weight_loader=loader))
        z = torch.empty(groups, n_part // pf, dtype=torch.int32)
        s = torch.empty(groups, n_part, dtype=params_dtype)
        if replicate:
            layer.register_parameter("scales", ChannelQuantScaleParameter(data=s, output_dim=1, weight_loader=loader))
            layer.register_parameter("qzeros", PackedColumnParameter(data=z, output_dim=1, packed_dim=1,
                                                                     packed_factor=pf, weight_loader=loader))
        else:
            layer.register_parameter("scales", GroupQuantScaleParameter(data=s, input_dim=0, output_dim=1,
                                                                        weight_loader=loader))
            layer.register_parameter("qzeros", PackedvLLMParameter(data=z, input_dim=0, output_dim=1, packed_dim=1,
                                                                   packed_factor=pf, weight_loader=loader))
        self.kernel = kernel_type(kcfg, w_q_param_name="qweight", w_s_param_name="scales",
                                  w_zp_param_name="qzeros", w_gidx_param_name="g_idx")

    def process_weights_after_loading(self, layer) -> None:
        self.kernel.process_weights_after_loading(layer)

    def apply(self, layer, x, bias=None):
        return self.kernel.apply_weights(layer, x, bias)
