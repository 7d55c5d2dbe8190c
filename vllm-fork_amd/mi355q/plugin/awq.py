"""AWQ checkpoints (AutoAWQ `quant_config.json` / `quantization_config`).

Mirrors vllm/model_executor/layers/quantization/auto_awq.py:171-410 (AutoAWQConfig),
:840-958 (weight creation and the linear method).  The reference's ROCm route keeps the
checkpoint tensors and calls awq_gemm / awq_dequantize + matmul (:947-955); here the tensors are
repacked once into the MI355X tile layout -- nibble order [0,2,4,6,1,3,5,7] of both qweight and
qzeros is undone by the repack (awq_triton.py:54-67) -- and the same two routes exist inside
ops.w4a16_gemm.
"""
from typing import Any

import torch

from .. import _lib
from .linear import LinearBase, LinearMethodBase, ParallelLMHead, UnquantizedLinearMethod
from .mp_kernel import MI355XW4A16LinearKernel, MPLinearLayerConfig, choose_mp_linear_kernel
from .parameter import GroupQuantScaleParameter, PackedvLLMParameter
from .registry import QuantizationConfig, register_quantization_config
from .scalar_type import scalar_types


def is_layer_skipped(prefix: str, ignored: list[str], fused_mapping: dict[str, list[str]] | None = None) -> bool:
    """Substring match against `modules_to_not_convert`; the shards of a fused module must
    agree (quantization/utils/quant_utils.py `is_layer_skipped`, skip_with_substr=True)."""
    proj = prefix.split(".")[-1]
    shards = (fused_mapping or {}).get(proj)
    if not shards:
        return any(i in prefix for i in ignored)
    verdicts = {any(i in prefix.replace(proj, s) for i in ignored) for s in shards}
    if len(verdicts) != 1:
        raise ValueError(f"Detected some but not all shards of {prefix} are quantized. "
                         "All shards of fused layers to have the same precision.")
    return verdicts.pop()


@register_quantization_config("auto_awq")
class AutoAWQConfig(QuantizationConfig):
    TYPE_MAP = {4: scalar_types.uint4}

    def __init__(self, weight_bits: int, group_size: int, zero_point: bool, lm_head_quantized: bool,
                 modules_to_not_convert: list[str] | None = None, full_config: dict[str, Any] | None = None) -> None:
        super().__init__()
        if weight_bits not in self.TYPE_MAP:
            raise ValueError(f"Unsupported num_bits = {weight_bits}. Supported: "
                             f"{', '.join(str(k) for k in self.TYPE_MAP)}.")
        self.weight_bits, self.group_size, self.zero_point = weight_bits, group_size, zero_point
        self.pack_factor = 32 // weight_bits
        self.lm_head_quantized = lm_head_quantized
        self.modules_to_not_convert = modules_to_not_convert or []
        self.full_config = full_config or {}
        self.quant_type = self.TYPE_MAP[weight_bits]

    def __repr__(self) -> str:
        return (f"AutoAWQConfig(quant_type={self.quant_type}, group_size={self.group_size}, "
                f"zero_point={self.zero_point}, lm_head_quantized={self.lm_head_quantized}, "
                f"modules_to_not_convert={self.modules_to_not_convert})")

    @classmethod
    def get_name(cls) -> str:
        return "auto_awq"

    @classmethod
    def get_supported_act_dtypes(cls) -> list[torch.dtype]:
        return [torch.half, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 75

    @classmethod
    def get_config_filenames(cls) -> list[str]:
        return ["quantize_config.json", "quant_config.json"]

    @classmethod
    def from_config(cls, config: dict[str, Any]) -> "AutoAWQConfig":
        full = dict(config)
        full["quant_method"] = "awq"
        return cls(cls.get_from_keys(config, ["w_bit", "bits"]), cls.get_from_keys(config, ["q_group_size", "group_size"]),
                   cls.get_from_keys(config, ["zero_point"]), cls.get_from_keys_or(config, ["lm_head"], default=False),
                   cls.get_from_keys_or(config, ["modules_to_not_convert"], None), full)

    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant, hf_config=None) -> str | None:
        if str(hf_quant_cfg.get("quant_method", "")).lower() != "awq":
            return None
        if user_quant is None or user_quant in ("awq", "awq_marlin", "auto_awq", "marlin"):
            return cls.get_name()
        return None

    def get_quant_method(self, layer: torch.nn.Module, prefix: str):
        if not (isinstance(layer, LinearBase) or (isinstance(layer, ParallelLMHead) and self.lm_head_quantized)):
            return None     # auto_awq.py:288-290
        if is_layer_skipped(prefix, self.modules_to_not_convert, self.packed_modules_mapping):
            return UnquantizedLinearMethod()
        return AutoAWQLinearMethod(self)

    def maybe_update_config(self, model_name: str, hf_config=None, revision: str | None = None,
                            safetensors_metadata: dict[str, dict] | None = None) -> None:
        """modules_to_not_convert := modules whose tensors are all f16/bf16/f32 in the
        safetensors header, when the config did not say (auto_awq.py:392-410)."""
        if self.modules_to_not_convert or safetensors_metadata is None:
            return
        plain = {"F16", "BF16", "F32"}
        layers = {n.rsplit(".", 1)[0] for n in safetensors_metadata}
        quant = {n.rsplit(".", 1)[0] for n, info in safetensors_metadata.items()
                 if info.get("dtype") and info["dtype"] not in plain}
        self.modules_to_not_convert = sorted(layers - quant)


class AutoAWQLinearMethod(LinearMethodBase):
    def __init__(self, quant_config: AutoAWQConfig):
        self.quant_config = quant_config

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs):
        qc = self.quant_config
        G = qc.group_size if qc.group_size != -1 else input_size
        n_part = sum(output_partition_sizes)
        if input_size_per_partition % G:
            raise ValueError("The input size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        if n_part % qc.pack_factor:
            raise ValueError("The output size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        loader, pf = extra_weight_attrs.get("weight_loader"), qc.pack_factor
        groups = input_size_per_partition // G
        layer.register_parameter("qweight", PackedvLLMParameter(
            data=torch.empty(input_size_per_partition, n_part // pf, dtype=torch.int32),
            input_dim=0, output_dim=1, packed_dim=1, packed_factor=pf, weight_loader=loader))
        layer.register_parameter("qzeros", PackedvLLMParameter(
            data=torch.empty(groups, n_part // pf, dtype=torch.int32),
            input_dim=0, output_dim=1, packed_dim=1, packed_factor=pf, weight_loader=loader))
        layer.register_parameter("scales", GroupQuantScaleParameter(
            data=torch.empty(groups, n_part, dtype=params_dtype), input_dim=0, output_dim=1, weight_loader=loader))
        kcfg = MPLinearLayerConfig(full_weight_shape=(input_size, output_size),
                                   partition_weight_shape=(input_size_per_partition, n_part),
                                   weight_type=qc.quant_type, act_type=params_dtype, group_size=qc.group_size,
                                   zero_points=True, has_g_idx=False)
        kernel_type = choose_mp_linear_kernel(kcfg)
        if not issubclass(kernel_type, MI355XW4A16LinearKernel):
            raise ValueError(f"{kernel_type.__name__} cannot take AWQ-ordered tensors")
        self.kernel = kernel_type(kcfg, w_q_param_name="qweight", w_s_param_name="scales", w_zp_param_name="qzeros",
                                  src_layout=_lib.W_AWQ, zp_mode=_lib.ZP_AWQ)

    def process_weights_after_loading(self, layer) -> None:
        self.kernel.process_weights_after_loading(layer)

    def apply(self, layer, x, bias=None):
        return self.kernel.apply_weights(layer, x, bias)
