"""AutoRound / Intel Neural Compressor checkpoints (`quant_method: "auto-round"`).

Mirrors vllm/model_executor/layers/quantization/inc/ of the reference: `INCConfig`
(inc.py:32-261), the per-layer resolution of `INCConfigParser` / `INCLayerConfig`
(config_parser.py:17-196), `INCLinearMethod` (inc_linear.py:14-47) and the weight-only integer
scheme `INCWna16Scheme` / `INCWNA16LinearScheme` (schemes/inc_wna16_scheme.py:26-107,
schemes/inc_wna16_linear.py:28-160): an AutoRound checkpoint is packed in the AutoGPTQ or the AutoAWQ
layout, so each quantized layer is handed to this package's GPTQ / AWQ linear method (and from
there to the MI355X W4A16 kernels), routed experts to `MoeWNA16Method`.

The mx_fp schemes of the reference (W4A4 / W8A8 microscaling) are a different arithmetic path and
are not built here: such configs parse, and asking for their method raises NotImplementedError.
"""
import re
from dataclasses import dataclass
from fractions import Fraction
from typing import Any

import torch

from .awq import AutoAWQConfig, AutoAWQLinearMethod
from .gptq import AutoGPTQConfig, AutoGPTQLinearMethod
from .linear import LinearBase, LinearMethodBase, UnquantizedLinearMethod
from .registry import QuantizationConfig, register_quantization_config

_REGEX_SPECIAL_CHARS = set(r"*+?^$()[]{}|\\")


@dataclass(frozen=True)
class INCLayerConfig:
    """What one layer resolves to (config_parser.py:17-47)."""
    bits: int
    group_size: int
    sym: bool
    packing_format: str
    backend: str
    data_type: str
    quantized: bool

    @property
    def is_gptq(self) -> bool:
        return "gptq" in self.packing_format or "gptq" in self.backend

    @property
    def is_awq(self) -> bool:
        return "awq" in self.packing_format or "awq" in self.backend

    @property
    def is_wna16_int(self) -> bool:
        return self.data_type == "int" and self.quantized

    @property
    def is_mxfp4(self) -> bool:
        return "mx_fp" in self.data_type and self.bits == 4

    @property
    def is_mxfp8(self) -> bool:
        return "mx_fp" in self.data_type and self.bits == 8


def _is_lm_head(layer) -> bool:
    # the reference tests isinstance(layer, ParallelLMHead); this package has no such class of its
    # own, so any class of that name (vLLM's, when present) counts
    return any(c.__name__ == "ParallelLMHead" for c in type(layer).__mro__)


def _is_fused_moe(layer) -> bool:
    return "fusedmoe" in layer.__class__.__name__.lower()


class INCConfigParser:
    """(bits, group_size, sym) of a layer from the global values, `extra_config` (exact names,
    suffixes, regular expressions), `block_name_to_quantize` and the fused-module mapping
    (config_parser.py:50-196, same order of precedence)."""

    def __init__(self, config: "INCConfig") -> None:
        self._config = config

    def resolve(self, layer, layer_name: str) -> INCLayerConfig:
        bits, group_size, sym = self._resolve_raw(layer, layer_name)
        c = self._config
        return INCLayerConfig(bits=bits, group_size=group_size, sym=sym, packing_format=c.packing_format,
                              backend=c.backend, data_type=c.data_type, quantized=bits < 16)

    def get_layer_config(self, layer, layer_name: str) -> tuple[int, int, bool]:
        lc = self.resolve(layer, layer_name)
        return lc.bits, lc.group_size, lc.sym

    # -- helpers -----------------------------------------------------------------------------
    def _defaults(self, quantized: bool) -> tuple[int, int, bool]:
        c = self._config
        return (c.weight_bits if quantized else 16, c.group_size if quantized else -1, c.sym if quantized else True)

    def _from_entry(self, cfg: dict, quantized: bool) -> tuple[int, int, bool]:
        b, g, s = self._defaults(quantized)
        return cfg.get("bits", b), cfg.get("group_size", g), cfg.get("sym", s)

    def _pattern_entries(self):
        for pattern, cfg in (self._config.extra_config or {}).items():
            if isinstance(pattern, str) and any(ch in _REGEX_SPECIAL_CHARS for ch in pattern):
                yield pattern, cfg

    def _matches(self, pattern: str, name: str) -> bool:
        try:
            return re.search(re.compile(pattern), name) is not None
        except re.error:   # a key that merely looks like a pattern
            return False

    def _explicit(self, name: str) -> bool:
        extra = self._config.extra_config
        if not extra:
            return False
        return name in extra or any(self._matches(p, name) for p, _ in self._pattern_entries())

    def _get(self, name: str, quantized: bool = True) -> tuple[int, int, bool]:
        extra = self._config.extra_config
        if not extra:
            return self._defaults(quantized)
        if name in extra:
            return self._from_entry(extra[name], quantized)
        for pattern, cfg in self._pattern_entries():
            if self._matches(pattern, name):
                return self._from_entry(cfg, quantized)
        return self._defaults(quantized)

    def _resolve_raw(self, layer, layer_name: str) -> tuple[int, int, bool]:
        c = self._config
        extra = c.extra_config
        if extra and layer_name in extra:
            return self._get(layer_name)
        # short keys ("lm_head") against fully qualified layer names
        if extra:
            for key in extra:
                if layer_name.endswith(f".{key}"):
                    return self._get(key)
        quantized = not _is_lm_head(layer)
        if c.block_name_to_quantize:
            quantized = any(layer_name.startswith(name) for name in c.block_name_to_quantize)
        if extra and _is_fused_moe(layer):
            moe = [self._get(name, quantized) for name in extra if name.startswith(layer_name)]
            if moe:
                if len(set(moe)) == 1:
                    return moe[0]
                raise ValueError(f"Fused MoE layer '{layer_name}' requires consistent quant config for all sub-layers")
        if extra:
            for fusion_key, sub_keys in c.packed_modules_mapping.items():
                if fusion_key in layer_name and layer_name.count(fusion_key) == 1:
                    sub_names = [layer_name.replace(fusion_key, sk) for sk in sub_keys]
                    # only when a shard is configured explicitly: a short fusion key may be a mere
                    # substring of an unrelated layer name
                    if not any(self._explicit(n) for n in sub_names):
                        continue
                    subs = [self._get(n, quantized) for n in sub_names]
                    if len(set(subs)) == 1:
                        return subs[0]
                    raise ValueError(f"Fused module '{layer_name}' requires consistent quant config for {sub_names}")
        return self._get(layer_name, quantized)


class INCLinearScheme:
    """What an INC linear method delegates to (schemes/inc_scheme.py)."""

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs):
        raise NotImplementedError

    def process_weights_after_loading(self, layer) -> None:
        raise NotImplementedError

    def apply_weights(self, layer, x, bias=None):
        raise NotImplementedError


class INCLinearMethod(LinearMethodBase):
    """inc_linear.py:14-47."""

    def __init__(self, scheme: INCLinearScheme) -> None:
        self.scheme = scheme

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs):
        return self.scheme.create_weights(layer=layer, input_size_per_partition=input_size_per_partition,
                                          output_partition_sizes=output_partition_sizes, input_size=input_size,
                                          output_size=output_size, params_dtype=params_dtype, **extra_weight_attrs)

    def process_weights_after_loading(self, layer) -> None:
        return self.scheme.process_weights_after_loading(layer)

    def apply(self, layer, x, bias=None):
        return self.scheme.apply_weights(layer, x, bias)


class INCWNA16LinearScheme(INCLinearScheme):
    """schemes/inc_wna16_linear.py:28-160: the layer becomes an AutoGPTQ or AutoAWQ layer.  Where the
    reference first asks whether Marlin can take the layer, this build has one kernel family for
    both layouts; what it cannot run (2 / 3 bits, asymmetric GPTQ packing, 8-bit AWQ) is refused here
    with the reference's wording."""

    def __init__(self, layer_config: INCLayerConfig) -> None:
        self.layer_config = layer_config
        self.inner_method = self._build_inner_method()

    @classmethod
    def get_min_capability(cls) -> int:
        return 60

    def _build_inner_method(self):
        lc = self.layer_config
        if lc.is_gptq:
            if (lc.bits, lc.sym) not in ((4, True), (8, True)):
                raise NotImplementedError(
                    f"INC quantization with bits={lc.bits}, sym={lc.sym} is not supported. "
                    "Only 4-bit and 8-bit symmetric quantization is supported with the MI355X kernels.")
            return AutoGPTQLinearMethod(AutoGPTQConfig(weight_bits=lc.bits, group_size=lc.group_size, desc_act=False,
                                                       is_sym=lc.sym, lm_head_quantized=False, dynamic={},
                                                       full_config={}))
        if lc.is_awq:
            if lc.bits != 4:
                raise NotImplementedError(f"INC quantization with bits={lc.bits} in the AWQ layout is not supported.")
            return AutoAWQLinearMethod(AutoAWQConfig(weight_bits=lc.bits, group_size=lc.group_size,
                                                     zero_point=not lc.sym, lm_head_quantized=False,
                                                     modules_to_not_convert=[], full_config={}))
        raise NotImplementedError(f"WNA16 linear scheme does not support {lc}")

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs) -> None:
        return self.inner_method.create_weights(layer, input_size_per_partition, output_partition_sizes, input_size,
                                                output_size, params_dtype, **extra_weight_attrs)

    def process_weights_after_loading(self, layer) -> None:
        return self.inner_method.process_weights_after_loading(layer)

    def apply_weights(self, layer, x, bias=None):
        return self.inner_method.apply(layer, x, bias)


class INCScheme:
    @staticmethod
    def can_handle(layer_config: INCLayerConfig) -> bool:
        raise NotImplementedError


class INCWna16Scheme(INCScheme):
    """schemes/inc_wna16_scheme.py:26-107 (the branch of a GPU platform)."""

    @staticmethod
    def can_handle(layer_config: INCLayerConfig) -> bool:
        return layer_config.is_wna16_int

    def get_linear_method(self, config: "INCConfig", layer, prefix: str, layer_config: INCLayerConfig):
        return INCLinearMethod(INCWNA16LinearScheme(layer_config))

    def get_moe_method(self, config: "INCConfig", layer, prefix: str, layer_config: INCLayerConfig):
        from .moe import MoeWNA16Config, MoeWNA16Method
        if layer_config.is_gptq:     # _resolve_gptq_moe (:110-158), the non-Marlin branch
            return MoeWNA16Method(MoeWNA16Config.from_config({
                "quant_method": "gptq", "bits": layer_config.bits, "group_size": layer_config.group_size,
                "sym": layer_config.sym, "lm_head": False}))
        if layer_config.is_awq:      # _resolve_awq_moe (:161-199)
            return MoeWNA16Method(MoeWNA16Config.from_config({
                "quant_method": "awq", "bits": layer_config.bits, "group_size": layer_config.group_size,
                "zero_point": not layer_config.sym, "lm_head": False}))
        raise NotImplementedError(f"WNA16 MoE does not support config {layer_config}")


class _INCMxfpScheme(INCScheme):
    what = "mx_fp"

    def _refuse(self, prefix: str):
        raise NotImplementedError(f"INC {self.what} layers ({prefix}) need the microscaling kernels, which this "
                                  "build does not provide; the weight-only integer formats are supported")

    def get_linear_method(self, config, layer, prefix, layer_config):
        self._refuse(prefix)

    def get_moe_method(self, config, layer, prefix, layer_config):
        self._refuse(prefix)


class INCMxfp8Scheme(_INCMxfpScheme):
    what = "MXFP8"

    @staticmethod
    def can_handle(layer_config: INCLayerConfig) -> bool:
        return layer_config.is_mxfp8


class INCMxfp4Scheme(_INCMxfpScheme):
    what = "MXFP4"

    @staticmethod
    def can_handle(layer_config: INCLayerConfig) -> bool:
        return layer_config.is_mxfp4


def resolve_scheme(layer_config: INCLayerConfig) -> INCScheme:
    """schemes/factory.py:11-26, same order."""
    for cls in (INCMxfp8Scheme, INCWna16Scheme, INCMxfp4Scheme):
        if cls.can_handle(layer_config):
            return cls()
    raise NotImplementedError(f"No INC scheme found for layer config: {layer_config}")


# What a checkpoint-level AutoRound config may say.  The accepted values are the reference's
# (inc/inc.py:35-52); the checks below are driven from these tables.
_CHOICES: dict[str, tuple] = {
    "weight_bits": (2, 3, 4, 8),
    "packing_format": ("auto_round:auto_gptq", "auto_round:auto_awq", "auto_round:llm_compressor"),
    "backend": ("auto", "gptq", "gptq:marlin", "awq", "awq:marlin", "marlin"),
}
_PLAIN_DATA_TYPES = ("int", "mx_fp")      # any other name containing "mx_fp" is a microscaling variant and passes
_MX_PACKING = "auto_round:llm_compressor"
# 8-bit microscaling pins these: weight side from the constructor arguments, activation side from
# the raw checkpoint dict (a key that is absent counts as the pinned value)
_MX8_WEIGHT_SIDE = {"group_size": 32, "sym": True, "packing_format": _MX_PACKING, "backend": "auto"}
_MX8_ACT_SIDE = {"act_bits": 8, "act_data_type": "mx_fp", "act_group_size": 32, "act_sym": True,
                 "act_dynamic": True, "enable_quanted_input": False}
# checkpoint key(s) -> constructor argument, default when absent (no default: the key is required)
_FROM_CHECKPOINT: tuple = (
    (("bits",), "weight_bits"),
    (("group_size",), "group_size"),
    (("sym",), "sym"),
    (("packing_format",), "packing_format", "auto_round:auto_gptq"),
    (("block_name_to_quantize", "to_quant_block_names"), "block_name_to_quantize", None),
    (("extra_config",), "extra_config", None),
    (("data_type",), "data_type", "int"),
    (("backend", "vllm_backend"), "backend", "auto"),
)


def _require_choice(what: str, value, allowed) -> None:
    if value not in allowed:
        raise ValueError(f"auto-round config: {what}={value!r} is not one of {list(allowed)}")


def _require_pinned(what: str, got: dict, pinned: dict) -> None:
    for key, want in pinned.items():
        if got[key] != want:
            raise AssertionError(f"auto-round config: {what} needs {key}={want!r}, the checkpoint has {key}={got[key]!r}")


@register_quantization_config("inc")
class INCConfig(QuantizationConfig):
    """Checkpoint-level AutoRound settings; same constructor arguments, attributes, accepted values and
    exception types as the reference's INCConfig (inc/inc.py:32-261)."""

    def __init__(self, weight_bits: int, group_size: int, sym: bool = True,
                 packing_format: str = "auto_round:auto_gptq",
                 block_name_to_quantize: str | list[str] | None = None,
                 extra_config: dict[str, Any] | None = None, data_type: str = "int", backend: str = "auto") -> None:
        super().__init__()
        given = {"weight_bits": weight_bits, "packing_format": packing_format, "backend": backend}
        # reported in the reference's order: bits, data type, packing format, backend
        _require_choice("weight_bits", weight_bits, _CHOICES["weight_bits"])
        if "mx_fp" not in data_type:
            _require_choice("data_type", data_type, _PLAIN_DATA_TYPES)
        for name in ("packing_format", "backend"):
            _require_choice(name, given[name], _CHOICES[name])
        if isinstance(block_name_to_quantize, str):
            block_name_to_quantize = block_name_to_quantize.split(",")
        self.weight_bits, self.group_size, self.sym = weight_bits, group_size, sym
        self.packing_format, self.backend, self.data_type = packing_format, backend, data_type
        self.block_name_to_quantize = block_name_to_quantize
        self.extra_config = extra_config
        self.pack_factor = Fraction(32, weight_bits)
        self.config_parser = INCConfigParser(self)
        if self.is_mxfp8:
            _require_pinned("8-bit microscaling", vars(self), _MX8_WEIGHT_SIDE)
        elif packing_format == _MX_PACKING and not self.is_mxfp:
            raise ValueError(f"auto-round config: packing_format={_MX_PACKING!r} only goes with a microscaling "
                             f"('mx_fp') data_type, not {data_type!r}")

    def __repr__(self) -> str:
        return f"INCConfig(weight_bits={self.weight_bits}, group_size={self.group_size}, sym={self.sym})"

    @property
    def is_mxfp(self) -> bool:
        return "mx_fp" in self.data_type

    @property
    def is_mxfp8(self) -> bool:
        return self.is_mxfp and self.weight_bits == 8

    @classmethod
    def get_name(cls) -> str:
        return "inc"

    @classmethod
    def get_supported_act_dtypes(cls) -> list[torch.dtype]:
        return [torch.half, torch.bfloat16]

    @classmethod
    def get_min_capability(cls) -> int:
        return 60

    @classmethod
    def get_config_filenames(cls) -> list[str]:
        return ["quantization_config.json"]

    @classmethod
    def from_config(cls, config: dict[str, Any]) -> "INCConfig":
        kwargs = {}
        for keys, arg, *default in _FROM_CHECKPOINT:
            kwargs[arg] = cls.get_from_keys_or(config, list(keys), default[0]) if default else cls.get_from_keys(config, list(keys))
        qc = cls(**kwargs)
        if qc.is_mxfp8:
            seen = {key: cls.get_from_keys_or(config, [key], want) for key, want in _MX8_ACT_SIDE.items()}
            _require_pinned("8-bit microscaling", seen, _MX8_ACT_SIDE)
        return qc

    def get_layer_config(self, layer, layer_name: str):
        return self.config_parser.get_layer_config(layer, layer_name)

    def get_quant_method(self, layer: torch.nn.Module, prefix: str):
        is_linear = isinstance(layer, LinearBase) or _is_lm_head(layer)
        is_moe = _is_fused_moe(layer) or type(layer).__name__ == "RoutedExperts"
        # layers listed (with or without the "model." prefix) at 16 bits or more stay unquantized
        if prefix and self.extra_config:
            for name, cfg in self.extra_config.items():
                if (name == prefix or name == f"model.{prefix}") and cfg.get("bits", 16) >= 16:
                    if is_moe:
                        raise NotImplementedError(f"unquantized routed experts ({prefix}) are outside this build")
                    return UnquantizedLinearMethod()
        layer_config = self.config_parser.resolve(layer, prefix)
        if not layer_config.quantized:
            if is_linear:
                return UnquantizedLinearMethod()
            if is_moe:
                raise NotImplementedError(f"unquantized routed experts ({prefix}) are outside this build")
            return None
        scheme = resolve_scheme(layer_config)
        if is_linear:
            return scheme.get_linear_method(self, layer, prefix, layer_config)
        if is_moe:
            return scheme.get_moe_method(self, layer, prefix, layer_config)
        return None

    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant, hf_config=None) -> str | None:
        """`quant_method: "auto-round"` checkpoints resolve to this config."""
        return cls.get_name() if hf_quant_cfg.get("quant_method", None) == "auto-round" else None
