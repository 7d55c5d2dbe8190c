"""Quantization-method plug-in surface: the abstract config / method classes and the
name -> config-class registry.

Mirrors vllm/model_executor/layers/quantization/base_config.py:20-276 (QuantizeMethodBase,
QuantizationConfig) and quantization/__init__.py:59-183 (register_quantization_config,
get_quantization_config): registering an existing name replaces it, unknown names raise
ValueError, and the class must derive from QuantizationConfig.
"""
import re
from abc import ABC, abstractmethod
from typing import Any

import torch


class QuantizeMethodBase(ABC):
    uses_meta_device: bool = False

    @abstractmethod
    def create_weights(self, layer: torch.nn.Module, *weight_args, **extra_weight_attrs):
        raise NotImplementedError

    @abstractmethod
    def apply(self, layer: torch.nn.Module, *args, **kwargs) -> torch.Tensor:
        raise NotImplementedError

    def embedding(self, layer: torch.nn.Module, *args, **kwargs) -> torch.Tensor:
        raise NotImplementedError

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        return None


class QuantizationConfig(ABC):
    def __init__(self):
        # filled in by models: fused module name -> checkpoint shard names
        self.packed_modules_mapping: dict[str, list[str]] = {}

    @classmethod
    @abstractmethod
    def get_name(cls) -> str: ...

    @classmethod
    @abstractmethod
    def get_supported_act_dtypes(cls) -> list[torch.dtype]: ...

    @classmethod
    @abstractmethod
    def get_min_capability(cls) -> int: ...

    @classmethod
    @abstractmethod
    def get_config_filenames(cls) -> list[str]: ...

    @classmethod
    @abstractmethod
    def from_config(cls, config: dict[str, Any]) -> "QuantizationConfig": ...

    @classmethod
    def override_quantization_method(cls, hf_quant_cfg: dict[str, Any], user_quant: str | None,
                                     hf_config: Any = None) -> str | None:
        return None

    @staticmethod
    def get_from_keys(config: dict[str, Any], keys: list[str]) -> Any:
        for k in keys:
            if k in config:
                return config[k]
        raise ValueError(f"Cannot find any of {keys} in the model's quantization config.")

    @staticmethod
    def get_from_keys_or(config: dict[str, Any], keys: list[str], default: Any) -> Any:
        for k in keys:
            if k in config:
                return config[k]
        return default

    @abstractmethod
    def get_quant_method(self, layer: torch.nn.Module, prefix: str) -> QuantizeMethodBase | None: ...

    # checkpoint name -> model name for KV-cache scales (base_config.py:194-227)
    _CACHE_SCALE_RULES = (
        (re.compile(r"\.kv_scale$"), r".attn.k_scale"),
        (re.compile(r"\.self_attn\.[kv]_proj\.([kv])_scale$"), r".self_attn.attn.\1_scale"),
        (re.compile(r"\.self_attn\.qk(?:qk)?v_proj\.([kv])_scale$"), r".self_attn.attn.\1_scale"),
        (re.compile(r"(?<!\.attn)\.([qkv])_scale$"), r".attn.\1_scale"),
        (re.compile(r"(?<!\.attn)\.([qkv])_zero_point$"), r".attn.\1_zero_point"),
    )

    @classmethod
    def map_cache_scale_name(cls, name: str) -> str:
        """Apply the first matching KV-scale renaming rule (what the reference's
        get_cache_scale_mapper() feeds to its weight loader)."""
        for pat, repl in cls._CACHE_SCALE_RULES:
            new, n = pat.subn(repl, name)
            if n:
                return new
        return name

    def apply_vllm_mapper(self, hf_to_vllm_mapper) -> None:
        return None

    def maybe_update_config(self, model_name: str, hf_config=None, revision: str | None = None) -> None:
        return None


# name -> config class.  Filled by register_quantization_config at import of the format modules.
_REGISTRY: dict[str, type[QuantizationConfig]] = {}
QUANTIZATION_METHODS: list[str] = []


def register_quantization_config(quantization: str):
    """Class decorator (quantization/__init__.py:59-106).  Re-registering a name swaps the
    class in (the reference only logs at debug), so a plug-in can take over `auto_gptq`."""

    def _wrap(cls):
        if not (isinstance(cls, type) and issubclass(cls, QuantizationConfig)):
            raise ValueError("The quantization config must be a subclass of `QuantizationConfig`.")
        if quantization not in QUANTIZATION_METHODS:
            QUANTIZATION_METHODS.append(quantization)
        _REGISTRY[quantization] = cls
        return cls

    return _wrap


def get_quantization_config(quantization: str) -> type[QuantizationConfig]:
    if quantization not in _REGISTRY:
        raise ValueError(f"Invalid quantization method: {quantization}")
    return _REGISTRY[quantization]


def resolve_quantization(hf_quant_cfg: dict[str, Any], user_quant: str | None = None) -> str:
    """Checkpoint `quantization_config` (+ optional user choice) -> registry name, letting
    each registered config claim the checkpoint through override_quantization_method
    (the loop in vllm/config/model.py `_verify_quantization`)."""
    method = str(hf_quant_cfg.get("quant_method", "")).lower()
    for name in QUANTIZATION_METHODS:
        got = _REGISTRY[name].override_quantization_method(hf_quant_cfg, user_quant)
        if got is not None:
            return got
    chosen = user_quant or method
    if chosen not in _REGISTRY:
        raise ValueError(f"Unknown quantization method: {chosen}. Must be one of {QUANTIZATION_METHODS}.")
    return chosen
