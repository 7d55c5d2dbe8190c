"""The MI355X attention backend as real subclasses of the fork's attention ABCs.

Only importable inside a vLLM installation (it imports vllm at module level); vLLM resolves
the class by path after `register_backend(AttentionBackendEnum.CUSTOM, "<this module>.…")`
(v1/attention/backends/registry.py:242-294).  What the fork calls on these classes:

  * `validate_configuration(...)` on the backend class while selecting a backend
    (platforms/rocm.py:574,601) -- inherited from `AttentionBackend`
    (v1/attention/backend.py:320-393); it consults the predicates defined here;
  * `get_kv_cache_shape`, `get_builder_cls`, `get_impl_cls`, `get_name` (the name must be a
    member of AttentionBackendEnum: attention.py:423);
  * builder `__init__(kv_cache_spec, layer_names, vllm_config, device)` and
    `build(common_prefix_len, common_attn_metadata, fast_build)` (rocm_attn.py:76-162);
  * impl `__init__` with the positional arguments of attention.py:409-422, `forward`,
    `do_kv_cache_update`, `do_rope_and_kv_cache_update` (rocm_attn.py:267-562).

The arithmetic lives in plugin/attention.py (Mi355AttentionImpl); these classes only give it
the fork's base classes and constructor signatures.
"""
from typing import ClassVar

import torch
from vllm.v1.attention.backend import (AttentionBackend, AttentionCGSupport, AttentionImpl, AttentionMetadataBuilder,
                                       AttentionType, CommonAttentionMetadata, MultipleOf)

from . import attention as _impl


class MI355XAttentionMetadataBuilder(AttentionMetadataBuilder[_impl.Mi355AttentionMetadata]):
    _cudagraph_support: ClassVar[AttentionCGSupport] = AttentionCGSupport.ALWAYS

    def __init__(self, kv_cache_spec, layer_names, vllm_config, device):
        super().__init__(kv_cache_spec, layer_names, vllm_config, device)
        self.block_size = kv_cache_spec.block_size
        # The partition scratch of the attention kernels is one buffer per device, shared by every
        # layer.  Size it here -- before the memory-profiling run and before graph capture -- for
        # the largest batch the scheduler can hand over, so the KV-cache sizing accounts for it.
        mc, sc = getattr(vllm_config, "model_config", None), getattr(vllm_config, "scheduler_config", None)
        if mc is not None and sc is not None and torch.device(device).type == "cuda":
            heads = mc.get_num_attention_heads(vllm_config.parallel_config)
            _impl.reserve_partition_scratch(int(sc.max_num_batched_tokens), int(mc.max_model_len), heads,
                                            mc.get_head_size(), mc.dtype, device)

    def build(self, common_prefix_len: int, common_attn_metadata: CommonAttentionMetadata,
              fast_build: bool = False) -> _impl.Mi355AttentionMetadata:
        if common_prefix_len > 0:
            raise NotImplementedError("cascade attention is not supported (use_cascade_attention() is False)")
        m = common_attn_metadata
        return _impl.Mi355AttentionMetadata(m.num_actual_tokens, m.max_query_len, m.query_start_loc, m.max_seq_len,
                                            m.seq_lens, m.block_table_tensor, m.slot_mapping, False, 0,
                                            getattr(m, "causal", True))

    def build_for_cudagraph_capture(self, common_attn_metadata: CommonAttentionMetadata):
        md = self.build(0, common_attn_metadata)
        md.seq_lens.fill_(1)   # capture with the shortest context; replay reads the live values
        return md


class MI355XAttentionImpl(AttentionImpl[_impl.Mi355AttentionMetadata]):
    """AttentionImpl whose methods are those of plugin/attention.py::Mi355AttentionImpl."""

    def __init__(self, num_heads: int, head_size: int, scale: float, num_kv_heads: int | None = None,
                 alibi_slopes: list[float] | None = None, sliding_window: int | None = None,
                 kv_cache_dtype: str = "auto", logits_soft_cap: float | None = None,
                 attn_type: str = AttentionType.DECODER, kv_sharing_target_layer_name: str | None = None,
                 sinks: torch.Tensor | None = None) -> None:
        if kv_sharing_target_layer_name is not None:
            raise NotImplementedError("KV-cache sharing between layers is not supported")
        _impl.Mi355AttentionImpl.__init__(self, num_heads, head_size, scale, num_kv_heads or num_heads, alibi_slopes,
                                          sliding_window, kv_cache_dtype, logits_soft_cap,
                                          "decoder" if attn_type == AttentionType.DECODER else str(attn_type),
                                          None, sinks)

    def forward(self, layer, query, key, value, kv_cache, attn_metadata, output, output_scale=None,
                output_block_scale=None):
        return _impl.Mi355AttentionImpl.forward(self, layer, query, key, value, kv_cache, attn_metadata, output,
                                                output_scale, output_block_scale)

    def fused_rope_kvcache_supported(self):
        return True

    # the helpers forward() and the cache hooks use
    _caches = _impl.Mi355AttentionImpl._caches
    _partition_scratch = _impl.Mi355AttentionImpl._partition_scratch
    do_kv_cache_update = _impl.Mi355AttentionImpl.do_kv_cache_update
    do_rope_and_kv_cache_update = _impl.Mi355AttentionImpl.do_rope_and_kv_cache_update
    decode_fused_supported = _impl.Mi355AttentionImpl.decode_fused_supported
    forward_decode_fused = _impl.Mi355AttentionImpl.forward_decode_fused


class MI355XAttentionBackend(AttentionBackend):
    """Registered as AttentionBackendEnum.CUSTOM; selected with `--attention-backend CUSTOM`."""
    supported_dtypes: ClassVar[list[torch.dtype]] = list(_impl.Mi355AttentionBackend.supported_dtypes)
    supported_kv_cache_dtypes: ClassVar[list[str]] = list(_impl.Mi355AttentionBackend.supported_kv_cache_dtypes)
    forward_includes_kv_cache_update: bool = False

    @staticmethod
    def get_name() -> str:
        return "CUSTOM"

    @staticmethod
    def get_supported_kernel_block_sizes() -> list[int | MultipleOf]:
        return [16, 32]

    @classmethod
    def get_supported_head_sizes(cls) -> list[int]:
        return _impl.Mi355AttentionBackend.get_supported_head_sizes()

    @classmethod
    def supports_kv_connector(cls) -> bool:
        return False   # (2, num_blocks, ...) layout, as ROCM_ATTN (rocm_attn.py:212-215)

    @classmethod
    def supports_compute_capability(cls, capability) -> bool:
        try:
            return capability is None or capability.to_int() >= 95   # gfx950
        except AttributeError:
            return True

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int, head_size: int,
                           cache_dtype_str: str = "auto") -> tuple[int, ...]:
        return _impl.Mi355AttentionBackend.get_kv_cache_shape(num_blocks, block_size, num_kv_heads, head_size,
                                                              cache_dtype_str)

    @staticmethod
    def use_cascade_attention(*args, **kwargs) -> bool:
        return False

    @staticmethod
    def get_impl_cls():
        return MI355XAttentionImpl

    @staticmethod
    def get_builder_cls():
        return MI355XAttentionMetadataBuilder


class MI355XRocmAttnBackend(MI355XAttentionBackend):
    """The same backend under the name ROCM_ATTN, for deployments that override the fork's default
    ROCm backend (`register_backend(AttentionBackendEnum.ROCM_ATTN, "<this class>")`) instead of
    passing `--attention-backend CUSTOM`."""

    @staticmethod
    def get_name() -> str:
        return "ROCM_ATTN"
