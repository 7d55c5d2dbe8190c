"""Attention backend for the paged KV cache on MI355X.

Mirrors the ROCM_ATTN backend contract, vllm/v1/attention/backends/rocm_attn.py:43-562:
metadata fields :43-73, builder :76-162, backend capabilities and KV-cache shape :165-264,
impl.forward :367-469, impl.do_kv_cache_update :471-517; the split of the cache tensor into
K [blocks, kv_heads, head/x, block, x] and V [blocks, kv_heads, head, block] views follows
vllm/v1/attention/ops/paged_attn.py:16-29, and the prefill-rows / decode-rows split follows
vllm/v1/attention/ops/chunked_prefill_paged_decode.py:271-493.  Unlike the reference nothing
is allocated per call: the partition scratch is one buffer per device, shared by the impls of all
layers (their launches are ordered on one stream), reserved up front by the metadata builder and only
ever grown outside graph capture.
"""
from dataclasses import dataclass

import torch

from .. import _lib, ops
from .kv_cache import BaseKVCacheMethod, is_quantized_kv_cache, set_default_quant_scales


@dataclass
class CommonAttentionMetadata:
    """What the model runner knows about a batch (v1/attention/backend.py `CommonAttentionMetadata`)."""
    query_start_loc: torch.Tensor      # int32 [S+1], device
    seq_lens: torch.Tensor             # int32 [S], device (context + new tokens)
    num_actual_tokens: int
    max_query_len: int
    max_seq_len: int
    block_table_tensor: torch.Tensor   # int32 [S, max_blocks]
    slot_mapping: torch.Tensor         # int64 [T]
    causal: bool = True


@dataclass
class Mi355AttentionMetadata:
    num_actual_tokens: int
    max_query_len: int
    query_start_loc: torch.Tensor
    max_seq_len: int
    seq_lens: torch.Tensor
    block_table: torch.Tensor
    slot_mapping: torch.Tensor
    use_cascade: bool = False
    common_prefix_len: int = 0
    causal: bool = True


class Mi355AttentionMetadataBuilder:
    cudagraph_support = "ALWAYS"   # AttentionCGSupport.ALWAYS, rocm_attn.py:77

    def __init__(self, block_size: int, device: torch.device | str = "cuda"):
        self.block_size, self.device = block_size, device

    def build(self, common_prefix_len: int, common_attn_metadata: CommonAttentionMetadata,
              fast_build: bool = False) -> Mi355AttentionMetadata:
        if common_prefix_len > 0:
            raise NotImplementedError("cascade attention is not supported (use_cascade_attention() is False)")
        m = common_attn_metadata
        return Mi355AttentionMetadata(m.num_actual_tokens, m.max_query_len, m.query_start_loc, m.max_seq_len,
                                      m.seq_lens, m.block_table_tensor, m.slot_mapping, False, 0, m.causal)

    def build_for_cudagraph_capture(self, common_attn_metadata: CommonAttentionMetadata) -> Mi355AttentionMetadata:
        md = self.build(0, common_attn_metadata)
        md.seq_lens.fill_(1)   # capture with the shortest context; replay reads the live values
        return md


# (device, dtype) -> (tmp_out [rows, heads, P, head], exp_sums [rows, heads, P], max_logits [same]) as flat buffers.
# One per device for every attention layer: a layer's launch has consumed its partials before the next
# layer's launch starts (one stream), so the layers can share them.
_SCRATCH: dict = {}
_SCRATCH_RETIRED: list = []   # captured graphs may still hold the address of an outgrown buffer
# Upper bound of what reserve_partition_scratch takes up front (bytes of tmp_out); longer batches grow it
# at their first eager run instead.
SCRATCH_RESERVE_LIMIT = int(__import__("os").environ.get("MI355Q_ATTN_SCRATCH_LIMIT", str(1 << 30)))


def _scratch_views(rows: int, heads: int, parts: int, head_size: int, dtype, device):
    device = torch.device(device)
    if device.index is None and device.type == "cuda":
        device = torch.device("cuda", torch.cuda.current_device())
    need = rows * heads * parts
    key = (device, dtype)
    s = _SCRATCH.get(key)
    if s is None or s[0].numel() < need * head_size or s[1].numel() < need:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("attention scratch must be sized before graph capture (reserve_partition_scratch, "
                               "or run one eager step)")
        if s is not None:
            _SCRATCH_RETIRED.append(s)
        s = (torch.empty(need * head_size, dtype=dtype, device=device),
             torch.empty(need, dtype=torch.float32, device=device),
             torch.empty(need, dtype=torch.float32, device=device))
        _SCRATCH[key] = s
    return (s[0][:need * head_size].view(rows, heads, parts, head_size), s[1][:need].view(rows, heads, parts),
            s[2][:need].view(rows, heads, parts))


def reserve_partition_scratch(max_tokens: int, max_seq_len: int, num_heads: int, head_size: int, dtype,
                              device) -> int:
    """Size the shared partition scratch for batches of up to `max_tokens` query rows over contexts of up
    to `max_seq_len` tokens, capped at SCRATCH_RESERVE_LIMIT bytes; returns the rows reserved.  Called by
    the metadata builder before the memory-profiling run, so the KV-cache sizing sees the allocation."""
    parts = (max_seq_len + _lib.PA_PARTITION - 1) // _lib.PA_PARTITION
    per_row = num_heads * parts * head_size * torch.empty((), dtype=dtype).element_size()
    rows = max(1, min(max_tokens, SCRATCH_RESERVE_LIMIT // max(per_row, 1)))
    _scratch_views(rows, num_heads, parts, head_size, dtype, device)
    return rows


class Mi355AttentionBackend:
    supported_dtypes = [torch.float16, torch.bfloat16]
    supported_kv_cache_dtypes = ["auto", "float16", "bfloat16", "fp8", "fp8_e4m3"]
    forward_includes_kv_cache_update = False

    @staticmethod
    def get_name() -> str:
        return "ROCM_ATTN"

    @staticmethod
    def get_supported_kernel_block_sizes() -> list[int]:
        return [16, 32]

    @classmethod
    def get_supported_head_sizes(cls) -> list[int]:
        return [64, 128]

    @classmethod
    def supports_sink(cls) -> bool:
        return False

    @classmethod
    def supports_sliding_window(cls) -> bool:
        return False

    @staticmethod
    def use_cascade_attention(*args, **kwargs) -> bool:
        return False

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int, head_size: int,
                           cache_dtype_str: str = "auto") -> tuple[int, ...]:
        if block_size % 16 != 0:
            raise ValueError("Block size must be a multiple of 16.")
        return (2, num_blocks, block_size, num_kv_heads, head_size)

    @staticmethod
    def get_impl_cls():
        return Mi355AttentionImpl

    @staticmethod
    def get_builder_cls():
        return Mi355AttentionMetadataBuilder


def split_kv_cache(kv_cache: torch.Tensor, num_kv_heads: int, head_size: int):
    """[2, blocks, block*kv_heads*head] storage -> the K and V views the kernels index."""
    x = 16 // kv_cache.element_size()
    nblk = kv_cache.shape[1]
    k = kv_cache[0].view(nblk, num_kv_heads, head_size // x, -1, x)
    v = kv_cache[1].view(nblk, num_kv_heads, head_size, -1)
    return k, v


class Mi355AttentionImpl:
    def __init__(self, num_heads: int, head_size: int, scale: float, num_kv_heads: int,
                 alibi_slopes: list[float] | None = None, sliding_window: int | None = None,
                 kv_cache_dtype: str = "auto", logits_soft_cap: float | None = None, attn_type: str = "decoder",
                 kv_sharing_target_layer_name=None, sinks=None) -> None:
        if sliding_window is not None or sinks is not None or logits_soft_cap:
            raise NotImplementedError("sliding window, sinks and logit soft-capping are not supported")
        if attn_type != "decoder":
            raise NotImplementedError("only decoder self-attention is supported")
        if head_size not in Mi355AttentionBackend.get_supported_head_sizes():
            raise ValueError(f"head size {head_size} not in {Mi355AttentionBackend.get_supported_head_sizes()}")
        if num_heads % num_kv_heads or not 1 <= num_heads // num_kv_heads <= 16:
            raise ValueError("query heads per KV head must be between 1 and 16")
        self.num_heads, self.head_size, self.scale, self.num_kv_heads = num_heads, head_size, float(scale), num_kv_heads
        self.alibi_slopes = None if alibi_slopes is None else torch.tensor(alibi_slopes, dtype=torch.float32)
        self.kv_cache_dtype = kv_cache_dtype
        self.num_queries_per_kv = num_heads // num_kv_heads

    # -- scratch -----------------------------------------------------------
    def _partition_scratch(self, rows: int, max_seq_len: int, dtype, device):
        P = (max_seq_len + _lib.PA_PARTITION - 1) // _lib.PA_PARTITION
        return _scratch_views(rows, self.num_heads, P, self.head_size, dtype, device)

    def _caches(self, kv_cache):
        k, v = split_kv_cache(kv_cache, self.num_kv_heads, self.head_size)
        if self.alibi_slopes is not None and self.alibi_slopes.device != kv_cache.device:
            self.alibi_slopes = self.alibi_slopes.to(kv_cache.device)
        return k, v

    # -- contract ----------------------------------------------------------
    def do_kv_cache_update(self, layer, key, value, kv_cache, slot_mapping) -> None:
        k, v = self._caches(kv_cache)
        T = slot_mapping.numel()
        ops.reshape_and_cache(key[:T].view(T, self.num_kv_heads, self.head_size),
                              value[:T].view(T, self.num_kv_heads, self.head_size), k, v, slot_mapping,
                              self.kv_cache_dtype, layer._k_scale, layer._v_scale)

    def forward(self, layer, query, key, value, kv_cache, attn_metadata: Mi355AttentionMetadata | None, output,
                output_scale=None, output_block_scale=None):
        if output_scale is not None or output_block_scale is not None:
            raise NotImplementedError("fused output quantization is not supported")
        if attn_metadata is None:     # memory-profiling run
            return output.fill_(0)
        if not attn_metadata.causal:
            raise NotImplementedError("non-causal attention is not supported")
        md = attn_metadata
        T = md.num_actual_tokens
        q = query[:T].view(T, self.num_heads, self.head_size)
        o = output[:T].view(T, self.num_heads, self.head_size)
        k, v = self._caches(kv_cache)
        bs = v.shape[3]
        P = (md.max_seq_len + _lib.PA_PARTITION - 1) // _lib.PA_PARTITION
        per_row = self.num_heads * P * self.head_size * q.element_size()
        cap = max(16, SCRATCH_RESERVE_LIMIT // per_row)
        if md.max_query_len > 1 and P > 1 and T > cap and not torch.cuda.is_current_stream_capturing():
            self._prefill_in_row_chunks(layer, q, o, k, v, bs, md, cap)
            return output
        tmp, es, ml = self._partition_scratch(T, md.max_seq_len, q.dtype, q.device)
        if md.max_query_len > 1:
            ops.paged_prefill_attention(o, es, ml, tmp, q, k, v, self.num_kv_heads, self.scale, md.block_table,
                                        md.seq_lens, md.query_start_loc, md.max_query_len, bs, md.max_seq_len,
                                        self.alibi_slopes, self.kv_cache_dtype, layer._k_scale, layer._v_scale)
        else:
            ops.paged_attention_rocm(o, es, ml, tmp, q, k, v, self.num_kv_heads, self.scale, md.block_table,
                                     md.seq_lens, md.query_start_loc, bs, md.max_seq_len, self.alibi_slopes,
                                     self.kv_cache_dtype, layer._k_scale, layer._v_scale)
        return output

    def _prefill_in_row_chunks(self, layer, q, o, k, v, bs, md, cap: int) -> None:
        """Long prompts over long contexts: the per-row partition scratch is kept to `cap` query rows by
        running the batch as consecutive row ranges.  Rows [a, b) of a sequence with query rows
        [qs, qe) and context length L form a sequence of their own with query length rb - ra and
        context length L - (qe - rb): row j then attends [0, L - (qe - qs) + (ra - qs) + j], as before."""
        qsl = md.query_start_loc.cpu().tolist()
        lens = md.seq_lens.cpu().tolist()
        T = md.num_actual_tokens
        for a in range(0, T, cap):
            b = min(T, a + cap)
            idx, new_qsl, new_lens = [], [0], []
            for s_i, L in enumerate(lens):
                qs, qe = qsl[s_i], qsl[s_i + 1]
                ra, rb = max(a, qs), min(b, qe)
                if ra >= rb:
                    continue
                idx.append(s_i)
                new_qsl.append(new_qsl[-1] + rb - ra)
                new_lens.append(L - (qe - rb))
            dev = q.device
            sel = torch.tensor(idx, dtype=torch.long, device=dev)
            c_qsl = torch.tensor(new_qsl, dtype=torch.int32, device=dev)
            c_lens = torch.tensor(new_lens, dtype=torch.int32, device=dev)
            c_maxq = max(new_qsl[i + 1] - new_qsl[i] for i in range(len(idx)))
            c_maxl = max(new_lens)
            tmp, es, ml = self._partition_scratch(b - a, c_maxl, q.dtype, dev)
            ops.paged_prefill_attention(o[a:b], es, ml, tmp, q[a:b], k, v, self.num_kv_heads, self.scale,
                                        md.block_table.index_select(0, sel).contiguous(), c_lens, c_qsl, c_maxq, bs,
                                        c_maxl, self.alibi_slopes, self.kv_cache_dtype, layer._k_scale, layer._v_scale)

    def fused_rope_kvcache_supported(self) -> bool:
        return True

    def do_rope_and_kv_cache_update(self, layer, query, key, value, positions, cos_sin_cache, is_neox, kv_cache,
                                    layer_slot_mapping) -> None:
        """RoPE on query / key and the cache write of the rotated key and of value in one launch
        (rocm_attn.py:522-562)."""
        k, v = self._caches(kv_cache)
        T = layer_slot_mapping.numel()
        ops.rope_and_cache(positions[:T], query[:T], key[:T], value[:T], self.head_size, cos_sin_cache, is_neox, k, v,
                           layer_slot_mapping, self.kv_cache_dtype, layer._k_scale, layer._v_scale)

    def decode_fused_supported(self) -> bool:
        return self.head_size == 128 or not is_quantized_kv_cache(self.kv_cache_dtype)

    def forward_decode_fused(self, layer, qkv, positions, cos_sin_cache, kv_cache,
                             attn_metadata: Mi355AttentionMetadata, output):
        """Decode rows only: NeoX RoPE, KV-cache write and attention in one launch from the fused
        qkv projection output (what do_rope_and_kv_cache_update + forward do in two steps,
        rocm_attn.py:522-562,367-469)."""
        md = attn_metadata
        S = md.num_actual_tokens
        k, v = self._caches(kv_cache)
        tmp, es, ml = self._partition_scratch(S, md.max_seq_len, qkv.dtype, qkv.device)
        ops.decode_attention_fused(output[:S].view(S, self.num_heads, self.head_size), es, ml, tmp, qkv[:S], positions,
                                   cos_sin_cache, k, v, self.num_heads, self.num_kv_heads, self.head_size, self.scale,
                                   md.block_table, md.seq_lens, v.shape[3], md.max_seq_len, self.kv_cache_dtype,
                                   layer._k_scale, layer._v_scale)
        return output


class Attention(torch.nn.Module):
    """The layer object models hold (vllm/model_executor/layers/attention/attention.py:477-567):
    owns the scale buffers and the impl; forward = cache update, then attention."""

    def __init__(self, num_heads: int, head_size: int, scale: float, num_kv_heads: int | None = None,
                 cache_dtype: str = "auto", quant_config=None, prefix: str = "", alibi_slopes=None):
        super().__init__()
        self.num_heads, self.head_size = num_heads, head_size
        self.num_kv_heads = num_kv_heads or num_heads
        self.kv_cache_dtype = cache_dtype
        self.layer_name = prefix
        set_default_quant_scales(self, register_buffer=True)
        method = quant_config.get_quant_method(self, prefix) if quant_config is not None else None
        if not isinstance(method, BaseKVCacheMethod):
            method = BaseKVCacheMethod(quant_config) if is_quantized_kv_cache(cache_dtype) else None
        self.quant_method = method
        if self.quant_method is not None:
            self.quant_method.create_weights(self)
        self.impl = Mi355AttentionImpl(num_heads, head_size, scale, self.num_kv_heads, alibi_slopes, None, cache_dtype)
        self.kv_cache: torch.Tensor | None = None

    def process_weights_after_loading(self) -> None:
        if self.quant_method is not None:
            self.quant_method.process_weights_after_loading(self)

    def forward(self, query, key, value, attn_metadata: Mi355AttentionMetadata, output=None):
        if output is None:
            output = torch.empty((query.shape[0], self.num_heads * self.head_size), dtype=query.dtype,
                                 device=query.device)
        if attn_metadata is not None:
            self.impl.do_kv_cache_update(self, key, value, self.kv_cache, attn_metadata.slot_mapping)
        return self.impl.forward(self, query, key, value, self.kv_cache, attn_metadata, output)
