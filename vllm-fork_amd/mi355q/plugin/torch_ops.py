"""torch.library registration of the kernel entry points under the `mi355q::` namespace.

The reference exposes its kernels as dispatcher ops with schemas that mark in-place arguments
(`Tensor!`) and pairs each with a fake implementation so Dynamo and graph capture can trace
through them (csrc/torch_bindings.cpp:369-377,417-421,653-660,889-894;
vllm/_custom_ops.py:580-595).  The same is done here: functional ops return a new tensor,
in-place ops mutate the marked arguments and return nothing, and every op has a
shape-only fake.  Argument order follows the reference schema of the op it stands in for.
"""
import torch
from torch.library import Library, impl, register_fake

from .. import ops

_lib = Library("mi355q", "FRAGMENT")
_DEFINED = False


def _w(packed, scales, zc, K, N, G, tile) -> ops.PackedW4:
    return ops.PackedW4(packed, scales, zc, K, N, G, tile)


def register_torch_ops() -> None:
    """Idempotent: the plug-in entry point runs in every process (front end, engine, workers)."""
    global _DEFINED
    if _DEFINED:
        return
    _DEFINED = True
    d = _lib.define
    d("w4a16_gemm(Tensor x, Tensor packed, Tensor scales, Tensor? zc, Tensor? bias, int K, int N, int G, int tile) -> Tensor")
    d("w4a16_gemm_silu_mul(Tensor x, Tensor packed, Tensor scales, Tensor? zc, Tensor? bias, int K, int N, int G, int tile) -> Tensor")
    d("w4a16_dequantize(Tensor packed, Tensor scales, Tensor? zc, int K, int N, int G, int tile) -> Tensor")
    d("rms_norm(Tensor! result, Tensor input, Tensor? weight, float epsilon) -> ()")
    d("fused_add_rms_norm(Tensor! input, Tensor! residual, Tensor? weight, float epsilon) -> ()")
    d("rotary_embedding(Tensor positions, Tensor! query, Tensor!? key, int head_size, Tensor cos_sin_cache, bool is_neox) -> ()")
    d("silu_and_mul(Tensor! result, Tensor input) -> ()")
    d("reshape_and_cache(Tensor key, Tensor value, Tensor! key_cache, Tensor! value_cache, Tensor slot_mapping, "
      "str kv_cache_dtype, Tensor k_scale, Tensor v_scale) -> ()")
    d("reshape_and_cache_flash(Tensor key, Tensor value, Tensor! key_cache, Tensor! value_cache, Tensor slot_mapping, "
      "str kv_cache_dtype, Tensor k_scale, Tensor v_scale) -> ()")
    d("paged_attention(Tensor! out, Tensor! exp_sums, Tensor! max_logits, Tensor! tmp_out, Tensor query, "
      "Tensor key_cache, Tensor value_cache, int num_kv_heads, float scale, Tensor block_tables, Tensor seq_lens, "
      "Tensor? query_start_loc, int block_size, int max_seq_len, Tensor? alibi_slopes, str kv_cache_dtype, "
      "Tensor k_scale, Tensor v_scale) -> ()")
    d("argmax(Tensor logits) -> Tensor")
    d("apply_repetition_penalties_(Tensor! logits, Tensor prompt_mask, Tensor output_mask, Tensor repetition_penalties) -> ()")

    @impl(_lib, "w4a16_gemm", "CUDA")
    def _(x, packed, scales, zc, bias, K, N, G, tile):
        return ops.w4a16_gemm(x, _w(packed, scales, zc, K, N, G, tile), bias=bias)

    @register_fake("mi355q::w4a16_gemm")
    def _(x, packed, scales, zc, bias, K, N, G, tile):
        return x.new_empty(x.shape[:-1] + (N,))

    @impl(_lib, "w4a16_gemm_silu_mul", "CUDA")
    def _(x, packed, scales, zc, bias, K, N, G, tile):
        return ops.w4a16_gemm(x, _w(packed, scales, zc, K, N, G, tile), bias=bias, fuse_silu_mul=True)

    @register_fake("mi355q::w4a16_gemm_silu_mul")
    def _(x, packed, scales, zc, bias, K, N, G, tile):
        return x.new_empty(x.shape[:-1] + (N,))

    @impl(_lib, "w4a16_dequantize", "CUDA")
    def _(packed, scales, zc, K, N, G, tile):
        return ops.w4a16_dequantize(_w(packed, scales, zc, K, N, G, tile))

    @register_fake("mi355q::w4a16_dequantize")
    def _(packed, scales, zc, K, N, G, tile):
        return scales.new_empty((K, N))

    @impl(_lib, "rms_norm", "CUDA")
    def _(result, input, weight, epsilon):
        ops.rms_norm(result, input, weight, epsilon)

    @impl(_lib, "fused_add_rms_norm", "CUDA")
    def _(input, residual, weight, epsilon):
        ops.fused_add_rms_norm(input, residual, weight, epsilon)

    @impl(_lib, "rotary_embedding", "CUDA")
    def _(positions, query, key, head_size, cos_sin_cache, is_neox):
        ops.rotary_embedding(positions, query, key, head_size, cos_sin_cache, is_neox)

    @impl(_lib, "silu_and_mul", "CUDA")
    def _(result, input):
        ops.silu_and_mul(result, input)

    @impl(_lib, "reshape_and_cache", "CUDA")
    def _(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype, k_scale, v_scale):
        ops.reshape_and_cache(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype, k_scale, v_scale)

    @impl(_lib, "reshape_and_cache_flash", "CUDA")
    def _(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype, k_scale, v_scale):
        ops.reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype, k_scale, v_scale)

    @impl(_lib, "paged_attention", "CUDA")
    def _(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_kv_heads, scale, block_tables,
          seq_lens, query_start_loc, block_size, max_seq_len, alibi_slopes, kv_cache_dtype, k_scale, v_scale):
        ops.paged_attention_rocm(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_kv_heads,
                                 scale, block_tables, seq_lens, query_start_loc, block_size, max_seq_len,
                                 alibi_slopes, kv_cache_dtype, k_scale, v_scale)

    @impl(_lib, "apply_repetition_penalties_", "CUDA")
    def _(logits, prompt_mask, output_mask, repetition_penalties):
        ops.apply_repetition_penalties(logits, prompt_mask, output_mask, repetition_penalties)

    for name in ("rms_norm", "fused_add_rms_norm", "rotary_embedding", "silu_and_mul", "reshape_and_cache",
                 "reshape_and_cache_flash", "paged_attention", "apply_repetition_penalties_"):
        register_fake(f"mi355q::{name}")(lambda *a, **k: None)

    @impl(_lib, "argmax", "CUDA")
    def _(logits):
        return ops.argmax(logits)

    @register_fake("mi355q::argmax")
    def _(logits):
        return logits.new_empty((logits.shape[0],), dtype=torch.int64)
