"""ctypes binding of libmi355q.so (the C ABI declared in include/mi355q.h).

The library is loaded from mi355q/lib/ inside the source tree.  There is no CPU
fallback: if the shared object is missing, import of the ops raises.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355Q_LIB_PATH") or os.path.join(_HERE, "lib", "libmi355q.so")   # override: kernel experiments
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "include", "mi355q.h"))

OK, EINVAL, ENOSPACE, EHIP = 0, -1, -2, -3
BF16, FP16, FP32 = 0, 1, 2
W_GPTQ, W_NSEQ, W_AWQ = 0, 1, 2
W_PAIR_HALVES = 0x100
ZP_CONST, ZP_NSEQ, ZP_AWQ, ZP_GPTQ_V1 = 0, 1, 2, 3
PA_PARTITION = 256

_lib = None

c_int, c_i64, c_f32, c_sz, c_vp = (ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t,
                                    ctypes.c_void_p)

# argtypes of every entry point, in header order
_SIGS = {
    "mi355q_last_error": (ctypes.c_char_p, []),
    "mi355q_version": (c_int, []),
    "mi355q_set_option": (c_int, [ctypes.c_char_p, ctypes.c_char_p]),
    "mi355q_reset_library_state": (c_int, [c_vp]),
    "mi355q_w4a16_choose_tile": (c_int, [c_int, c_int, c_int]),
    "mi355q_w4a16_choose_strip": (c_int, [c_int, c_int, c_int]),
    "mi355q_moe_w4a16_choose_strip": (c_int, [c_int, c_int, c_int, c_int]),
    "mi355q_w4a16_repack": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "mi355q_w4a16_repack_host": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "mi355q_w4a16_unpack_host": (c_int, [c_vp, c_vp, c_int, c_int, c_int]),
    "mi355q_w4a16_packed_bytes": (c_sz, [c_int, c_int]),
    "mi355q_w4a16_workspace_bytes": (c_sz, [c_int, c_int, c_int, c_int]),
    "mi355q_workspace_init": (c_int, [c_vp, c_sz, c_vp]),
    "mi355q_w4a16_gemm_add_rms_norm": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp,
                                               c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_sz, c_int, c_vp]),
    "mi355q_w4a16_gemm_moe_sum_add_rms_norm_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "mi355q_w4a16_gemm_moe_sum_add_rms_norm": (c_int, [c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp,
                                                       c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "mi355q_w4a16_gemm": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_vp, c_sz, c_int, c_vp]),
    "mi355q_w4a16_gemm_silu_mul": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int,
                                           c_int, c_int, c_int, c_vp, c_sz, c_int, c_vp]),
    "mi355q_w4a16_dequantize": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "mi355q_w4a16_gate_up_silu": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, ctypes.c_float, c_vp, c_vp, c_vp, c_vp,
                                          c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "mi355q_w4a16_gate_up_silu_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "mi355q_w4a16_mm": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_int, c_vp, c_sz, c_vp]),
    "mi355q_w4a16_mm_min_rows": (c_int, [c_int, c_int, c_int, c_int]),
    "mi355q_w4a16_mm_streams": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "mi355q_w4a16_mm_add_rms_norm_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "mi355q_w4a16_mm_add_rms_norm_workspace_bytes": (c_sz, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "mi355q_w4a16_mm_add_rms_norm": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_int, c_int,
                                             c_int, c_int, c_int, c_int, c_vp, c_sz, c_vp]),
    "mi355q_w4a16_mm_workspace_bytes": (c_sz, [c_int, c_int, c_int, c_int, c_int]),
    "mi355q_w4a16_mm_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "mi355q_topk_softmax": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "mi355q_peer_ar_handle_bytes": (c_int, []),
    "mi355q_peer_ar_init": (c_int, [c_int, c_int, c_sz, ctypes.POINTER(c_vp)]),
    "mi355q_peer_ar_get_handle": (c_int, [c_vp, c_vp]),
    "mi355q_peer_ar_open_handles": (c_int, [c_vp, c_vp]),
    "mi355q_peer_ar_max_bytes": (c_sz, [c_vp]),
    "mi355q_peer_ar_all_reduce": (c_int, [c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
    "mi355q_peer_ar_all_gather": (c_int, [c_vp, c_vp, c_vp, c_int, c_sz, c_int, c_vp]),
    "mi355q_peer_ar_timeouts": (c_int, [c_vp, ctypes.POINTER(c_int)]),
    "mi355q_peer_ar_destroy": (c_int, [c_vp]),
    "mi355q_moe_router_topk_supported": (c_int, [c_int, c_int]),
    "mi355q_moe_router_topk": (c_int, [c_vp, c_i64, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int,
                                       c_int, c_int, c_int, c_vp]),
    "mi355q_moe_w4a16_gemm": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int,
                                      c_int, c_int, c_int, c_int, c_vp]),
    "mi355q_moe_align_block_size": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    "mi355q_moe_grouped_block_rows": (c_int, []),
    "mi355q_moe_w4a16_grouped_mm_workspace_bytes": (c_sz, [c_int, c_int, c_int, c_int, c_int]),
    "mi355q_moe_w4a16_grouped_mm": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int,
                                            c_int, c_int, c_int, c_int, c_vp, c_sz, c_vp]),
    "mi355q_moe_sum": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "mi355q_w8a16_gemm": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "mi355q_w8a16_dequantize": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "mi355q_wi8a16_gemm": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "mi355q_wi8a16_dequantize": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "mi355q_static_scaled_fp8_quant": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_int, c_int, c_int, c_int, c_i64, c_i64,
                                               c_int, c_vp]),
    "mi355q_dynamic_scaled_fp8_quant": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_int, c_int, c_int, c_vp]),
    "mi355q_dynamic_per_token_scaled_fp8_quant": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_int,
                                                          c_vp]),
    "mi355q_rms_norm_static_fp8_quant": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_f32, c_int, c_int, c_int, c_vp]),
    "mi355q_fused_add_rms_norm_static_fp8_quant": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_f32, c_int, c_int,
                                                           c_int, c_vp]),
    "mi355q_rms_norm_dynamic_per_token_quant": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_f32, c_int, c_int,
                                                        c_int, c_vp]),
    "mi355q_silu_and_mul_quant": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "mi355q_fp8_scaled_mm": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_i64, c_int,
                                     c_int, c_int, c_int, c_vp]),
    "mi355q_per_token_group_fp8_quant": (c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_f32,
                                                 c_int, c_vp]),
    "mi355q_fp8_block_scaled_mm": (c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_int,
                                           c_vp, c_vp, c_i64, c_int, c_int, c_int, c_int, c_vp]),
    "mi355q_linear_skinny": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "mi355q_rms_norm": (c_int, [c_vp, c_vp, c_i64, c_vp, c_f32, c_int, c_int, c_int, c_vp]),
    "mi355q_fused_add_rms_norm": (c_int, [c_vp, c_i64, c_vp, c_vp, c_f32, c_int, c_int, c_int, c_vp]),
    "mi355q_splitk_add_rms_norm": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_f32, c_int, c_int, c_int, c_vp]),
    "mi355q_rotary_embedding": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_int,
                                        c_int, c_int, c_int, c_int, c_vp]),
    "mi355q_silu_and_mul": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "mi355q_silu_and_mul_pairs": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "mi355q_reshape_and_cache": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_int, c_int, c_int,
                                         c_int, c_i64, c_int, c_int, c_vp, c_vp, c_vp]),
    "mi355q_reshape_and_cache_flash": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int,
                                               c_i64, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_int, c_vp]),
    "mi355q_rope_and_cache": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_int,
                                      c_int, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_i64, c_int, c_int, c_vp, c_vp,
                                      c_vp]),
    "mi355q_splitk_rope_and_cache": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp,
                                             c_vp, c_int, c_int, c_i64, c_int, c_int, c_vp, c_vp, c_vp]),
    "mi355q_w4a16_mm_rope_and_cache": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int,
                                               c_int, c_vp, c_sz, c_vp, c_int, c_int, c_int, c_vp, c_int, c_int, c_int,
                                               c_vp, c_vp, c_vp, c_int, c_i64, c_int, c_vp, c_vp, c_vp]),
    "mi355q_convert_fp8": (c_int, [c_vp, c_vp, c_i64, c_f32, c_int, c_int, c_vp]),
    "mi355q_paged_attention": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_int,
                                       c_int, c_int, c_int, c_f32, c_vp, c_int, c_vp, c_vp, c_int, c_int,
                                       c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "mi355q_paged_attention_noalloc": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_int,
                                               c_f32, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_int, c_vp,
                                               c_vp, c_vp]),
    "mi355q_paged_prefill_attention": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64,
                                               c_int, c_int, c_int, c_int, c_f32, c_vp, c_int, c_vp, c_vp,
                                               c_int, c_int, c_int, c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "mi355q_decode_attention_fused": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64,
                                              c_int, c_int, c_int, c_int, c_f32, c_vp, c_int, c_vp, c_int, c_int,
                                              c_int, c_int, c_vp, c_vp, c_vp]),
    "mi355q_argmax": (c_int, [c_vp, c_i64, c_vp, c_int, c_int, c_int, c_vp]),
    "mi355q_argmax_decode_step": (c_int, [c_vp, c_i64, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64,
                                          c_int, c_vp]),
    "mi355q_gptq_expand_bits": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "mi355q_sample_topk_topp": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    "mi355q_apply_repetition_penalties": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
}


def declared_symbols():
    """Names of the entry points declared in include/mi355q.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(mi355q_[a-z0-9_]+)\s*\(", text)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C vllm-fork_amd/csrc` "
                "(there is no CPU fallback for the MI355X kernels)")
        # torch ships its own HIP runtime under the same SONAME; it must be the one already
        # loaded when this library resolves libamdhip64, so that kernels, streams and
        # allocations all belong to one runtime instance
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def set_option(name: str, value: str | None) -> None:
    """Set or clear one of the library's MI355Q_* knobs for the rest of the process (the library reads
    the environment only once)."""
    check(lib().mi355q_set_option(name.encode(), None if value is None else value.encode()))


def check(status):
    if status != OK:
        msg = lib().mi355q_last_error().decode()
        raise RuntimeError(f"mi355q error {status}: {msg}")
