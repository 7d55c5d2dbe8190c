"""Tensor-level ops of the MI355X quantized-decode path.

Function names, argument order and in-place/functional behaviour follow
vllm/_custom_ops.py of the reference fork (cited per function), so call sites in
the fork can be pointed here unchanged.  Every op launches on torch's current
stream and never synchronises; outputs come from torch's caching allocator, so
the ops are HIP-graph capturable.  All arithmetic runs in libmi355q.so; there is
no eager/CPU fallback.
"""
import os

import torch

from . import _lib
from ._lib import BF16, FP16, FP32, check

_DT = {torch.bfloat16: BF16, torch.float16: FP16, torch.float32: FP32}


def _dt(t: torch.Tensor, allow_fp32: bool = False) -> int:
    if t.dtype not in _DT or (t.dtype == torch.float32 and not allow_fp32):
        raise RuntimeError(f"unsupported dtype {t.dtype}")
    return _DT[t.dtype]


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mi355q ops need device tensors (MI355X); no CPU fallback exists")


# ---------------------------------------------------------------------------
# W4A16
# ---------------------------------------------------------------------------
class PackedW4:
    """Kernel-layout 4-bit weight of one linear layer (output of `w4a16_repack`)."""

    __slots__ = ("packed", "scales", "zc", "K", "N", "G", "tile", "pair_halves", "dense")

    def __init__(self, packed, scales, zc, K, N, G, tile, pair_halves=False):
        self.packed, self.scales, self.zc, self.K, self.N, self.G, self.tile = packed, scales, zc, K, N, G, tile
        self.pair_halves = pair_halves
        self.dense = None   # expanded [K, N] copy, kept only under MI355Q_PROMPT_DENSE_RESIDENT=1


def w4a16_repack(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor | None,
                 group_size: int, src_layout: int = _lib.W_GPTQ, zp_mode: int = _lib.ZP_CONST,
                 tile: int = 0, pair_halves: bool = False) -> PackedW4:
    """Checkpoint tensors -> kernel layout.  Stands in for
    MPLinearKernel.process_weights_after_loading (mixed_precision/triton_w4a16.py:338-426).
    Works on device tensors (HIP kernel) and on CPU tensors (host arithmetic, for load
    paths that repack before upload).  pair_halves interleaves the two halves of the output
    channels (a merged gate_up projection: gate_i next to up_i) for `w4a16_gate_up_silu`."""
    if src_layout == _lib.W_GPTQ:
        K, N = qweight.shape[0] * 8, qweight.shape[1]
    else:
        K, N = qweight.shape[0], qweight.shape[1] * 8
    G = K if group_size in (-1, 0) else group_size
    if scales.shape != (K // G, N):
        raise RuntimeError(f"scales shape {tuple(scales.shape)} != ({K // G}, {N})")
    if qweight.dtype != torch.int32 or not qweight.is_contiguous():
        raise RuntimeError("qweight must be a contiguous int32 tensor")
    act = _dt(scales)
    packed = torch.empty(K * N // 8, dtype=torch.int32, device=qweight.device)
    zc = None
    if zp_mode != _lib.ZP_CONST:
        if qzeros is None or qzeros.shape != (K // G, N // 8) or qzeros.dtype != torch.int32:
            raise RuntimeError("qzeros must be int32 [K/G, N/8] for asymmetric formats")
        qzeros = qzeros.contiguous()
        zc = torch.empty((K // G, N), dtype=scales.dtype, device=qweight.device)
    L = _lib.lib()
    if tile == 0:
        tile = L.mi355q_w4a16_choose_strip(K, N, G)
    if pair_halves:
        src_layout |= _lib.W_PAIR_HALVES
        scales = scales.view(K // G, 2, N // 2).transpose(1, 2).reshape(K // G, N)
    if qweight.is_cuda:
        check(L.mi355q_w4a16_repack(qweight.data_ptr(), _ptr(qzeros if zc is not None else None),
                                    packed.data_ptr(), _ptr(zc), K, N, G, tile, src_layout, zp_mode, act, _stream()))
    else:
        check(L.mi355q_w4a16_repack_host(qweight.data_ptr(), _ptr(qzeros if zc is not None else None),
                                         packed.data_ptr(), _ptr(zc), K, N, G, tile, src_layout, zp_mode, act))
    # strip-major scales, [N / tile][K / G][tile]: one contiguous run per workgroup (zc comes out that way)
    scales = scales.reshape(K // G, N // tile, tile).permute(1, 0, 2).contiguous()
    if zc is not None:
        zc = zc.view(N // tile, K // G, tile)
    return PackedW4(packed, scales, zc, K, N, G, tile, pair_halves)


def scales_row_major(w: PackedW4, zc: bool = False) -> torch.Tensor:
    """The [K/G, N] view of a repacked weight's scales (or zc) -- inverse of the strip-major storage."""
    t = w.zc if zc else w.scales
    return t.reshape(w.N // w.tile, w.K // w.G, w.tile).permute(1, 0, 2).reshape(w.K // w.G, w.N)


_WS: dict = {}
_WS_RETIRED: list = []
DEQUANT_MATMUL_MIN_TOKENS = 32
# token range served by the tiled matrix-core W4A16 kernel (below: the decode kernels; above: dequantise
# once + dense GEMM, compute-bound and best left to the BLAS).  Upper end by measurement on Llama-3-8B
# (TTFT, tiled / dequantise + BLAS): 900 tokens 19.8 / 21.3 ms, 1024 19.8 / 21.1, 1280 27.5 / 24.5.
MM_MIN_TOKENS = _MM_MIN_TOKENS_DEFAULT = 17
MM_MAX_TOKENS = int(os.environ.get("MI355Q_MM_MAX_TOKENS", "1024"))
# Above MM_MAX_TOKENS every call expands the layer's weights into the staging buffer first (a quarter of the
# route's time at 2048 tokens).  With 288 GB per GPU a deployment may prefer to keep the expanded copy of
# every layer next to the packed one (4 x the packed bytes: 14 GB for Llama-3-8B): decode still streams the
# 4-bit weights, long prompts skip the expansion.  Off by default -- the packed weight stays the only copy.
DENSE_RESIDENT = os.environ.get("MI355Q_PROMPT_DENSE_RESIDENT", "0") == "1"


COUNTER_BYTES = 64 * 1024   # arrival counters at the head of the workspace; partial tiles follow


def _workspace(dev, nbytes: int) -> torch.Tensor:
    """Per-device split-K workspace (arrival counters + fp32 partial tiles).  Created zeroed,
    grown on demand outside graph capture and then reused, so its address is stable under
    capture; GEMMs leave the counters zero."""
    ws = _WS.get(dev)
    if ws is None or ws.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("split-K workspace must be sized before graph capture (run the op once eagerly)")
        if ws is not None:
            _WS_RETIRED.append(ws)   # captured graphs may still hold its address
        ws = torch.zeros(max(nbytes, 1 << 24), dtype=torch.uint8, device=dev)
        _WS[dev] = ws
    return ws


def w4a16_gemm(x: torch.Tensor, w: PackedW4, bias: torch.Tensor | None = None, split_k: int = 0,
               fuse_silu_mul: bool = False, out: torch.Tensor | None = None) -> torch.Tensor:
    """x[..., K] @ dequant(w)[K, N] (+ bias).  Stands in for ops.gptq_gemm
    (vllm/_custom_ops.py:558-577), ops.awq_gemm (:525-536) and triton_w4a16_gemm
    (mixed_precision/triton_w4a16.py:164-270).  With fuse_silu_mul, x is the
    [..., 2K] gate_up output and SiluAndMul is applied on the fly."""
    _need_cuda(x, w.packed)
    Kx = w.K * (2 if fuse_silu_mul else 1)
    if x.shape[-1] != Kx:
        raise RuntimeError(f"input feature size {x.shape[-1]} != {Kx}")
    if x.dtype != w.scales.dtype:
        raise RuntimeError(f"activation dtype {x.dtype} != scales dtype {w.scales.dtype}")
    x2 = x.reshape(-1, Kx)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    M = x2.shape[0]
    if out is not None and (out.shape != (M, w.N) or out.dtype != x.dtype or not out.is_contiguous()):
        raise RuntimeError(f"out must be a contiguous [{M}, {w.N}] tensor of the activation dtype")
    if M <= MM_MAX_TOKENS and not fuse_silu_mul and M >= mm_min_tokens(w) and _mm_supported(w):
        return w4a16_mm(x2, w, bias, out=out).reshape(x.shape[:-1] + (w.N,))
    if M > DEQUANT_MATMUL_MIN_TOKENS and not torch.cuda.is_current_stream_capturing():
        # many tokens (prompt processing): dequantise once and run a dense GEMM, as the reference
        # does from 256 tokens on (auto_awq.py:947-955; exllama reconstruct path q_gemm.cu:1509-1522)
        if fuse_silu_mul:
            act = torch.empty((M, w.K), dtype=x.dtype, device=x.device)
            silu_and_mul(act, x2.contiguous())
            x2 = act
        if DENSE_RESIDENT:
            if w.dense is None:
                w.dense = w4a16_dequantize(w)
            dense = w.dense
        else:
            dense = w4a16_dequantize(w, out=_dequant_stage(w))
        res = torch.matmul(x2, dense, out=out)
        if bias is not None:
            res.add_(bias)
        return res.reshape(x.shape[:-1] + (w.N,))
    if out is None:
        out = torch.empty((M, w.N), dtype=x.dtype, device=x.device)
    L = _lib.lib()
    ws_bytes = L.mi355q_w4a16_workspace_bytes(M, w.N, w.K, w.G)
    ws = _workspace(x.device, ws_bytes)
    fn = L.mi355q_w4a16_gemm_silu_mul if fuse_silu_mul else L.mi355q_w4a16_gemm
    check(fn(x2.data_ptr(), x2.stride(0), w.packed.data_ptr(), w.scales.data_ptr(), _ptr(w.zc), _ptr(bias),
             out.data_ptr(), M, w.N, w.K, w.G, w.tile, _dt(x), ws.data_ptr(), ws.numel(), split_k, _stream()))
    return out.reshape(x.shape[:-1] + (w.N,))


def _mm_supported(w: PackedW4) -> bool:
    return bool(_lib.lib().mi355q_w4a16_mm_supported(w.N, w.K, w.G, w.tile))


def mm_min_tokens(w: PackedW4) -> int:
    """Row count from which `w4a16_mm` beats the few-row entries on this weight: MM_MIN_TOKENS, or fewer where the
    streaming rows kernel takes over earlier (112-wide strips: the single-pass GEMV kernel runs out of LDS for the
    activation rows at 13)."""
    if MM_MIN_TOKENS != _MM_MIN_TOKENS_DEFAULT:   # a caller moved the line (tests of the few-row kernels at every M)
        return MM_MIN_TOKENS
    return min(MM_MIN_TOKENS, int(_lib.lib().mi355q_w4a16_mm_min_rows(w.N, w.K, w.G, w.tile)))


def w4a16_mm(x: torch.Tensor, w: PackedW4, bias: torch.Tensor | None = None, silu_pairs: bool = False,
             out: torch.Tensor | None = None) -> torch.Tensor:
    """[tokens, K] @ dequant(w) for prompt-sized token counts on the tiled matrix-core kernel: the packed
    weight is read once per 64 tokens and never expanded in memory.  Stands in for the many-token
    routes of ops.gptq_gemm (q_gemm.cu:1509-1522) and awq (auto_awq.py:947-955).  With silu_pairs
    (weight repacked with pair_halves) the output is SiluAndMul of the projection, [tokens, N/2]."""
    _need_cuda(x, w.packed)
    if x.dim() != 2 or x.shape[1] != w.K or x.stride(1) != 1:
        raise RuntimeError(f"w4a16_mm: input must be [tokens, {w.K}] with unit inner stride")
    if x.dtype != w.scales.dtype:
        raise RuntimeError(f"activation dtype {x.dtype} != scales dtype {w.scales.dtype}")
    if silu_pairs and not w.pair_halves:
        raise RuntimeError("w4a16_mm: silu_pairs needs a weight repacked with pair_halves=True")
    M = x.shape[0]
    L = _lib.lib()
    if out is None:
        out = torch.empty((M, w.N // 2 if silu_pairs else w.N), dtype=x.dtype, device=x.device)
    need = L.mi355q_w4a16_mm_workspace_bytes(M, w.N, w.K, w.G, w.tile)
    ws = _workspace(x.device, need + COUNTER_BYTES) if need else None
    check(L.mi355q_w4a16_mm(x.data_ptr(), x.stride(0), w.packed.data_ptr(), w.scales.data_ptr(), _ptr(w.zc), _ptr(bias),
                            out.data_ptr(), M, w.N, w.K, w.G, w.tile, _dt(x), int(silu_pairs),
                            ws.data_ptr() + COUNTER_BYTES if ws is not None else None,
                            ws.numel() - COUNTER_BYTES if ws is not None else 0, _stream()))
    return out


def w4a16_mm_add_rms_norm_supported(tokens: int, w: PackedW4) -> bool:
    return bool(_lib.lib().mi355q_w4a16_mm_add_rms_norm_supported(tokens, w.N, w.K, w.G, w.tile, int(w.zc is not None)))


def w4a16_mm_add_rms_norm_preferred(tokens: int, w: PackedW4) -> bool:
    """The fused projection + add + norm is served AND faster than `w4a16_mm` followed by `fused_add_rms_norm`: not
    so where the projection streams its 16-wide strips without K-slices (nothing to merge, o_proj at 17-32 rows)."""
    if not w4a16_mm_add_rms_norm_supported(tokens, w):
        return False
    return not (w.tile == 16 and _lib.lib().mi355q_w4a16_mm_streams(tokens, w.N, w.K, w.G, w.tile))


def w4a16_mm_add_rms_norm(x: torch.Tensor, w: PackedW4, residual: torch.Tensor, norm_weight: torch.Tensor,
                          epsilon: float, bias: torch.Tensor | None = None) -> torch.Tensor:
    """w4a16_mm(x, w, bias) followed by fused_add_rms_norm(result, residual): `residual` is updated in
    place and the normalised rows are returned.  With K-slices the slice sums, the residual add and
    the norm are one kernel (no separate reduction launch)."""
    _need_cuda(x, w.packed, residual, norm_weight)
    if x.dim() != 2 or x.shape[1] != w.K or x.stride(1) != 1:
        raise RuntimeError(f"w4a16_mm_add_rms_norm: input must be [tokens, {w.K}] with unit inner stride")
    if x.dtype != w.scales.dtype or residual.dtype != x.dtype or norm_weight.dtype != x.dtype:
        raise RuntimeError("w4a16_mm_add_rms_norm: activation, residual, norm weight and scales must share a dtype")
    M = x.shape[0]
    if residual.shape != (M, w.N) or not residual.is_contiguous() or norm_weight.numel() != w.N:
        raise RuntimeError(f"w4a16_mm_add_rms_norm: residual must be contiguous [{M}, {w.N}] and the norm weight [{w.N}]")
    L = _lib.lib()
    if not L.mi355q_w4a16_mm_add_rms_norm_supported(M, w.N, w.K, w.G, w.tile, int(w.zc is not None)):
        raise RuntimeError(f"w4a16_mm_add_rms_norm: {M} rows of a [{w.K}, {w.N}] projection (strip {w.tile}) are not served")
    out = torch.empty((M, w.N), dtype=x.dtype, device=x.device)
    need = L.mi355q_w4a16_mm_add_rms_norm_workspace_bytes(M, w.N, w.K, w.G, w.tile, int(w.zc is not None))
    ws = _workspace(x.device, need + COUNTER_BYTES) if need else None
    check(L.mi355q_w4a16_mm_add_rms_norm(
        x.data_ptr(), x.stride(0), w.packed.data_ptr(), w.scales.data_ptr(), _ptr(w.zc), _ptr(bias), residual.data_ptr(),
        norm_weight.data_ptr(), epsilon, out.data_ptr(), M, w.N, w.K, w.G, w.tile, _dt(x),
        ws.data_ptr() + COUNTER_BYTES if ws is not None else None, ws.numel() - COUNTER_BYTES if ws is not None else 0,
        _stream()))
    return out


def w4a16_mm_rope_and_cache(x: torch.Tensor, w: PackedW4, positions: torch.Tensor, head_size: int, num_heads: int,
                            num_kv_heads: int, cos_sin_cache: torch.Tensor, is_neox: bool, key_cache: torch.Tensor,
                            value_cache: torch.Tensor, slot_mapping: torch.Tensor, kv_cache_dtype: str, k_scale, v_scale,
                            bias: torch.Tensor | None = None) -> torch.Tensor:
    """The fused qkv projection of prompt tokens, w4a16_mm(x, w, bias), followed by rope_and_cache on its
    q / k / v column ranges; returns qkv [tokens, (heads + 2 kv_heads) * head] with q and k rotated.  When
    the projection is computed in K-slices, their merge, the rotation and the cache write are one launch."""
    _need_cuda(x, w.packed, positions, cos_sin_cache, key_cache, value_cache, slot_mapping)
    if x.dim() != 2 or x.shape[1] != w.K or x.stride(1) != 1:
        raise RuntimeError(f"w4a16_mm_rope_and_cache: input must be [tokens, {w.K}] with unit inner stride")
    if x.dtype != w.scales.dtype:
        raise RuntimeError(f"activation dtype {x.dtype} != scales dtype {w.scales.dtype}")
    M = x.shape[0]
    if w.N != (num_heads + 2 * num_kv_heads) * head_size:
        raise RuntimeError(f"w4a16_mm_rope_and_cache: N = {w.N} is not (heads + 2 kv_heads) * head_size")
    if positions.dtype != torch.int64 or slot_mapping.dtype != torch.int64 or positions.numel() != M or \
            slot_mapping.numel() != M:
        raise RuntimeError("w4a16_mm_rope_and_cache: positions and slot_mapping must be int64, one entry per token")
    if key_cache.shape[1] != num_kv_heads or value_cache.shape[1] != num_kv_heads:
        raise RuntimeError("w4a16_mm_rope_and_cache: cache head count differs from num_kv_heads")
    L = _lib.lib()
    qkv = torch.empty((M, w.N), dtype=x.dtype, device=x.device)
    need = L.mi355q_w4a16_mm_workspace_bytes(M, w.N, w.K, w.G, w.tile)
    ws = _workspace(x.device, need + COUNTER_BYTES) if need else None
    check(L.mi355q_w4a16_mm_rope_and_cache(
        x.data_ptr(), x.stride(0), w.packed.data_ptr(), w.scales.data_ptr(), _ptr(w.zc), _ptr(bias), qkv.data_ptr(),
        M, w.N, w.K, w.G, w.tile, _dt(x), ws.data_ptr() + COUNTER_BYTES if ws is not None else None,
        ws.numel() - COUNTER_BYTES if ws is not None else 0, positions.data_ptr(), head_size, num_heads, num_kv_heads,
        cos_sin_cache.data_ptr(), cos_sin_cache.shape[1], _dt(cos_sin_cache, True), int(is_neox), key_cache.data_ptr(),
        value_cache.data_ptr(), slot_mapping.data_ptr(), value_cache.shape[3], key_cache.stride(0),
        int(_is_fp8(kv_cache_dtype)), _ptr(k_scale), _ptr(v_scale), _stream()))
    return qkv


def w4a16_gemm_add_rms_norm(x: torch.Tensor, residual: torch.Tensor | None, norm_weight: torch.Tensor | None,
                            epsilon: float, w: PackedW4, bias: torch.Tensor | None = None, split_k: int = 0):
    """fused_add_rms_norm(x, residual) followed by the W4A16 GEMM, in one kernel.
    Returns (gemm output, new residual); x and residual are left untouched."""
    _need_cuda(x, w.packed)
    if x.shape[-1] != w.K:
        raise RuntimeError(f"input feature size {x.shape[-1]} != {w.K}")
    x2 = x.reshape(-1, w.K)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    M = x2.shape[0]
    if residual is not None and (not residual.is_contiguous() or residual.numel() != M * w.K):
        raise RuntimeError("residual must be a contiguous [tokens, hidden] tensor")
    out = torch.empty((M, w.N), dtype=x.dtype, device=x.device)
    res_out = torch.empty((M, w.K), dtype=x.dtype, device=x.device)
    L = _lib.lib()
    ws = _workspace(x.device, L.mi355q_w4a16_workspace_bytes(M, w.N, w.K, w.G))
    check(L.mi355q_w4a16_gemm_add_rms_norm(
        x2.data_ptr(), x2.stride(0), _ptr(residual), res_out.data_ptr(), _ptr(norm_weight), epsilon,
        w.packed.data_ptr(), w.scales.data_ptr(), _ptr(w.zc), _ptr(bias), out.data_ptr(), M, w.N, w.K, w.G,
        w.tile, _dt(x), ws.data_ptr(), ws.numel(), split_k, _stream()))
    return out.reshape(x.shape[:-1] + (w.N,)), res_out.reshape(x.shape)


def w4a16_gemm_moe_sum_add_rms_norm_supported(num_tokens: int, topk: int, w: PackedW4) -> bool:
    return bool(_lib.lib().mi355q_w4a16_gemm_moe_sum_add_rms_norm_supported(num_tokens, topk, w.N, w.K, w.G, w.tile))


def w4a16_gemm_moe_sum_add_rms_norm(expert_out: torch.Tensor, topk_weights: torch.Tensor, residual: torch.Tensor,
                                    norm_weight: torch.Tensor | None, epsilon: float, w: PackedW4,
                                    bias: torch.Tensor | None = None):
    """moe_sum(expert_out, topk_weights) -> fused_add_rms_norm(., residual) -> W4A16 GEMM as one kernel: expert_out
    [1, 2, K] (the selected experts' outputs of one token), topk_weights fp32 [1, 2].  Returns (gemm output [1, N],
    new residual [1, K]) with the bits of the three ops run one after the other."""
    _need_cuda(expert_out, topk_weights, w.packed)
    if expert_out.dim() != 3 or expert_out.shape[2] != w.K or not expert_out.is_contiguous():
        raise RuntimeError(f"expert_out must be a contiguous [tokens, topk, {w.K}] tensor")
    T, topk = expert_out.shape[0], expert_out.shape[1]
    if topk_weights.dtype != torch.float32 or tuple(topk_weights.shape) != (T, topk) or not topk_weights.is_contiguous():
        raise RuntimeError("topk_weights must be a contiguous fp32 [tokens, topk] tensor")
    if residual is None or not residual.is_contiguous() or residual.numel() != T * w.K:
        raise RuntimeError("residual must be a contiguous [tokens, hidden] tensor")
    out = torch.empty((T, w.N), dtype=expert_out.dtype, device=expert_out.device)
    res_out = torch.empty((T, w.K), dtype=expert_out.dtype, device=expert_out.device)
    check(_lib.lib().mi355q_w4a16_gemm_moe_sum_add_rms_norm(
        expert_out.data_ptr(), topk_weights.data_ptr(), topk, residual.data_ptr(), res_out.data_ptr(), _ptr(norm_weight),
        epsilon, w.packed.data_ptr(), w.scales.data_ptr(), _ptr(w.zc), _ptr(bias), out.data_ptr(), T, w.N, w.K, w.G,
        w.tile, _dt(expert_out), _stream()))
    return out, res_out


def w4a16_gate_up_silu(x: torch.Tensor, residual: torch.Tensor | None, norm_weight: torch.Tensor | None,
                       epsilon: float, w: PackedW4, add_norm: bool = True):
    """[fused_add_rms_norm ->] gate_up projection -> SiluAndMul in one kernel, for a weight repacked with
    pair_halves.  Returns (activation [tokens, N/2], new residual or None)."""
    _need_cuda(x, w.packed)
    if not w.pair_halves:
        raise RuntimeError("w4a16_gate_up_silu needs a weight repacked with pair_halves=True")
    x2 = x.reshape(-1, w.K)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    M = x2.shape[0]
    out = torch.empty((M, w.N // 2), dtype=x.dtype, device=x.device)
    res_out = torch.empty((M, w.K), dtype=x.dtype, device=x.device) if add_norm else None
    check(_lib.lib().mi355q_w4a16_gate_up_silu(
        x2.data_ptr(), x2.stride(0), _ptr(residual), _ptr(res_out), _ptr(norm_weight), epsilon, w.packed.data_ptr(),
        w.scales.data_ptr(), _ptr(w.zc), out.data_ptr(), M, w.N, w.K, w.G, w.tile, _dt(x), _stream()))
    return out.reshape(x.shape[:-1] + (w.N // 2,)), (None if res_out is None else res_out.reshape(x.shape))


def w4a16_gate_up_silu_supported(K: int, N: int, group_size: int) -> bool:
    """Can a [K, N] gate_up layer take the pair_halves layout and the fused kernel?"""
    L = _lib.lib()
    G = K if group_size in (-1, 0) else group_size
    return bool(L.mi355q_w4a16_gate_up_silu_supported(K, N, G, L.mi355q_w4a16_choose_strip(K, N, G)))


def unpair_halves(y: torch.Tensor) -> torch.Tensor:
    """Output of a pair_halves projection [.., N] (g0, u0, g1, u1, ...) -> (g | u) halves."""
    return torch.cat((y[..., 0::2], y[..., 1::2]), dim=-1)


_DQ_STAGE: dict = {}


def _dequant_stage(w: PackedW4) -> torch.Tensor:
    """Per-device staging buffer for the expanded weights of ONE layer call on the many-token route
    (sized for the largest layer seen, reused by every later call: the route allocates nothing in steady
    state).  Calls on one device share it, so they must be stream-ordered -- they are: the buffer is
    written and consumed inside one w4a16_gemm call."""
    key = (w.packed.device, w.scales.dtype)
    need = w.K * w.N
    buf = _DQ_STAGE.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.empty(need, dtype=w.scales.dtype, device=w.packed.device)
        _DQ_STAGE[key] = buf
    return buf[:need].view(w.K, w.N)


def w4a16_dequantize(w: PackedW4, out: torch.Tensor | None = None) -> torch.Tensor:
    """[K, N] dequantised weight.  Stands in for ops.awq_dequantize (vllm/_custom_ops.py:491-505)."""
    _need_cuda(w.packed)
    if out is None:
        out = torch.empty((w.K, w.N), dtype=w.scales.dtype, device=w.packed.device)
    elif out.shape != (w.K, w.N) or out.dtype != w.scales.dtype or not out.is_contiguous():
        raise RuntimeError(f"out must be a contiguous [{w.K}, {w.N}] tensor of the scales' dtype")
    check(_lib.lib().mi355q_w4a16_dequantize(w.packed.data_ptr(), w.scales.data_ptr(), _ptr(w.zc),
                                             out.data_ptr(), w.K, w.N, w.G, w.tile, _dt(w.scales), _stream()))
    return out


def gptq_expand_bits(t: torch.Tensor, bits: int, add: int = 0, axis: int = 0) -> torch.Tensor:
    """2- / 3-bit AutoGPTQ tensor (qweight: axis 0, bit strings along K; qzeros: axis 1, along N) -> the 4-bit
    layout of the same values (+ add)."""
    _need_cuda(t)
    if t.dtype != torch.int32 or t.dim() != 2 or not t.is_contiguous():
        raise RuntimeError("gptq_expand_bits: a contiguous 2-d int32 tensor is expected")
    packed, other = (t.shape[0], t.shape[1]) if axis == 0 else (t.shape[1], t.shape[0])
    if (packed * 32) % bits:
        raise RuntimeError(f"gptq_expand_bits: {packed} words do not hold whole {bits}-bit values")
    values = packed * 32 // bits
    out = torch.empty((values // 8, other) if axis == 0 else (other, values // 8), dtype=torch.int32, device=t.device)
    check(_lib.lib().mi355q_gptq_expand_bits(t.data_ptr(), out.data_ptr(), values, other, bits, add, axis, _stream()))
    return out


def gptq_repack(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor | None, group_size: int, bits: int = 4,
                use_v2_format: bool = False, tile: int = 0) -> PackedW4:
    """What gptq_shuffle prepares for gptq_gemm (vllm/_custom_ops.py:558-583), for 2-, 3- and 4-bit AutoGPTQ
    tensors: qweight int32 [K * bits / 32, N]; qzeros int32 [K/G, N * bits / 32] or None (symmetric: zero point
    2^(bits-1)); stored zero points are z - 1 unless use_v2_format (q_gemm.cu:202)."""
    if bits not in (2, 3, 4):
        raise RuntimeError(f"gptq_repack: {bits}-bit weights (8 bits: gptq8_repack)")
    zp_mode = _lib.ZP_CONST if qzeros is None else (_lib.ZP_NSEQ if use_v2_format else _lib.ZP_GPTQ_V1)
    if bits != 4:
        qweight = gptq_expand_bits(qweight, bits, add=(8 - (1 << (bits - 1))) if qzeros is None else 0, axis=0)
        if qzeros is not None:
            qzeros = gptq_expand_bits(qzeros, bits, add=0, axis=1)
    return w4a16_repack(qweight, scales, qzeros, group_size, _lib.W_GPTQ, zp_mode, tile=tile)


def gptq_gemm(a, b_q_weight: PackedW4, b_gptq_qzeros, b_gptq_scales, b_g_idx, use_exllama: bool,
              use_v2_format: bool, bit: int) -> torch.Tensor:
    """Signature of ops.gptq_gemm (vllm/_custom_ops.py:558-577).  b_q_weight is the
    repacked weight (the reference passes the gptq_shuffle'd tensor here); zeros and scales
    travel inside it; g_idx must be empty here -- checkpoints with activation reordering are served
    one level up (MI355XW4A16LinearKernel sorts the rows at load time and gathers the inputs)."""
    if bit not in (2, 3, 4):   # 2 / 3 bits: b_q_weight comes from gptq_repack(bits=...), widened to the 4-bit layout
        raise RuntimeError(f"gptq_gemm: {bit}-bit weights go through wi8a16_gemm (8 bits) or are not supported")
    if b_g_idx is not None and b_g_idx.numel() > 0:
        raise RuntimeError("gptq_gemm: pass activation-reordered layers through MI355XW4A16LinearKernel (g_idx must be empty here)")
    return w4a16_gemm(a, b_q_weight)


def awq_gemm(input, qweight: PackedW4, scales, qzeros, split_k_iters: int) -> torch.Tensor:
    """Signature of ops.awq_gemm (vllm/_custom_ops.py:525-536); split_k_iters is accepted
    and ignored (the kernel chooses its own deterministic split)."""
    return w4a16_gemm(input, qweight)


def awq_dequantize(qweight: PackedW4, scales, zeros, split_k_iters: int, thx: int, thy: int) -> torch.Tensor:
    """Signature of ops.awq_dequantize (vllm/_custom_ops.py:491-505)."""
    return w4a16_dequantize(qweight)


# ---------------------------------------------------------------------------
# mixture of experts (4-bit experts)
# ---------------------------------------------------------------------------
class PackedMoeW4:
    """Kernel-layout 4-bit weights of all experts of one projection (output of `moe_w4a16_repack`)."""

    __slots__ = ("packed", "scales", "zc", "E", "K", "N", "G", "tile", "pair_halves")

    def __init__(self, packed, scales, zc, E, K, N, G, tile, pair_halves=False):
        self.packed, self.scales, self.zc, self.E, self.K, self.N, self.G, self.tile = packed, scales, zc, E, K, N, G, tile
        self.pair_halves = pair_halves

    def expert(self, e: int) -> PackedW4:
        return PackedW4(self.packed[e], self.scales[e], None if self.zc is None else self.zc[e], self.K, self.N,
                        self.G, self.tile, self.pair_halves)


def moe_w4a16_repack(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor | None,
                     group_size: int, pair_halves: bool = False, tile: int = 0, decode_slots: int = 1) -> PackedMoeW4:
    """MoE WNA16 checkpoint-side layout (moe_wna16.py:204-330 after its loader conversions,
    :516-638) -> kernel layout.  qweight uint8 [E, N, K/2], two 4-bit values per byte along K with
    the even k in the low nibble; scales [E, N, K/G]; qzeros uint8 [E, N/2, K/G] (even n low) or
    None for symmetric uint4b8.  pair_halves (the merged w13 projection): gate_i next to up_i, for
    the SiluAndMul epilogue -- granted only where the layer shape supports it (see the result's
    `pair_halves`).  decode_slots: the layer's top-k, for the choice of the strip width."""
    if qweight.dtype != torch.uint8 or qweight.dim() != 3:
        raise RuntimeError("moe qweight must be uint8 [E, N, K/2]")
    E, N, K2 = qweight.shape
    K = 2 * K2
    G = K if group_size in (-1, 0) else group_size
    if scales.shape != (E, N, K // G):
        raise RuntimeError(f"moe scales shape {tuple(scales.shape)} != ({E}, {N}, {K // G})")
    if qzeros is not None and (qzeros.dtype != torch.uint8 or qzeros.shape != (E, N // 2, K // G)):
        raise RuntimeError(f"moe qzeros must be uint8 [{E}, {N // 2}, {K // G}]")
    if tile == 0:    # decode_slots: the (token, expert) slots of a decode launch at batch 1, i.e. the layer's top-k
        tile = _lib.lib().mi355q_moe_w4a16_choose_strip(K, N, G, decode_slots)
    pair_halves = bool(pair_halves) and w4a16_gate_up_silu_supported(K, N, G)
    packed = torch.empty((E, K * N // 8), dtype=torch.int32, device=qweight.device)
    sc = torch.empty((E, N // tile, K // G, tile), dtype=scales.dtype, device=qweight.device)
    zc = None if qzeros is None else torch.empty((E, N // tile, K // G, tile), dtype=scales.dtype, device=qweight.device)
    for e in range(E):
        # four bytes = eight consecutive k, least significant nibble first: the GPTQ word, transposed
        qw = qweight[e].contiguous().view(torch.int32).t().contiguous()              # [K/8, N]
        qz = None
        if qzeros is not None:
            # [K/G, N/8], sequential nibbles (reshape gives canonical strides even when K/G == 1)
            qz = qzeros[e].t().contiguous().reshape(-1).view(K // G, N // 2).view(torch.int32)
        w = w4a16_repack(qw, scales[e].t().contiguous(), qz, G, _lib.W_GPTQ,
                         _lib.ZP_CONST if qz is None else _lib.ZP_NSEQ, tile, pair_halves)
        packed[e].copy_(w.packed)
        sc[e].copy_(w.scales)
        if zc is not None:
            zc[e].copy_(w.zc.view(N // tile, K // G, tile))
    return PackedMoeW4(packed, sc, zc, E, K, N, G, tile, pair_halves)


def topk_softmax(topk_weights: torch.Tensor, topk_ids: torch.Tensor, token_expert_indices: torch.Tensor | None,
                 gating_output: torch.Tensor, renormalize: bool = False) -> None:
    """ops.topk_softmax (vllm/_custom_ops.py:2385): fills topk_weights (fp32) and topk_ids (int32)
    in place; token_expert_indices is accepted and left untouched."""
    _need_cuda(topk_weights, topk_ids, gating_output)
    T, E = gating_output.shape
    if topk_weights.dtype != torch.float32 or topk_ids.dtype != torch.int32 or topk_weights.shape != topk_ids.shape:
        raise RuntimeError("topk_softmax: topk_weights must be fp32 and topk_ids int32, both [tokens, topk]")
    if not gating_output.is_contiguous() or not topk_weights.is_contiguous() or not topk_ids.is_contiguous():
        raise RuntimeError("topk_softmax: tensors must be contiguous")
    check(_lib.lib().mi355q_topk_softmax(topk_weights.data_ptr(), topk_ids.data_ptr(), gating_output.data_ptr(), T, E,
                                         topk_weights.shape[1], int(renormalize), _dt(gating_output, True), _stream()))


def moe_router_topk_supported(num_experts: int, hidden: int) -> bool:
    return bool(_lib.lib().mi355q_moe_router_topk_supported(num_experts, hidden))


def moe_router_topk(x: torch.Tensor, residual: torch.Tensor, norm_weight: torch.Tensor | None, epsilon: float,
                    router_weight: torch.Tensor, topk: int, renormalize: bool, return_logits: bool = False):
    """fused_add_rms_norm(x, residual, norm_weight) in place, logits = x @ router_weight.T, then topk_softmax of
    the logits, as one launch (mixtral.py:284,144-146 -> fused_topk).  Returns (fp32 weights [T, topk], int32 ids
    [T, topk]) and, on request, the logits [T, E]; every result holds the bits the three ops give one after the
    other."""
    _need_cuda(x, residual, router_weight)
    if x.dim() != 2 or x.stride(1) != 1 or residual.shape != x.shape or not residual.is_contiguous():
        raise RuntimeError("moe_router_topk: x must be [tokens, hidden] with unit inner stride, residual contiguous of the same shape")
    E, H = router_weight.shape
    if H != x.shape[1] or not router_weight.is_contiguous() or router_weight.dtype != x.dtype or residual.dtype != x.dtype:
        raise RuntimeError("moe_router_topk: router_weight must be a contiguous [experts, hidden] tensor of the activation dtype")
    T = x.shape[0]
    w = torch.empty((T, topk), dtype=torch.float32, device=x.device)
    ids = torch.empty((T, topk), dtype=torch.int32, device=x.device)
    logits = torch.empty((T, E), dtype=x.dtype, device=x.device) if return_logits else None
    check(_lib.lib().mi355q_moe_router_topk(x.data_ptr(), x.stride(0), residual.data_ptr(), _ptr(norm_weight), epsilon,
                                            router_weight.data_ptr(), _ptr(logits), w.data_ptr(), ids.data_ptr(), T, E, H,
                                            topk, int(renormalize), _dt(x), _stream()))
    return (w, ids, logits) if return_logits else (w, ids)


def moe_w4a16_gemm(x: torch.Tensor, w: PackedMoeW4, expert_ids: torch.Tensor, x_row_div: int,
                   fuse_silu_mul: bool = False, silu_pairs: bool = False) -> torch.Tensor:
    """Row r of the result = x[r // x_row_div] (or silu-mul of x[r]) @ dequant(expert expert_ids[r]).
    Stands in for ops.moe_wna16_gemm (vllm/_custom_ops.py:2307-2342) at decode sizes.  silu_pairs
    (experts repacked with pair_halves): SiluAndMul of the projection in the epilogue, [slots, N/2]."""
    _need_cuda(x, w.packed, expert_ids)
    if expert_ids.dtype != torch.int32 or not expert_ids.is_contiguous():
        raise RuntimeError("moe_w4a16_gemm: expert_ids must be a contiguous int32 tensor")
    if silu_pairs and (fuse_silu_mul or not w.pair_halves):
        raise RuntimeError("moe_w4a16_gemm: silu_pairs needs experts repacked with pair_halves and no prologue")
    slots = expert_ids.numel()
    Kx = w.K * (2 if fuse_silu_mul else 1)
    if x.dim() != 2 or x.shape[1] != Kx or x.stride(1) != 1 or x.shape[0] * x_row_div < slots:
        raise RuntimeError(f"moe_w4a16_gemm: input must be [>= {-(-slots // x_row_div)}, {Kx}]")
    out = torch.empty((slots, w.N // 2 if silu_pairs else w.N), dtype=x.dtype, device=x.device)
    mode = 2 if silu_pairs else int(fuse_silu_mul)
    check(_lib.lib().mi355q_moe_w4a16_gemm(x.data_ptr(), x.stride(0), x_row_div, w.packed.data_ptr(),
                                           w.scales.data_ptr(), _ptr(w.zc), expert_ids.data_ptr(), out.data_ptr(),
                                           slots, w.E, w.N, w.K, w.G, w.tile, _dt(x), mode, _stream()))
    return out


def moe_align_block_size(expert_ids: torch.Tensor, num_experts: int, block_size: int | None = None):
    """ops.moe_align_block_size (vllm/_custom_ops.py:2237-2255) for the grouped expert GEMM: the slots
    (`expert_ids` int32 [slots], negative = expert held elsewhere) sorted by expert, every expert's run padded
    to whole tiles.  Returns (sorted_ids [max_tiles * block], tile_expert [max_tiles], num_tiles [1],
    slot_pos [slots]); padding rows of sorted_ids hold `slots`."""
    _need_cuda(expert_ids)
    if expert_ids.dtype != torch.int32 or not expert_ids.is_contiguous():
        raise RuntimeError("moe_align_block_size: expert_ids must be a contiguous int32 tensor")
    L = _lib.lib()
    block = block_size or L.mi355q_moe_grouped_block_rows()
    slots = expert_ids.numel()
    max_tiles = max(1, (slots + min(slots, num_experts) * (block - 1) + block - 1) // block)
    dev = expert_ids.device
    sorted_ids = torch.empty(max_tiles * block, dtype=torch.int32, device=dev)
    tile_expert = torch.empty(max_tiles, dtype=torch.int32, device=dev)
    num_tiles = torch.empty(1, dtype=torch.int32, device=dev)
    slot_pos = torch.empty(slots, dtype=torch.int32, device=dev)
    check(L.mi355q_moe_align_block_size(expert_ids.data_ptr(), slots, num_experts, block, sorted_ids.data_ptr(),
                                        tile_expert.data_ptr(), num_tiles.data_ptr(), slot_pos.data_ptr(), max_tiles,
                                        _stream()))
    return sorted_ids, tile_expert, num_tiles, slot_pos


def moe_grouped_mm_supported(w: PackedMoeW4) -> bool:
    return _mm_supported(w.expert(0))


def moe_w4a16_grouped_mm(x_sorted: torch.Tensor, w: PackedMoeW4, tile_expert: torch.Tensor, num_tiles: torch.Tensor,
                         silu_pairs: bool = False) -> torch.Tensor:
    """Row r of the result = x_sorted[r] @ dequant(expert tile_expert[r // block]) for the rows of the first
    num_tiles tiles (the rest is left uninitialised).  Stands in for ops.moe_wna16_gemm
    (vllm/_custom_ops.py:2307-2342) at prompt sizes, on rows sorted by `moe_align_block_size`."""
    _need_cuda(x_sorted, w.packed, tile_expert, num_tiles)
    if silu_pairs and not w.pair_halves:
        raise RuntimeError("moe_w4a16_grouped_mm: silu_pairs needs experts repacked with pair_halves")
    rows = x_sorted.shape[0]
    if x_sorted.dim() != 2 or x_sorted.shape[1] != w.K or x_sorted.stride(1) != 1:
        raise RuntimeError(f"moe_w4a16_grouped_mm: input must be [rows, {w.K}]")
    L = _lib.lib()
    if tile_expert.numel() * L.mi355q_moe_grouped_block_rows() != rows:
        raise RuntimeError("moe_w4a16_grouped_mm: one tile_expert entry per row tile is expected")
    out = torch.empty((rows, w.N // 2 if silu_pairs else w.N), dtype=x_sorted.dtype, device=x_sorted.device)
    need = L.mi355q_moe_w4a16_grouped_mm_workspace_bytes(rows, w.N, w.K, w.G, w.tile)
    ws = _workspace(x_sorted.device, need + COUNTER_BYTES) if need else None
    check(L.mi355q_moe_w4a16_grouped_mm(x_sorted.data_ptr(), x_sorted.stride(0), w.packed.data_ptr(), w.scales.data_ptr(),
                                        _ptr(w.zc), tile_expert.data_ptr(), num_tiles.data_ptr(), out.data_ptr(), rows,
                                        w.E, w.N, w.K, w.G, w.tile, _dt(x_sorted), int(silu_pairs),
                                        ws.data_ptr() + COUNTER_BYTES if ws is not None else None,
                                        ws.numel() - COUNTER_BYTES if ws is not None else 0, _stream()))
    return out


def moe_sum(input: torch.Tensor, output: torch.Tensor, topk_weights: torch.Tensor | None = None) -> None:
    """ops.moe_sum(input [T, topk, H], output [T, H]) (vllm/_custom_ops.py:2228); with
    topk_weights (fp32 [T, topk]) the rows are weighted in fp32 before the sum."""
    _need_cuda(input, output)
    T, topk, H = input.shape
    if not input.is_contiguous() or not output.is_contiguous():
        raise RuntimeError("moe_sum: tensors must be contiguous")
    if topk_weights is not None and (topk_weights.dtype != torch.float32 or not topk_weights.is_contiguous()):
        raise RuntimeError("moe_sum: topk_weights must be contiguous fp32")
    check(_lib.lib().mi355q_moe_sum(output.data_ptr(), input.data_ptr(), _ptr(topk_weights), T, topk, H, _dt(input),
                                    _stream()))


# ---------------------------------------------------------------------------
# fp8 weight-only
# ---------------------------------------------------------------------------
def _w8_blocks(weight: torch.Tensor, scales: torch.Tensor, block: tuple[int, int] | None):
    N, K = weight.shape
    if weight.element_size() != 1:
        raise RuntimeError("fp8 weight must be a 1-byte tensor [N, K]")
    if scales.dtype != torch.float32 or not scales.is_contiguous():
        raise RuntimeError("fp8 weight scales must be a contiguous fp32 tensor")
    if block is None:
        if scales.numel() != N:
            raise RuntimeError(f"per-channel scales must have {N} elements, got {scales.numel()}")
        return 1, K
    bn, bk = block
    if tuple(scales.shape) != ((N + bn - 1) // bn, (K + bk - 1) // bk):
        raise RuntimeError(f"block scales shape {tuple(scales.shape)} != "
                           f"({(N + bn - 1) // bn}, {(K + bk - 1) // bk}) for blocks {block}")
    return bn, bk


def w8a16_dequantize(weight: torch.Tensor, scales: torch.Tensor, block: tuple[int, int] | None,
                     dtype: torch.dtype) -> torch.Tensor:
    """[N, K] 16-bit weight = round(fp8 * scale) (the tensor the reference's fallback builds,
    vllm/model_executor/layers/quantization/fp8.py:433-454)."""
    _need_cuda(weight, scales)
    bn, bk = _w8_blocks(weight, scales, block)
    N, K = weight.shape
    out = torch.empty((N, K), dtype=dtype, device=weight.device)
    check(_lib.lib().mi355q_w8a16_dequantize(weight.data_ptr(), scales.data_ptr(), out.data_ptr(), N, K, bn, bk,
                                             _DT[dtype], _stream()))
    return out


def w8a16_gemm(x: torch.Tensor, weight: torch.Tensor, scales: torch.Tensor, block: tuple[int, int] | None = None,
               bias: torch.Tensor | None = None) -> torch.Tensor:
    """x[..., K] @ (fp8 weight[N, K] * scales)^T (+ bias) with 16-bit activations.  `scales` is
    fp32 [N] (per output channel) or, with block=(block_n, block_k), [ceil(N/bn), ceil(K/bk)]."""
    _need_cuda(x, weight, scales)
    bn, bk = _w8_blocks(weight, scales, block)
    N, K = weight.shape
    if x.shape[-1] != K:
        raise RuntimeError(f"input feature size {x.shape[-1]} != {K}")
    x2 = x.reshape(-1, K)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    M = x2.shape[0]
    if M > DEQUANT_MATMUL_MIN_TOKENS and not torch.cuda.is_current_stream_capturing():
        out = torch.nn.functional.linear(x2, w8a16_dequantize(weight, scales, block, x.dtype), bias)
        return out.reshape(x.shape[:-1] + (N,))
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    check(_lib.lib().mi355q_w8a16_gemm(x2.data_ptr(), x2.stride(0), weight.data_ptr(), scales.data_ptr(), _ptr(bias),
                                       out.data_ptr(), M, N, K, bn, bk, _dt(x), _stream()))
    return out.reshape(x.shape[:-1] + (N,))


# ---------------------------------------------------------------------------
# fp8 activations x fp8 weights (W8A8)
# ---------------------------------------------------------------------------
FP8_DTYPE = torch.float8_e4m3fn   # gfx950 is OCP fp8 (vllm/platforms/rocm.py: fp8_dtype())
# From this many rows on, a product with per-tensor scales is a plain library GEMM and goes to hipBLASLt
# through torch._scaled_mm -- the call the reference makes above four rows
# (kernels/linear/scaled_mm/rocm.py:42-51); measured 1.6-2.7 x the tiled kernel of this library there.
FP8_MM_BLAS_MIN_TOKENS = int(os.environ.get("MI355Q_FP8_MM_BLAS_MIN_TOKENS", "128"))


def _rows2d(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.ndim != 2 or t.stride(-1) != 1:
        raise RuntimeError(f"{what} must be 2-D with a contiguous last dimension")
    return t


def scaled_fp8_quant(input: torch.Tensor, scale: torch.Tensor | None = None, num_token_padding: int | None = None,
                     scale_ub: torch.Tensor | None = None, use_per_token_if_dynamic: bool = False,
                     output: torch.Tensor | None = None,
                     group_shape: tuple[int, int] | None = None) -> tuple[torch.Tensor, torch.Tensor]:
    """ops.scaled_fp8_quant (vllm/_custom_ops.py:1832-1898): static quantisation with the given scale
    (per tensor, 1-D per token / per channel with `group_shape`, or 2-D groups), else dynamic per tensor
    or per token (optionally capped by `scale_ub`).  Returns (fp8 tensor, scale)."""
    _need_cuda(input, scale, scale_ub, output)
    x = _rows2d(input, "scaled_fp8_quant input")
    T, H = x.shape
    shape = (max(num_token_padding, T), H) if num_token_padding else (T, H)
    if output is None:
        output = torch.empty(shape, device=x.device, dtype=FP8_DTYPE)
    elif num_token_padding is not None or output.dtype != FP8_DTYPE:
        raise RuntimeError("scaled_fp8_quant: a given output must be float8_e4m3fn and excludes padding")
    _rows2d(output, "scaled_fp8_quant output")
    L, dt, st = _lib.lib(), _dt(x, allow_fp32=True), _stream()
    if scale is None:
        if use_per_token_if_dynamic:
            scale = torch.empty((shape[0], 1), device=x.device, dtype=torch.float32)
            check(L.mi355q_dynamic_per_token_scaled_fp8_quant(output.data_ptr(), output.stride(0), x.data_ptr(),
                                                              x.stride(0), scale.data_ptr(), _ptr(scale_ub), T, H, dt, st))
        else:
            scale = torch.empty(1, device=x.device, dtype=torch.float32)
            check(L.mi355q_dynamic_scaled_fp8_quant(output.data_ptr(), output.stride(0), x.data_ptr(), x.stride(0),
                                                    scale.data_ptr(), T, H, dt, st))
        return output, scale
    if scale.dtype != torch.float32:
        raise RuntimeError("scaled_fp8_quant: scale must be fp32")
    if scale.ndim == 0 or scale.numel() == 1:
        gm, gn, si, sj = T, H, 0, 0
    elif scale.ndim == 1:
        if group_shape is None or len(group_shape) != 2:
            raise RuntimeError("1D scale requires explicit group_shape to disambiguate per-channel vs per-token "
                               "quantization. Use group_shape=(-1, 1) for per-channel or group_shape=(1, -1) for "
                               "per-token.")
        gm = T if group_shape[0] == -1 else group_shape[0]
        gn = H if group_shape[1] == -1 else group_shape[1]
        if T % gm or H % gn or scale.numel() != (T // gm) * (H // gn):
            raise RuntimeError(f"1D scale length ({scale.numel()}) does not match group_shape {tuple(group_shape)} "
                               f"with input shape ({T}, {H})")
        if T // gm == 1:
            si, sj = 0, scale.stride(0)
        elif H // gn == 1:
            si, sj = scale.stride(0), 0
        else:
            raise RuntimeError("1D scale can only be used when one of the scale dimensions is 1. For 2D group "
                               "scaling, use a 2D scale tensor.")
    elif scale.ndim == 2:
        if T % scale.shape[0] or H % scale.shape[1]:
            raise RuntimeError(f"input shape ({T}, {H}) must be divisible by the scale shape {tuple(scale.shape)}")
        gm, gn = T // scale.shape[0], H // scale.shape[1]
        if group_shape is not None:
            egm = T if group_shape[0] == -1 else group_shape[0]
            egn = H if group_shape[1] == -1 else group_shape[1]
            if (egm, egn) != (gm, gn):
                raise RuntimeError(f"Explicit group_shape {tuple(group_shape)} does not match inferred group shape "
                                   f"({gm}, {gn}) from 2D scale tensor shape {tuple(scale.shape)}")
        si, sj = scale.stride(0), scale.stride(1)
    else:
        raise RuntimeError(f"scale must be 0D, 1D, or 2D tensor, but got {scale.ndim}D")
    check(L.mi355q_static_scaled_fp8_quant(output.data_ptr(), output.stride(0), x.data_ptr(), x.stride(0),
                                           scale.data_ptr(), T, H, gm, gn, si, sj, dt, st))
    return output, scale


def rms_norm_static_fp8_quant(out: torch.Tensor, input: torch.Tensor, weight: torch.Tensor, scale: torch.Tensor,
                              epsilon: float) -> None:
    """torch.ops._C.rms_norm_static_fp8_quant (csrc/libtorch_stable/layernorm_quant_kernels.cu:206-246)."""
    _need_cuda(out, input, weight, scale)
    H = input.shape[-1]
    x = input.reshape(-1, H)
    if not out.is_contiguous() or x.stride(-1) != 1:
        raise RuntimeError("rms_norm_static_fp8_quant: out must be contiguous, input contiguous in its last dimension")
    check(_lib.lib().mi355q_rms_norm_static_fp8_quant(out.data_ptr(), x.data_ptr(), x.stride(0), weight.data_ptr(),
                                                      scale.data_ptr(), float(epsilon), x.shape[0], H, _dt(x), _stream()))


def fused_add_rms_norm_static_fp8_quant(out: torch.Tensor, input: torch.Tensor, residual: torch.Tensor,
                                        weight: torch.Tensor, scale: torch.Tensor, epsilon: float) -> None:
    """torch.ops._C.fused_add_rms_norm_static_fp8_quant (layernorm_quant_kernels.cu:264-309): residual is
    updated in place to input + residual."""
    _need_cuda(out, input, residual, weight, scale)
    H = input.shape[-1]
    x = input.reshape(-1, H)
    if not out.is_contiguous() or not residual.is_contiguous() or x.stride(-1) != 1:
        raise RuntimeError("fused_add_rms_norm_static_fp8_quant: out and residual must be contiguous")
    if residual.dtype != input.dtype or weight.dtype != input.dtype:
        raise RuntimeError("fused_add_rms_norm_static_fp8_quant: residual and weight must have the input's dtype")
    check(_lib.lib().mi355q_fused_add_rms_norm_static_fp8_quant(out.data_ptr(), x.data_ptr(), x.stride(0),
                                                                residual.data_ptr(), weight.data_ptr(), scale.data_ptr(),
                                                                float(epsilon), x.shape[0], H, _dt(x), _stream()))


def rms_norm_dynamic_per_token_quant(input: torch.Tensor, weight: torch.Tensor, epsilon: float,
                                     quant_dtype: torch.dtype = FP8_DTYPE, scale_ub: torch.Tensor | None = None,
                                     residual: torch.Tensor | None = None) -> tuple[torch.Tensor, torch.Tensor]:
    """ops.rms_norm_dynamic_per_token_quant (vllm/_custom_ops.py:361-…) for fp8 outputs: returns
    (fp8 tensor, scales [T, 1]); residual, when given, is updated in place."""
    _need_cuda(input, weight, scale_ub, residual)
    if quant_dtype != FP8_DTYPE:
        raise RuntimeError("rms_norm_dynamic_per_token_quant: only float8_e4m3fn outputs are built")
    H = input.shape[-1]
    x = input.reshape(-1, H)
    if x.stride(-1) != 1 or (residual is not None and (not residual.is_contiguous() or residual.dtype != input.dtype)):
        raise RuntimeError("rms_norm_dynamic_per_token_quant: input must be contiguous in its last dimension and "
                           "the residual contiguous, of the same dtype")
    out = torch.empty(input.shape, dtype=FP8_DTYPE, device=input.device)
    scales = torch.empty((x.shape[0], 1), dtype=torch.float32, device=input.device)
    check(_lib.lib().mi355q_rms_norm_dynamic_per_token_quant(out.data_ptr(), scales.data_ptr(), x.data_ptr(), x.stride(0),
                                                             weight.data_ptr(), _ptr(scale_ub), _ptr(residual),
                                                             float(epsilon), x.shape[0], H, _dt(x), _stream()))
    return out, scales


def silu_and_mul_quant(out: torch.Tensor, input: torch.Tensor, scale: torch.Tensor) -> None:
    """torch.ops._C.silu_and_mul_quant (csrc/libtorch_stable/quantization/activation_kernels.cu:584-…)."""
    _need_cuda(out, input, scale)
    d = input.shape[-1] // 2
    if not input.is_contiguous() or not out.is_contiguous() or out.shape[-1] != d or out.dtype != FP8_DTYPE:
        raise RuntimeError("silu_and_mul_quant: input [..., 2d] and out [..., d] (float8_e4m3fn) must be contiguous")
    check(_lib.lib().mi355q_silu_and_mul_quant(out.data_ptr(), input.data_ptr(), scale.data_ptr(), input.numel() // (2 * d),
                                               d, _dt(input), _stream()))


def fp8_scaled_mm(a: torch.Tensor, weight: torch.Tensor, out_dtype: torch.dtype, scale_a: torch.Tensor,
                  scale_b: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """(a[M, K] @ weight[N, K]^T) * scale_a * scale_b + bias with fp8 operands and fp32 accumulation:
    torch._scaled_mm(a, weight.t(), ...) / ops.wvSplitKQ(weight, a, ...) of
    vllm/model_executor/kernels/linear/scaled_mm/rocm.py:19-52.  scale_a: 1 or M elements, scale_b: 1 or N."""
    _need_cuda(a, weight, scale_a, scale_b, bias)
    if a.dtype != FP8_DTYPE or weight.dtype != FP8_DTYPE:
        raise RuntimeError("fp8_scaled_mm: operands must be float8_e4m3fn")
    _rows2d(a, "fp8_scaled_mm a")
    _rows2d(weight, "fp8_scaled_mm weight")
    (M, K), N = a.shape, weight.shape[0]
    if weight.shape[1] != K:
        raise RuntimeError(f"fp8_scaled_mm: a has {K} columns, weight {weight.shape[1]}")
    if scale_a.dtype != torch.float32 or scale_b.dtype != torch.float32:
        raise RuntimeError("fp8_scaled_mm: scales must be fp32")
    if bias is not None and (bias.dtype != out_dtype or bias.numel() != N):
        raise RuntimeError("fp8_scaled_mm: bias must have the output dtype and N elements")
    sa, sb = scale_a.reshape(-1), scale_b.reshape(-1)
    if not sa.is_contiguous() or not sb.is_contiguous():
        sa, sb = sa.contiguous(), sb.contiguous()
    if M >= FP8_MM_BLAS_MIN_TOKENS and sa.numel() == 1 and sb.numel() == 1 and weight.is_contiguous() and a.is_contiguous():
        return torch._scaled_mm(a, weight.t(), out_dtype=out_dtype, scale_a=sa.reshape(()), scale_b=sb.reshape(()),
                                bias=bias)
    out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    check(_lib.lib().mi355q_fp8_scaled_mm(a.data_ptr(), a.stride(0), weight.data_ptr(), weight.stride(0), sa.data_ptr(),
                                          sa.numel(), sb.data_ptr(), sb.numel(), _ptr(bias), out.data_ptr(), N, M, N, K,
                                          _DT[out_dtype], _stream()))
    return out


def per_token_group_quant_fp8(x: torch.Tensor, group_size: int, eps: float = 1e-10, dtype: torch.dtype | None = None,
                              column_major_scales: bool = False, out_q: torch.Tensor | None = None,
                              use_ue8m0: bool = False) -> tuple[torch.Tensor, torch.Tensor]:
    """per_token_group_quant_fp8 (vllm/model_executor/layers/quantization/utils/fp8_utils.py:533-659): one fp32
    scale per (token, group of `group_size` elements), scale = max(eps, max |x|) / 448, x_q = fp8(clamp(x / scale)).
    Returns (x_q like x, scales x.shape[:-1] + (H / group_size,)); with column_major_scales the scale tensor has
    that shape over column-major storage (stride 1 along the tokens), as the reference returns it."""
    _need_cuda(x, out_q)
    dtype = FP8_DTYPE if dtype is None else dtype
    if dtype != FP8_DTYPE:
        raise RuntimeError("per_token_group_quant_fp8: only float8_e4m3fn outputs are built")
    if use_ue8m0:
        raise RuntimeError("per_token_group_quant_fp8: power-of-two (ue8m0) scales are a DeepGEMM option, not built")
    H = x.shape[-1]
    if H % group_size:
        raise RuntimeError(f"the last dimension of `x` {H} must be divisible by `group_size` {group_size}")
    if x.stride(-1) != 1:
        raise RuntimeError("`x` groups must be contiguous")
    x2 = x.reshape(-1, H)
    T, G = x2.shape[0], H // group_size
    if out_q is not None and (out_q.shape != x.shape or out_q.dtype != FP8_DTYPE or not out_q.is_contiguous()):
        raise RuntimeError("per_token_group_quant_fp8: out_q must be a contiguous float8_e4m3fn tensor of x's shape")
    x_q = out_q if out_q is not None else torch.empty(x.shape, dtype=FP8_DTYPE, device=x.device)
    if column_major_scales:
        if x.dim() != 2:
            raise RuntimeError("per_token_group_quant_fp8: column-major scales are built for 2-D inputs")
        x_s = torch.empty((G, T), dtype=torch.float32, device=x.device).permute(-1, -2)
        st, sg = 1, T
    else:
        x_s = torch.empty(x.shape[:-1] + (G,), dtype=torch.float32, device=x.device)
        st, sg = G, 1
    check(_lib.lib().mi355q_per_token_group_fp8_quant(x_q.data_ptr(), H, x2.data_ptr(), x2.stride(0), x_s.data_ptr(), st, sg,
                                                      T, H, group_size, float(eps), _dt(x2, True), _stream()))
    return x_q, x_s


def w8a8_block_scaled_mm(A: torch.Tensor, B: torch.Tensor, As: torch.Tensor, Bs: torch.Tensor, block_size: list[int],
                         output_dtype: torch.dtype = torch.bfloat16, bias: torch.Tensor | None = None) -> torch.Tensor:
    """w8a8_triton_block_scaled_mm (fp8_utils.py:859-990): A fp8 [..., K] with per-token-group scales As
    [..., ceil(K / block_k)], B fp8 [N, K] with per-block scales Bs [ceil(N / block_n), ceil(K / block_k)];
    returns (sum over k blocks of (A_b . B_b^T) * As_b * Bs_b) in output_dtype, shape [..., N]."""
    _need_cuda(A, B, As, Bs, bias)
    if len(block_size) != 2:
        raise RuntimeError("w8a8_block_scaled_mm: block_size must be [block_n, block_k]")
    block_n, block_k = int(block_size[0]), int(block_size[1])
    if A.dtype != FP8_DTYPE or B.dtype != FP8_DTYPE:
        raise RuntimeError("w8a8_block_scaled_mm: operands must be float8_e4m3fn")
    if As.dtype != torch.float32 or Bs.dtype != torch.float32:
        raise RuntimeError("w8a8_block_scaled_mm: scales must be fp32")
    K = A.shape[-1]
    if B.dim() != 2 or Bs.dim() != 2 or B.shape[1] != K:
        raise RuntimeError("w8a8_block_scaled_mm: B must be [N, K] and Bs 2-D")
    N = B.shape[0]
    kt, nt = -(-K // block_k), -(-N // block_n)
    if As.shape[:-1] != A.shape[:-1] or As.shape[-1] != kt or tuple(Bs.shape) != (nt, kt):
        raise RuntimeError(f"w8a8_block_scaled_mm: scale shapes {tuple(As.shape)} / {tuple(Bs.shape)} do not match "
                           f"A {tuple(A.shape)}, B {tuple(B.shape)} and blocks {block_size}")
    if output_dtype not in _DT or (bias is not None and (bias.dtype != output_dtype or bias.numel() != N)):
        raise RuntimeError("w8a8_block_scaled_mm: output must be bf16 or fp16, the bias of that dtype with N elements")
    A2 = A.reshape(-1, K)
    As2 = As.reshape(-1, kt)
    if K % 16 or A2.stride(0) % 16 or B.stride(0) % 16:
        # rows that do not start on 16-byte boundaries (the reference test's K = 3884; no checkpoint has such a
        # layer): zero-padded copies up to the next multiple of 16 -- zeros add nothing to a block's dot product
        # and the padding never opens a new scale block
        pad = (-K) % 16
        A2 = torch.nn.functional.pad(A2.view(torch.uint8), (0, pad)).view(FP8_DTYPE)
        B = torch.nn.functional.pad(B.view(torch.uint8), (0, pad)).view(FP8_DTYPE)
    _rows2d(A2, "w8a8_block_scaled_mm A")
    _rows2d(B, "w8a8_block_scaled_mm B")
    M = A2.shape[0]
    out = torch.empty((M, N), dtype=output_dtype, device=A.device)
    check(_lib.lib().mi355q_fp8_block_scaled_mm(A2.data_ptr(), A2.stride(0), As2.data_ptr(), As2.stride(0), As2.stride(1),
                                                B.data_ptr(), B.stride(0), Bs.data_ptr(), Bs.stride(0), Bs.stride(1),
                                                block_n, block_k, _ptr(bias), out.data_ptr(), N, M, N, A2.shape[1],
                                                _DT[output_dtype], _stream()))
    return out.reshape(A.shape[:-1] + (N,))


def wvSplitKQ(a: torch.Tensor, b: torch.Tensor, out_dtype: torch.dtype, scale_a: torch.Tensor, scale_b: torch.Tensor,
              cu_count: int, bias: torch.Tensor | None = None) -> torch.Tensor:
    """ops.wvSplitKQ (vllm/_custom_ops.py:2213-…): `a` is the weight [N, K], `b` the activations [M, K],
    scale_a the activations' scale, scale_b the weight's; cu_count is not needed here."""
    return fp8_scaled_mm(b, a, out_dtype, scale_a, scale_b, bias)


# ---------------------------------------------------------------------------
# 8-bit integer weights (GPTQ 8 bits, symmetric)
# ---------------------------------------------------------------------------
def gptq8_repack(qweight: torch.Tensor, scales: torch.Tensor, group_size: int):
    """GPTQ 8-bit checkpoint tensors -> kernel operands.  qweight int32 [K/4, N] (four consecutive k
    per word, least significant byte first; auto_gptq.py:381-445 with pack_factor 4), scales
    [K/G, N] -> (uint8 [N, K] values still biased by 128, fp32 [N, K/G] scales)."""
    if qweight.dtype != torch.int32 or qweight.dim() != 2:
        raise RuntimeError("gptq8 qweight must be int32 [K/4, N]")
    K4, N = qweight.shape
    K = 4 * K4
    G = K if group_size in (-1, 0) else group_size
    if K % G or G % 16 or scales.shape != (K // G, N):
        raise RuntimeError(f"gptq8: scales shape {tuple(scales.shape)} / group size {G} do not fit K={K}, N={N}")
    w = qweight.contiguous().view(torch.uint8).view(K4, N, 4).permute(1, 0, 2).reshape(N, K).contiguous()
    return w, scales.t().to(torch.float32).contiguous()


def wi8a16_dequantize(weight: torch.Tensor, scales: torch.Tensor, group_size: int, dtype: torch.dtype) -> torch.Tensor:
    """[N, K] 16-bit weight = round((q - 128) * scale): the reference's reconstruct route for 8-bit
    GPTQ (q_gemm.cu:1509-1522)."""
    _need_cuda(weight, scales)
    N, K = weight.shape
    G = K if group_size in (-1, 0) else group_size
    if weight.dtype != torch.uint8 or scales.dtype != torch.float32 or scales.shape != (N, K // G):
        raise RuntimeError("wi8a16: weight must be uint8 [N, K] and scales fp32 [N, K/G]")
    out = torch.empty((N, K), dtype=dtype, device=weight.device)
    check(_lib.lib().mi355q_wi8a16_dequantize(weight.data_ptr(), scales.data_ptr(), out.data_ptr(), N, K, G,
                                              _DT[dtype], _stream()))
    return out


def wi8a16_gemm(x: torch.Tensor, weight: torch.Tensor, scales: torch.Tensor, group_size: int,
                bias: torch.Tensor | None = None) -> torch.Tensor:
    """x[..., K] @ ((weight[N, K] - 128) * scales)^T (+ bias): ops.gptq_gemm with bit = 8
    (vllm/_custom_ops.py:558-577).  Operands from `gptq8_repack`."""
    _need_cuda(x, weight, scales)
    N, K = weight.shape
    G = K if group_size in (-1, 0) else group_size
    if weight.dtype != torch.uint8 or scales.dtype != torch.float32 or scales.shape != (N, K // G) or \
            not scales.is_contiguous() or not weight.is_contiguous():
        raise RuntimeError("wi8a16: weight must be contiguous uint8 [N, K] and scales contiguous fp32 [N, K/G]")
    if x.shape[-1] != K:
        raise RuntimeError(f"input feature size {x.shape[-1]} != {K}")
    x2 = x.reshape(-1, K)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    M = x2.shape[0]
    if M > DEQUANT_MATMUL_MIN_TOKENS and not torch.cuda.is_current_stream_capturing():
        out = torch.nn.functional.linear(x2, wi8a16_dequantize(weight, scales, G, x.dtype), bias)
        return out.reshape(x.shape[:-1] + (N,))
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    check(_lib.lib().mi355q_wi8a16_gemm(x2.data_ptr(), x2.stride(0), weight.data_ptr(), scales.data_ptr(), _ptr(bias),
                                        out.data_ptr(), M, N, K, G, _dt(x), _stream()))
    return out.reshape(x.shape[:-1] + (N,))


def linear_skinny(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """x[..., K] @ weight[N, K]^T (+ bias) for small token counts.  Stands in for
    ops.wvSplitK / ops.LLMM1 (vllm/_custom_ops.py:2161-2213) as used by
    rocm_unquantized_gemm (vllm/model_executor/layers/utils.py:116-186)."""
    _need_cuda(x, weight)
    N, K = weight.shape
    x2 = x.reshape(-1, K)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    M = x2.shape[0]
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    check(_lib.lib().mi355q_linear_skinny(x2.data_ptr(), x2.stride(0), weight.data_ptr(), _ptr(bias),
                                          out.data_ptr(), M, N, K, _dt(x), _stream()))
    return out.reshape(x.shape[:-1] + (N,))


# ---------------------------------------------------------------------------
# norms / rope / activation
# ---------------------------------------------------------------------------
def rms_norm(out: torch.Tensor, input: torch.Tensor, weight: torch.Tensor | None, epsilon: float) -> None:
    """ops.rms_norm (vllm/_custom_ops.py:229-235)."""
    _need_cuda(out, input)
    H = input.shape[-1]
    x2 = input.reshape(-1, H) if input.is_contiguous() else input
    if x2.dim() != 2 or x2.stride(-1) != 1:
        raise RuntimeError("rms_norm: input must be [tokens, hidden] with a contiguous last dim")
    check(_lib.lib().mi355q_rms_norm(out.data_ptr(), x2.data_ptr(), x2.stride(0), _ptr(weight), epsilon,
                                     x2.shape[0], H, _dt(input), _stream()))


def fused_add_rms_norm(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor | None,
                       epsilon: float) -> None:
    """ops.fused_add_rms_norm (vllm/_custom_ops.py:269-276): in place on both tensors."""
    _need_cuda(input, residual)
    H = input.shape[-1]
    x2 = input.view(-1, H)
    if not residual.is_contiguous():
        raise RuntimeError("fused_add_rms_norm: residual must be contiguous")
    check(_lib.lib().mi355q_fused_add_rms_norm(x2.data_ptr(), x2.stride(0), residual.data_ptr(), _ptr(weight),
                                               epsilon, x2.shape[0], H, _dt(input), _stream()))


def rotary_embedding(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor | None, head_size: int,
                     cos_sin_cache: torch.Tensor, is_neox: bool) -> None:
    """ops.rotary_embedding (vllm/_custom_ops.py:201-225): in place on query and key."""
    _need_cuda(positions, query, cos_sin_cache)
    T = positions.numel()
    if positions.dtype != torch.int64:
        raise RuntimeError("rotary_embedding: positions must be int64")
    q2 = query.view(T, -1)
    nh = q2.shape[1] // head_size
    k2 = None if key is None else key.view(T, -1)
    nkv = 0 if k2 is None else k2.shape[1] // head_size
    check(_lib.lib().mi355q_rotary_embedding(
        positions.data_ptr(), q2.data_ptr(), q2.stride(0), _ptr(k2), 0 if k2 is None else k2.stride(0),
        head_size, nh, nkv, cos_sin_cache.data_ptr(), cos_sin_cache.shape[1], _dt(cos_sin_cache, True),
        int(is_neox), T, _dt(query), _stream()))


def silu_and_mul(out: torch.Tensor, input: torch.Tensor) -> None:
    """torch.ops._C.silu_and_mul(out, input) (csrc/libtorch_stable/activation_kernels.cu:98-155)."""
    _need_cuda(out, input)
    d = input.shape[-1] // 2
    if not input.is_contiguous() or not out.is_contiguous():
        raise RuntimeError("silu_and_mul: tensors must be contiguous")
    check(_lib.lib().mi355q_silu_and_mul(out.data_ptr(), input.data_ptr(), input.numel() // (2 * d), d,
                                         _dt(input), _stream()))


def silu_and_mul_pairs(out: torch.Tensor, input: torch.Tensor) -> None:
    """silu_and_mul on the pair-interleaved output (g0, u0, g1, u1, ...) of a gate_up projection repacked
    with pair_halves: out[..., i] = silu(input[..., 2 i]) * input[..., 2 i + 1] -- silu_and_mul(out,
    unpair_halves(input)) without the copy."""
    _need_cuda(out, input)
    d = input.shape[-1] // 2
    if not input.is_contiguous() or not out.is_contiguous() or out.shape[-1] != d:
        raise RuntimeError("silu_and_mul_pairs: tensors must be contiguous, out [.., d] for input [.., 2 d]")
    check(_lib.lib().mi355q_silu_and_mul_pairs(out.data_ptr(), input.data_ptr(), input.numel() // (2 * d), d,
                                               _dt(input), _stream()))


# ---------------------------------------------------------------------------
# KV cache + attention
# ---------------------------------------------------------------------------
def _is_fp8(kv_cache_dtype: str) -> bool:
    if kv_cache_dtype in ("auto", "bfloat16", "float16", "half"):
        return False
    if kv_cache_dtype in ("fp8", "fp8_e4m3"):
        return True
    raise RuntimeError(f"unsupported kv_cache_dtype {kv_cache_dtype!r}")


def reshape_and_cache(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype: str, k_scale, v_scale) -> None:
    """ops.reshape_and_cache (vllm/_custom_ops.py:2592-2611).  key/value [T, kv_heads, head];
    key_cache [blocks, kv_heads, head/x, block_size, x]; value_cache [blocks, kv_heads, head, block_size]."""
    _need_cuda(key, value, key_cache, value_cache, slot_mapping)
    T, kvh, head = key.shape
    bs = value_cache.shape[3]
    if slot_mapping.dtype != torch.int64:
        raise RuntimeError("reshape_and_cache: slot_mapping must be int64")
    check(_lib.lib().mi355q_reshape_and_cache(
        key.data_ptr(), value.data_ptr(), key.stride(0), value.stride(0), key_cache.data_ptr(),
        value_cache.data_ptr(), slot_mapping.data_ptr(), slot_mapping.numel(), kvh, head, bs, key_cache.stride(0),
        _dt(key), int(_is_fp8(kv_cache_dtype)), _ptr(k_scale), _ptr(v_scale), _stream()))


def reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype: str, k_scale, v_scale) -> None:
    """ops.reshape_and_cache_flash (vllm/_custom_ops.py:2614-2637).  key/value [T, heads, head]; both caches
    [blocks, block_size, heads, head], or the [blocks, heads, block_size, head] storage viewed that way (the
    strides tell); k_scale / v_scale hold one value or one per head."""
    _need_cuda(key, value, key_cache, value_cache, slot_mapping)
    T, heads, head = key.shape
    if key_cache.dim() != 4 or key_cache.shape[2] != heads or key_cache.shape[3] != head or key_cache.stride(3) != 1:
        raise RuntimeError("reshape_and_cache_flash: key_cache must be [blocks, block_size, heads, head] with a contiguous last dim")
    if key_cache.stride() != value_cache.stride() or key_cache.shape != value_cache.shape:
        raise RuntimeError("reshape_and_cache_flash: key_cache and value_cache must share shape and strides")
    if slot_mapping.dtype != torch.int64:
        raise RuntimeError("reshape_and_cache_flash: slot_mapping must be int64")
    fp8 = _is_fp8(kv_cache_dtype)
    per_head = 0
    if fp8:
        if k_scale is None or v_scale is None or k_scale.shape != v_scale.shape or k_scale.numel() not in (1, heads):
            raise RuntimeError("reshape_and_cache_flash: k_scale and v_scale must be of shape [1] or [num_heads]")
        per_head = int(k_scale.numel() > 1)
    check(_lib.lib().mi355q_reshape_and_cache_flash(
        key.data_ptr(), value.data_ptr(), key.stride(0), value.stride(0), key_cache.data_ptr(), value_cache.data_ptr(),
        slot_mapping.data_ptr(), slot_mapping.numel(), heads, head, key_cache.shape[1], key_cache.stride(0),
        key_cache.stride(1), key_cache.stride(2), _dt(key), int(fp8), _ptr(k_scale), _ptr(v_scale), per_head, _stream()))


def rope_and_cache(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                   head_size: int, cos_sin_cache: torch.Tensor, is_neox: bool, key_cache: torch.Tensor,
                   value_cache: torch.Tensor, slot_mapping: torch.Tensor, kv_cache_dtype: str, k_scale, v_scale) -> None:
    """rotary_embedding + reshape_and_cache in one launch (the arguments of
    rocm_aiter_ops.triton_rope_and_cache as called at vllm/v1/attention/backends/rocm_attn.py:548-562):
    query / key [T, heads * head] are rotated in place, the rotated key and value [T, kv_heads * head]
    go to the paged cache."""
    _need_cuda(positions, query, key, value, cos_sin_cache, key_cache, value_cache, slot_mapping)
    T = positions.numel()
    if positions.dtype != torch.int64 or slot_mapping.dtype != torch.int64:
        raise RuntimeError("rope_and_cache: positions and slot_mapping must be int64")
    if slot_mapping.numel() != T:
        raise RuntimeError(f"rope_and_cache: {slot_mapping.numel()} slots for {T} positions")
    q2, k2, v2 = query.view(T, -1), key.view(T, -1), value.view(T, -1)
    if q2.stride(1) != 1 or k2.stride(1) != 1 or v2.stride(1) != 1:
        raise RuntimeError("rope_and_cache: query / key / value rows must be contiguous")
    nh, nkv = q2.shape[1] // head_size, k2.shape[1] // head_size
    if v2.shape[1] != nkv * head_size or key_cache.shape[1] != nkv or value_cache.shape[1] != nkv:
        raise RuntimeError("rope_and_cache: key / value / cache head counts differ")
    check(_lib.lib().mi355q_rope_and_cache(
        positions.data_ptr(), q2.data_ptr(), q2.stride(0), k2.data_ptr(), k2.stride(0), v2.data_ptr(), v2.stride(0),
        head_size, nh, nkv, cos_sin_cache.data_ptr(), cos_sin_cache.shape[1], _dt(cos_sin_cache, True), int(is_neox),
        key_cache.data_ptr(), value_cache.data_ptr(), slot_mapping.data_ptr(), T, value_cache.shape[3],
        key_cache.stride(0), _dt(query), int(_is_fp8(kv_cache_dtype)), _ptr(k_scale), _ptr(v_scale), _stream()))


def convert_fp8(output: torch.Tensor, input: torch.Tensor, scale: float = 1.0, kv_dtype: str = "fp8") -> None:
    """ops.convert_fp8 (vllm/_custom_ops.py:2908): fp8 cache <-> 16-bit tensor."""
    _need_cuda(output, input)
    to_fp8 = output.element_size() == 1
    ref = input if to_fp8 else output
    check(_lib.lib().mi355q_convert_fp8(output.data_ptr(), input.data_ptr(), input.numel(), scale, _dt(ref),
                                        int(to_fp8), _stream()))


def paged_attention_rocm(out, exp_sum, max_logits, tmp_out, query, key_cache, value_cache, num_kv_heads: int,
                         scale: float, block_tables, seq_lens, query_start_loc, block_size: int, max_seq_len: int,
                         alibi_slopes, kv_cache_dtype: str, k_scale, v_scale, fp8_out_scale=None,
                         mfma_type: str = "f16") -> None:
    """ops.paged_attention_rocm (vllm/_custom_ops.py:120-163), decode rows only."""
    _need_cuda(out, query, key_cache, value_cache, block_tables, seq_lens)
    if fp8_out_scale is not None:
        raise RuntimeError("paged_attention_rocm: fp8 output is not supported")
    if block_tables.dtype != torch.int32 or seq_lens.dtype != torch.int32:
        raise RuntimeError("paged_attention_rocm: block_tables and seq_lens must be int32")
    S = block_tables.shape[0]
    H, D = query.shape[1], query.shape[2]
    check(_lib.lib().mi355q_paged_attention(
        out.data_ptr(), _ptr(exp_sum), _ptr(max_logits), _ptr(tmp_out), query.data_ptr(), query.stride(0),
        key_cache.data_ptr(), value_cache.data_ptr(), key_cache.stride(0), key_cache.stride(1), S, H,
        num_kv_heads, D, scale, block_tables.data_ptr(), block_tables.shape[1], seq_lens.data_ptr(),
        _ptr(query_start_loc), block_size, max_seq_len, _ptr(alibi_slopes), _dt(query),
        int(_is_fp8(kv_cache_dtype)), _ptr(k_scale), _ptr(v_scale), _stream()))


def paged_attention_noalloc(out, query, key_cache, value_cache, num_kv_heads: int, scale: float, block_tables,
                            seq_lens, query_start_loc, block_size: int, max_seq_len: int, alibi_slopes,
                            kv_cache_dtype: str, k_scale, v_scale) -> None:
    """paged_attention_rocm with the partition scratch taken from a device array of the library
    (mi355q_paged_attention_noalloc): for callers that cannot allocate.  Raises when sequences x heads x
    partitions exceeds the built-in scratch."""
    _need_cuda(out, query, key_cache, value_cache, block_tables, seq_lens)
    if block_tables.dtype != torch.int32 or seq_lens.dtype != torch.int32:
        raise RuntimeError("paged_attention_noalloc: block_tables and seq_lens must be int32")
    S = block_tables.shape[0]
    H, D = query.shape[1], query.shape[2]
    check(_lib.lib().mi355q_paged_attention_noalloc(
        out.data_ptr(), query.data_ptr(), query.stride(0), key_cache.data_ptr(), value_cache.data_ptr(),
        key_cache.stride(0), key_cache.stride(1), S, H, num_kv_heads, D, scale, block_tables.data_ptr(),
        block_tables.shape[1], seq_lens.data_ptr(), _ptr(query_start_loc), block_size, max_seq_len,
        _ptr(alibi_slopes), _dt(query), int(_is_fp8(kv_cache_dtype)), _ptr(k_scale), _ptr(v_scale), _stream()))


def paged_prefill_attention(out, exp_sum, max_logits, tmp_out, query, key_cache, value_cache, num_kv_heads: int,
                            scale: float, block_tables, seq_lens, query_start_loc, max_query_len: int,
                            block_size: int, max_seq_len: int, alibi_slopes, kv_cache_dtype: str, k_scale,
                            v_scale) -> None:
    """Causal attention for rows of any query length, K/V read from the paged cache
    (the prompt rows of chunked_prefill_paged_decode.py:303-326)."""
    _need_cuda(out, query, key_cache, value_cache, block_tables, seq_lens, query_start_loc)
    S = block_tables.shape[0]
    H, D = query.shape[1], query.shape[2]
    check(_lib.lib().mi355q_paged_prefill_attention(
        out.data_ptr(), _ptr(exp_sum), _ptr(max_logits), _ptr(tmp_out), query.data_ptr(), query.stride(0),
        key_cache.data_ptr(), value_cache.data_ptr(), key_cache.stride(0), key_cache.stride(1), S, H,
        num_kv_heads, D, scale, block_tables.data_ptr(), block_tables.shape[1], seq_lens.data_ptr(),
        query_start_loc.data_ptr(), max_query_len, block_size, max_seq_len, _ptr(alibi_slopes), _dt(query),
        int(_is_fp8(kv_cache_dtype)), _ptr(k_scale), _ptr(v_scale), _stream()))


def decode_attention_fused(out, exp_sum, max_logits, tmp_out, qkv, positions, cos_sin_cache, key_cache, value_cache,
                           num_heads: int, num_kv_heads: int, head_size: int, scale: float, block_tables, seq_lens,
                           block_size: int, max_seq_len: int, kv_cache_dtype: str, k_scale, v_scale) -> None:
    """rotary_embedding + reshape_and_cache + paged_attention_rocm of one decode step in one
    launch (the sequence of vllm/v1/attention/backends/rocm_attn.py:367-517).  qkv [S, (H+2KVH)*D]
    is read only; the new key/value land in the cache at position seq_lens - 1."""
    _need_cuda(out, qkv, key_cache, value_cache, block_tables, seq_lens, positions, cos_sin_cache)
    if cos_sin_cache.dtype != qkv.dtype or cos_sin_cache.shape[1] != head_size:
        raise RuntimeError("decode_attention_fused: cos_sin_cache must be [max_pos, head_size] in the activation dtype")
    S = qkv.shape[0]
    check(_lib.lib().mi355q_decode_attention_fused(
        out.data_ptr(), _ptr(exp_sum), _ptr(max_logits), _ptr(tmp_out), qkv.data_ptr(), qkv.stride(0),
        positions.data_ptr(), cos_sin_cache.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
        key_cache.stride(0), key_cache.stride(1), S, num_heads, num_kv_heads, head_size, scale,
        block_tables.data_ptr(), block_tables.shape[1], seq_lens.data_ptr(), block_size, max_seq_len, _dt(qkv),
        int(_is_fp8(kv_cache_dtype)), _ptr(k_scale), _ptr(v_scale), _stream()))


# ---------------------------------------------------------------------------
# sampler inner loop
# ---------------------------------------------------------------------------
def argmax(logits: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """logits.float().argmax(-1) with first-index ties (vllm/v1/sample/sampler.py:96,240-241).
    `out` (int64 [rows], contiguous) receives the ids when given."""
    _need_cuda(logits)
    if logits.dim() != 2 or logits.stride(1) != 1:
        raise RuntimeError("argmax: logits must be [rows, vocab] with a contiguous last dim")
    if out is not None:
        _need_cuda(out)
        if out.dtype != torch.int64 or out.shape != (logits.shape[0],) or not out.is_contiguous():
            raise RuntimeError("argmax: out must be a contiguous int64 tensor with one entry per row")
    ids = out if out is not None else torch.empty((logits.shape[0],), dtype=torch.int64, device=logits.device)
    check(_lib.lib().mi355q_argmax(logits.data_ptr(), logits.stride(0), ids.data_ptr(), logits.shape[0],
                                   logits.shape[1], _dt(logits, True), _stream()))
    return ids


def argmax_decode_step(logits: torch.Tensor, ids: torch.Tensor, positions: torch.Tensor | None, seq_lens: torch.Tensor | None,
                       embedding: torch.Tensor | None = None, hidden: torch.Tensor | None = None) -> torch.Tensor:
    """argmax into `ids` (int64 [rows]) and, in the same launch, positions += 1 (int64), seq_lens += 1 (int32) and
    hidden[b] = embedding[ids[b]]: the bookkeeping of a greedy decode step (sampler.py:240-241 followed by the model
    runner's position update and the next step's embedding lookup)."""
    _need_cuda(logits, ids)
    if logits.dim() != 2 or logits.stride(1) != 1:
        raise RuntimeError("argmax_decode_step: logits must be [rows, vocab] with a contiguous last dim")
    B = logits.shape[0]
    if ids.dtype != torch.int64 or ids.shape != (B,) or not ids.is_contiguous():
        raise RuntimeError("argmax_decode_step: ids must be a contiguous int64 tensor with one entry per row")
    if positions is not None and (positions.dtype != torch.int64 or positions.shape != (B,) or not positions.is_contiguous()):
        raise RuntimeError("argmax_decode_step: positions must be contiguous int64 [rows]")
    if seq_lens is not None and (seq_lens.dtype != torch.int32 or seq_lens.shape != (B,) or not seq_lens.is_contiguous()):
        raise RuntimeError("argmax_decode_step: seq_lens must be contiguous int32 [rows]")
    H = 0
    if embedding is not None or hidden is not None:
        if embedding is None or hidden is None or embedding.dtype != hidden.dtype or embedding.dim() != 2 or \
                hidden.shape != (B, embedding.shape[1]) or embedding.stride(1) != 1 or hidden.stride(1) != 1 or \
                embedding.shape[0] < logits.shape[1] or embedding.element_size() != 2:
            raise RuntimeError("argmax_decode_step: embedding [vocab, hidden] and hidden [rows, hidden] must be 16-bit, "
                               "same dtype, contiguous rows, one embedding row per logit")
        H = embedding.shape[1]
    check(_lib.lib().mi355q_argmax_decode_step(
        logits.data_ptr(), logits.stride(0), ids.data_ptr(), B, logits.shape[1], _dt(logits, True), _ptr(positions),
        _ptr(seq_lens), _ptr(embedding), embedding.stride(0) if embedding is not None else 0, _ptr(hidden),
        hidden.stride(0) if hidden is not None else 0, H, _stream()))
    return ids


def sample_topk_topp(logits: torch.Tensor, temperature: torch.Tensor | None, top_k: torch.Tensor | None,
                     top_p: torch.Tensor | None, noise: torch.Tensor) -> torch.Tensor:
    """argmax(softmax(top_k_top_p_mask(logits / temperature)) / noise) per row, without sorting the vocabulary.
    Stands in for apply_top_k_top_p + random_sample (vllm/v1/sample/ops/topk_topp_sampler.py:349-407, 441-466);
    `noise` holds the Exp(1) draws the caller made with the request generators.  fp32 logits [rows, vocab]."""
    _need_cuda(logits, noise)
    if logits.dim() != 2 or logits.stride(1) != 1 or logits.dtype != torch.float32:
        raise RuntimeError("sample_topk_topp: logits must be fp32 [rows, vocab] with a contiguous last dim")
    B, V = logits.shape
    if noise.shape != (B, V) or noise.dtype != torch.float32 or not noise.is_contiguous():
        raise RuntimeError("sample_topk_topp: noise must be a contiguous fp32 tensor of the logits' shape")

    def per_row(t, dtype, what):
        if t is None:
            return None
        _need_cuda(t)
        t = t.to(dtype).contiguous()
        if t.shape != (B,):
            raise RuntimeError(f"sample_topk_topp: {what} must have one entry per row")
        return t
    temperature = per_row(temperature, torch.float32, "temperature")
    top_k = per_row(top_k, torch.int32, "top_k")
    top_p = per_row(top_p, torch.float32, "top_p")
    ids = torch.empty((B,), dtype=torch.int64, device=logits.device)
    check(_lib.lib().mi355q_sample_topk_topp(logits.data_ptr(), logits.stride(0), _ptr(temperature), _ptr(top_k), _ptr(top_p),
                                             noise.data_ptr(), ids.data_ptr(), B, V, _stream()))
    return ids


def apply_repetition_penalties(logits, prompt_mask, output_mask, repetition_penalties) -> None:
    """ops.apply_repetition_penalties (vllm/_custom_ops.py:336-357): in place."""
    _need_cuda(logits, prompt_mask, output_mask, repetition_penalties)
    if not logits.is_contiguous():
        raise RuntimeError("apply_repetition_penalties: logits must be contiguous")
    pen = repetition_penalties.to(logits.dtype)
    check(_lib.lib().mi355q_apply_repetition_penalties(
        logits.data_ptr(), prompt_mask.contiguous().data_ptr(), output_mask.contiguous().data_ptr(),
        pen.data_ptr(), logits.shape[0], logits.shape[1], _dt(logits, True), _stream()))
