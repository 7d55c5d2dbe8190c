"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/liboracle.so (built from oracle/oracle.c by
`make -C oracle`).  Only tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py may import this package; the product package never does.

All 16-bit float tensors cross this boundary as numpy uint16 bit patterns,
fp8 as uint8.  dt: 0 = bf16, 1 = fp16.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BF16, FP16 = 0, 1
W_GPTQ, W_NSEQ, W_AWQ = 0, 1, 2
ZP_CONST, ZP_NSEQ, ZP_AWQ, ZP_GPTQ_V1 = 0, 1, 2, 3


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be C-contiguous"
    return a.ctypes.data_as(ctypes.c_void_p)


_i = ctypes.c_int
_f = ctypes.c_float


# ---- bit helpers -------------------------------------------------------
def f32_to_h16(x, dt=BF16):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty(x.shape, dtype=np.uint16)
    lib().orc_f32_to_h16(_p(x), _p(y), ctypes.c_int64(x.size), _i(dt))
    return y


def h16_to_f32(x, dt=BF16):
    x = np.ascontiguousarray(x, dtype=np.uint16)
    y = np.empty(x.shape, dtype=np.float32)
    lib().orc_h16_to_f32(_p(x), _p(y), ctypes.c_int64(x.size), _i(dt))
    return y


def f32_to_e4m3(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty(x.shape, dtype=np.uint8)
    lib().orc_f32_to_e4m3(_p(x), _p(y), ctypes.c_int64(x.size))
    return y


def e4m3_to_f32(x):
    x = np.ascontiguousarray(x, dtype=np.uint8)
    y = np.empty(x.shape, dtype=np.float32)
    lib().orc_e4m3_to_f32(_p(x), _p(y), ctypes.c_int64(x.size))
    return y


# ---- packers -----------------------------------------------------------
def gptq_pack(q):
    q = np.ascontiguousarray(q, dtype=np.int32)
    K, N = q.shape
    out = np.empty((K // 8, N), dtype=np.int32)
    lib().orc_gptq_pack(_p(q), _p(out), _i(K), _i(N))
    return out


def pack_cols(q, awq_order=False):
    q = np.ascontiguousarray(q, dtype=np.int32)
    R, N = q.shape
    out = np.empty((R, N // 8), dtype=np.int32)
    lib().orc_pack_cols(_p(q), _p(out), _i(R), _i(N), _i(int(awq_order)))
    return out


def unpack_rows(p):
    p = np.ascontiguousarray(p, dtype=np.int32)
    K8, N = p.shape
    out = np.empty((K8 * 8, N), dtype=np.int32)
    lib().orc_unpack_rows(_p(p), _p(out), _i(K8 * 8), _i(N))
    return out


def unpack_cols(p, awq_order=False):
    p = np.ascontiguousarray(p, dtype=np.int32)
    R, N8 = p.shape
    out = np.empty((R, N8 * 8), dtype=np.int32)
    lib().orc_unpack_cols(_p(p), _p(out), _i(R), _i(N8 * 8), _i(int(awq_order)))
    return out


# ---- W4A16 -------------------------------------------------------------
def gptq_pack_bits(q, bits, axis=0):
    """AutoGPTQ packing of b-bit values (b = 2, 3, 4, 8): one little-endian bit string per column along
    K (axis 0: qweight [K * b / 32, N]) or per row along N (axis 1: qzeros [K/G, N * b / 32]); value i at
    bits [b i, b i + b).  For b = 3 this is the 32-values-in-3-words layout drawn in
    csrc/libtorch_stable/quantization/gptq/qdq_3.cuh:21-23; for b = 4 it equals gptq_pack."""
    q = np.asarray(q, dtype=np.uint64)
    if axis == 1:
        return np.ascontiguousarray(gptq_pack_bits(q.T, bits, 0).T)
    K, N = q.shape
    assert (K * bits) % 32 == 0
    out = np.zeros((K * bits // 32, N), dtype=np.uint64)
    for k in range(K):
        bit = k * bits
        w, sh = bit // 32, bit % 32
        v = q[k] << np.uint64(sh)
        out[w] |= v & np.uint64(0xffffffff)
        if sh + bits > 32:
            out[w + 1] |= v >> np.uint64(32)
    return out.astype(np.uint32).view(np.int32)


def gptq_unpack_bits(p, bits, axis=0):
    """Inverse of gptq_pack_bits: int32 values in [0, 2^bits)."""
    p = np.asarray(p).view(np.uint32).astype(np.uint64)
    if axis == 1:
        return np.ascontiguousarray(gptq_unpack_bits(p.T.astype(np.uint32).view(np.int32), bits, 0).T)
    W, N = p.shape
    K = W * 32 // bits
    q = np.zeros((K, N), dtype=np.int32)
    mask = np.uint64((1 << bits) - 1)
    for k in range(K):
        bit = k * bits
        w, sh = bit // 32, bit % 32
        v = p[w] >> np.uint64(sh)
        if sh + bits > 32:
            v |= p[w + 1] << np.uint64(32 - sh)
        q[k] = (v & mask).astype(np.int32)
    return q


def _wshape(qw, layout):
    if layout == W_GPTQ:
        return qw.shape[0] * 8, qw.shape[1]
    return qw.shape[0], qw.shape[1] * 8


def w4a16_dequant(qw, scales, qz=None, G=128, layout=W_GPTQ, zp_mode=ZP_CONST, zp_const=8,
                  dt=BF16, round_weight=False):
    qw = np.ascontiguousarray(qw, dtype=np.int32)
    K, N = _wshape(qw, layout)
    W = np.empty((K, N), dtype=np.float32)
    lib().orc_w4a16_dequant(_p(qw), _p(np.ascontiguousarray(scales)), _p(qz), _p(W), _i(K), _i(N),
                            _i(G), _i(layout), _i(zp_mode), _i(zp_const), _i(dt),
                            _i(int(round_weight)))
    return W


def w4a16_gemm(a, qw, scales, qz=None, bias=None, G=128, layout=W_GPTQ, zp_mode=ZP_CONST,
               zp_const=8, dt=BF16, round_weight=False):
    a = np.ascontiguousarray(a, dtype=np.uint16)
    qw = np.ascontiguousarray(qw, dtype=np.int32)
    K, N = _wshape(qw, layout)
    M = a.shape[0]
    assert a.shape[1] == K
    c = np.empty((M, N), dtype=np.uint16)
    lib().orc_w4a16_gemm(_p(a), _p(qw), _p(np.ascontiguousarray(scales)), _p(qz), _p(bias), _p(c),
                         _i(M), _i(N), _i(K), _i(G), _i(layout), _i(zp_mode), _i(zp_const),
                         _i(dt), _i(int(round_weight)))
    return c


def w8a16_dequant(w, scales, block=None, dt=BF16, round_weight=False):
    w = np.ascontiguousarray(w, dtype=np.uint8)
    N, K = w.shape
    bn, bk = block or (1, K)
    out = np.empty((N, K), dtype=np.float32)
    lib().orc_w8a16_dequant(_p(w), _p(np.ascontiguousarray(scales, dtype=np.float32)), _p(out), _i(N), _i(K),
                            _i(bn), _i(bk), _i(dt), _i(int(round_weight)))
    return out


def w8a16_gemm(a, w, scales, bias=None, block=None, dt=BF16, round_weight=False):
    a = np.ascontiguousarray(a, dtype=np.uint16)
    w = np.ascontiguousarray(w, dtype=np.uint8)
    N, K = w.shape
    M = a.shape[0]
    assert a.shape[1] == K
    bn, bk = block or (1, K)
    c = np.empty((M, N), dtype=np.uint16)
    lib().orc_w8a16_gemm(_p(a), _p(w), _p(np.ascontiguousarray(scales, dtype=np.float32)), _p(bias), _p(c),
                         _i(M), _i(N), _i(K), _i(bn), _i(bk), _i(dt), _i(int(round_weight)))
    return c


def cpu_repack_gptq(qw):
    qw = np.ascontiguousarray(qw, dtype=np.int32)
    K, N = qw.shape[0] * 8, qw.shape[1]
    out = np.empty((N // 16, K * 2), dtype=np.int32)
    lib().orc_cpu_repack_gptq(_p(qw), _p(out), _i(K), _i(N))
    return out


def cpu_gemm_wna16(a, qwb, scales, zeros=None, bias=None, G=128, dt=BF16):
    a = np.ascontiguousarray(a, dtype=np.uint16)
    M, K = a.shape
    N = qwb.shape[0] * 16
    c = np.empty((M, N), dtype=np.uint16)
    lib().orc_cpu_gemm_wna16(_p(a), _p(qwb), _p(np.ascontiguousarray(scales)), _p(zeros), _p(bias),
                             _p(c), _i(M), _i(N), _i(K), _i(G), _i(dt))
    return c


def linear_h16(a, w, dt=BF16):
    a = np.ascontiguousarray(a, dtype=np.uint16)
    w = np.ascontiguousarray(w, dtype=np.uint16)
    M, K = a.shape
    N = w.shape[0]
    c = np.empty((M, N), dtype=np.uint16)
    lib().orc_linear_h16(_p(a), _p(w), _p(c), _i(M), _i(N), _i(K), _i(dt))
    return c


# ---- norms / rope / activation ----------------------------------------
def rms_norm(x, w, eps, dt=BF16):
    x = np.ascontiguousarray(x, dtype=np.uint16)
    T, H = x.shape
    out = np.empty_like(x)
    lib().orc_rms_norm(_p(x), _p(w), _p(out), _i(T), _i(H), _f(eps), _i(dt))
    return out


def fused_add_rms_norm(x, residual, w, eps, dt=BF16):
    x = np.array(x, dtype=np.uint16, copy=True, order="C")
    r = np.array(residual, dtype=np.uint16, copy=True, order="C")
    T, H = x.shape
    lib().orc_fused_add_rms_norm(_p(x), _p(r), _p(w), _i(T), _i(H), _f(eps), _i(dt))
    return x, r


def rotary_embedding(positions, q, k, cache, head, is_neox, dt=BF16):
    q = np.array(q, dtype=np.uint16, copy=True, order="C")
    k = None if k is None else np.array(k, dtype=np.uint16, copy=True, order="C")
    positions = np.ascontiguousarray(positions, dtype=np.int64)
    T = positions.shape[0]
    nh = q.size // T // head
    nkv = 0 if k is None else k.size // T // head
    rot = cache.shape[1]
    lib().orc_rotary_embedding(_p(positions), _p(q), _p(k), _p(np.ascontiguousarray(cache)), _i(T),
                               _i(nh), _i(nkv), _i(head), _i(rot), _i(int(is_neox)), _i(dt))
    return q, k


def silu_and_mul(x, dt=BF16):
    x = np.ascontiguousarray(x, dtype=np.uint16)
    T, d2 = x.shape
    out = np.empty((T, d2 // 2), dtype=np.uint16)
    lib().orc_silu_and_mul(_p(x), _p(out), _i(T), _i(d2 // 2), _i(dt))
    return out


# ---- KV cache / attention ---------------------------------------------
def reshape_and_cache(key, value, kc, vc, slot_mapping, fp8=False, k_scale=1.0, v_scale=1.0,
                      dt=BF16):
    """kc [blocks, kvh, head/x, bs, x], vc [blocks, kvh, head, bs]; in place."""
    key = np.ascontiguousarray(key, dtype=np.uint16)
    value = np.ascontiguousarray(value, dtype=np.uint16)
    T, kvh, head = key.shape
    bs, x = kc.shape[3], kc.shape[4]
    sm = np.ascontiguousarray(slot_mapping, dtype=np.int64)
    lib().orc_reshape_and_cache(_p(key), _p(value), _p(kc), _p(vc), _p(sm), _i(T), _i(kvh),
                                _i(head), _i(bs), _i(x), _i(int(fp8)), _f(k_scale), _f(v_scale),
                                _i(dt))


def reshape_and_cache_flash(key, value, kc, vc, slot_mapping, hnd=False, fp8=False, k_scale=(1.0,), v_scale=(1.0,),
                            dt=BF16):
    """kc / vc: [blocks, bs, heads, head] (hnd=False) or [blocks, heads, bs, head] (hnd=True) arrays, written in
    place; k_scale / v_scale: one value or one per head."""
    key = np.ascontiguousarray(key, dtype=np.uint16)
    value = np.ascontiguousarray(value, dtype=np.uint16)
    T, heads, head = key.shape
    bs = kc.shape[2] if hnd else kc.shape[1]
    block_stride = bs * heads * head
    page_stride, head_stride = (head, bs * head) if hnd else (heads * head, head)
    sm = np.ascontiguousarray(slot_mapping, dtype=np.int64)
    ks = np.ascontiguousarray(k_scale, dtype=np.float32).reshape(-1)
    vs = np.ascontiguousarray(v_scale, dtype=np.float32).reshape(-1)
    assert ks.size == vs.size and ks.size in (1, heads)
    lib().orc_reshape_and_cache_flash(_p(key), _p(value), _p(kc), _p(vc), _p(sm), _i(len(sm)), _i(heads), _i(head), _i(bs),
                                      ctypes.c_int64(block_stride), ctypes.c_int64(page_stride), ctypes.c_int64(head_stride),
                                      _i(int(fp8)), _p(ks), _p(vs), _i(int(ks.size > 1)), _i(dt))


def paged_attention(q, kc, vc, block_tables, seq_lens, scale, alibi=None, fp8=False, k_scale=1.0,
                    v_scale=1.0, dt=BF16):
    q = np.ascontiguousarray(q, dtype=np.uint16)
    S, H, D = q.shape
    kvh, bs, x = kc.shape[1], kc.shape[3], kc.shape[4]
    bt = np.ascontiguousarray(block_tables, dtype=np.int32)
    sl = np.ascontiguousarray(seq_lens, dtype=np.int32)
    if alibi is not None:
        alibi = np.ascontiguousarray(alibi, dtype=np.float32)
    out = np.empty_like(q)
    lib().orc_paged_attention(_p(q), _p(np.ascontiguousarray(kc)), _p(np.ascontiguousarray(vc)),
                              _p(bt), _p(sl), _p(alibi), _p(out), _i(S), _i(H), _i(kvh), _i(D),
                              _i(bs), _i(bt.shape[1]), _i(x), _i(int(fp8)), _f(scale),
                              _f(k_scale), _f(v_scale), _i(dt))
    return out


def prefill_attention(q, k, v, scale, dt=BF16):
    q = np.ascontiguousarray(q, dtype=np.uint16)
    k = np.ascontiguousarray(k, dtype=np.uint16)
    v = np.ascontiguousarray(v, dtype=np.uint16)
    T, H, D = q.shape
    kvh = k.shape[1]
    out = np.empty_like(q)
    lib().orc_prefill_attention(_p(q), _p(k), _p(v), _p(out), _i(T), _i(H), _i(kvh), _i(D),
                                _f(scale), _i(dt))
    return out


# ---- sampler ------------------------------------------------------------
def argmax_h16(logits, dt=BF16):
    logits = np.ascontiguousarray(logits, dtype=np.uint16)
    B, V = logits.shape
    ids = np.empty((B,), dtype=np.int64)
    lib().orc_argmax_h16(_p(logits), _p(ids), _i(B), _i(V), _i(dt))
    return ids


def argmax_f32(logits):
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    B, V = logits.shape
    ids = np.empty((B,), dtype=np.int64)
    lib().orc_argmax_f32(_p(logits), _p(ids), _i(B), _i(V))
    return ids


def top_k_top_p_mask(logits, top_k=None, top_p=None):
    """apply_top_k_top_p_pytorch (vllm/v1/sample/ops/topk_topp_sampler.py:367-407) on fp32 [B, V]: sort
    ascending, mask everything below the k-th largest, softmax + cumulative sum of the sorted row, mask
    where the sum is <= 1 - p (never the last), scatter back.  Returns the masked copy."""
    x = np.array(logits, dtype=np.float32, copy=True)
    B, V = x.shape
    for b in range(B):
        idx = np.argsort(x[b], kind="stable")
        srt = x[b][idx]
        if top_k is not None:
            kth = srt[V - int(top_k[b])]
            srt[srt < kth] = -np.inf
        if top_p is not None:
            e = np.exp(srt - srt[-1], dtype=np.float32)
            probs = e / e.sum(dtype=np.float32)
            drop = np.cumsum(probs, dtype=np.float32) <= np.float32(1.0) - np.float32(top_p[b])
            drop[-1] = False
            srt[drop] = -np.inf
        x[b][idx] = srt
    return x


def sample_topk_topp(logits, temperature, top_k, top_p, noise):
    """Sampler.sample for random rows (vllm/v1/sample/sampler.py:263-302): logits / temperature, the masks
    above, softmax in fp32, argmax(probs / noise) with noise ~ Exp(1) (topk_topp_sampler.py:427-438)."""
    x = np.asarray(logits, dtype=np.float32)
    if temperature is not None:
        x = x / np.asarray(temperature, dtype=np.float32)[:, None]
    m = top_k_top_p_mask(x, top_k, top_p)
    e = np.exp(m - m.max(axis=1, keepdims=True), dtype=np.float32)
    probs = e / e.sum(axis=1, keepdims=True, dtype=np.float32)
    return np.argmax(probs / np.asarray(noise, dtype=np.float32), axis=1).astype(np.int64)


def apply_repetition_penalties(logits, prompt_mask, output_mask, penalties):
    lg = np.array(logits, dtype=np.float32, copy=True, order="C")
    B, V = lg.shape
    pm = np.ascontiguousarray(prompt_mask, dtype=np.uint8)
    om = np.ascontiguousarray(output_mask, dtype=np.uint8)
    pen = np.ascontiguousarray(penalties, dtype=np.float32)
    lib().orc_apply_repetition_penalties(_p(lg), _p(pm), _p(om), _p(pen), _i(B), _i(V))
    return lg


# ---- GPTQ 8 bits (symmetric) ---------------------------------------------------
def gptq8_pack(q):
    """[K, N] values 0..255 -> int32 [K/4, N], four consecutive k per word, least significant byte
    first (pack_rows with 8 bits: vllm/model_executor/layers/quantization/utils/quant_utils.py:828-849)."""
    q = np.ascontiguousarray(q, dtype=np.uint32)
    K, N = q.shape
    r = q.reshape(K // 4, 4, N)
    return (r[:, 0] | (r[:, 1] << 8) | (r[:, 2] << 16) | (r[:, 3] << 24)).astype(np.uint32).view(np.int32)


def gptq8_dequant(q, scales_bits, G, dt=BF16, round_weight=False):
    """W[k, n] = (q[k, n] - 128) * s[k // G, n] in fp32 (uint8b128: csrc/quantization/gptq/qdq_8.cuh
    of the reference, zero point 128 for symmetric checkpoints); optionally rounded to the activation
    dtype as the reconstruct route does (q_gemm.cu:1509-1522)."""
    s = h16_to_f32(scales_bits, dt)
    W = (np.asarray(q, dtype=np.float32) - np.float32(128)) * np.repeat(s, G, axis=0)
    return h16_to_f32(f32_to_h16(W, dt), dt) if round_weight else W


def gptq8_gemm(a_bits, q, scales_bits, G, bias_bits=None, dt=BF16, round_weight=False):
    """round(a @ W) (+ bias added to the rounded product, as output.add_(bias) does), fp32 products
    summed in float64 (order-free stand-in for the kernels' fp32 accumulation)."""
    W = gptq8_dequant(q, scales_bits, G, dt, round_weight).astype(np.float64)
    y = h16_to_f32(a_bits, dt).astype(np.float64) @ W
    out = f32_to_h16(y.astype(np.float32), dt)
    if bias_bits is not None:
        out = f32_to_h16(h16_to_f32(out, dt) + h16_to_f32(bias_bits, dt)[None, :], dt)
    return out


# ---- fp8 activation quantisers and the fp8 x fp8 scaled product ---------------
FP32 = 2
_i64 = ctypes.c_int64


def _in(x, dt):
    """2-D input: fp32 array for dt = FP32, uint16 bit patterns otherwise (C-contiguous)."""
    return np.ascontiguousarray(x, dtype=np.float32 if dt == FP32 else np.uint16)


def static_scaled_fp8_quant(x, scale, group_shape=None, dt=BF16):
    """scale: 0-D / [1] (per tensor), 1-D with group_shape (-1, 1) or (1, -1), or 2-D [T/gm, H/gn]."""
    x = _in(x, dt)
    T, H = x.shape
    scale = np.ascontiguousarray(scale, dtype=np.float32)
    if scale.size == 1:
        gm, gn, si, sj = T, H, 0, 0
    elif scale.ndim == 1:
        gm = T if group_shape[0] == -1 else group_shape[0]
        gn = H if group_shape[1] == -1 else group_shape[1]
        si, sj = (0, 1) if T // gm == 1 else (1, 0)
    else:
        gm, gn = T // scale.shape[0], H // scale.shape[1]
        si, sj = scale.shape[1], 1
    out = np.empty((T, H), dtype=np.uint8)
    lib().orc_static_scaled_fp8_quant(_p(x), _i64(H), _p(scale), _p(out), _i(T), _i(H), _i(gm), _i(gn), _i64(si),
                                      _i64(sj), _i(dt))
    return out


def dynamic_scaled_fp8_quant(x, dt=BF16):
    x = _in(x, dt)
    T, H = x.shape
    out, scale = np.empty((T, H), dtype=np.uint8), np.zeros(1, dtype=np.float32)
    lib().orc_dynamic_scaled_fp8_quant(_p(x), _i64(H), _p(scale), _p(out), _i(T), _i(H), _i(dt))
    return out, scale


def dynamic_per_token_scaled_fp8_quant(x, scale_ub=None, dt=BF16):
    x = _in(x, dt)
    T, H = x.shape
    out, scales = np.empty((T, H), dtype=np.uint8), np.zeros((T, 1), dtype=np.float32)
    ub = None if scale_ub is None else np.asarray([scale_ub], dtype=np.float32)
    lib().orc_dynamic_per_token_scaled_fp8_quant(_p(x), _i64(H), _p(scales), _p(ub), _p(out), _i(T), _i(H), _i(dt))
    return out, scales


def rms_norm_static_fp8_quant(x, w, scale, eps, residual=None, dt=BF16):
    """Returns (fp8 bytes, new residual or None)."""
    x = np.ascontiguousarray(x, dtype=np.uint16)
    T, H = x.shape
    res = None if residual is None else np.array(residual, dtype=np.uint16, order="C")
    out = np.empty((T, H), dtype=np.uint8)
    sc = np.asarray([scale], dtype=np.float32).reshape(1)
    lib().orc_rms_norm_static_fp8_quant(_p(x), _i64(H), _p(res), _p(np.ascontiguousarray(w, dtype=np.uint16)), _p(sc),
                                        _p(out), _i(T), _i(H), _f(eps), _i(dt))
    return out, res


def rms_norm_dynamic_per_token_quant(x, w, eps, scale_ub=None, residual=None, dt=BF16):
    """Returns (fp8 bytes, scales [T, 1], new residual or None)."""
    x = np.ascontiguousarray(x, dtype=np.uint16)
    T, H = x.shape
    res = None if residual is None else np.array(residual, dtype=np.uint16, order="C")
    out, scales = np.empty((T, H), dtype=np.uint8), np.zeros((T, 1), dtype=np.float32)
    ub = None if scale_ub is None else np.asarray([scale_ub], dtype=np.float32)
    lib().orc_rms_norm_dynamic_per_token_quant(_p(x), _i64(H), _p(res), _p(np.ascontiguousarray(w, dtype=np.uint16)),
                                               _p(ub), _p(scales), _p(out), _i(T), _i(H), _f(eps), _i(dt))
    return out, scales, res


def silu_and_mul_quant(x, scale, dt=BF16):
    x = np.ascontiguousarray(x, dtype=np.uint16)
    T, d2 = x.shape
    out = np.empty((T, d2 // 2), dtype=np.uint8)
    sc = np.asarray([scale], dtype=np.float32).reshape(1)
    lib().orc_silu_and_mul_quant(_p(x), _p(sc), _p(out), _i(T), _i(d2 // 2), _i(dt))
    return out


def fp8_scaled_mm(a, w, scale_a, scale_b, bias=None, dt=BF16):
    """a: fp8 bytes [M, K]; w: fp8 bytes [N, K]; scales fp32 with 1 or M / N elements; bias: 16-bit patterns."""
    a, w = np.ascontiguousarray(a, dtype=np.uint8), np.ascontiguousarray(w, dtype=np.uint8)
    (M, K), N = a.shape, w.shape[0]
    sa = np.ascontiguousarray(scale_a, dtype=np.float32).reshape(-1)
    sb = np.ascontiguousarray(scale_b, dtype=np.float32).reshape(-1)
    c = np.empty((M, N), dtype=np.uint16)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.uint16)
    lib().orc_fp8_scaled_mm(_p(a), _p(w), _p(sa), _i(sa.size), _p(sb), _i(sb.size), _p(b), _p(c), _i(M), _i(N), _i(K),
                            _i(dt))
    return c


def per_token_group_quant_fp8(x, group_size, eps=1e-10, dt=BF16):
    """x: [T, H] (bit patterns, or fp32 for dt = FP32); returns (fp8 bytes [T, H], scales fp32 [T, H / group_size])."""
    x = _in(x, dt)
    T, H = x.shape
    out, scales = np.empty((T, H), dtype=np.uint8), np.zeros((T, H // group_size), dtype=np.float32)
    lib().orc_per_token_group_quant_fp8(_p(x), _i64(H), _p(scales), _p(out), _i(T), _i(H), _i(group_size), _f(eps), _i(dt))
    return out, scales


def w8a8_block_scaled_mm(a, b, a_scales, b_scales, block_size, dt=BF16, bias=None):
    """a: fp8 bytes [M, K]; b: fp8 bytes [N, K]; a_scales fp32 [M, ceil(K / bk)]; b_scales fp32 [ceil(N / bn), ceil(K / bk)]."""
    a, b = np.ascontiguousarray(a, dtype=np.uint8), np.ascontiguousarray(b, dtype=np.uint8)
    (M, K), N = a.shape, b.shape[0]
    sa = np.ascontiguousarray(a_scales, dtype=np.float32)
    sb = np.ascontiguousarray(b_scales, dtype=np.float32)
    bn, bk = block_size
    assert sa.shape == (M, -(-K // bk)) and sb.shape == (-(-N // bn), -(-K // bk))
    c = np.empty((M, N), dtype=np.uint16)
    bb = None if bias is None else np.ascontiguousarray(bias, dtype=np.uint16)
    lib().orc_w8a8_block_scaled_mm(_p(a), _p(b), _p(sa), _p(sb), _p(bb), _p(c), _i(M), _i(N), _i(K), _i(bn), _i(bk), _i(dt))
    return c
