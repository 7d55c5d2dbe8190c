"""Test-side restatement of the expert computation with the CPU oracle (not a product path).
Follows tests/kernels/utils.py:912-1078 of the reference: per (token, expert) slot
y = (silu(x W1^T) * (x W3^T)) W2^T, each GEMM rounded to the activation dtype, then
out = round(sum_j weight_j * y_j) in fp32."""
import numpy as np

import oracle as orc


def unpack_moe_weight(qw_u8):
    """uint8 [N, K/2] (even k in the low nibble) -> int32 [K, N]."""
    n, k2 = qw_u8.shape
    q = np.empty((n, 2 * k2), dtype=np.int32)
    q[:, 0::2] = qw_u8 & 0xF
    q[:, 1::2] = qw_u8 >> 4
    return np.ascontiguousarray(q.T)


def unpack_moe_zeros(qz_u8):
    """uint8 [N/2, K/G] (even n in the low nibble) -> int32 [K/G, N]."""
    n2, g = qz_u8.shape
    z = np.empty((2 * n2, g), dtype=np.int32)
    z[0::2] = qz_u8 & 0xF
    z[1::2] = qz_u8 >> 4
    return np.ascontiguousarray(z.T)


def expert_gemm(a_bits, qw_u8, s_bits, qz_u8, G, dt=orc.BF16, round_weight=False):
    q = unpack_moe_weight(qw_u8)
    s = np.ascontiguousarray(s_bits.T)                      # [K/G, N]
    if qz_u8 is None:
        return orc.w4a16_gemm(a_bits, orc.gptq_pack(q), s, None, None, G=G, dt=dt, round_weight=round_weight)
    z = orc.pack_cols(unpack_moe_zeros(qz_u8))
    return orc.w4a16_gemm(a_bits, orc.gptq_pack(q), s, z, None, G=G, zp_mode=orc.ZP_NSEQ, dt=dt,
                          round_weight=round_weight)


def route(score_bits, topk, renormalize=False, dt=orc.BF16):
    s = orc.h16_to_f32(score_bits, dt).astype(np.float32)
    e = np.exp(s - s.max(axis=1, keepdims=True), dtype=np.float32)
    p = e / e.sum(axis=1, keepdims=True, dtype=np.float32)
    ids = np.argsort(-p, axis=1, kind="stable")[:, :topk]
    w = np.take_along_axis(p, ids, axis=1)
    if renormalize:
        w = w / w.sum(axis=1, keepdims=True)
    return w.astype(np.float32), ids.astype(np.int32)


def moe(a_bits, w1, w2, weights, ids, G, dt=orc.BF16, round_weight=False, expert_map=None):
    """w1 = (qweight u8 [E, 2I, H/2], scales bits [E, 2I, H/G], zeros u8 or None); w2 likewise."""
    T, topk = ids.shape
    H = a_bits.shape[1]
    acc = np.zeros((T, H), dtype=np.float32)
    for t in range(T):
        for j in range(topk):
            e = int(ids[t, j])
            if expert_map is not None:
                e = int(expert_map[e])
                if e < 0:
                    continue
            h = expert_gemm(a_bits[t:t + 1], w1[0][e], w1[1][e], None if w1[2] is None else w1[2][e], G, dt, round_weight)
            act = orc.silu_and_mul(h, dt)
            y = expert_gemm(act, w2[0][e], w2[1][e], None if w2[2] is None else w2[2][e], G, dt, round_weight)
            acc[t] += weights[t, j] * orc.h16_to_f32(y[0], dt)
    return orc.f32_to_h16(acc, dt)
