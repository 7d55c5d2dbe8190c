"""Parity at the full sizes of the benchmark configurations, through the calls the decode runner makes.

  * Llama-3-8B AWQ g128, one token: the four projections exactly as LlamaRunner._layer launches them
    (add + RMSNorm prologue on qkv, plain o_proj, add + RMSNorm prologue and SiLU epilogue on gate_up,
    plain down_proj on the activated input);
  * Llama-3-70B GPTQ g128 tensor-parallel over 8: the four per-rank shapes at 1 and 8 tokens, and as a
    128-token prompt through the tiled prompt kernel;
  * one Mixtral-8x7B-sized expert layer (8 experts, hidden 4096, intermediate 14336, top-2, AWQ) at
    1, 2 and 8 tokens on every route of plugin/moe.py;
  * greedy tokens of a two-layer model with Llama-3-8B widths against the oracle model;
  * reference edge cases of RMSNorm (tokens 83 / 4096, hidden 769 / 5120) and RoPE (head 80 / 120 / 256).

Shapes: benchmarks/kernels/weight_shapes.py:28-33,58-63 of the reference.  Tolerances: rtol = atol =
1e-2 for the dequant-GEMMs (tests/kernels/quantization/test_triton_w4a16.py:162), atol 2e-2 for the
expert layer (tests/kernels/moe/test_moe.py:698), and bit-level agreement rates on top.
"""
import numpy as np
import pytest
import torch

import oracle as orc
from mi355q import _lib, ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EPS = 1e-5


def to_dev(bits):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(torch.bfloat16).to(DEV)


def to_bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def f32(bits):
    return orc.h16_to_f32(bits, orc.BF16)


def i32(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(DEV)


def _words(rng, shape):
    """Random packed int32 words: every nibble is uniform in [0, 16)."""
    return rng.integers(-2**31, 2**31, size=shape, dtype=np.int64).astype(np.int32)


class _Layer:
    """Checkpoint tensors of one projection, its repacked device weight and the oracle's GEMM on it."""

    def __init__(self, rng, K, N, G, awq, pair=False):
        self.K, self.N, self.G, self.awq = K, N, G, awq
        gain = 1.0 / (21.25 * K) ** 0.5
        self.s = orc.f32_to_h16((gain * (0.5 + rng.random((K // G, N)))).astype(np.float32))
        if awq:
            self.qw, self.qz = _words(rng, (K, N // 8)), _words(rng, (K // G, N // 8))
            self.w = ops.w4a16_repack(i32(self.qw), to_dev(self.s), i32(self.qz), G, _lib.W_AWQ, _lib.ZP_AWQ,
                                      pair_halves=pair)
        else:
            self.qw, self.qz = _words(rng, (K // 8, N)), None
            self.w = ops.w4a16_repack(i32(self.qw), to_dev(self.s), None, G, pair_halves=pair)

    def ref(self, a_bits):
        if self.awq:
            return orc.w4a16_gemm(a_bits, self.qw, self.s, self.qz, None, G=self.G, layout=orc.W_AWQ, zp_mode=orc.ZP_AWQ)
        return orc.w4a16_gemm(a_bits, self.qw, self.s, None, None, G=self.G)


def _close(got_bits, want_bits, what, min_equal=0.93):
    """Reference tolerance, plus a floor on bit-identical outputs.  The kernel sums a (bias + q) products on
    the matrix cores and removes bias * sum(a) afterwards; over K = 4096..14336 that leaves about one output
    in twenty a single 16-bit rounding step away from the oracle's plain fp32 (q - z) * s sum."""
    np.testing.assert_allclose(f32(got_bits), f32(want_bits), rtol=1e-2, atol=1e-2, err_msg=what)
    step = np.abs(got_bits.astype(np.int32) - want_bits.astype(np.int32))
    assert np.all((step <= 1) | np.isclose(f32(got_bits), f32(want_bits), rtol=0, atol=1e-3)), what
    assert (got_bits == want_bits).mean() > min_equal, what


def _decode_layer_calls(rng, M, hidden, qkv_n, o_k, gate_up_n, down_k, awq, what):
    """The four projections of one decoder layer as LlamaRunner._layer issues them for a few tokens,
    each checked against the oracle's sequence on the same tensors."""
    G = 128
    qkv = _Layer(rng, hidden, qkv_n, G, awq)
    o = _Layer(rng, o_k, hidden, G, awq)
    pair = ops.w4a16_gate_up_silu_supported(hidden, gate_up_n, G)
    gate_up = _Layer(rng, hidden, gate_up_n, G, awq, pair=pair)
    down = _Layer(rng, down_k, hidden, G, awq)
    x = orc.f32_to_h16(rng.standard_normal((M, hidden)).astype(np.float32))
    res = orc.f32_to_h16(rng.standard_normal((M, hidden)).astype(np.float32))
    ln = orc.f32_to_h16((1 + 0.1 * rng.standard_normal(hidden)).astype(np.float32))

    # qkv: residual add + RMSNorm in the prologue (ops.w4a16_gemm_add_rms_norm)
    got, new_res = ops.w4a16_gemm_add_rms_norm(to_dev(x), to_dev(res), to_dev(ln), EPS, qkv.w)
    xn, res_ref = orc.fused_add_rms_norm(x, res, ln, EPS)
    assert np.array_equal(to_bits(new_res), res_ref), what + " qkv residual"
    _close(to_bits(got), qkv.ref(xn), what + " qkv")

    # o_proj: plain GEMM on the attention output
    attn = orc.f32_to_h16((0.5 * rng.standard_normal((M, o_k))).astype(np.float32))
    _close(to_bits(ops.w4a16_gemm(to_dev(attn), o.w)), o.ref(attn), what + " o_proj")

    # gate_up: add + RMSNorm prologue, SiluAndMul epilogue where the layer supports it
    if pair:
        act, new_res = ops.w4a16_gate_up_silu(to_dev(x), to_dev(res), to_dev(ln), EPS, gate_up.w)
        assert np.array_equal(to_bits(new_res), res_ref), what + " gate_up residual"
        act_ref = orc.silu_and_mul(gate_up.ref(xn))
        # SiLU and the product are rounded again on top of the projection's rounding
        np.testing.assert_allclose(f32(to_bits(act)), f32(act_ref), rtol=2e-2, atol=1e-2, err_msg=what + " gate_up")
        assert (to_bits(act) == act_ref).mean() > 0.88, what + " gate_up"    # two projections' steps and SiLU's
    else:
        gu, new_res = ops.w4a16_gemm_add_rms_norm(to_dev(x), to_dev(res), to_dev(ln), EPS, gate_up.w)
        _close(to_bits(gu), gate_up.ref(xn), what + " gate_up")

    # down_proj on the activated input; and with SiluAndMul in its prologue, as batches past the fused path run it
    a_in = orc.f32_to_h16((0.5 * rng.standard_normal((M, down_k))).astype(np.float32))
    _close(to_bits(ops.w4a16_gemm(to_dev(a_in), down.w)), down.ref(a_in), what + " down")
    gu_in = orc.f32_to_h16(rng.standard_normal((M, 2 * down_k)).astype(np.float32))
    got = ops.w4a16_gemm(to_dev(gu_in), down.w, fuse_silu_mul=True)
    _close(to_bits(got), down.ref(orc.silu_and_mul(gu_in)), what + " down(silu)")


def test_llama3_8b_awq_single_token_layer_calls():
    _decode_layer_calls(np.random.default_rng(8), 1, 4096, 6144, 4096, 28672, 14336, True, "8B awq M=1")


def test_llama3_8b_gptq_single_token_layer_calls():
    _decode_layer_calls(np.random.default_rng(9), 1, 4096, 6144, 4096, 28672, 14336, False, "8B gptq M=1")


@pytest.mark.parametrize("M", [1, 8])
def test_llama3_70b_tp8_rank_layer_calls(M):
    """Per-rank shapes of Llama-3-70B split over 8 GPUs: qkv 8192 x 1280, o 1024 x 8192, gate_up
    8192 x 7168, down 3584 x 8192."""
    _decode_layer_calls(np.random.default_rng(70 + M), M, 8192, 1280, 1024, 7168, 3584, False, f"70B tp8 M={M}")


# ---------------------------------------------------------------------------
# Mixtral-8x7B-sized expert layer
# ---------------------------------------------------------------------------
def _expert_ref(a_bits, w_u8, s_bits, z_u8, G):
    """One expert GEMM on the oracle.  The MoE layout's bytes (two nibbles along K, even k low) are the
    little-endian bytes of GPTQ words, so [N, K/2] uint8 viewed as int32 [N, K/8] and transposed IS the
    GPTQ tensor."""
    qw = np.ascontiguousarray(np.ascontiguousarray(w_u8).view(np.int32).T)
    s = np.ascontiguousarray(s_bits.T)
    if z_u8 is None:
        return orc.w4a16_gemm(a_bits, qw, s, None, None, G=G)
    n2, g = z_u8.shape
    z = np.empty((2 * n2, g), dtype=np.int32)
    z[0::2], z[1::2] = z_u8 & 0xF, z_u8 >> 4
    return orc.w4a16_gemm(a_bits, qw, s, orc.pack_cols(np.ascontiguousarray(z.T)), None, G=G, zp_mode=orc.ZP_NSEQ)


@pytest.mark.parametrize("K,N", [(8192, 1280), (1024, 8192), (8192, 7168), (3584, 8192)])
def test_llama3_70b_tp8_rank_prompt_gemms(K, N):
    """The per-rank projections of Llama-3-70B over 8 GPUs on a 128-token prompt: the strip width the library
    chooses for the shape, the prompt kernel's form for it (K-slices, two-strip form, ...), against the oracle."""
    M, G = 128, 128
    rng = np.random.default_rng(K + N)
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    a = orc.f32_to_h16(0.25 * rng.standard_normal((M, K)).astype(np.float32))
    s = orc.f32_to_h16((0.05 * rng.random((K // G, N))).astype(np.float32))
    qw = orc.gptq_pack(q)
    ref = orc.w4a16_gemm(a, qw, s, None, None, G=G)
    w = ops.w4a16_repack(torch.from_numpy(qw).to(DEV), to_dev(s), None, G)
    out = ops.w4a16_gemm(to_dev(a), w)   # the route a prompt of 128 tokens takes (tiled kernel where the shape allows)
    np.testing.assert_allclose(f32(to_bits(out)), f32(ref), rtol=1e-2, atol=1e-2)
    assert (to_bits(out) == ref).mean() > 0.98


def test_mixtral_sized_expert_layer(monkeypatch):
    from mi355q.plugin import fused_experts, moe as moe_mod
    rng = np.random.default_rng(87)
    E, H, I, G, topk, Tmax = 8, 4096, 14336, 128, 2, 8

    def mk(n, k):
        gain = 1.0 / (21.25 * k) ** 0.5
        return (rng.integers(0, 256, size=(E, n, k // 2), dtype=np.uint8),
                orc.f32_to_h16((gain * (0.5 + rng.random((E, n, k // G)))).astype(np.float32)),
                rng.integers(0, 256, size=(E, n // 2, k // G), dtype=np.uint8))

    w1, w2 = mk(2 * I, H), mk(H, I)

    def dev(w, pair_halves):
        return ops.moe_w4a16_repack(torch.from_numpy(w[0]).to(DEV), to_dev(w[1]), torch.from_numpy(w[2]).to(DEV), G,
                                    pair_halves=pair_halves)
    pair = ops.w4a16_gate_up_silu_supported(H, 2 * I, G)
    d13, d2 = dev(w1, pair), dev(w2, False)
    a = orc.f32_to_h16(rng.standard_normal((Tmax, H)).astype(np.float32))
    ids = np.stack([rng.permutation(E)[:topk] for _ in range(Tmax)]).astype(np.int32)
    tw = rng.random((Tmax, topk)).astype(np.float32)
    tw /= tw.sum(axis=1, keepdims=True)
    # the reference, token by token (independent of the batch a token arrives in)
    acc = np.zeros((Tmax, H), dtype=np.float32)
    for t in range(Tmax):
        for j in range(topk):
            e = int(ids[t, j])
            h = _expert_ref(a[t:t + 1], w1[0][e], w1[1][e], w1[2][e], G)
            y = _expert_ref(orc.silu_and_mul(h), w2[0][e], w2[1][e], w2[2][e], G)
            acc[t] += tw[t, j] * f32(y[0])
    ref = f32(orc.f32_to_h16(acc))
    for T in (1, 2, 8):
        for route in ("slots", "all_experts", "grouped", None):
            monkeypatch.setattr(moe_mod, "FORCE_ROUTE", route)
            out = fused_experts(to_dev(a[:T]), d13, d2, torch.from_numpy(tw[:T]).to(DEV), torch.from_numpy(ids[:T]).to(DEV))
            np.testing.assert_allclose(f32(to_bits(out)), ref[:T], rtol=1e-2, atol=2e-2, err_msg=f"T={T} route={route}")


# ---------------------------------------------------------------------------
# end to end at Llama-3-8B widths
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("quant,kv", [("gptq", "auto"), ("awq", "fp8")])
def test_greedy_tokens_at_8b_widths_match_oracle(quant, kv):
    """Two decoder layers with the hidden / intermediate / head / vocabulary sizes of Llama-3-8B: prefill
    of 8 prompt tokens and 3 graph-captured decode steps against the oracle model, token for token
    (a sequence may leave the oracle's path only at a logit near-tie, see assert_tokens_match)."""
    from mi355q.model import LlamaConfig, LlamaRunner
    from oracle_model import assert_tokens_match, oracle_generate
    cfg = LlamaConfig("llama3-8b-2layer", 4096, 14336, 2, 32, 8, 128, 128256)
    runner = LlamaRunner(cfg, quant=quant, kv_cache_dtype=kv, device=DEV, max_batch=1, max_seq_len=64,
                         keep_checkpoint=True, seed=11)
    assert runner.layers[0]["gate_up"].w.pair_halves
    prompt = torch.randint(0, cfg.vocab, (1, 8), generator=torch.Generator().manual_seed(5))
    got = [runner.prefill(prompt).cpu().tolist()]
    runner.capture()
    for _ in range(3):
        runner.decode_step()
        got.append(runner.tokens.cpu().tolist())
    want, logits = oracle_generate(runner, prompt, 4, with_logits=True)
    assert_tokens_match(got, want, logits)
    assert got[0] == want[0]


# ---------------------------------------------------------------------------
# reference edge cases: tests/kernels/core/test_layernorm.py:21-30, test_pos_encoding.py:14-124
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("tokens,hidden", [(83, 769), (83, 5120), (4096, 769), (4096, 5120), (7, 5125), (7, 8)])
def test_rmsnorm_reference_shapes(tokens, hidden):
    rng = np.random.default_rng(tokens + hidden)
    x = orc.f32_to_h16(rng.standard_normal((tokens, hidden)).astype(np.float32))
    r = orc.f32_to_h16(rng.standard_normal((tokens, hidden)).astype(np.float32))
    w = orc.f32_to_h16((1 + 0.1 * rng.standard_normal(hidden)).astype(np.float32))
    y = torch.empty((tokens, hidden), dtype=torch.bfloat16, device=DEV)
    ops.rms_norm(y, to_dev(x), to_dev(w), EPS)
    ref = orc.rms_norm(x, w, EPS)
    assert (to_bits(y) == ref).mean() > 0.995       # fp32 sum order differs; one rounding either way
    np.testing.assert_allclose(f32(to_bits(y)), f32(ref), rtol=1.6e-2, atol=1e-3)   # vllm/ir/tolerances.py bf16
    x2, r2 = to_dev(x), to_dev(r)
    ops.fused_add_rms_norm(x2, r2, to_dev(w), EPS)
    ref_x, ref_r = orc.fused_add_rms_norm(x, r, w, EPS)
    assert np.array_equal(to_bits(r2), ref_r)
    assert (to_bits(x2) == ref_x).mean() > 0.995
    np.testing.assert_allclose(f32(to_bits(x2)), f32(ref_x), rtol=1.6e-2, atol=1e-3)


@pytest.mark.parametrize("neox", [True, False])
@pytest.mark.parametrize("head,rot", [(80, 80), (120, 120), (256, 256), (256, 64), (80, 32)])
def test_rope_reference_head_sizes(head, rot, neox):
    rng = np.random.default_rng(head + rot)
    T, nh, kvh, max_pos = 13, 5, 1, 128
    q = orc.f32_to_h16(rng.standard_normal((T, nh * head)).astype(np.float32))
    k = orc.f32_to_h16(rng.standard_normal((T, kvh * head)).astype(np.float32))
    pos = rng.integers(0, max_pos, size=T).astype(np.int64)
    inv = 1.0 / (10000.0 ** (np.arange(0, rot, 2, dtype=np.float32) / rot))
    fr = np.arange(max_pos, dtype=np.float32)[:, None] * inv[None, :]
    cache = orc.f32_to_h16(np.concatenate([np.cos(fr), np.sin(fr)], axis=1).astype(np.float32))
    qd, kd = to_dev(q), to_dev(k)
    ops.rotary_embedding(torch.from_numpy(pos).to(DEV), qd, kd, head, to_dev(cache), neox)
    qo, ko = orc.rotary_embedding(pos, q, k, cache, head, neox)
    assert np.array_equal(to_bits(qd), qo) and np.array_equal(to_bits(kd), ko)
    # untouched tail of a partial rotation
    if rot < head:
        assert np.array_equal(to_bits(qd).reshape(T, nh, head)[:, :, rot:], q.reshape(T, nh, head)[:, :, rot:])
