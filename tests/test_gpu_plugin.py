"""The plug-in surface end to end on the GPU: checkpoint tensors -> weight_loader ->
process_weights_after_loading -> quant_method.apply, the attention layer/backend, the layer
wrappers, the sampler and the dispatcher ops, each checked against the CPU oracle."""
import numpy as np
import pytest
import torch

import oracle as orc
from mi355q import ops
from mi355q.plugin import AutoAWQConfig, AutoGPTQConfig, register
from mi355q.plugin.attention import (Attention, CommonAttentionMetadata, Mi355AttentionBackend,
                                     Mi355AttentionMetadataBuilder)
from mi355q.plugin.layers import RMSNorm, SiluAndMul, get_rope
from mi355q.plugin.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
from mi355q.plugin.sampler import Sampler, SamplingMetadata

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def to_dev(bits, dt=torch.bfloat16):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(dt).to(DEV)


def to_bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def f32(bits, dt=orc.BF16):
    return orc.h16_to_f32(bits, dt)


@pytest.fixture(autouse=True)
def _kernel_route(monkeypatch):
    monkeypatch.setattr(ops, "DEQUANT_MATMUL_MIN_TOKENS", 1 << 30)


def test_gptq_linear_methods_match_oracle():
    rng = np.random.default_rng(0)
    hidden, inter, G, T = 512, 1024, 128, 5
    cfg = AutoGPTQConfig.from_config({"bits": 4, "group_size": G, "desc_act": False, "sym": True})
    with torch.device(DEV):   # parameters are created on the device, as vLLM's model loader does
        gu = MergedColumnParallelLinear(hidden, [inter, inter], bias=False, params_dtype=torch.bfloat16,
                                        quant_config=cfg, prefix="mlp.gate_up_proj", return_bias=False)
        down = RowParallelLinear(inter, hidden, bias=True, params_dtype=torch.bfloat16, quant_config=cfg,
                                 prefix="mlp.down_proj", return_bias=False)
    ck = {}
    for name, (K, N) in {"gate": (hidden, inter), "up": (hidden, inter), "down": (inter, hidden)}.items():
        ck[name] = (rng.integers(0, 16, size=(K, N), dtype=np.int32),
                    orc.f32_to_h16((0.02 * rng.random((K // G, N)) + 0.005).astype(np.float32)))
    bias = orc.f32_to_h16(rng.standard_normal(hidden).astype(np.float32))
    for i, n in enumerate(("gate", "up")):
        gu.weight_loader(gu.qweight, torch.from_numpy(orc.gptq_pack(ck[n][0])), i)
        gu.weight_loader(gu.scales, to_dev(ck[n][1]).cpu(), i)
    down.weight_loader(down.qweight, torch.from_numpy(orc.gptq_pack(ck["down"][0])))
    down.weight_loader(down.scales, to_dev(ck["down"][1]).cpu())
    down.weight_loader(down.bias, to_dev(bias).cpu())
    gu.process_weights_after_loading()
    down.process_weights_after_loading()
    assert gu.qweight.is_cuda and gu.mi355q_weight.packed.data_ptr() == gu.qweight.data_ptr()

    x = orc.f32_to_h16(rng.standard_normal((T, hidden)).astype(np.float32))
    g = orc.w4a16_gemm(x, orc.gptq_pack(ck["gate"][0]), ck["gate"][1], None, None, G=G)
    u = orc.w4a16_gemm(x, orc.gptq_pack(ck["up"][0]), ck["up"][1], None, None, G=G)
    ref_gu = np.concatenate([g, u], axis=1)
    got_gu = gu(to_dev(x))
    assert (to_bits(got_gu) == ref_gu).mean() > 0.98
    np.testing.assert_allclose(f32(to_bits(got_gu)), f32(ref_gu), rtol=1e-2, atol=1e-2)
    act = SiluAndMul()(got_gu)
    ref_act = orc.silu_and_mul(to_bits(got_gu))
    assert np.array_equal(to_bits(act), ref_act)
    ref = orc.w4a16_gemm(ref_act, orc.gptq_pack(ck["down"][0]), ck["down"][1], None, bias, G=G)
    got = down(act)
    np.testing.assert_allclose(f32(to_bits(got)), f32(ref), rtol=1e-2, atol=1e-2)
    assert (to_bits(got) == ref).mean() > 0.98
    # the fused prologue route of the kernel object gives the same numbers as SiluAndMul + apply
    fused = down.quant_method.kernel.apply_weights_silu_mul(down, got_gu, down.bias)
    np.testing.assert_allclose(f32(to_bits(fused)), f32(ref), rtol=1e-2, atol=1e-2)


def test_auto_round_checkpoint_runs_on_the_gptq_route():
    """An AutoRound (`quant_method: "auto-round"`, auto_gptq packing) layer: INCConfig -> INCLinearMethod
    -> the GPTQ method and kernel; same numbers as the oracle, a listed 16-bit layer stays a plain linear."""
    from mi355q.plugin import INCConfig, INCLinearMethod
    from mi355q.plugin.linear import UnquantizedLinearMethod
    rng = np.random.default_rng(3)
    hidden, inter, G, T = 512, 1536, 128, 3
    cfg = INCConfig.from_config({"quant_method": "auto-round", "bits": 4, "group_size": G, "sym": True,
                                 "packing_format": "auto_round:auto_gptq",
                                 "extra_config": {"model.layers.0.mlp.gate": {"bits": 16, "data_type": "float"}}})
    with torch.device(DEV):
        down = RowParallelLinear(inter, hidden, bias=False, params_dtype=torch.bfloat16, quant_config=cfg,
                                 prefix="model.layers.0.mlp.down_proj", return_bias=False)
        gate = RowParallelLinear(hidden, 64, bias=False, params_dtype=torch.bfloat16, quant_config=cfg,
                                 prefix="model.layers.0.mlp.gate", return_bias=False)
    assert isinstance(down.quant_method, INCLinearMethod)
    assert isinstance(gate.quant_method, UnquantizedLinearMethod)
    q = rng.integers(0, 16, size=(inter, hidden), dtype=np.int32)
    s = orc.f32_to_h16((0.02 * rng.random((inter // G, hidden)) + 0.005).astype(np.float32))
    down.weight_loader(down.qweight, torch.from_numpy(orc.gptq_pack(q)))
    down.weight_loader(down.scales, to_dev(s).cpu())
    down.process_weights_after_loading()
    x = orc.f32_to_h16(rng.standard_normal((T, inter)).astype(np.float32))
    ref = orc.w4a16_gemm(x, orc.gptq_pack(q), s, None, None, G=G)
    got = down(to_dev(x))
    np.testing.assert_allclose(f32(to_bits(got)), f32(ref), rtol=1e-2, atol=1e-2)
    assert (to_bits(got) == ref).mean() > 0.98


def test_gptq_act_order_checkpoint():
    """desc_act=True: row k of the checkpoint belongs to group g_idx[k].  Reference semantics
    (q_gemm.cu:189-228: weights gathered through the sorted permutation, scales by group):
    W[k, n] = (q[k, n] - 8) * s[g_idx[k], n]; checked against that definition computed directly."""
    rng = np.random.default_rng(8)
    hidden, out, G, T = 1024, 512, 128, 4
    cfg = AutoGPTQConfig.from_config({"bits": 4, "group_size": G, "desc_act": True, "sym": True})
    with torch.device(DEV):
        lin = MergedColumnParallelLinear(hidden, [out // 2, out // 2], bias=False, params_dtype=torch.bfloat16,
                                         quant_config=cfg, prefix="mlp.gate_up_proj", return_bias=False)
    q = rng.integers(0, 16, size=(hidden, out), dtype=np.int32)
    s = orc.f32_to_h16((0.02 * rng.random((hidden // G, out)) + 0.005).astype(np.float32))
    g_idx = rng.permutation(np.arange(hidden) // G).astype(np.int32)
    half = out // 2
    for i in range(2):
        lin.weight_loader(lin.qweight, torch.from_numpy(orc.gptq_pack(q[:, i * half:(i + 1) * half])), i)
        lin.weight_loader(lin.scales, to_dev(s[:, i * half:(i + 1) * half]).cpu(), i)
    lin.weight_loader(lin.g_idx, torch.from_numpy(g_idx))
    lin.process_weights_after_loading()
    assert lin.mi355q_perm is not None
    x = orc.f32_to_h16(rng.standard_normal((T, hidden)).astype(np.float32))
    W = (q - 8).astype(np.float64) * f32(s)[g_idx].astype(np.float64)
    ref = f32(x).astype(np.float64) @ W
    got = f32(to_bits(lin(to_dev(x))))
    np.testing.assert_allclose(got, ref, rtol=1e-2, atol=2e-2)
    # the fused-prologue entry points of the kernel object serve reordered layers too
    k = lin.quant_method.kernel
    res = orc.f32_to_h16(rng.standard_normal((T, hidden)).astype(np.float32))
    nw = orc.f32_to_h16((1 + 0.1 * rng.standard_normal(hidden)).astype(np.float32))
    y, r2 = k.apply_weights_add_rms_norm(lin, to_dev(x), to_dev(res), to_dev(nw), 1e-5)
    xn, rn = orc.fused_add_rms_norm(x, res, nw, 1e-5)
    assert np.array_equal(to_bits(r2), rn)
    np.testing.assert_allclose(f32(to_bits(y)), f32(xn).astype(np.float64) @ W, rtol=1e-2, atol=2e-2)
    # a trivial g_idx (checkpoint written without reordering) takes the plain route
    with torch.device(DEV):
        lin2 = RowParallelLinear(hidden, out, bias=False, params_dtype=torch.bfloat16, quant_config=cfg,
                                 prefix="mlp.down_proj", return_bias=False)
    lin2.weight_loader(lin2.qweight, torch.from_numpy(orc.gptq_pack(q)))
    lin2.weight_loader(lin2.scales, to_dev(s).cpu())
    lin2.weight_loader(lin2.g_idx, torch.from_numpy((np.arange(hidden) // G).astype(np.int32)))
    lin2.process_weights_after_loading()
    assert lin2.mi355q_perm is None
    ref2 = orc.w4a16_gemm(x, orc.gptq_pack(q), s, None, None, G=G)
    assert (to_bits(lin2(to_dev(x))) == ref2).mean() > 0.98


def test_gptq_act_order_checkpoint_with_k_sharded(monkeypatch):
    """desc_act on a row-parallel layer at TP = 4 (auto_gptq.py:367-378: g_idx sharded, scales held in full):
    every rank's partial product against the definition W[k, n] = (q[k, n] - 8) * s[g_idx[k], n] on its rows,
    and the ranks' partial products summed against the unsharded layer's output."""
    from mi355q.plugin.mp_kernel import MI355XW4A16LinearKernel, MPLinearLayerConfig
    from mi355q.plugin.scalar_type import scalar_types
    rng = np.random.default_rng(23)
    K_full, N, G, TP, T = 2048, 256, 128, 4, 3
    K = K_full // TP
    g_full = rng.permutation(np.arange(K_full) // G).astype(np.int32)
    q = rng.integers(0, 16, size=(K_full, N), dtype=np.int32)
    s = orc.f32_to_h16((0.01 + 0.02 * rng.random((K_full // G, N))).astype(np.float32))
    x = orc.f32_to_h16(rng.standard_normal((T, K_full)).astype(np.float32))
    W = (q - 8).astype(np.float64) * f32(s)[g_full].astype(np.float64)
    total = np.zeros((T, N))
    for rank in range(TP):
        rows = slice(rank * K, (rank + 1) * K)
        cfg = MPLinearLayerConfig((K_full, N), (K, N), scalar_types.uint4b8, torch.bfloat16, G, False, True)
        layer = torch.nn.Module()
        for name, t in (("qweight", torch.from_numpy(orc.gptq_pack(q[rows])).to(DEV)), ("scales", to_dev(s)),
                        ("qzeros", torch.zeros((K_full // G, N // 8), dtype=torch.int32, device=DEV)),
                        ("g_idx", torch.from_numpy(g_full[rows]).to(DEV))):
            layer.register_parameter(name, torch.nn.Parameter(t, requires_grad=False))
        kernel = MI355XW4A16LinearKernel(cfg, "qweight", "scales", "qzeros", "g_idx")
        kernel.process_weights_after_loading(layer)
        got = f32(to_bits(kernel.apply_weights(layer, to_dev(x[:, rows]))))
        want = f32(x[:, rows]).astype(np.float64) @ W[rows]
        np.testing.assert_allclose(got, want, rtol=1e-2, atol=2e-2)
        total += got
    np.testing.assert_allclose(total, f32(x).astype(np.float64) @ W, rtol=2e-2, atol=5e-2)


def test_gptq_8bit_checkpoint():
    """bits=8 GPTQ config -> uint8b128 -> the 8-bit kernel class; bias added to the rounded product."""
    rng = np.random.default_rng(15)
    hidden, out_f, G, T = 1024, 768, 128, 3
    cfg = AutoGPTQConfig.from_config({"bits": 8, "group_size": G, "desc_act": False, "sym": True})
    with torch.device(DEV):
        lin = RowParallelLinear(hidden, out_f, bias=True, params_dtype=torch.bfloat16, quant_config=cfg,
                                prefix="mlp.down_proj", return_bias=False)
    assert type(lin.quant_method.kernel).__name__ == "MI355XW8A16LinearKernel"
    assert lin.qweight.shape == (hidden // 4, out_f)
    q = rng.integers(0, 256, size=(hidden, out_f), dtype=np.int32)
    s = orc.f32_to_h16((0.004 * rng.random((hidden // G, out_f)) + 0.001).astype(np.float32))
    bias = orc.f32_to_h16(rng.standard_normal(out_f).astype(np.float32))
    lin.weight_loader(lin.qweight, torch.from_numpy(orc.gptq8_pack(q)))
    lin.weight_loader(lin.scales, to_dev(s).cpu())
    lin.weight_loader(lin.bias, to_dev(bias).cpu())
    lin.process_weights_after_loading()
    x = orc.f32_to_h16(rng.standard_normal((T, hidden)).astype(np.float32))
    ref = orc.gptq8_gemm(x, q, s, G, bias)
    got = lin(to_dev(x))
    np.testing.assert_allclose(f32(to_bits(got)), f32(ref), rtol=1e-2, atol=1e-2)
    assert (to_bits(got) == ref).mean() > 0.97


def test_awq_linear_method_matches_oracle():
    rng = np.random.default_rng(1)
    K, N, G, T = 512, 768, 128, 3
    cfg = AutoAWQConfig.from_config({"w_bit": 4, "q_group_size": G, "zero_point": True})
    with torch.device(DEV):
        lin = RowParallelLinear(K, N, bias=False, params_dtype=torch.float16, quant_config=cfg, prefix="o_proj",
                                return_bias=False)
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    z = rng.integers(0, 16, size=(K // G, N), dtype=np.int32)
    s = orc.f32_to_h16((0.02 * rng.random((K // G, N)) + 0.005).astype(np.float32), orc.FP16)
    qw, qz = orc.pack_cols(q, awq_order=True), orc.pack_cols(z, awq_order=True)
    lin.weight_loader(lin.qweight, torch.from_numpy(qw))
    lin.weight_loader(lin.qzeros, torch.from_numpy(qz))
    lin.weight_loader(lin.scales, to_dev(s, torch.float16).cpu())
    lin.process_weights_after_loading()
    x = orc.f32_to_h16(rng.standard_normal((T, K)).astype(np.float32), orc.FP16)
    ref = orc.w4a16_gemm(x, qw, s, qz, None, G=G, layout=orc.W_AWQ, zp_mode=orc.ZP_AWQ, dt=orc.FP16)
    got = lin(to_dev(x, torch.float16))
    np.testing.assert_allclose(f32(to_bits(got), orc.FP16), f32(ref, orc.FP16), rtol=1e-2, atol=1e-2)
    # fp16 keeps 3 more mantissa bits than bf16, so fp32 summation-order differences show more often
    assert (to_bits(got) == ref).mean() > 0.9


@pytest.mark.parametrize("kv_dtype", ["auto", "fp8"])
def test_attention_layer_prefill_then_decode(kv_dtype):
    """Attention.forward = cache update + attention, driven by backend metadata, for a prompt
    step followed by a decode step; compared with dense causal attention of the oracle."""
    rng = np.random.default_rng(3)
    H, KVH, D, bs, nblk = 8, 2, 128, 16, 32
    fp8 = kv_dtype == "fp8"
    lens = [21, 50]
    S, scale = len(lens), 1.0 / D ** 0.5
    with torch.device(DEV):
        attn = Attention(H, D, scale, KVH, cache_dtype=kv_dtype, prefix="model.layers.0.self_attn.attn")
    ks = vs = 1.0
    if fp8:   # scales arrive from the checkpoint
        ks, vs = 1 / 32, 1 / 16
        attn.k_scale.weight_loader(attn.k_scale, torch.tensor(ks))
        attn.v_scale.weight_loader(attn.v_scale, torch.tensor([vs]))
    attn.process_weights_after_loading()
    assert not hasattr(attn, "k_scale") and abs(attn._k_scale_float - ks) < 1e-9 and attn._v_scale.is_cuda
    shape = Mi355AttentionBackend.get_kv_cache_shape(nblk, bs, KVH, D)
    attn.kv_cache = torch.zeros(shape, dtype=torch.uint8 if fp8 else torch.bfloat16, device=DEV)
    builder = Mi355AttentionMetadataBuilder(bs, DEV)
    maxb = (max(lens) + 1 + bs - 1) // bs
    bt = rng.permutation(nblk)[: S * maxb].reshape(S, maxb).astype(np.int32)

    def slots(s, lo, hi):
        return [int(bt[s, j // bs]) * bs + j % bs for j in range(lo, hi)]

    def dq(bits, sc):
        if not fp8:
            return bits
        return orc.f32_to_h16(orc.e4m3_to_f32(orc.f32_to_e4m3(orc.h16_to_f32(bits) / np.float32(sc))) * np.float32(sc))

    qs = [orc.f32_to_h16(rng.uniform(-1, 1, (L + 1, H, D)).astype(np.float32)) for L in lens]
    kf = [orc.f32_to_h16(rng.uniform(-1, 1, (L + 1, KVH, D)).astype(np.float32)) for L in lens]
    vf = [orc.f32_to_h16(rng.uniform(-1, 1, (L + 1, KVH, D)).astype(np.float32)) for L in lens]
    refs = [orc.prefill_attention(qs[s], dq(kf[s], ks), dq(vf[s], vs), scale) for s in range(S)]

    # prompt step
    T = sum(lens)
    md = builder.build(0, CommonAttentionMetadata(
        query_start_loc=torch.tensor(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), device=DEV),
        seq_lens=torch.tensor(lens, dtype=torch.int32, device=DEV), num_actual_tokens=T, max_query_len=max(lens),
        max_seq_len=max(lens), block_table_tensor=torch.from_numpy(bt).to(DEV),
        slot_mapping=torch.tensor(sum((slots(s, 0, lens[s]) for s in range(S)), []), dtype=torch.int64, device=DEV)))
    q = to_dev(np.concatenate([qs[s][: lens[s]] for s in range(S)])).view(T, H * D)
    k = to_dev(np.concatenate([kf[s][: lens[s]] for s in range(S)])).view(T, KVH * D)
    v = to_dev(np.concatenate([vf[s][: lens[s]] for s in range(S)])).view(T, KVH * D)
    out = attn(q, k, v, md)
    ref = np.concatenate([refs[s][: lens[s]] for s in range(S)]).reshape(T, H * D)
    atol = 1e-2 if fp8 else 2e-3
    np.testing.assert_allclose(f32(to_bits(out)), f32(ref), rtol=1e-5, atol=atol)

    # decode step
    md = builder.build(0, CommonAttentionMetadata(
        query_start_loc=torch.arange(S + 1, dtype=torch.int32, device=DEV),
        seq_lens=torch.tensor([L + 1 for L in lens], dtype=torch.int32, device=DEV), num_actual_tokens=S,
        max_query_len=1, max_seq_len=max(lens) + 1, block_table_tensor=torch.from_numpy(bt).to(DEV),
        slot_mapping=torch.tensor([slots(s, lens[s], lens[s] + 1)[0] for s in range(S)], dtype=torch.int64, device=DEV)))
    q = to_dev(np.stack([qs[s][lens[s]] for s in range(S)])).view(S, H * D)
    k = to_dev(np.stack([kf[s][lens[s]] for s in range(S)])).view(S, KVH * D)
    v = to_dev(np.stack([vf[s][lens[s]] for s in range(S)])).view(S, KVH * D)
    out = attn(q, k, v, md)
    ref = np.stack([refs[s][lens[s]] for s in range(S)]).reshape(S, H * D)
    np.testing.assert_allclose(f32(to_bits(out)), f32(ref), rtol=1e-5, atol=atol)
    # memory-profiling call
    assert float(attn.impl.forward(attn, q, k, v, attn.kv_cache, None, out.clone()).abs().sum()) == 0.0
    # the backend's fused RoPE + cache-update hook = rotary_embedding, then do_kv_cache_update
    assert attn.impl.fused_rope_kvcache_supported()
    pos = torch.tensor([L for L in lens], dtype=torch.int64, device=DEV)
    inv = 1.0 / (10000.0 ** (np.arange(0, D, 2, dtype=np.float32) / D))
    fr = np.arange(64, dtype=np.float32)[:, None] * inv[None, :]
    cos_sin = to_dev(orc.f32_to_h16(np.concatenate([np.cos(fr), np.sin(fr)], axis=1).astype(np.float32)))
    q1, k1, q2, k2 = q.clone(), k.clone(), q.clone(), k.clone()
    cache1, cache2 = torch.zeros_like(attn.kv_cache), torch.zeros_like(attn.kv_cache)
    ops.rotary_embedding(pos, q1, k1, D, cos_sin, True)
    attn.impl.do_kv_cache_update(attn, k1, v, cache1, md.slot_mapping)
    attn.impl.do_rope_and_kv_cache_update(attn, q2, k2, v, pos, cos_sin, True, cache2, md.slot_mapping)
    assert torch.equal(q1, q2) and torch.equal(k1, k2) and torch.equal(cache1, cache2)
    assert int((cache2 != 0).sum()) > 0


def test_layer_wrappers_match_oracle():
    rng = np.random.default_rng(4)
    T, hidden, D, H, KVH = 6, 1024, 128, 4, 2
    x = orc.f32_to_h16(rng.standard_normal((T, hidden)).astype(np.float32))
    r = orc.f32_to_h16(rng.standard_normal((T, hidden)).astype(np.float32))
    w = orc.f32_to_h16((1 + 0.1 * rng.standard_normal(hidden)).astype(np.float32))
    norm = RMSNorm(hidden, eps=1e-5, dtype=torch.bfloat16).to(DEV)
    norm.weight.data.copy_(to_dev(w))
    assert np.array_equal(to_bits(norm(to_dev(x))), orc.rms_norm(x, w, 1e-5))
    xo, ro = norm(to_dev(x), to_dev(r))
    ref_x, ref_r = orc.fused_add_rms_norm(x, r, w, 1e-5)
    assert np.array_equal(to_bits(xo), ref_x) and np.array_equal(to_bits(ro), ref_r)

    rope = get_rope(D, 512, True, {"rope_theta": 500000.0, "rope_type": "llama3", "factor": 8.0,
                                   "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                   "original_max_position_embeddings": 128}, dtype=torch.bfloat16)
    plain = get_rope(D, 512, True, {"rope_theta": 500000.0}, dtype=torch.bfloat16)
    # highest frequency untouched, lowest divided by the factor
    inv_l3, inv = rope._compute_inv_freq(rope.base), plain._compute_inv_freq(plain.base)
    assert inv_l3[0] == inv[0] and abs(float(inv_l3[-1] / inv[-1]) - 1 / 8) < 1e-6
    pos = rng.integers(0, 512, size=T).astype(np.int64)
    q = orc.f32_to_h16(rng.standard_normal((T, H * D)).astype(np.float32))
    k = orc.f32_to_h16(rng.standard_normal((T, KVH * D)).astype(np.float32))
    cache = to_bits(rope.cos_sin_cache)
    ref_q, ref_k = orc.rotary_embedding(pos, q, k, cache, D, True)
    qd, kd = rope(torch.from_numpy(pos).to(DEV), to_dev(q), to_dev(k))
    assert np.array_equal(to_bits(qd), ref_q) and np.array_equal(to_bits(kd), ref_k)


def test_sampler_greedy_penalties_and_random():
    rng = np.random.default_rng(5)
    B, V = 4, 4096
    logits = orc.f32_to_h16((3 * rng.standard_normal((B, V))).astype(np.float32))
    logits[1, 77] = logits[1, 4000] = orc.f32_to_h16(np.float32([50.0]))[0]   # tie: first index wins
    sampler = Sampler()
    out = sampler(to_dev(logits), SamplingMetadata())
    assert out.sampled_token_ids.dtype == torch.int32 and out.sampled_token_ids.shape == (B, 1)
    assert out.sampled_token_ids.view(-1).tolist() == orc.argmax_h16(logits).tolist()
    assert out.sampled_token_ids[1, 0] == 77
    # repetition penalty moves the arg-max off an already generated token
    lf = orc.h16_to_f32(logits)
    pm = np.zeros((B, V), dtype=bool)
    om = np.zeros((B, V), dtype=bool)
    om[1, 77] = om[1, 4000] = True
    pen = np.full(B, 100.0, dtype=np.float32)
    ref = orc.argmax_f32(orc.apply_repetition_penalties(lf.copy(), pm, om, pen))
    md = SamplingMetadata(no_penalties=False, repetition_penalties=torch.from_numpy(pen).to(DEV),
                          prompt_mask=torch.from_numpy(pm).to(DEV), output_mask=torch.from_numpy(om).to(DEV))
    got = sampler(to_dev(logits), md).sampled_token_ids.view(-1).tolist()
    assert got == ref.tolist() and got[1] not in (77, 4000)
    # random rows: top_k = 1 must reproduce greedy whatever the noise; greedy rows stay greedy
    temp = torch.tensor([1.0, 0.0, 0.7, 1.3], device=DEV)
    md = SamplingMetadata(temperature=temp, all_greedy=False, all_random=False,
                          top_k=torch.ones(B, dtype=torch.int64, device=DEV), top_p=torch.full((B,), 0.9, device=DEV),
                          generators={0: torch.Generator(device=DEV).manual_seed(1)})
    assert sampler(to_dev(logits), md).sampled_token_ids.view(-1).tolist() == orc.argmax_h16(logits).tolist()
    # top-p: only tokens inside the nucleus are ever drawn
    md = SamplingMetadata(temperature=torch.ones(B, device=DEV), all_greedy=False, all_random=True,
                          top_p=torch.full((B,), 0.5, device=DEV))
    probs = torch.softmax(torch.from_numpy(lf), -1)
    srt, idx = probs.sort(-1, descending=True)
    # tokens with EQUAL logits at the nucleus boundary are kept together by the HIP sampler (the reference
    # keeps whichever its sort happens to place last): row 1 has such a pair
    inside = []
    for b in range(B):
        n = int((srt[b].cumsum(0) < 0.5).sum()) + 1
        inside.append(set(idx[b, : int((srt[b] >= srt[b, n - 1]).sum())].tolist()))
    for _ in range(5):
        ids = sampler(to_dev(logits), md).sampled_token_ids.view(-1).tolist()
        assert all(ids[b] in inside[b] for b in range(B))


def test_dispatcher_ops_and_graph_capture():
    """torch.ops.mi355q.* reach the same kernels, and a captured graph of them replays."""
    register()
    rng = np.random.default_rng(6)
    K, N, G = 512, 256, 128
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    s = orc.f32_to_h16((0.02 * rng.random((K // G, N)) + 0.005).astype(np.float32))
    w = ops.w4a16_repack(torch.from_numpy(orc.gptq_pack(q)).to(DEV), to_dev(s), None, G)
    x = to_dev(orc.f32_to_h16(rng.standard_normal((2, K)).astype(np.float32)))
    direct = ops.w4a16_gemm(x, w)
    via = torch.ops.mi355q.w4a16_gemm(x, w.packed, w.scales, None, None, K, N, G, w.tile)
    assert torch.equal(direct, via)
    assert torch.equal(torch.ops.mi355q.w4a16_dequantize(w.packed, w.scales, None, K, N, G, w.tile), ops.w4a16_dequantize(w))
    out = torch.empty_like(direct)
    torch.ops.mi355q.rms_norm(out, direct, None, 1e-5)
    ref = torch.empty_like(direct)
    ops.rms_norm(ref, direct, None, 1e-5)
    assert torch.equal(out, ref)
    assert torch.equal(torch.ops.mi355q.argmax(direct), ops.argmax(direct))
    # shape-only fakes
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode() as mode:
        fx = mode.from_tensor(x)
        fo = torch.ops.mi355q.w4a16_gemm(fx, mode.from_tensor(w.packed), mode.from_tensor(w.scales), None, None, K, N, G, w.tile)
        assert fo.shape == (2, N) and fo.dtype == torch.bfloat16
    g = torch.cuda.CUDAGraph()
    static_x = x.clone()
    s0 = torch.cuda.Stream()
    s0.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s0):
        torch.ops.mi355q.w4a16_gemm(static_x, w.packed, w.scales, None, None, K, N, G, w.tile)
    torch.cuda.current_stream().wait_stream(s0)
    with torch.cuda.graph(g):
        y = torch.ops.mi355q.w4a16_gemm(static_x, w.packed, w.scales, None, None, K, N, G, w.tile)
    static_x.copy_(x * 2)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, ops.w4a16_gemm(x * 2, w))


@pytest.mark.parametrize("block", [None, [128, 128]])
def test_fp8_weight_only_linear_method(block):
    from mi355q.plugin import Fp8Config
    rng = np.random.default_rng(9)
    hidden, inter, T = 512, 768, 2
    cfg = Fp8Config(True, "dynamic", weight_block_size=block, linear_mode="w8a16")
    with torch.device(DEV):
        gu = MergedColumnParallelLinear(hidden, [inter, inter], bias=False, params_dtype=torch.bfloat16,
                                        quant_config=cfg, prefix="mlp.gate_up_proj", return_bias=False)
    ws, ss = [], []
    for i in range(2):
        w8 = orc.f32_to_e4m3((rng.standard_normal((inter, hidden)) * 30).astype(np.float32))
        gu.weight_loader(gu.weight, torch.from_numpy(w8).view(torch.float8_e4m3fn), i)
        if block is None:
            s = np.float32(0.002 * (i + 1))
            gu.weight_loader(gu.weight_scale, torch.tensor(s), i)
            ss.append(np.full(inter, s, dtype=np.float32))
        else:
            s = (0.001 + 0.002 * rng.random((inter // 128, hidden // 128))).astype(np.float32)
            gu.weight_loader(gu.weight_scale_inv, torch.from_numpy(s), i)
            ss.append(s)
        ws.append(w8)
    gu.process_weights_after_loading()
    x = orc.f32_to_h16(rng.standard_normal((T, hidden)).astype(np.float32))
    ref = orc.w8a16_gemm(x, np.concatenate(ws), np.concatenate(ss), None, block=None if block is None else tuple(block))
    got = gu(to_dev(x))
    np.testing.assert_allclose(f32(to_bits(got)), f32(ref), rtol=1e-2, atol=1e-2)
    assert (to_bits(got) == ref).mean() > 0.97


@pytest.mark.parametrize("T", [3, 70])
def test_fp8_block_w8a8_linear_method(T):
    """A block-quantised fp8 checkpoint ([128, 128] weight blocks) in w8a8 mode, the reference's W8A8BlockFp8LinearOp:
    activations quantised per (token, 128-wide group), every k block of the product scaled on its own.  Against the
    oracle's restatement of both steps on the same bytes, on the decode kernel (3 rows) and the tiled one (70)."""
    from mi355q.plugin import Fp8Config
    rng = np.random.default_rng(23)
    hidden, inter = 512, 768
    cfg = Fp8Config(True, "dynamic", weight_block_size=[128, 128], linear_mode="w8a8")
    with torch.device(DEV):
        gu = MergedColumnParallelLinear(hidden, [inter, inter], bias=False, params_dtype=torch.bfloat16,
                                        quant_config=cfg, prefix="mlp.gate_up_proj", return_bias=False)
    assert gu.quant_method.w8a8_block
    ws, ss = [], []
    for i in range(2):
        w8 = orc.f32_to_e4m3((rng.standard_normal((inter, hidden)) * 30).astype(np.float32))
        s = (0.001 + 0.002 * rng.random((inter // 128, hidden // 128))).astype(np.float32)
        gu.weight_loader(gu.weight, torch.from_numpy(w8).view(torch.float8_e4m3fn), i)
        gu.weight_loader(gu.weight_scale_inv, torch.from_numpy(s), i)
        ws.append(w8)
        ss.append(s)
    gu.process_weights_after_loading()
    x = orc.f32_to_h16(rng.standard_normal((T, hidden)).astype(np.float32))
    xq, xs = orc.per_token_group_quant_fp8(x, 128)
    ref = orc.w8a8_block_scaled_mm(xq, np.concatenate(ws), xs, np.concatenate(ss), [128, 128])
    got = gu(to_dev(x))
    np.testing.assert_allclose(f32(to_bits(got)), f32(ref), rtol=1e-2, atol=1e-2 * float(np.abs(f32(ref)).max()))
    assert (to_bits(got) == ref).mean() > 0.9


@pytest.mark.parametrize("scheme", ["dynamic", "static"])
def test_fp8_w8a8_linear_method(scheme):
    """The reference's ROCm route for per-tensor fp8 checkpoints: fused shards requantised to the largest
    weight scale, activations quantised per tensor (run-time scale or the checkpoint's input_scale), scaled
    fp8 x fp8 product.  Against the oracle's restatement of each step on the same bytes."""
    from mi355q.plugin import Fp8Config
    rng = np.random.default_rng(19)
    hidden, inter = 512, 384
    cfg = Fp8Config(True, scheme)
    with torch.device(DEV):
        gu = MergedColumnParallelLinear(hidden, [inter, inter], bias=True, params_dtype=torch.bfloat16,
                                        quant_config=cfg, prefix="mlp.gate_up_proj", return_bias=False)
    ws = [orc.f32_to_e4m3((rng.standard_normal((inter, hidden)) * 30).astype(np.float32)) for _ in range(2)]
    wscales, xscales = [0.004, 0.002], [0.02, 0.03]
    for i in range(2):
        gu.weight_loader(gu.weight, torch.from_numpy(ws[i]).view(torch.float8_e4m3fn), i)
        gu.weight_loader(gu.weight_scale, torch.tensor(wscales[i]), i)
        if scheme == "static":
            gu.weight_loader(gu.input_scale, torch.tensor(xscales[i]), i)
    bias = orc.f32_to_h16(rng.standard_normal(2 * inter).astype(np.float32) * 0.1)
    gu.bias.data.copy_(to_dev(bias))
    gu.process_weights_after_loading()
    # shard 1 requantised from scale 0.002 to 0.004 (through fp16, as per_tensor_dequantize does)
    dq = (torch.from_numpy(ws[1]).view(torch.float8_e4m3fn).to(torch.float16) * np.float32(0.002)).float().numpy()
    w1 = orc.static_scaled_fp8_quant(dq, np.array([0.004], np.float32), None, orc.FP32)
    w_all = np.concatenate([ws[0], w1])
    assert np.array_equal(gu.weight.data.view(torch.uint8).cpu().numpy(), w_all)
    assert gu.weight_scale.tolist() == [np.float32(0.004)]
    for T in (1, 3, 40, 150):
        x = orc.f32_to_h16(rng.standard_normal((T, hidden)).astype(np.float32))
        if scheme == "static":
            xs = np.array([0.03], np.float32)
            xq = orc.static_scaled_fp8_quant(x, xs)
        else:
            xq, xs = orc.dynamic_scaled_fp8_quant(x)
        ref = orc.fp8_scaled_mm(xq, w_all, xs, np.array([0.004], np.float32), bias)
        got = gu(to_dev(x))
        np.testing.assert_allclose(f32(to_bits(got)), f32(ref), rtol=1e-2, atol=1e-2)
        assert (to_bits(got) == ref).mean() > 0.97


def test_fp8_online_quantisation_of_16bit_checkpoint():
    from mi355q.plugin import Fp8Config
    rng = np.random.default_rng(10)
    K, N, T = 256, 128, 3
    with torch.device(DEV):
        lin = RowParallelLinear(K, N, bias=False, params_dtype=torch.bfloat16, quant_config=Fp8Config(False, "dynamic", linear_mode="w8a16"),
                                prefix="o_proj", return_bias=False)
    w = torch.from_numpy((rng.standard_normal((N, K)) * 0.05).astype(np.float32)).to(torch.bfloat16)
    lin.weight_loader(lin.weight, w)
    lin.process_weights_after_loading()
    assert lin.weight.dtype == torch.float8_e4m3fn and lin.weight_scale.shape == (N,)
    x = torch.from_numpy(rng.standard_normal((T, K)).astype(np.float32)).to(torch.bfloat16)
    ref = torch.nn.functional.linear(x.float(), w.float())
    got = lin(x.to(DEV)).float().cpu()
    # e4m3 keeps 3 mantissa bits: relative weight error <= 2^-4, averaged down over K terms
    assert float((got - ref).abs().max()) < 0.06 * float(ref.abs().max())


def test_fused_moe_layer_awq_checkpoint():
    """FusedMoE built from an AWQ MoE checkpoint config: loader conversion, repack, routing and
    the expert GEMMs, against the oracle restatement of the reference's torch_moe."""
    import moe_ref
    from mi355q.plugin import FusedMoE, MoeWNA16Config
    rng = np.random.default_rng(13)
    E, H, I, G, T, topk = 4, 256, 128, 64, 2, 2
    cfg = MoeWNA16Config.from_config({"quant_method": "awq", "bits": 4, "group_size": G, "zero_point": True})
    with torch.device(DEV):
        moe = FusedMoE(E, topk, H, I, params_dtype=torch.bfloat16, renormalize=True, quant_config=cfg,
                       prefix="model.layers.0.block_sparse_moe.experts")
    ck = {}
    for e in range(E):
        for sid, (K, N) in (("w1", (H, I)), ("w3", (H, I)), ("w2", (I, H))):
            q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
            z = rng.integers(0, 16, size=(K // G, N), dtype=np.int32)
            s = orc.f32_to_h16((0.02 * rng.random((K // G, N)) + 0.005).astype(np.float32))
            ck[e, sid] = (q, z, s)
            name = f"experts.{e}.{sid}"
            target = ("w2" if sid == "w2" else "w13")
            moe.weight_loader(getattr(moe, target + "_qweight"), torch.from_numpy(orc.pack_cols(q, awq_order=True)),
                              name + ".qweight", sid, e)
            moe.weight_loader(getattr(moe, target + "_qzeros"), torch.from_numpy(orc.pack_cols(z, awq_order=True)),
                              name + ".qzeros", sid, e)
            moe.weight_loader(getattr(moe, target + "_scales"), to_dev(s).cpu(), name + ".scales", sid, e)
    moe.process_weights_after_loading()
    x = orc.f32_to_h16(rng.standard_normal((T, H)).astype(np.float32))
    logits = orc.f32_to_h16(rng.standard_normal((T, E)).astype(np.float32))
    got = moe(to_dev(x), to_dev(logits))

    def packed(sids):   # byte layout of the reference loader from the logical values
        qs, zs, ss = [], [], []
        for e in range(E):
            q = np.concatenate([ck[e, s][0] for s in sids], axis=1)          # [K, N]
            z = np.concatenate([ck[e, s][1] for s in sids], axis=1)          # [K/G, N]
            sc = np.concatenate([ck[e, s][2] for s in sids], axis=1)
            qt, zt = q.T, z.T                                                # [N, K], [N, K/G]
            qs.append((qt[:, 1::2] * 16 + qt[:, ::2]).astype(np.uint8))
            zs.append((zt[1::2] * 16 + zt[::2]).astype(np.uint8))
            ss.append(np.ascontiguousarray(sc.T))
        return np.stack(qs), np.stack(ss), np.stack(zs)

    w, ids = moe_ref.route(logits, topk, renormalize=True)
    ref = moe_ref.moe(x, packed(("w1", "w3")), packed(("w2",)), w, ids, G)
    np.testing.assert_allclose(f32(to_bits(got)), f32(ref), rtol=1e-2, atol=2e-3)


# -- compressed-tensors checkpoints -------------------------------------------------------------------
_CT_FUSED = {"qkv_proj": ["q_proj", "k_proj", "v_proj"], "gate_up_proj": ["gate_proj", "up_proj"]}


def _ct_config(fmt="pack-quantized", **weights):
    from mi355q.plugin import CompressedTensorsConfig
    w = {"num_bits": 4, "type": "int", "symmetric": True, "strategy": "group", "group_size": 128, "dynamic": False}
    w.update(weights)
    cfg = CompressedTensorsConfig.from_config({
        "quant_method": "compressed-tensors", "format": fmt, "ignore": ["lm_head"],
        "config_groups": {"group_0": {"targets": ["Linear"], "weights": w, "input_activations": None}}})
    cfg.packed_modules_mapping = _CT_FUSED
    return cfg


@pytest.mark.parametrize("T,sym", [(2, True), (40, True), (3, False), (40, False)])
def test_fused_moe_layer_compressed_tensors_checkpoint(T, sym):
    """Routed experts of a pack-quantized compressed-tensors checkpoint (compressed_tensors_moe_wna16.py:52-330):
    transposed parameters, repack, routing and the expert GEMMs (slot route at 2-3 tokens, token sort + grouped
    GEMM at 40), symmetric and with zero points, against the oracle restatement of the reference's torch_moe
    on the same values."""
    import moe_ref
    from mi355q.plugin import FusedMoE, CompressedTensorsWNA16MoEMethod
    rng = np.random.default_rng(17 + T)
    E, H, I, G, topk = 4, 512, 256, 128, 2
    cfg = _ct_config(symmetric=sym)
    with torch.device(DEV):
        moe = FusedMoE(E, topk, H, I, params_dtype=torch.bfloat16, renormalize=True, quant_config=cfg,
                       prefix="model.layers.0.mlp.experts")
    assert isinstance(moe.quant_method, CompressedTensorsWNA16MoEMethod)

    def words(v):   # eight consecutive entries of the last dim per int32, least significant nibble first
        out = np.zeros(v.shape[:-1] + (v.shape[-1] // 8,), dtype=np.int64)
        for j in range(8):
            out |= v[..., j::8] << (4 * j)
        return torch.from_numpy(out.astype(np.uint32).view(np.int32))

    ck = {}
    for e in range(E):
        for sid, (K, N) in (("w1", (H, I)), ("w3", (H, I)), ("w2", (I, H))):
            q = rng.integers(0, 16, size=(N, K), dtype=np.int64)                       # stored nibble, output-major
            z = rng.integers(0, 16, size=(N, K // G), dtype=np.int64)
            s = orc.f32_to_h16((0.02 * rng.random((N, K // G)) + 0.005).astype(np.float32))
            ck[e, sid] = (q, s, z)
            tgt, name = ("w2" if sid == "w2" else "w13"), f"experts.{e}.{sid}"
            moe.weight_loader(getattr(moe, tgt + "_weight_packed"), words(q), name + ".weight_packed", sid, e)
            moe.weight_loader(getattr(moe, tgt + "_weight_scale"), to_dev(s).cpu(), name + ".weight_scale", sid, e)
            moe.weight_loader(getattr(moe, tgt + "_weight_shape"), torch.tensor([N, K]), name + ".weight_shape", sid, e)
            if not sym:   # [N/8, K/G]: packed along the output channels
                moe.weight_loader(getattr(moe, tgt + "_weight_zero_point"), words(np.ascontiguousarray(z.T)).t().contiguous(),
                                  name + ".weight_zero_point", sid, e)
    moe.process_weights_after_loading()
    x = orc.f32_to_h16(rng.standard_normal((T, H)).astype(np.float32))
    logits = orc.f32_to_h16(rng.standard_normal((T, E)).astype(np.float32))
    got = moe(to_dev(x), to_dev(logits))

    def packed(sids):   # the byte layouts the oracle reads: [E, N, K/2] even k low, zeros [E, N/2, K/G] even n low
        qs, ss, zs = [], [], []
        for e in range(E):
            q = np.concatenate([ck[e, s][0] for s in sids], axis=0)
            z = np.concatenate([ck[e, s][2] for s in sids], axis=0)
            qs.append((q[:, 1::2] * 16 + q[:, ::2]).astype(np.uint8))
            zs.append((z[1::2] * 16 + z[::2]).astype(np.uint8))
            ss.append(np.concatenate([ck[e, s][1] for s in sids], axis=0))
        return np.stack(qs), np.stack(ss), (None if sym else np.stack(zs))

    w, ids = moe_ref.route(logits, topk, renormalize=True)
    ref = moe_ref.moe(x, packed(("w1", "w3")), packed(("w2",)), w, ids, G)
    # outputs reach magnitude 4-8 here, where one bf16 step is 0.03: the absolute term scales with them as in the
    # kernel-level test of the grouped route
    scale_mag = max(1.0, float(np.abs(f32(ref)).max()) / 4)
    np.testing.assert_allclose(f32(to_bits(got)), f32(ref), rtol=1e-2, atol=5e-3 * scale_mag)
    assert (to_bits(got) == ref).mean() > 0.9


@pytest.mark.parametrize("sym", [True, False])
def test_compressed_tensors_w4a16_checkpoint(sym):
    """pack-quantized INT4: weight_packed int32 [N, K/8] (8 consecutive k per word), weight_scale [N, K/G],
    weight_zero_point int32 [N/8, K/G] (8 consecutive n per word).  Same unsigned values as a GPTQ
    checkpoint, so the oracle's GPTQ-layout GEMM over the transposed tensors is the expected result."""
    rng = np.random.default_rng(21 + sym)
    hidden, inter, G, T = 512, 1024, 128, 3
    cfg = _ct_config(symmetric=sym)
    with torch.device(DEV):
        gu = MergedColumnParallelLinear(hidden, [inter, inter], bias=False, params_dtype=torch.bfloat16,
                                        quant_config=cfg, prefix="model.layers.0.mlp.gate_up_proj", return_bias=False)
    qs, ss, zs = [], [], []
    for i in range(2):
        q = rng.integers(0, 16, size=(hidden, inter), dtype=np.int32)
        s = orc.f32_to_h16((0.02 * rng.random((hidden // G, inter)) + 0.005).astype(np.float32))
        gu.weight_loader(gu.weight_packed, torch.from_numpy(np.ascontiguousarray(orc.gptq_pack(q).T)), i)
        gu.weight_loader(gu.weight_scale, to_dev(np.ascontiguousarray(s.T)).cpu(), i)
        gu.weight_loader(gu.weight_shape, torch.tensor([inter, hidden]), i)
        if not sym:
            z = rng.integers(0, 16, size=(hidden // G, inter), dtype=np.int32)
            gu.weight_loader(gu.weight_zero_point, torch.from_numpy(np.ascontiguousarray(orc.pack_cols(z).T)), i)
            zs.append(z)
        qs.append(q)
        ss.append(s)
    gu.process_weights_after_loading()
    q, s = np.concatenate(qs, axis=1), np.concatenate(ss, axis=1)
    x = orc.f32_to_h16(rng.standard_normal((T, hidden)).astype(np.float32))
    if sym:
        ref = orc.w4a16_gemm(x, orc.gptq_pack(q), s, None, None, G=G)
    else:
        ref = orc.w4a16_gemm(x, orc.gptq_pack(q), s, orc.pack_cols(np.concatenate(zs, axis=1)), None, G=G,
                             zp_mode=orc.ZP_NSEQ)
    got = gu(to_dev(x))
    np.testing.assert_allclose(f32(to_bits(got)), f32(ref), rtol=1e-2, atol=2e-2)
    assert (to_bits(got) == ref).mean() > 0.97


def test_compressed_tensors_w8a16_int_channelwise_and_grouped():
    """pack-quantized INT8 (uint8b128): weight_packed int32 [N, K/4]; channel strategy has weight_scale [N, 1]."""
    rng = np.random.default_rng(23)
    hidden, out_f, T = 1024, 768, 3
    for G, weights in ((hidden, {"num_bits": 8, "strategy": "channel", "group_size": None}),
                       (128, {"num_bits": 8})):
        cfg = _ct_config(**weights)
        with torch.device(DEV):
            lin = RowParallelLinear(hidden, out_f, bias=True, params_dtype=torch.bfloat16, quant_config=cfg,
                                    prefix="model.layers.0.mlp.down_proj", return_bias=False)
        assert type(lin.scheme.kernel).__name__ == "MI355XW8A16LinearKernel"
        q = rng.integers(0, 256, size=(hidden, out_f), dtype=np.int32)
        s = orc.f32_to_h16((0.004 * rng.random((hidden // G, out_f)) + 0.001).astype(np.float32))
        bias = orc.f32_to_h16(rng.standard_normal(out_f).astype(np.float32))
        lin.weight_loader(lin.weight_packed, torch.from_numpy(np.ascontiguousarray(orc.gptq8_pack(q).T)))
        lin.weight_loader(lin.weight_scale, to_dev(np.ascontiguousarray(s.T)).cpu())
        lin.weight_loader(lin.bias, to_dev(bias).cpu())
        lin.process_weights_after_loading()
        x = orc.f32_to_h16(rng.standard_normal((T, hidden)).astype(np.float32))
        ref = orc.gptq8_gemm(x, q, s, G, bias)
        got = lin(to_dev(x))
        np.testing.assert_allclose(f32(to_bits(got)), f32(ref), rtol=1e-2, atol=1e-2)
        assert (to_bits(got) == ref).mean() > 0.97


def test_compressed_tensors_actorder_group():
    """actorder "group": weight_g_idx[k] names the scale group of input feature k;
    W[k, n] = (q[k, n] - 8) * s[g_idx[k], n], computed directly."""
    rng = np.random.default_rng(24)
    hidden, out_f, G, T = 1024, 512, 128, 4
    cfg = _ct_config(actorder="group")
    with torch.device(DEV):
        lin = RowParallelLinear(hidden, out_f, bias=False, params_dtype=torch.bfloat16, quant_config=cfg,
                                prefix="model.layers.0.mlp.down_proj", return_bias=False)
    q = rng.integers(0, 16, size=(hidden, out_f), dtype=np.int32)
    s = orc.f32_to_h16((0.02 * rng.random((hidden // G, out_f)) + 0.005).astype(np.float32))
    g_idx = rng.permutation(np.arange(hidden) // G).astype(np.int32)
    lin.weight_loader(lin.weight_packed, torch.from_numpy(np.ascontiguousarray(orc.gptq_pack(q).T)))
    lin.weight_loader(lin.weight_scale, to_dev(np.ascontiguousarray(s.T)).cpu())
    lin.weight_loader(lin.weight_g_idx, torch.from_numpy(g_idx))
    lin.process_weights_after_loading()
    assert lin.mi355q_perm is not None
    x = orc.f32_to_h16(rng.standard_normal((T, hidden)).astype(np.float32))
    W = (q - 8).astype(np.float64) * f32(s)[g_idx].astype(np.float64)
    ref = f32(x).astype(np.float64) @ W
    np.testing.assert_allclose(f32(to_bits(lin(to_dev(x)))), ref, rtol=1e-2, atol=2e-2)


@pytest.mark.parametrize("strategy", ["tensor", "channel", "block"])
def test_compressed_tensors_fp8_weight_only(strategy, monkeypatch):
    """float-quantized fp8 weights (a W8A8 checkpoint with dynamic activation arguments included) under
    MI355Q_FP8_LINEAR=w8a16: served weight-only, x @ (fp8(w) * scale)^T."""
    from mi355q.plugin import CompressedTensorsConfig
    monkeypatch.setenv("MI355Q_FP8_LINEAR", "w8a16")
    rng = np.random.default_rng(25)
    hidden, inter, T = 512, 768, 2
    w_args = {"num_bits": 8, "type": "float", "symmetric": True, "strategy": strategy, "dynamic": False}
    if strategy == "block":
        w_args["block_structure"] = [128, 128]
    act = {"num_bits": 8, "type": "float", "symmetric": True, "strategy": "token", "dynamic": True}
    cfg = CompressedTensorsConfig.from_config({
        "quant_method": "compressed-tensors", "format": "float-quantized", "ignore": ["lm_head"],
        "config_groups": {"group_0": {"targets": ["Linear"], "weights": w_args,
                                      "input_activations": None if strategy == "block" else act}}})
    cfg.packed_modules_mapping = _CT_FUSED
    with torch.device(DEV):
        gu = MergedColumnParallelLinear(hidden, [inter, inter], bias=False, params_dtype=torch.bfloat16,
                                        quant_config=cfg, prefix="model.layers.0.mlp.gate_up_proj", return_bias=False)
    ws, ss = [], []
    for i in range(2):
        w8 = orc.f32_to_e4m3((rng.standard_normal((inter, hidden)) * 30).astype(np.float32))
        gu.weight_loader(gu.weight, torch.from_numpy(w8).view(torch.float8_e4m3fn), i)
        if strategy == "tensor":
            s = np.float32(0.002 * (i + 1))
            gu.weight_loader(gu.weight_scale, torch.tensor(s), i)
            ss.append(np.full(inter, s, dtype=np.float32))
        elif strategy == "channel":
            s = (0.001 + 0.002 * rng.random((inter, 1))).astype(np.float32)
            gu.weight_loader(gu.weight_scale, torch.from_numpy(s), i)
            ss.append(s.reshape(-1))
        else:
            s = (0.001 + 0.002 * rng.random((inter // 128, hidden // 128))).astype(np.float32)
            gu.weight_loader(gu.weight_scale, torch.from_numpy(s), i)
            ss.append(s)
        ws.append(w8)
    gu.process_weights_after_loading()
    x = orc.f32_to_h16(rng.standard_normal((T, hidden)).astype(np.float32))
    ref = orc.w8a16_gemm(x, np.concatenate(ws), np.concatenate(ss), None,
                         block=(128, 128) if strategy == "block" else None)
    got = gu(to_dev(x))
    np.testing.assert_allclose(f32(to_bits(got)), f32(ref), rtol=1e-2, atol=1e-2)
    assert (to_bits(got) == ref).mean() > 0.97


@pytest.mark.parametrize("T", [2, 90])
def test_compressed_tensors_fp8_w8a8_block(T, monkeypatch):
    """float-quantized fp8 checkpoint with 128 x 128 weight blocks and dynamic activation groups
    (schemes/compressed_tensors_w8a8_fp8.py:69-79,167-174): one run-time scale per token and 128-wide group,
    then the block-scaled product.  Against the oracle's restatement of both steps on the same bytes."""
    from mi355q.plugin import CompressedTensorsConfig
    from mi355q.plugin.compressed_tensors import CompressedTensorsW8A8Fp8
    monkeypatch.delenv("MI355Q_FP8_LINEAR", raising=False)
    rng = np.random.default_rng(43)
    hidden, inter = 512, 768
    w_args = {"num_bits": 8, "type": "float", "symmetric": True, "strategy": "block", "dynamic": False,
              "block_structure": [128, 128]}
    act = {"num_bits": 8, "type": "float", "symmetric": True, "strategy": "group", "group_size": 128, "dynamic": True}
    cfg = CompressedTensorsConfig.from_config({
        "quant_method": "compressed-tensors", "format": "float-quantized", "ignore": ["lm_head"],
        "config_groups": {"group_0": {"targets": ["Linear"], "weights": w_args, "input_activations": act}}})
    cfg.packed_modules_mapping = _CT_FUSED
    with torch.device(DEV):
        gu = MergedColumnParallelLinear(hidden, [inter, inter], bias=False, params_dtype=torch.bfloat16,
                                        quant_config=cfg, prefix="model.layers.0.mlp.gate_up_proj", return_bias=False)
    assert type(gu.scheme) is CompressedTensorsW8A8Fp8 and gu.scheme.strategy == "block"
    ws, ss = [], []
    for i in range(2):
        w8 = orc.f32_to_e4m3((rng.standard_normal((inter, hidden)) * 30).astype(np.float32))
        s = (0.001 + 0.002 * rng.random((inter // 128, hidden // 128))).astype(np.float32)
        gu.weight_loader(gu.weight, torch.from_numpy(w8).view(torch.float8_e4m3fn), i)
        gu.weight_loader(gu.weight_scale, torch.from_numpy(s), i)
        ws.append(w8)
        ss.append(s)
    gu.process_weights_after_loading()
    assert gu.weight_scale.shape == (2 * inter // 128, hidden // 128)
    x = orc.f32_to_h16(rng.standard_normal((T, hidden)).astype(np.float32))
    xq, xs = orc.per_token_group_quant_fp8(x, 128)
    ref = orc.w8a8_block_scaled_mm(xq, np.concatenate(ws), xs, np.concatenate(ss), [128, 128])
    got = gu(to_dev(x))
    np.testing.assert_allclose(f32(to_bits(got)), f32(ref), rtol=1e-2, atol=1e-2 * float(np.abs(f32(ref)).max()))
    assert (to_bits(got) == ref).mean() > 0.9


@pytest.mark.parametrize("T", [3, 200])
@pytest.mark.parametrize("strategy,dynamic", [("tensor", False), ("tensor", True), ("channel", False),
                                              ("channel", True)])
def test_compressed_tensors_fp8_w8a8(strategy, dynamic, T, monkeypatch):
    """float-quantized fp8 W8A8 checkpoint (schemes/compressed_tensors_w8a8_fp8.py:60-207): activations
    quantised with the checkpoint's largest input_scale (static) or one scale per token (dynamic), fused
    per-tensor shards requantised to the largest weight scale, then the scaled fp8 x fp8 product.  Checked
    against the same steps on the CPU."""
    from mi355q.plugin import CompressedTensorsConfig
    from mi355q.plugin.compressed_tensors import CompressedTensorsW8A8Fp8
    monkeypatch.delenv("MI355Q_FP8_LINEAR", raising=False)
    rng = np.random.default_rng(41)
    hidden, inter = 512, 768
    w_args = {"num_bits": 8, "type": "float", "symmetric": True, "strategy": strategy, "dynamic": False}
    act = {"num_bits": 8, "type": "float", "symmetric": True, "strategy": "token" if dynamic else "tensor",
           "dynamic": dynamic}
    cfg = CompressedTensorsConfig.from_config({
        "quant_method": "compressed-tensors", "format": "float-quantized", "ignore": ["lm_head"],
        "config_groups": {"group_0": {"targets": ["Linear"], "weights": w_args, "input_activations": act}}})
    cfg.packed_modules_mapping = _CT_FUSED
    with torch.device(DEV):
        gu = MergedColumnParallelLinear(hidden, [inter, inter], bias=False, params_dtype=torch.bfloat16,
                                        quant_config=cfg, prefix="model.layers.0.mlp.gate_up_proj", return_bias=False)
    assert type(gu.scheme) is CompressedTensorsW8A8Fp8
    ws, ss, in_scales = [], [], [0.02, 0.035]
    for i in range(2):
        w8 = orc.f32_to_e4m3((rng.standard_normal((inter, hidden)) * 30).astype(np.float32))
        gu.weight_loader(gu.weight, torch.from_numpy(w8).view(torch.float8_e4m3fn), i)
        if strategy == "tensor":
            s = np.float32(0.002 * (i + 1))
            gu.weight_loader(gu.weight_scale, torch.tensor(s), i)
            ss.append(s)
        else:
            s = (0.001 + 0.002 * rng.random((inter, 1))).astype(np.float32)
            gu.weight_loader(gu.weight_scale, torch.from_numpy(s), i)
            ss.append(s.reshape(-1))
        if not dynamic:
            gu.weight_loader(gu.input_scale, torch.tensor(in_scales[i]), i)
        ws.append(w8)
    gu.process_weights_after_loading()
    x = orc.f32_to_h16(rng.standard_normal((T, hidden)).astype(np.float32))
    # CPU restatement of the same steps
    if strategy == "tensor":
        smax = np.float32(max(ss))
        deq = [(torch.from_numpy(w).view(torch.float8_e4m3fn).to(torch.float16) * s).float().numpy()
               for w, s in zip(ws, ss)]                                                   # through fp16
        wq = np.concatenate([orc.static_scaled_fp8_quant(d, np.array([smax], np.float32), None, orc.FP32) for d in deq])
        assert np.array_equal(gu.weight.data.view(torch.uint8).cpu().numpy(), wq)
        sb = np.array([smax], np.float32)
    else:
        wq, sb = np.concatenate(ws), np.concatenate(ss)
    if dynamic:
        xq, sa = orc.dynamic_per_token_scaled_fp8_quant(x)
    else:
        sa = np.array([max(in_scales)], np.float32)
        xq = orc.static_scaled_fp8_quant(x, sa)
    ref = orc.fp8_scaled_mm(xq, wq, sa, sb)
    got = to_bits(gu(to_dev(x)))
    np.testing.assert_allclose(f32(got), f32(ref), rtol=1e-2, atol=1e-2)
    assert (got == ref).mean() > 0.97
