#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (authoring container only).

Runs the reference's own pure-torch helpers (imported from /root/reference with
stub modules for absent third-party packages) on seeded inputs and stores the
inputs and expected outputs as small .npz files.  Only numeric data is stored.
bf16/fp16 tensors are stored as their raw 16-bit patterns (uint16), fp8 as uint8.

Where a reference helper cannot be imported on a GPU-less box (module-level
skips or msgspec metaclasses), its arithmetic is restated here in torch and the
file it follows is cited; those fixtures are marked restated=1.

    python tests/golden/gen_golden.py
"""
import importlib
import os
import random
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

REF = os.environ.get("MI355Q_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)


def imp(name):
    while True:
        try:
            return importlib.import_module(name)
        except ModuleNotFoundError as e:
            if e.name is None or e.name.startswith(("vllm", "tests")):
                raise
            m = MagicMock(name=e.name)
            m.__path__ = []
            m.__spec__ = None
            sys.modules[e.name] = m


qu = imp("vllm.model_executor.layers.quantization.utils.quant_utils")
st = imp("vllm.scalar_type").scalar_types
ln = imp("vllm.ir.ops.layernorm")
rope_base = imp("vllm.model_executor.layers.rotary_embedding.base")


def bits16(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def bits8(t):
    return t.contiguous().view(torch.uint8).numpy().copy()


def save(name, **kw):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **kw)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


# --------------------------------------------------------------------------
# G1: integer packers (quant_utils.py:537-577, 828-937)
# --------------------------------------------------------------------------
def gen_pack():
    torch.manual_seed(0)
    K, N = 64, 48
    q = torch.randint(0, 16, (K, N), dtype=torch.int32)
    z = torch.randint(0, 16, (K // 32, N), dtype=torch.int32)
    save(
        "pack",
        q=q.numpy(),
        z=z.numpy(),
        gptq_pack=qu.gptq_pack(q, 4, K, N).numpy(),
        pack_cols=qu.pack_cols(q, 4, K, N).numpy(),
        awq_pack=qu.awq_pack(q, 4, K, N).numpy(),
        z_pack_cols=qu.pack_cols(z, 4, K // 32, N).numpy(),
        z_awq_pack=qu.awq_pack(z, 4, K // 32, N).numpy(),
        pack_i32_dim0=qu.pack_quantized_values_into_int32(q, st.uint4, 0).numpy(),
        pack_i32_dim1=qu.pack_quantized_values_into_int32(q, st.uint4, 1).numpy(),
        unpack_cols=qu.unpack_cols(qu.pack_cols(q, 4, K, N), 4, K, N).numpy(),
    )


# --------------------------------------------------------------------------
# G2: quantize_weights / gptq_quantize_weights (quant_utils.py:680-810)
# --------------------------------------------------------------------------
def gen_quantize():
    out = {}
    for tag, qt, G, zp in [
        ("u4b8_g128", st.uint4b8, 128, False),
        ("u4b8_g32", st.uint4b8, 32, False),
        ("u4b8_gm1", st.uint4b8, -1, False),
        ("u4_zp_g64", st.uint4, 64, True),
        ("u8b128_g128", st.uint8b128, 128, False),
    ]:
        torch.manual_seed(0)
        w = torch.randn(256, 32, dtype=torch.float32) / 10
        w_ref, w_q, w_s, w_zp = qu.quantize_weights(w, qt, G, zp)
        out[f"{tag}_w"] = w.numpy()
        out[f"{tag}_w_ref"] = w_ref.numpy()
        out[f"{tag}_w_q"] = w_q.numpy()
        out[f"{tag}_w_s"] = w_s.numpy()
        if w_zp is not None:
            out[f"{tag}_w_zp"] = w_zp.numpy()
    save("quantize", **out)


# --------------------------------------------------------------------------
# G3: W4A16 GEMM.  (a) inputs and reference of
# tests/kernels/quantization/test_triton_w4a16.py:66-162 (restated: the module
# skips itself at import on a non-ROCm platform); (b) GPTQ checkpoint route via
# gptq_quantize_weights + gptq_pack with out = a @ w_ref (imported helpers).
# --------------------------------------------------------------------------
def _pack_n(w):  # test_triton_w4a16.py:33-43
    K, N = w.shape
    shifts = torch.arange(8, dtype=torch.int32) * 4
    return torch.sum((w.view(K, N // 8, 8) & 0xF) << shifts, dim=2, dtype=torch.int32)


def _unpack_n(p):  # test_triton_w4a16.py:46-52
    K, N8 = p.shape
    shifts = torch.arange(8, dtype=torch.int32) * 4
    return ((p.unsqueeze(-1) >> shifts) & 0xF).reshape(K, N8 * 8)


def _w4a16_reference(a, b_packed, scales, G, qzeros, zp_bias):  # :66-107
    M, K = a.shape
    w = _unpack_n(b_packed)
    N = w.shape[1]
    if qzeros is None:
        z = torch.full((K, N), zp_bias, dtype=torch.int32)
    else:
        z = _unpack_n(qzeros).repeat_interleave(G, dim=0)
    s = scales.repeat_interleave(G, dim=0).to(torch.float32)
    w_fp = (w - z).to(torch.float32) * s
    return (a.to(torch.float32) @ w_fp).to(a.dtype)


def gen_w4a16():
    out = {}
    cases = [
        (1, 256, 256, 32, False),
        (17, 256, 512, 32, False),
        (32, 512, 256, 64, False),
        (33, 512, 512, 128, False),
        (64, 1024, 256, 256, False),
        (128, 256, 1024, 32, True),
        (64, 512, 512, 64, True),
    ]
    names = []
    for dtype, dn in [(torch.bfloat16, "bf16"), (torch.float16, "fp16")]:
        for M, K, N, G, has_zp in cases:
            if dn == "fp16" and M > 17:
                continue
            torch.manual_seed(0)
            a = (0.25 * torch.randn((M, K), dtype=torch.float32)).to(dtype)
            w = torch.randint(0, 16, (K, N), dtype=torch.int32)
            bp = _pack_n(w)
            s = (0.05 * torch.rand((K // G, N), dtype=torch.float32)).to(dtype)
            qz = None
            if has_zp:
                qz = _pack_n(torch.randint(0, 16, (K // G, N), dtype=torch.int32))
            ref = _w4a16_reference(a, bp, s, G, qz, 8)
            tag = f"{dn}_m{M}_k{K}_n{N}_g{G}_zp{int(has_zp)}"
            names.append(tag)
            out[tag + "_a"] = bits16(a)
            out[tag + "_b"] = bp.numpy()
            out[tag + "_s"] = bits16(s)
            if qz is not None:
                out[tag + "_z"] = qz.numpy()
            out[tag + "_ref"] = bits16(ref)
    out["names"] = np.array(names)
    save("w4a16_triton_cases", **out)

    # GPTQ checkpoint route (imported reference quantiser + packer).
    out = {}
    names = []
    for M, K, N, G in [(1, 256, 128, 128), (5, 512, 64, 32), (16, 256, 64, -1)]:
        torch.manual_seed(0)
        w = torch.randn(K, N, dtype=torch.float32) / 10
        w_ref, w_q, w_s, g_idx, _ = qu.gptq_quantize_weights(
            w.to(torch.bfloat16), st.uint4b8, G if G != -1 else K, False
        )
        qweight = qu.gptq_pack(w_q, 4, K, N)
        a = (0.25 * torch.randn((M, K), dtype=torch.float32)).to(torch.bfloat16)
        ref = (a.float() @ w_ref.float()).to(torch.bfloat16)
        tag = f"m{M}_k{K}_n{N}_g{G}"
        names.append(tag)
        out[tag + "_a"] = bits16(a)
        out[tag + "_qweight"] = qweight.numpy()
        out[tag + "_scales"] = bits16(w_s.to(torch.bfloat16))
        out[tag + "_w_ref"] = bits16(w_ref)
        out[tag + "_ref"] = bits16(ref)
    out["names"] = np.array(names)
    save("w4a16_gptq_route", **out)


# --------------------------------------------------------------------------
# G4: AWQ.  quantize_weights(uint4, zero points) + awq_pack (imported); the
# dequantised weight expected from the packed tensors is w_ref.  The AWQ
# dequantize reference of tests/kernels/quantization/test_awq_triton.py:27-74
# (reverse order [0,4,1,5,2,6,3,7]) is restated to produce `deq`.
# --------------------------------------------------------------------------
def _awq_dequantize_torch(qweight, scales, qzeros, G):  # test_awq_triton.py:27-74
    rev = [0, 4, 1, 5, 2, 6, 3, 7]
    shifts = torch.arange(0, 32, 4)

    def unpack(t):
        u = torch.bitwise_right_shift(t[:, :, None], shifts[None, None, :]).to(torch.int8)
        u = u.view(t.shape[0], -1)
        order = torch.arange(u.shape[-1], dtype=torch.int32).view(-1, 8)[:, rev].reshape(-1)
        return u[:, order] & 0xF

    iw = unpack(qweight)
    z = unpack(qzeros).repeat_interleave(G, dim=0)
    s = scales.repeat_interleave(G, dim=0)
    return (iw - z) * s


def gen_awq():
    out = {}
    names = []
    for M, K, N, G in [(1, 256, 128, 128), (7, 256, 64, 32), (16, 512, 64, 64)]:
        torch.manual_seed(0)
        w = torch.randn(K, N, dtype=torch.float32) / 10
        w_ref, w_q, w_s, w_zp = qu.quantize_weights(w.to(torch.float16), st.uint4, G, True)
        qweight = qu.awq_pack(w_q, 4, K, N)
        qzeros = qu.awq_pack(w_zp, 4, K // G, N)
        deq = _awq_dequantize_torch(qweight, w_s, qzeros, G)
        assert torch.equal(deq.to(torch.float16), w_ref.to(torch.float16))
        a = (0.25 * torch.randn((M, K), dtype=torch.float32)).to(torch.float16)
        ref = (a.float() @ w_ref.float()).to(torch.float16)
        tag = f"m{M}_k{K}_n{N}_g{G}"
        names.append(tag)
        out[tag + "_a"] = bits16(a)
        out[tag + "_qweight"] = qweight.numpy()
        out[tag + "_qzeros"] = qzeros.numpy()
        out[tag + "_scales"] = bits16(w_s.to(torch.float16))
        out[tag + "_deq"] = bits16(deq.to(torch.float16))
        out[tag + "_ref"] = bits16(ref)
    out["names"] = np.array(names)
    save("awq", **out)


# --------------------------------------------------------------------------
# G5: RMSNorm.  native IR impl (vllm/ir/ops/layernorm.py:9-62, imported) and the
# kernel-order fused variant (csrc/libtorch_stable/layernorm_kernels.cu:133-172:
# residual sum rounded to the activation dtype before the variance; restated).
# --------------------------------------------------------------------------
def _native(op):
    impls = getattr(op, "impls", None)
    if impls is not None and "native" in impls:
        impl = impls["native"]
        for a in ("impl_fn", "fn", "func"):
            if hasattr(impl, a):
                return getattr(impl, a)
        return impl
    return op


def gen_rmsnorm():
    rms = _native(ln.rms_norm)
    fused = _native(ln.fused_add_rms_norm)
    out = {}
    names = []
    for T, H in [(7, 768), (3, 4096), (2, 5125), (1, 8192)]:
        torch.manual_seed(0)
        x = torch.randn(T, H).to(torch.bfloat16)
        r = torch.randn(T, H).to(torch.bfloat16)
        w = torch.randn(H).to(torch.bfloat16)
        eps = 1e-5
        y = rms(x, w, eps)
        y2, r2 = fused(x.clone(), r.clone(), w, eps)
        # kernel-order: z = bf16(x + r); var from z; out = bf16(z * rs * w)
        z = (x.float() + r.float()).to(torch.bfloat16)
        var = z.float().pow(2).mean(dim=-1, keepdim=True)
        yk = (z.float() * torch.rsqrt(var + eps) * w.float()).to(torch.bfloat16)
        # single-rounding rms_norm of the CUDA kernel (layernorm_kernels.cu:89-97)
        var1 = x.float().pow(2).mean(dim=-1, keepdim=True)
        y1 = (x.float() * torch.rsqrt(var1 + eps) * w.float()).to(torch.bfloat16)
        tag = f"t{T}_h{H}"
        names.append(tag)
        out[tag + "_x"] = bits16(x)
        out[tag + "_r"] = bits16(r)
        out[tag + "_w"] = bits16(w)
        out[tag + "_native"] = bits16(y)
        out[tag + "_kernel"] = bits16(y1)
        out[tag + "_fused_native"] = bits16(y2)
        out[tag + "_fused_native_res"] = bits16(r2)
        out[tag + "_fused_kernel"] = bits16(yk)
        out[tag + "_fused_kernel_res"] = bits16(z)
    out["names"] = np.array(names)
    save("rmsnorm", **out)


# --------------------------------------------------------------------------
# G6: RoPE.  RotaryEmbedding.forward_static (rotary_embedding/base.py:160-201,
# imported) on a cos/sin cache built as base.py:80-103, cast to bf16.
# --------------------------------------------------------------------------
def gen_rope():
    out = {}
    names = []
    for head, rot, neox, nh, nkv in [
        (128, 128, True, 8, 2),
        (64, 64, False, 4, 4),
        (128, 32, True, 4, 1),
        (64, 32, False, 2, 2),
    ]:
        torch.manual_seed(0)
        T, max_pos, base = 5, 64, 10000.0
        inv_freq = 1.0 / (base ** (torch.arange(0, rot, 2, dtype=torch.float) / rot))
        t = torch.arange(max_pos, dtype=torch.float)
        freqs = torch.einsum("i,j -> ij", t, inv_freq)
        cache = torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(torch.bfloat16)
        pos = torch.randint(0, max_pos, (T,), dtype=torch.int64)
        q = torch.randn(T, nh * head).to(torch.bfloat16)
        k = torch.randn(T, nkv * head).to(torch.bfloat16)
        qo, ko = rope_base.RotaryEmbedding.forward_static(pos, q.clone(), k.clone(), head, rot, cache, neox)
        tag = f"h{head}_r{rot}_neox{int(neox)}"
        names.append(tag)
        out[tag + "_pos"] = pos.numpy()
        out[tag + "_q"] = bits16(q)
        out[tag + "_k"] = bits16(k)
        out[tag + "_cache"] = bits16(cache)
        out[tag + "_qo"] = bits16(qo)
        out[tag + "_ko"] = bits16(ko)
        out[tag + "_meta"] = np.array([head, rot, int(neox), nh, nkv])
    out["names"] = np.array(names)
    save("rope", **out)


# --------------------------------------------------------------------------
# G7: paged attention (decode).  Reference restated from
# tests/kernels/attention/test_attention.py:42-109 (module import needs the
# compiled _C_cuda_utils); KV layout of vllm/v1/attention/ops/paged_attn.py:16-29.
# fp8 KV: stored = float8_e4m3fn(x / scale), read = float(stored) * scale
# (kv_cache.py:95-151, csrc/quantization/w8a8/fp8/amd/quant_utils.cuh:29-49).
# --------------------------------------------------------------------------
def _ref_paged(q, kc, vc, block_tables, seq_lens, scale, alibi):
    S, H, D = q.shape
    KVH = vc.shape[1]
    bs = vc.shape[3]
    rep = H // KVH
    out = torch.empty_like(q)
    for i in range(S):
        L = int(seq_lens[i])
        ks, vs = [], []
        for j in range(L):
            b = int(block_tables[i, j // bs])
            o = j % bs
            ks.append(kc[b, :, :, o, :].reshape(KVH, D))
            vs.append(vc[b, :, :, o])
        keys = torch.stack(ks, 0)
        vals = torch.stack(vs, 0)
        if rep > 1:
            keys = torch.repeat_interleave(keys, rep, dim=1)
            vals = torch.repeat_interleave(vals, rep, dim=1)
        w = scale * torch.einsum("qhd,khd->hqk", q[i : i + 1], keys).float()
        if alibi is not None:
            bias = (torch.arange(L).int() - L + 1).float()
            w = w + alibi.view(-1, 1, 1) * bias.view(1, 1, -1)
        w = torch.softmax(w, dim=-1).to(vals.dtype)
        out[i] = torch.einsum("hqk,khd->qhd", w, vals).view(H, D)
    return out


def gen_paged_attn():
    out = {}
    names = []
    for tag, H, KVH, D, bs, lens, kvdt, use_alibi, nblk in [
        ("bf16_h8_kv2_d128_b16", 8, 2, 128, 16, [1, 17, 260], "auto", False, 24),
        ("bf16_h4_kv4_d64_b32", 4, 4, 64, 32, [33, 257], "auto", False, 12),
        ("fp8_h8_kv2_d128_b16", 8, 2, 128, 16, [5, 290], "fp8", False, 24),
        ("fp8_h16_kv2_d128_b32", 16, 2, 128, 32, [64, 520], "fp8", False, 20),
        ("bf16_alibi_h8_kv2_d128_b16", 8, 2, 128, 16, [40, 260], "auto", True, 20),
    ]:
        torch.manual_seed(0)
        random.seed(0)
        S = len(lens)
        scale = float(1.0 / (D**0.5))
        q = torch.empty(S, H, D).uniform_(-scale, scale).to(torch.bfloat16)
        x = 16 if kvdt == "fp8" else 8
        kc = torch.empty(nblk, KVH, D // x, bs, x).uniform_(-scale, scale)
        vc = torch.empty(nblk, KVH, D, bs).uniform_(-scale, scale)
        max_blocks = (max(lens) + bs - 1) // bs
        bt = torch.tensor(
            [[random.randint(0, nblk - 1) for _ in range(max_blocks)] for _ in range(S)],
            dtype=torch.int32,
        )
        sl = torch.tensor(lens, dtype=torch.int32)
        alibi = torch.randn(H, dtype=torch.float) if use_alibi else None
        if kvdt == "fp8":
            k_scale, v_scale = 0.5 / 16, 0.25 / 16
            kq = (kc / k_scale).to(torch.float8_e4m3fn)
            vq = (vc / v_scale).to(torch.float8_e4m3fn)
            # dequantise as convert_fp8 does, to the activation dtype
            kd = (kq.float() * k_scale).to(torch.bfloat16)
            vd = (vq.float() * v_scale).to(torch.bfloat16)
            # the reference reads K as [blk, kvh, D/x', bs, x'] with x' of bf16 (8)
            kd8 = kd.permute(0, 1, 3, 2, 4).reshape(nblk, KVH, bs, D)
            kd8 = kd8.reshape(nblk, KVH, bs, D // 8, 8).permute(0, 1, 3, 2, 4).contiguous()
            ref = _ref_paged(q, kd8, vd, bt, sl, scale, alibi)
            out[tag + "_kc"] = bits8(kq)
            out[tag + "_vc"] = bits8(vq)
            out[tag + "_kv_scale"] = np.array([k_scale, v_scale], dtype=np.float32)
        else:
            kcb = kc.to(torch.bfloat16)
            vcb = vc.to(torch.bfloat16)
            ref = _ref_paged(q, kcb, vcb, bt, sl, scale, alibi)
            out[tag + "_kc"] = bits16(kcb)
            out[tag + "_vc"] = bits16(vcb)
        names.append(tag)
        out[tag + "_q"] = bits16(q)
        out[tag + "_bt"] = bt.numpy()
        out[tag + "_sl"] = sl.numpy()
        out[tag + "_ref"] = bits16(ref)
        out[tag + "_meta"] = np.array([H, KVH, D, bs, int(kvdt == "fp8")])
        out[tag + "_scale"] = np.array([scale], dtype=np.float32)
        if alibi is not None:
            out[tag + "_alibi"] = alibi.numpy()
    out["names"] = np.array(names)
    save("paged_attn", **out)


# --------------------------------------------------------------------------
# G9b: random sampling with top-k / top-p -- the reference's own functions
# (vllm/v1/sample/ops/topk_topp_sampler.py: apply_top_k_top_p_pytorch :367-407,
# random_sample :441-466), imported and run on CPU with seeded per-request generators.
# --------------------------------------------------------------------------
def gen_sampler_topk_topp():
    tk = imp("vllm.v1.sample.ops.topk_topp_sampler")
    torch.manual_seed(1)
    B, V = 8, 8192
    logits = (torch.randn(B, V) * 2.5).float()
    logits[3, 100:140] = -float("inf")            # masked vocabulary entries
    temperature = torch.tensor([1.0, 0.7, 1.3, 1.0, 0.5, 2.0, 1.0, 0.9])
    k = torch.tensor([V, 50, 1, 40, V, 1000, 5, 300], dtype=torch.int32)
    p = torch.tensor([1.0, 1.0, 0.9, 0.95, 0.8, 0.5, 0.3, 0.99])
    scaled = logits / temperature.unsqueeze(1)
    masked = tk.apply_top_k_top_p_pytorch(scaled.clone(), k, p)
    generators = {i: torch.Generator().manual_seed(100 + i) for i in range(B)}
    # the noise the reference draws (same generators, same call order as random_sample)
    noise = torch.empty(B, V)
    for i in range(B):
        noise[i].exponential_(generator=torch.Generator().manual_seed(100 + i))
    probs = masked.softmax(dim=-1, dtype=torch.float32)
    ids = tk.random_sample(probs.clone(), generators)
    assert torch.equal(ids, (probs / noise).argmax(dim=-1)), "noise replay differs from random_sample"
    # how decisive every draw is: ratio of the second-best score to the best
    sc = probs / noise
    top2 = sc.topk(2, dim=-1).values
    save("sampler_topk_topp", logits=logits.numpy(), temperature=temperature.numpy(), top_k=k.numpy(), top_p=p.numpy(),
         noise=noise.numpy(), masked=masked.numpy(), ids=ids.numpy(), runner_up_ratio=(top2[:, 1] / top2[:, 0]).numpy())


# --------------------------------------------------------------------------
# G9: sampler inner loop.  Greedy = logits.float().argmax(-1)
# (vllm/v1/sample/sampler.py:96,240-241; restated: the module derives from
# msgspec.Struct and cannot be imported with a stub) and repetition penalties
# (vllm/_custom_ops.py:309-322, imported below when possible).
# --------------------------------------------------------------------------
def gen_sampler():
    torch.manual_seed(0)
    B, V = 4, 1000
    logits = torch.randn(B, V).to(torch.bfloat16)
    logits[1, 17] = logits[1].max() + 1
    logits[1, 900] = logits[1, 17]  # tie: first index wins
    logits[2, :] = 0.5  # all equal -> 0
    ids = logits.float().argmax(dim=-1)
    prompt_mask = torch.rand(B, V) < 0.1
    output_mask = torch.rand(B, V) < 0.1
    pen = torch.tensor([1.0, 1.2, 2.0, 0.8])
    lg = logits.float().clone()
    try:
        ops = imp("vllm._custom_ops")
        ops.apply_repetition_penalties_torch(lg, prompt_mask, output_mask, pen)
        restated = 0
    except Exception:
        p = torch.where(prompt_mask | output_mask, pen.unsqueeze(1).repeat(1, V), 1.0)
        lg *= torch.where(lg > 0, 1.0 / p, p)
        restated = 1
    save(
        "sampler",
        logits=bits16(logits),
        argmax=ids.numpy(),
        prompt_mask=prompt_mask.numpy(),
        output_mask=output_mask.numpy(),
        penalties=pen.numpy(),
        penalized=lg.numpy(),
        restated=np.array([restated]),
    )


def _fp8_fallback_linear(x, weight_fp8, weight_scale, bias):
    """The dequantise-then-linear route of Fp8LinearMethod.apply (quantization/fp8.py:433-454):
    weight and scale go to bf16, are multiplied there (per-tensor scalar or per-row vector), and
    the product feeds a plain linear.  Restated: the branch sits behind VLLM_BATCH_INVARIANT and a
    kernel object that cannot be constructed without a GPU platform."""
    w = weight_fp8.to(torch.bfloat16)
    s = weight_scale.to(torch.bfloat16)
    w = w * (s if s.numel() == 1 else s.unsqueeze(1))
    return torch.nn.functional.linear(x, w, bias)


def gen_w8a16():
    torch.manual_seed(0)
    out, names = {}, []
    for tag, M, N, K, per_row, with_bias in [("t_m1", 1, 256, 512, False, False), ("t_m3_bias", 3, 192, 1024, False, True),
                                             ("row_m2", 2, 128, 768, True, False), ("row_m5_bias", 5, 320, 256, True, True)]:
        w = torch.randn(N, K) * 0.05
        scale = (w.abs().amax(dim=1) if per_row else w.abs().amax().reshape(1)) / 448.0
        wq = (w / (scale.unsqueeze(1) if per_row else scale)).to(torch.float8_e4m3fn)
        x = (torch.randn(M, K) * 0.5).to(torch.bfloat16)
        bias = torch.randn(N).to(torch.bfloat16) if with_bias else None
        ref = _fp8_fallback_linear(x, wq, scale, bias)
        names.append(tag)
        out[tag + "_w"] = bits8(wq)
        out[tag + "_scale"] = scale.numpy().astype(np.float32)
        out[tag + "_x"] = bits16(x)
        out[tag + "_ref"] = bits16(ref)
        if with_bias:
            out[tag + "_bias"] = bits16(bias)
    save("w8a16", names=np.array(names), restated=np.array([1]), **out)


def gen_moe():
    """Inputs of tests/kernels/moe/test_moe.py:590-652 (test_fused_moe_wn16: a, w/10, score, weights
    quantised by quantize_weights and packed two nibbles per byte, even index low) and the
    expected output of tests.kernels.utils.torch_moe on the dequantised reference weights
    (:696), computed here on CPU."""
    def restated_torch_moe(a, w1, w2, score, topk):   # tests/kernels/utils.py:912-1078, bf16 branch
        score = torch.softmax(score, dim=-1, dtype=torch.float32)
        topk_weight, topk_ids = torch.topk(score, topk)
        M, K = a.shape
        a = a.view(M, -1, K).repeat(1, topk, 1).reshape(-1, K)
        out = torch.zeros(M * topk, w2.shape[1], dtype=a.dtype)
        ids = topk_ids.view(-1)
        for i in range(w1.shape[0]):
            mask = ids == i
            if mask.sum():
                tmp1 = a[mask] @ w1[i].transpose(0, 1)
                d = tmp1.shape[-1] // 2
                tmp2 = torch.nn.functional.silu(tmp1[..., :d]) * tmp1[..., d:]
                out[mask] = tmp2 @ w2[i].transpose(0, 1)
        return (out.view(M, -1, w2.shape[1]).to(torch.float32) * topk_weight.view(M, -1, 1)).sum(dim=1).to(out.dtype)

    ref_moe, ctx = None, None
    try:
        ref_moe = imp("tests.kernels.utils").torch_moe
        cfg = imp("vllm.config")
        ctx = lambda: cfg.set_current_vllm_config(cfg.VllmConfig())  # noqa: E731
        with ctx():
            ref_moe(torch.zeros(1, 32, dtype=torch.bfloat16), torch.zeros(2, 64, 32, dtype=torch.bfloat16),
                    torch.zeros(2, 32, 32, dtype=torch.bfloat16), torch.zeros(1, 2, dtype=torch.bfloat16), 1)
        restated = 0
    except Exception as exc:
        print("reference torch_moe not runnable here (%s: %s); using its restatement" % (type(exc).__name__, str(exc)[:120]))
        ref_moe, restated = None, 1

    def torch_moe(a, w1, w2, score, topk):
        if ref_moe is None:
            return restated_torch_moe(a, w1, w2, score, topk)
        with ctx():
            got = ref_moe(a, w1, w2, score, topk)
        assert torch.equal(got, restated_torch_moe(a, w1, w2, score, topk)), "restatement drifted from the reference"
        return got

    out, names = {}, []
    dtype = torch.bfloat16
    for tag, m, n, k, e, topk, G, has_zp in [("m1_sym", 1, 128, 256, 8, 2, 64, False), ("m3_zp", 3, 128, 256, 8, 2, 128, True),
                                             ("m33_zp", 33, 64, 128, 4, 2, 64, True), ("m2_top6", 2, 64, 128, 8, 6, 32, False)]:
        torch.manual_seed(7)
        a = torch.randn((m, k), dtype=dtype) / 10
        w1 = torch.randn((e, 2 * n, k), dtype=dtype) / 10
        w2 = torch.randn((e, k, n), dtype=dtype) / 10
        score = torch.randn((m, e), dtype=dtype)
        qt = st.uint4 if has_zp else st.uint4b8
        packs = {}
        for which, w in (("w1", w1), ("w2", w2)):
            ref = torch.empty_like(w)
            qws, scs, zps = [], [], []
            for i in range(e):
                weight, qweight, scales, qzeros = qu.quantize_weights(w[i].T, qt, G, has_zp, False)
                ref[i] = weight.T
                qweight = qweight.T.contiguous().to(torch.uint8)
                qws.append(qweight[:, 1::2] * 16 + qweight[:, ::2])
                scs.append(scales.T.contiguous())
                if has_zp:
                    qz = qzeros.T.contiguous().to(torch.uint8)
                    zps.append(qz[1::2, :] * 16 + qz[::2, :])
            packs[which] = (ref, torch.stack(qws), torch.stack(scs), torch.stack(zps) if has_zp else None)
        expect = torch_moe(a, packs["w1"][0], packs["w2"][0], score, topk)
        names.append(tag)
        out[tag + "_meta"] = np.array([m, n, k, e, topk, G, int(has_zp)])
        out[tag + "_a"] = bits16(a)
        out[tag + "_score"] = bits16(score)
        for which in ("w1", "w2"):
            ref, qw, sc, zp = packs[which]
            out[f"{tag}_{which}_q"] = qw.numpy()
            out[f"{tag}_{which}_s"] = bits16(sc)
            if zp is not None:
                out[f"{tag}_{which}_z"] = zp.numpy()
        out[tag + "_ref"] = bits16(expect)
    save("moe", names=np.array(names), restated=np.array([restated]), **out)


def gen_fp8_w8a8():
    """fp8 activation quantisers and the scaled product.
    * dynamic per-token / per-tensor quantisation: the reference's own test oracles
      tests/kernels/quant_utils.py:22-88 (imported; their `as_float32_tensor` pins device="cuda", so it is
      pointed at the CPU here) on the inputs of tests/kernels/quantization/test_fp8_quant.py:57-108;
    * static quantisation with group shapes: quant_utils.scaled_quantize (imported), test_fp8_quant.py:135-221;
    * RMSNorm + per-token quantisation: the reference test's composite
      (tests/kernels/core/test_fused_quant_layernorm.py:136-182: native RMSNorm, then per-token quantisation);
    * scaled product: torch._scaled_mm on CPU with the inputs of
      tests/kernels/quantization/test_rocm_skinny_gemms.py:250-279."""
    tq = imp("tests.kernels.quant_utils")
    tq.as_float32_tensor = lambda x: torch.as_tensor(x, dtype=torch.float32, device="cpu")
    assert tq.FP8_DTYPE == torch.float8_e4m3fn, tq.FP8_DTYPE
    out, names = {}, []
    # dynamic quantisers
    for T, H, dtype, with_ub in [(1, 17, torch.bfloat16, False), (7, 1024, torch.bfloat16, True), (7, 1025, torch.float32, False),
                                 (7, 1026, torch.bfloat16, True), (3, 5137, torch.float32, True), (16, 1024, torch.bfloat16, False)]:
        torch.manual_seed(0)
        tag = f"dyn_{T}x{H}_{'bf16' if dtype == torch.bfloat16 else 'f32'}_{int(with_ub)}"
        x = torch.rand(T, H, dtype=dtype) + 1e-6
        ub = torch.mean(x).to(torch.float32) if with_ub else None
        q_tok, s_tok = tq.ref_dynamic_per_token_quant(x, tq.FP8_DTYPE, ub)
        q_ten, s_ten = tq.ref_dynamic_per_tensor_fp8_quant(x)
        names.append(tag)
        out[tag + "_x"] = bits16(x) if dtype == torch.bfloat16 else x.numpy()
        if with_ub:
            out[tag + "_ub"] = ub.reshape(1).numpy()
        out[tag + "_q_tok"], out[tag + "_s_tok"] = bits8(q_tok), s_tok.numpy()
        out[tag + "_q_ten"], out[tag + "_s_ten"] = bits8(q_ten), s_ten.numpy()
    # static quantisation with group shapes
    gnames = []
    for T, H, gs in [(128, 256, (-1, -1)), (128, 256, (-1, 1)), (128, 256, (1, -1)), (128, 256, (-1, 128)), (128, 256, (1, 128)),
                     (128, 256, (128, 128)), (128, 256, (1, 64)), (128, 256, (1, 16)), (128, 256, (4, 256))]:
        for dtype in (torch.bfloat16, torch.float32) if gs in ((-1, -1), (1, 16), (1, -1)) else (torch.bfloat16,):
            torch.manual_seed(0)
            tag = f"static_{T}x{H}_{gs[0]}_{gs[1]}_{'bf16' if dtype == torch.bfloat16 else 'f32'}"
            x = torch.rand(T, H, dtype=dtype)
            q, sc = qu.scaled_quantize(x, gs, tq.FP8_DTYPE, compute_dtype=torch.float32)
            gnames.append(tag)
            out[tag + "_x"] = bits16(x) if dtype == torch.bfloat16 else x.numpy()
            out[tag + "_gs"] = np.array(gs)
            out[tag + "_scale"] = sc.numpy()
            out[tag + "_q"] = bits8(q)
    # RMSNorm (+ residual) -> per-token fp8 (the reference test's composite)
    nnames = []
    for T, H, add_res, with_ub in [(1, 64, False, False), (1, 1024, True, False), (5, 1027, True, True), (4, 5120, False, True),
                                   (3, 5137, True, False)]:
        torch.manual_seed(0)
        tag = f"normq_{T}x{H}_{int(add_res)}_{int(with_ub)}"
        w = torch.empty(H, dtype=torch.bfloat16).normal_(mean=1.0, std=0.1)
        x = torch.randn(T, H, dtype=torch.bfloat16) * (1 / H)
        res = torch.randn(T, H, dtype=torch.bfloat16) * (1 / H) if add_res else None
        if add_res:
            normed, new_res = _native(ln.fused_add_rms_norm)(x.clone(), res.clone(), w, 1e-6)
        else:
            normed, new_res = _native(ln.rms_norm)(x, w, 1e-6), None
        ub = torch.mean(normed).to(torch.float32) if with_ub else None
        q, sc = tq.ref_dynamic_per_token_quant(normed, tq.FP8_DTYPE, ub)
        nnames.append(tag)
        out[tag + "_x"], out[tag + "_w"] = bits16(x), bits16(w)
        if add_res:
            out[tag + "_res"], out[tag + "_new_res"] = bits16(res), bits16(new_res)
        if with_ub:
            out[tag + "_ub"] = ub.reshape(1).numpy()
        out[tag + "_normed"] = bits16(normed)
        out[tag + "_q"], out[tag + "_s"] = bits8(q), sc.numpy()
    # scaled product
    mnames = []
    for n, k, m, dtype, biased in [(1, 64, 64, torch.bfloat16, False), (4, 80, 80, torch.float16, True), (3, 512, 528, torch.bfloat16, True),
                                   (4, 1024, 256, torch.bfloat16, False), (33, 512, 256, torch.bfloat16, True),
                                   (200, 256, 384, torch.float16, False)]:
        torch.manual_seed(0)
        tag = f"mm_{n}x{k}x{m}_{'bf16' if dtype == torch.bfloat16 else 'f16'}_{int(biased)}"
        xav = (2 / k) ** 0.5
        A = (torch.rand(n, k) * 2 - 1) * xav
        B = (torch.rand(m, k) * 2 - 1) * xav
        A8, sa = tq.ref_dynamic_per_tensor_fp8_quant(A)
        B8, sb = tq.ref_dynamic_per_tensor_fp8_quant(B)
        bias = (torch.rand(m, dtype=dtype) * 2 - 1) if biased else None
        ref = torch._scaled_mm(A8, B8.t(), out_dtype=dtype, scale_a=sa, scale_b=sb, bias=bias)
        mnames.append(tag)
        out[tag + "_a"], out[tag + "_b"] = bits8(A8), bits8(B8)
        out[tag + "_sa"], out[tag + "_sb"] = sa.numpy(), sb.numpy()
        if biased:
            out[tag + "_bias"] = bits16(bias)
        out[tag + "_ref"] = bits16(ref)
        out[tag + "_dt"] = np.array([0 if dtype == torch.bfloat16 else 1])
    # per-token-group quantisation and the block-scaled product: the reference test's own restatements
    # (tests/kernels/quant_utils.py:157-180 and :91-154, imported) on the inputs of
    # tests/kernels/quantization/test_block_fp8.py:86-153 (sizes cut to what a fixture can hold)
    pnames = []
    for T, H, gs in [(7, 512, 64), (7, 512, 128), (7, 4096, 128), (3, 5120, 512), (33, 512, 128)]:
        torch.manual_seed(0)
        tag = f"ptg_{T}x{H}_{gs}"
        x = torch.rand(T, H, dtype=torch.bfloat16)
        q, sc = tq.native_per_token_group_quant_fp8(x, gs)
        pnames.append(tag)
        out[tag + "_x"], out[tag + "_gs"] = bits16(x), np.array([gs])
        out[tag + "_q"], out[tag + "_s"] = bits8(q), sc.numpy()
    bnames = []
    for M_, N_, K_ in [(1, 128, 256), (7, 512, 256), (8, 576, 1024), (83, 128, 3884 // 4 * 4), (1, 576, 4096), (130, 512, 384)]:
        torch.manual_seed(0)
        tag = f"blockmm_{M_}x{N_}x{K_}"
        fmax = 448.0
        A8 = ((torch.rand(M_, K_, dtype=torch.float32) - 0.5) * 2 * fmax).clamp(min=-fmax, max=fmax).to(tq.FP8_DTYPE)
        B8 = ((torch.rand(N_, K_, dtype=torch.float32) - 0.5) * 2 * fmax).clamp(min=-fmax, max=fmax).to(tq.FP8_DTYPE)
        kt, nt = -(-K_ // 128), -(-N_ // 128)
        As = torch.rand(M_, kt, dtype=torch.float32) * 1e-2
        Bs = torch.rand(nt, kt, dtype=torch.float32) * 1e-2
        ref = tq.native_w8a8_block_matmul(A8, B8, As, Bs, [128, 128], torch.bfloat16)
        bnames.append(tag)
        out[tag + "_a"], out[tag + "_b"] = bits8(A8), bits8(B8)
        out[tag + "_as"], out[tag + "_bs"] = As.numpy(), Bs.numpy()
        out[tag + "_ref"] = bits16(ref)
    save("fp8_w8a8", dyn_names=np.array(names), static_names=np.array(gnames), normq_names=np.array(nnames),
         mm_names=np.array(mnames), ptg_names=np.array(pnames), blockmm_names=np.array(bnames), **out)


GENERATORS = {"pack": gen_pack, "quantize": gen_quantize, "w4a16": gen_w4a16, "awq": gen_awq, "rmsnorm": gen_rmsnorm,
              "rope": gen_rope, "paged_attn": gen_paged_attn, "sampler": gen_sampler, "sampler_topk_topp": gen_sampler_topk_topp, "w8a16": gen_w8a16, "moe": gen_moe, "fp8_w8a8": gen_fp8_w8a8}

if __name__ == "__main__":
    for name in (sys.argv[1:] or GENERATORS):
        GENERATORS[name]()
