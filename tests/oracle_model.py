"""Llama-shaped forward pass assembled from the CPU oracle's functions (test infrastructure).

Consumes the checkpoint-layout tensors a LlamaRunner kept (keep_checkpoint=True) and generates
greedy tokens one position at a time, so the HIP runner's prefill + decode can be compared
token for token."""
import numpy as np
import torch

import oracle as orc


def _bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def _gemm(lin, x):
    kind, qweight, scales, qzeros, G = lin.checkpoint
    if kind == "gptq":
        return orc.w4a16_gemm(x, qweight.numpy(), _bits(scales), None, None, G=G)
    return orc.w4a16_gemm(x, qweight.numpy(), _bits(scales), qzeros.numpy(), None, G=G, layout=orc.W_AWQ,
                          zp_mode=orc.ZP_AWQ)


def _moe(blk, x):
    """The sparse block of mixtral.py:126-130 with the oracle's pieces: 16-bit router, fp32 softmax top-k with
    renormalisation, experts and weighted sum as tests/kernels/utils.py:912-1078 states them."""
    import moe_ref
    (q13, s13, z13), (q2, s2, z2) = blk.checkpoint
    logits = orc.linear_h16(x, _bits(blk.router))
    w, ids = moe_ref.route(logits, blk.top_k, renormalize=True)
    np_ = lambda t: None if t is None else t.numpy()
    return moe_ref.moe(x, (q13.numpy(), _bits(s13), np_(z13)), (q2.numpy(), _bits(s2), np_(z2)), w, ids, blk.G)


def assert_tokens_match(got, want, logits):
    """got/want: [step][sequence] token ids; logits: [step][sequence] fp32 oracle logits.  Tokens must
    agree, except that a sequence may leave the oracle's path at a near-tie: fp32 sums are not
    associative, so a logit pair closer than two 16-bit ulps may order either way.  After such a
    step the sequence has a different history and is no longer compared."""
    B = len(want[0])
    for b in range(B):
        for i in range(len(want)):
            g, w = got[i][b], want[i][b]
            if g == w:
                continue
            lg = logits[i][b]
            tol = 2.0 ** -6 * max(abs(float(lg[w])), 1e-3)
            assert float(lg[w] - lg[g]) <= tol, (
                f"sequence {b} step {i}: token {g} instead of {w}; oracle logits {lg[g]} vs {lg[w]}")
            break


def oracle_generate(runner, prompt, n_new, with_logits=False):
    """Returns [[tok per sequence] for each of n_new generated positions] (and, with_logits, the
    fp32 logits behind every one of them in the same nesting)."""
    cfg = runner.cfg
    B, T = prompt.shape
    D, H, KVH = cfg.head_dim, runner.heads, runner.kv_heads
    fp8 = runner.fp8
    x = 16 if fp8 else 8
    bs = runner.block_size
    nblk = runner.blocks_per_seq * runner.max_batch
    cdt = np.uint8 if fp8 else np.uint16
    kc = [np.zeros((nblk, KVH, D // x, bs, x), dtype=cdt) for _ in runner.layers]
    vc = [np.zeros((nblk, KVH, D, bs), dtype=cdt) for _ in runner.layers]
    ks, vs = float(runner.k_scale.item()), float(runner.v_scale.item())
    bt = runner.block_tables.cpu().numpy()
    embed, lm_head = _bits(runner.embed), _bits(runner.lm_head)
    cache = _bits(runner.cos_sin)
    eps = cfg.rms_eps

    def step(b, tok, pos):
        h = embed[tok][None, :].copy()
        residual = None
        slot = np.array([bt[b, pos // bs] * bs + pos % bs], dtype=np.int64)
        for li, L in enumerate(runner.layers):
            if residual is None:
                residual = h.copy()
                xn = orc.rms_norm(h, _bits(L["ln1"]), eps)
            else:
                xn, residual = orc.fused_add_rms_norm(h, residual, _bits(L["ln1"]), eps)
            qkv = _gemm(L["qkv"], xn)
            q, k, v = qkv[:, :H * D], qkv[:, H * D:(H + KVH) * D], qkv[:, (H + KVH) * D:]
            q, k = orc.rotary_embedding(np.array([pos]), q, k, cache, D, True)
            orc.reshape_and_cache(k.reshape(1, KVH, D), np.ascontiguousarray(v).reshape(1, KVH, D), kc[li], vc[li],
                                  slot, fp8=fp8, k_scale=ks, v_scale=vs)
            attn = orc.paged_attention(q.reshape(1, H, D), kc[li], vc[li], bt[b:b + 1], np.array([pos + 1]),
                                       runner.scale, fp8=fp8, k_scale=ks, v_scale=vs)
            o = _gemm(L["o"], attn.reshape(1, H * D))
            o, residual = orc.fused_add_rms_norm(o, residual, _bits(L["ln2"]), eps)
            if "moe" in L:
                h = _moe(L["moe"], o)
                continue
            gu = _gemm(L["gate_up"], o)
            h = _gemm(L["down"], orc.silu_and_mul(gu))
        hn, _ = orc.fused_add_rms_norm(h, residual, _bits(runner.norm), eps)
        logits = orc.linear_h16(hn, lm_head)
        return int(orc.argmax_h16(logits)[0]), orc.h16_to_f32(logits[0])

    out = [[0] * B for _ in range(n_new)]
    lgs = [[None] * B for _ in range(n_new)]
    for b in range(B):
        nxt = lg = None
        for pos in range(T):
            nxt, lg = step(b, int(prompt[b, pos]), pos)
        out[0][b], lgs[0][b] = nxt, lg
        for i in range(1, n_new):
            nxt, lg = step(b, nxt, T + i - 1)
            out[i][b], lgs[i][b] = nxt, lg
    return (out, lgs) if with_logits else out
