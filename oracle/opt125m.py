"""TEST INFRASTRUCTURE -- CPU restatement of an OPT-125m-shaped bf16 greedy decode.

This is the checker/baseline side of the build: only tests, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it; nothing the product path runs does.

What it restates: the decoder the reference runs as its CPU-backend plumbing case,
vllm/model_executor/models/opt.py:
  * learned positions with the +2 offset (opt.py:59-69),
  * pre-LayerNorm decoder layers: LN -> qkv (bias) -> causal attention over the cache -> out_proj (bias)
    -> + residual; LN -> fc1 (bias) -> ReLU -> fc2 (bias) -> + residual (opt.py:146-196),
  * final LayerNorm (opt.py:247-251, 289-290), logits through the tied embedding matrix (opt.py:353),
  * greedy argmax.
Every tensor between two ops is bf16, every op accumulates in fp32 and rounds its output once, as the
torch CPU kernels behind the reference's CPU backend do.

Parity: the reference holds no fixture of OPT-125m activations or tokens that could be checked without
its checkpoint, so whole-model tokens are "parity unpinned".  The primitives are pinned:
tests/test_opt_oracle_cpu.py checks LayerNorm and linear+bias against the torch CPU bf16 kernels (the
ones the reference's CPU backend calls for this model) and that decoding with the KV cache equals
recomputing the whole prefix.  The module exists to time the shape of that workload on the host cores.
"""
import numpy as np

from . import f32_to_h16, h16_to_f32, linear_h16

SHAPE = dict(hidden=768, layers=12, heads=12, head_dim=64, ffn=3072, vocab=50272, max_pos=2048)


def _r(x):
    """Round fp32 values to bf16 and back (what storing a bf16 tensor does)."""
    return h16_to_f32(f32_to_h16(np.ascontiguousarray(x, dtype=np.float32)))


def _layer_norm(x, w, b, eps=1e-5):
    x = x.astype(np.float32)
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return _r((x - mu) / np.sqrt(var + eps) * w + b)


def _linear(x, w_bits, bias):
    """bf16 x (T, K) times bf16 weight (N, K)^T, fp32 accumulate, + bias, one rounding."""
    y = h16_to_f32(linear_h16(f32_to_h16(x), w_bits))
    # the CPU GEMM adds the bias in fp32 before the single rounding; linear_h16 has rounded already, so
    # the bias add rounds a second time -- at most one bf16 step from the fused form, irrelevant for timing
    return _r(y + bias) if bias is not None else y


def make_weights(seed=0, shape=SHAPE, std=0.02):
    """Random-init weights of the OPT-125m architecture (bf16 bit patterns for matrices)."""
    rng = np.random.default_rng(seed)
    H, F, V = shape["hidden"], shape["ffn"], shape["vocab"]

    def mat(n, k, s=std):
        return f32_to_h16((rng.standard_normal((n, k)) * s).astype(np.float32))

    def vec(n, s=0.02, base=0.0):
        return _r((rng.standard_normal(n) * s + base).astype(np.float32))

    w = {"embed": mat(V, H), "pos": _r((rng.standard_normal((shape["max_pos"] + 2, H)) * 0.02).astype(np.float32)),
         "ln_f": (vec(H, 0.02, 1.0), vec(H)), "layers": []}
    for _ in range(shape["layers"]):
        w["layers"].append({
            "ln1": (vec(H, 0.02, 1.0), vec(H)), "qkv": mat(3 * H, H), "qkv_b": vec(3 * H),
            "out": mat(H, H), "out_b": vec(H),
            "ln2": (vec(H, 0.02, 1.0), vec(H)), "fc1": mat(F, H), "fc1_b": vec(F), "fc2": mat(H, F), "fc2_b": vec(H)})
    return w


class OptDecoder:
    """Greedy decode, one sequence, contiguous KV cache (the CPU backend's block size of 128 is a layout
    detail that does not change the arithmetic)."""

    def __init__(self, weights, shape=SHAPE, max_len=256):
        self.w, self.s = weights, shape
        L, nh, hd = shape["layers"], shape["heads"], shape["head_dim"]
        self.k = np.zeros((L, max_len, nh, hd), dtype=np.float32)
        self.v = np.zeros((L, max_len, nh, hd), dtype=np.float32)
        self.len = 0

    def forward(self, token_ids):
        """Append `token_ids` (T,) to the sequence; returns the next token id after the last one."""
        s, w = self.s, self.w
        T, H, nh, hd = len(token_ids), s["hidden"], s["heads"], s["head_dim"]
        pos = np.arange(self.len, self.len + T)
        x = _r(h16_to_f32(w["embed"][token_ids]) + w["pos"][pos + 2])
        scale = hd ** -0.5
        for li, lw in enumerate(w["layers"]):
            h = _layer_norm(x, *lw["ln1"])
            qkv = _linear(h, lw["qkv"], lw["qkv_b"])
            q, k, v = (qkv[:, i * H:(i + 1) * H].reshape(T, nh, hd) for i in range(3))
            self.k[li, self.len:self.len + T] = k
            self.v[li, self.len:self.len + T] = v
            ctx = self.len + T
            kk, vv = self.k[li, :ctx], self.v[li, :ctx]
            logits = np.einsum("tnd,cnd->ntc", q, kk) * scale
            mask = np.arange(ctx)[None, :] > pos[:, None]
            logits = np.where(mask[None], -np.inf, logits)
            logits -= logits.max(axis=-1, keepdims=True)
            p = np.exp(logits)
            p /= p.sum(axis=-1, keepdims=True)
            a = _r(np.einsum("ntc,cnd->tnd", p, vv).reshape(T, H))
            x = _r(x + _linear(a, lw["out"], lw["out_b"]))
            h = _layer_norm(x, *lw["ln2"])
            h = np.maximum(_linear(h, lw["fc1"], lw["fc1_b"]), 0.0)
            x = _r(x + _linear(h, lw["fc2"], lw["fc2_b"]))
        self.len += T
        x = _layer_norm(x[-1:], *w["ln_f"])
        logits = h16_to_f32(linear_h16(f32_to_h16(x), w["embed"]))
        return int(np.argmax(logits[0]))


def greedy_decode(weights, prompt, steps, shape=SHAPE):
    """Prompt + `steps` greedy tokens; returns the generated ids."""
    d = OptDecoder(weights, shape, max_len=len(prompt) + steps + 1)
    tok = d.forward(np.asarray(prompt, dtype=np.int64))
    out = [tok]
    for _ in range(steps - 1):
        tok = d.forward(np.asarray([tok], dtype=np.int64))
        out.append(tok)
    return out
