"""The OPT-125m-shaped CPU decode of the oracle (the timed `cpu_baseline` for the reference's CPU plumbing
case): its LayerNorm and linear agree with the torch CPU bf16 kernels the reference's CPU backend calls,
and decoding with the KV cache equals recomputing the whole prefix."""
import numpy as np
import torch

from oracle import f32_to_h16, h16_to_f32, opt125m

SMALL = dict(hidden=64, layers=2, heads=4, head_dim=16, ffn=128, vocab=500, max_pos=64)


def _bf16(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16)


def test_layer_norm_matches_torch_cpu_bf16():
    rng = np.random.default_rng(0)
    x = opt125m._r(rng.standard_normal((7, 768)).astype(np.float32))
    w = opt125m._r((1 + 0.1 * rng.standard_normal(768)).astype(np.float32))
    b = opt125m._r((0.1 * rng.standard_normal(768)).astype(np.float32))
    got = opt125m._layer_norm(x, w, b)
    ref = torch.nn.functional.layer_norm(_bf16(x), (768,), _bf16(w), _bf16(b), 1e-5).float().numpy()
    # same fp32 statistics, one rounding: at most one bf16 step apart (2^-8 relative)
    assert np.all(np.abs(got - ref) <= np.abs(ref) * 2.0 ** -7 + 1e-6)
    assert np.mean(got == ref) > 0.98


def test_linear_with_bias_matches_torch_cpu_bf16():
    rng = np.random.default_rng(1)
    x = opt125m._r(rng.standard_normal((3, 768)).astype(np.float32))
    w = f32_to_h16((0.05 * rng.standard_normal((96, 768))).astype(np.float32))
    b = opt125m._r((0.1 * rng.standard_normal(96)).astype(np.float32))
    got = opt125m._linear(x, w, b)
    ref = torch.nn.functional.linear(_bf16(x), _bf16(h16_to_f32(w)), _bf16(b)).float().numpy()
    assert np.all(np.abs(got - ref) <= np.abs(ref) * 2.0 ** -6 + 2e-3)


def test_cached_decode_equals_prefix_recompute():
    w = opt125m.make_weights(1, SMALL, std=0.3)
    prompt = [3, 7, 11, 200, 45]
    cached = opt125m.greedy_decode(w, prompt, 6, SMALL)
    seq, full = list(prompt), []
    for _ in range(6):
        d = opt125m.OptDecoder(w, SMALL, max_len=32)
        full.append(d.forward(np.asarray(seq)))
        seq.append(full[-1])
    assert cached == full
    assert len(set(cached)) > 1   # the random model is not degenerate at this weight scale
