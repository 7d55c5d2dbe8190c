"""Parity of the HIP kernels (called through the C ABI via mi355q.ops) against the CPU
oracle and the golden fixtures.  Needs a real MI355X: run with `-m gpu`."""
import numpy as np
import pytest
import torch

import oracle as orc
from mi355q import _lib, ops

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TDT = {orc.BF16: torch.bfloat16, orc.FP16: torch.float16}


@pytest.fixture(autouse=True)
def _kernel_route(monkeypatch):
    """These tests check the skinny W4A16 kernel itself at every M; the dequantise + dense GEMM
    route for long prompts has its own test below."""
    monkeypatch.setattr(ops, "DEQUANT_MATMUL_MIN_TOKENS", 1 << 30)
    monkeypatch.setattr(ops, "MM_MIN_TOKENS", 1 << 30)


def to_dev(bits, dt=orc.BF16):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(TDT[dt]).to(DEV)


def to_bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def f32(bits, dt=orc.BF16):
    return orc.h16_to_f32(bits, dt)


def i32(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(DEV)


# ---------------------------------------------------------------------------
# W4A16
# ---------------------------------------------------------------------------
def test_w4a16_golden_triton_cases(golden):
    """tests/kernels/quantization/test_triton_w4a16.py:112-162 inputs; rtol=atol=1e-2 (:162)."""
    g = golden("w4a16_triton_cases")
    for tag in g["names"]:
        tag = str(tag)
        N = int(tag.split("_n")[1].split("_")[0])
        if N % 64:
            continue
        dt = orc.BF16 if tag.startswith("bf16") else orc.FP16
        G = int(tag.split("_g")[1].split("_")[0])
        has_zp = tag.endswith("zp1")
        w = ops.w4a16_repack(i32(g[tag + "_b"]), to_dev(g[tag + "_s"], dt), i32(g[tag + "_z"]) if has_zp else None,
                             G, _lib.W_NSEQ, _lib.ZP_NSEQ if has_zp else _lib.ZP_CONST)
        out = ops.w4a16_gemm(to_dev(g[tag + "_a"], dt), w)
        np.testing.assert_allclose(f32(to_bits(out), dt), f32(g[tag + "_ref"], dt), rtol=1e-2, atol=1e-2, err_msg=tag)


def test_w4a16_golden_gptq_and_awq_routes(golden):
    g = golden("w4a16_gptq_route")
    for tag in g["names"]:
        tag = str(tag)
        G = int(tag.split("_g")[1])
        w = ops.w4a16_repack(i32(g[tag + "_qweight"]), to_dev(g[tag + "_scales"]), None, G)
        out = ops.w4a16_gemm(to_dev(g[tag + "_a"]), w)
        np.testing.assert_allclose(f32(to_bits(out)), f32(g[tag + "_ref"]), rtol=1e-2, atol=1e-2, err_msg=tag)
        deq = ops.w4a16_dequantize(w)
        assert np.array_equal(to_bits(deq), g[tag + "_w_ref"]), tag
    g = golden("awq")
    for tag in g["names"]:
        tag = str(tag)
        G = int(tag.split("_g")[1])
        w = ops.w4a16_repack(i32(g[tag + "_qweight"]), to_dev(g[tag + "_scales"], orc.FP16), i32(g[tag + "_qzeros"]),
                             G, _lib.W_AWQ, _lib.ZP_AWQ)
        assert np.array_equal(to_bits(ops.w4a16_dequantize(w)), g[tag + "_deq"]), tag
        out = ops.w4a16_gemm(to_dev(g[tag + "_a"], orc.FP16), w)
        np.testing.assert_allclose(f32(to_bits(out), orc.FP16), f32(g[tag + "_ref"], orc.FP16), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("M,K,N,G,split", [(1, 4096, 4096, 128, 0), (1, 4096, 4096, 128, 1), (3, 1024, 1280, 128, 0),
                                           (8, 3584, 512, 128, 0), (16, 512, 256, 32, 3), (17, 256, 128, 64, 0),
                                           (40, 512, 192, -1, 2), (5, 1024, 64, 128, 8)])
def test_w4a16_vs_oracle(M, K, N, G, split):
    rng = np.random.default_rng(M * 1000 + N)
    Ge = K if G == -1 else G
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    a = orc.f32_to_h16(0.25 * rng.standard_normal((M, K)).astype(np.float32))
    s = orc.f32_to_h16((0.05 * rng.random((K // Ge, N))).astype(np.float32))
    bias = orc.f32_to_h16(rng.standard_normal(N).astype(np.float32))
    qw = orc.gptq_pack(q)
    ref = orc.w4a16_gemm(a, qw, s, None, bias, G=Ge)
    w = ops.w4a16_repack(i32(qw), to_dev(s), None, G)
    out = ops.w4a16_gemm(to_dev(a), w, bias=to_dev(bias), split_k=split)
    np.testing.assert_allclose(f32(to_bits(out)), f32(ref), rtol=1e-2, atol=1e-2)
    # fp32 math on both sides: all but sum-order ties are bit-equal
    assert (to_bits(out) == ref).mean() > 0.98
    # run-to-run determinism (no atomics in the split-K path)
    out2 = ops.w4a16_gemm(to_dev(a), w, bias=to_dev(bias), split_k=split)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("M,K,N,G,awq,dt", [(8, 14336, 4096, 128, False, orc.BF16), (16, 4096, 4096, 128, False, orc.BF16),
                                            (32, 14336, 4096, 128, False, orc.BF16), (21, 1024, 256, 256, True, orc.FP16),
                                            (5, 3584, 8192, 128, True, orc.BF16), (9, 1024, 8192, 512, False, orc.FP16),
                                            (33, 2048, 128, 128, False, orc.BF16)])
def test_w4a16_several_rows_on_narrow_strips(M, K, N, G, awq, dt):
    """5 or more rows against a 16-channel-strip weight (row-parallel o_proj / down_proj shapes at
    decode batch 8..32): adjacent strips side by side, K sliced over workgroups, fp32 partial
    tiles summed in slice order.  Same bar as the single-row kernel, and run-to-run identical."""
    rng = np.random.default_rng(M * 7 + N)
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    a = orc.f32_to_h16(0.25 * rng.standard_normal((M, K)).astype(np.float32), dt)
    s = orc.f32_to_h16((0.05 * rng.random((K // G, N))).astype(np.float32), dt)
    bias = orc.f32_to_h16(rng.standard_normal(N).astype(np.float32), dt)
    if awq:
        z = rng.integers(0, 16, size=(K // G, N), dtype=np.int32)
        qw, qz = orc.pack_cols(q, awq_order=True), orc.pack_cols(z, awq_order=True)
        ref = orc.w4a16_gemm(a, qw, s, qz, bias, G=G, dt=dt, layout=orc.W_AWQ, zp_mode=orc.ZP_AWQ)
        w = ops.w4a16_repack(i32(qw), to_dev(s, dt), i32(qz), G, _lib.W_AWQ, _lib.ZP_AWQ, tile=16)
    else:
        qw = orc.gptq_pack(q)
        ref = orc.w4a16_gemm(a, qw, s, None, bias, G=G, dt=dt)
        w = ops.w4a16_repack(i32(qw), to_dev(s, dt), None, G, tile=16)
    assert w.tile == 16
    out = ops.w4a16_gemm(to_dev(a, dt), w, bias=to_dev(bias, dt))
    np.testing.assert_allclose(f32(to_bits(out), dt), f32(ref, dt), rtol=1e-2, atol=1e-2)
    assert (to_bits(out) == ref).mean() > (0.98 if dt == orc.BF16 else 0.9)
    for _ in range(3):
        assert torch.equal(out, ops.w4a16_gemm(to_dev(a, dt), w, bias=to_dev(bias, dt)))
    # the arrival counters are back at zero: a single-row split-K GEMM on the same workspace still works
    a1 = a[:1]
    out1 = ops.w4a16_gemm(to_dev(a1, dt), w, bias=to_dev(bias, dt), split_k=4)
    assert np.array_equal(to_bits(out1), to_bits(out[:1])) or np.allclose(f32(to_bits(out1), dt), f32(to_bits(out[:1]), dt), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("M,K,N,G,tile,awq,dt", [
    (128, 4096, 28672, 128, 112, False, orc.BF16),   # gate_up: 112-wide strips, 2 K-slices
    (128, 14336, 4096, 128, 16, False, orc.BF16),    # down: 16-wide strips, 4 side by side per wave, K sliced
    (17, 4096, 6144, 128, 32, False, orc.BF16),      # qkv: 32-wide strips
    (200, 1024, 512, 256, 64, True, orc.FP16),       # 64-wide strips, zero points, ragged last row block
    (64, 512, 256, 128, 16, True, orc.BF16),
    (333, 2048, 896, 128, 112, False, orc.FP16),
])
def test_w4a16_mm_vs_oracle(M, K, N, G, tile, awq, dt):
    """Tiled matrix-core kernel for prompt-sized token counts: same fp32 (q - z) * s arithmetic as
    the decode kernels, so the same bar against the oracle (reference tolerance rtol = atol = 1e-2,
    tests/kernels/quantization/test_triton_w4a16.py:162, and nearly all outputs bit-equal)."""
    rng = np.random.default_rng(M + K + N)
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    a = orc.f32_to_h16(0.25 * rng.standard_normal((M, K)).astype(np.float32), dt)
    s = orc.f32_to_h16((0.05 * rng.random((K // G, N))).astype(np.float32), dt)
    bias = orc.f32_to_h16(rng.standard_normal(N).astype(np.float32), dt)
    if awq:
        z = rng.integers(0, 16, size=(K // G, N), dtype=np.int32)
        qw, qz = orc.pack_cols(q, awq_order=True), orc.pack_cols(z, awq_order=True)
        ref = orc.w4a16_gemm(a, qw, s, qz, bias, G=G, dt=dt, layout=orc.W_AWQ, zp_mode=orc.ZP_AWQ)
        w = ops.w4a16_repack(i32(qw), to_dev(s, dt), i32(qz), G, _lib.W_AWQ, _lib.ZP_AWQ, tile=tile)
    else:
        qw = orc.gptq_pack(q)
        ref = orc.w4a16_gemm(a, qw, s, None, bias, G=G, dt=dt)
        w = ops.w4a16_repack(i32(qw), to_dev(s, dt), None, G, tile=tile)
    assert w.tile == tile
    out = ops.w4a16_mm(to_dev(a, dt), w, bias=to_dev(bias, dt))
    # the bias is added to the ROUNDED product (reference semantics), so a one-ulp difference of a
    # large product can survive a cancelling bias as a large relative error of a small sum: allow
    # a few such outputs per million, each within one ulp of the product's magnitude
    got, want = f32(to_bits(out), dt), f32(ref, dt)
    bad = ~np.isclose(got, want, rtol=1e-2, atol=1e-2)
    assert bad.mean() < 1e-5
    prod_mag = np.abs(want - f32(bias, dt)[None, :]) + 1.0
    assert np.all(np.abs(got - want)[bad] <= prod_mag[bad] * (2.0 ** -7 if dt == orc.BF16 else 2.0 ** -10))
    assert (to_bits(out) == ref).mean() > (0.98 if dt == orc.BF16 else 0.9)
    assert torch.equal(out, ops.w4a16_mm(to_dev(a, dt), w, bias=to_dev(bias, dt)))


@pytest.mark.parametrize("tile,M,awq,options", [
    (16, 200, False, {"MI355Q_MM_RING_MIN": "129"}),       # wave halves split the rows of 128-row tiles
    (16, 200, True, {"MI355Q_MM_RING_MIN": "129"}),
    (16, 200, False, {"MI355Q_MM_RING_MIN": "100000"}),    # ... the chunks of 64-row tiles
    (64, 300, True, {"MI355Q_MM_RING_MIN": "100000"}),
    (32, 100, False, {"MI355Q_MM_RING_K": "0"}),           # no K-split form: the register-prefetch kernel
    (16, 200, False, {"MI355Q_MM_RING": "0"}),
    (112, 200, False, {}),                                 # 112-wide strips: subtile halves
    (112, 200, False, {"MI355Q_MM_RING_ROWS7": "1", "MI355Q_MM_RING_MIN": "129"}),
    (112, 100, True, {}),                                  # ... two strips per workgroup, K split inside
    (112, 100, False, {"MI355Q_MM_RING_SUBK": "0"}),
    (112, 100, False, {"MI355Q_MM_RING": "0"}),
])
def test_w4a16_mm_every_workgroup_form(tile, M, awq, options):
    """The prompt kernel's forms (what the two wave halves of a workgroup split, or the register-prefetch
    kernel) are chosen by shape; here each is forced on one small problem and held to the same bar."""
    K, N, G, dt = 2048, 896 if tile == 112 else 1024, 128, orc.BF16
    rng = np.random.default_rng(tile + M)
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    a = orc.f32_to_h16(0.25 * rng.standard_normal((M, K)).astype(np.float32), dt)
    s = orc.f32_to_h16((0.05 * rng.random((K // G, N))).astype(np.float32), dt)
    if awq:
        z = rng.integers(0, 16, size=(K // G, N), dtype=np.int32)
        qw, qz = orc.pack_cols(q, awq_order=True), orc.pack_cols(z, awq_order=True)
        ref = orc.w4a16_gemm(a, qw, s, qz, None, G=G, dt=dt, layout=orc.W_AWQ, zp_mode=orc.ZP_AWQ)
        w = ops.w4a16_repack(i32(qw), to_dev(s, dt), i32(qz), G, _lib.W_AWQ, _lib.ZP_AWQ, tile=tile)
    else:
        qw = orc.gptq_pack(q)
        ref = orc.w4a16_gemm(a, qw, s, None, None, G=G, dt=dt)
        w = ops.w4a16_repack(i32(qw), to_dev(s, dt), None, G, tile=tile)
    try:
        for k, v in options.items():
            _lib.set_option(k, v)
        out = ops.w4a16_mm(to_dev(a, dt), w)
        torch.cuda.synchronize()
    finally:
        for k in options:
            _lib.set_option(k, None)
    np.testing.assert_allclose(f32(to_bits(out), dt), f32(ref, dt), rtol=1e-2, atol=1e-2)
    assert (to_bits(out) == ref).mean() > 0.98


@pytest.mark.parametrize("M,K,N,with_bias", [(128, 4096, 4096, False), (32, 14336, 4096, True), (600, 1024, 4096, False),
                                              (20, 512, 8192, True), (8, 14336, 4096, False), (13, 14336, 4096, True)])
def test_w4a16_mm_add_rms_norm_equals_the_two_steps(M, K, N, with_bias):
    """Projection + residual add + RMSNorm as one call: the residual is bit-identical to
    w4a16_gemm followed by fused_add_rms_norm (same slice-order sums, same roundings); the normalised rows
    may differ by one rounding where the sum of squares was added in another order.  Also against the
    oracle's GEMM + fused_add_rms_norm."""
    rng = np.random.default_rng(M + K + N)
    G = 128
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    a = orc.f32_to_h16(0.25 * rng.standard_normal((M, K)).astype(np.float32))
    s = orc.f32_to_h16((0.05 * rng.random((K // G, N))).astype(np.float32))
    res = orc.f32_to_h16(rng.standard_normal((M, N)).astype(np.float32))
    nw = orc.f32_to_h16((1 + 0.1 * rng.standard_normal(N)).astype(np.float32))
    bias = orc.f32_to_h16(rng.standard_normal(N).astype(np.float32)) if with_bias else None
    qw = orc.gptq_pack(q)
    w = ops.w4a16_repack(i32(qw), to_dev(s), None, G)
    bias_d = None if bias is None else to_dev(bias)
    assert ops.w4a16_mm_add_rms_norm_supported(M, w)
    # the two-step reference on the same kernel: 5..16 rows K-sliced rows kernel, 17+ tiled kernel
    c = ops.w4a16_mm(to_dev(a), w, bias=bias_d) if M >= 17 else ops.w4a16_gemm(to_dev(a), w, bias=bias_d)
    r2 = to_dev(res).clone()
    ops.fused_add_rms_norm(c, r2, to_dev(nw), 1e-5)
    r1 = to_dev(res).clone()
    out = ops.w4a16_mm_add_rms_norm(to_dev(a), w, r1, to_dev(nw), 1e-5, bias=bias_d)
    assert torch.equal(r1, r2)
    assert (to_bits(out) == to_bits(c)).mean() > 0.999
    np.testing.assert_allclose(f32(to_bits(out)), f32(to_bits(c)), rtol=2.0 ** -7, atol=1e-6)
    ref_c = orc.w4a16_gemm(a, qw, s, None, bias, G=G)
    ref_x, ref_r = orc.fused_add_rms_norm(ref_c, res, nw, 1e-5)
    assert (to_bits(r1) == ref_r).mean() > 0.97
    np.testing.assert_allclose(f32(to_bits(out)), f32(ref_x), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("M,K,nh,nkv,head,with_bias", [(128, 4096, 32, 8, 128, False), (40, 1024, 8, 4, 64, True),
                                                        (300, 2048, 14, 1, 128, False)])
def test_w4a16_mm_rope_and_cache_equals_the_sequence(fp8, M, K, nh, nkv, head, with_bias):
    """qkv projection + RoPE + cache write as one call = w4a16_mm, then rope_and_cache on the column
    ranges: identical bits in qkv and in both caches (K-slices merged in the same order)."""
    rng = np.random.default_rng(M + K + nh)
    G, bs, max_pos = 128, 16, 512
    N = (nh + 2 * nkv) * head
    nblk = (M + bs - 1) // bs + 2
    x = 16 if fp8 else 8
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    a = to_dev(orc.f32_to_h16(0.25 * rng.standard_normal((M, K)).astype(np.float32)))
    s = orc.f32_to_h16((0.05 * rng.random((K // G, N))).astype(np.float32))
    bias = to_dev(orc.f32_to_h16(rng.standard_normal(N).astype(np.float32))) if with_bias else None
    w = ops.w4a16_repack(i32(orc.gptq_pack(q)), to_dev(s), None, G)
    pos = torch.from_numpy(rng.integers(0, max_pos, size=M).astype(np.int64)).to(DEV)
    inv = 1.0 / (10000.0 ** (np.arange(0, head, 2, dtype=np.float32) / head))
    fr = np.arange(max_pos, dtype=np.float32)[:, None] * inv[None, :]
    cache = to_dev(orc.f32_to_h16(np.concatenate([np.cos(fr), np.sin(fr)], axis=1).astype(np.float32)))
    slots_np = rng.permutation(nblk * bs)[:M].astype(np.int64)
    slots_np[M // 2] = -1
    slots = torch.from_numpy(slots_np).to(DEV)
    tdt = torch.uint8 if fp8 else torch.bfloat16
    kvd = "fp8" if fp8 else "auto"
    ks = torch.tensor(0.05 if fp8 else 1.0, dtype=torch.float32, device=DEV)
    vs = torch.tensor(0.02 if fp8 else 1.0, dtype=torch.float32, device=DEV)

    def caches():
        return (torch.zeros((nblk, nkv, head // x, bs, x), dtype=tdt, device=DEV),
                torch.zeros((nblk, nkv, head, bs), dtype=tdt, device=DEV))

    kc1, vc1 = caches()
    ref = ops.w4a16_mm(a, w, bias=bias)
    qs, ksz = nh * head, nkv * head
    ops.rope_and_cache(pos, ref[:, :qs], ref[:, qs:qs + ksz], ref[:, qs + ksz:], head, cache, True, kc1, vc1, slots, kvd,
                       ks, vs)
    kc2, vc2 = caches()
    got = ops.w4a16_mm_rope_and_cache(a, w, pos, head, nh, nkv, cache, True, kc2, vc2, slots, kvd, ks, vs, bias=bias)
    assert torch.equal(got, ref)
    assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2)
    assert int((kc2 != 0).sum()) > 0


def test_w4a16_mm_silu_pairs_and_routing(monkeypatch):
    """pair_halves weight + SiLU epilogue = silu_and_mul of the plain projection; w4a16_gemm routes
    17..1024 tokens to the tiled kernel and refuses nothing it used to serve."""
    rng = np.random.default_rng(5)
    M, K, I, G = 96, 1024, 1792, 128
    N = 2 * I
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    a = orc.f32_to_h16(0.5 * rng.standard_normal((M, K)).astype(np.float32))
    s = orc.f32_to_h16((0.03 * rng.random((K // G, N)) + 0.005).astype(np.float32))
    qw = orc.gptq_pack(q)
    y = orc.w4a16_gemm(a, qw, s, None, None, G=G)
    ref = orc.silu_and_mul(y)
    w = ops.w4a16_repack(i32(qw), to_dev(s), None, G, pair_halves=True)
    act = ops.w4a16_mm(to_dev(a), w, silu_pairs=True)
    assert act.shape == (M, I)
    np.testing.assert_allclose(f32(to_bits(act)), f32(ref), rtol=2e-2, atol=1e-2)
    assert (to_bits(act) == ref).mean() > 0.97
    monkeypatch.setattr(ops, "MM_MIN_TOKENS", 17)
    plain = ops.unpair_halves(ops.w4a16_gemm(to_dev(a), w))
    assert (to_bits(plain) == y).mean() > 0.98


@pytest.mark.parametrize("M,K,N,G,awq", [(1, 4096, 4096, 64, False), (3, 4096, 28672, 32, False), (1, 14336, 4096, 32, True),
                                          (2, 4096, 6144, 64, True), (16, 1024, 4096, 32, False)])
def test_w4a16_small_scale_groups_on_the_decode_kernel(M, K, N, G, awq):
    """Scale groups of 32 / 64 input channels (AutoRound and GPTQ checkpoints use them) on layers wide
    enough for the one-workgroup-per-strip kernel: group sums folded inside the 128-k chunk.  Same
    bar as group size 128; the add+RMSNorm prologue and the SiLU-pair epilogue work on them too."""
    rng = np.random.default_rng(K + N + G)
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    a = orc.f32_to_h16(0.25 * rng.standard_normal((M, K)).astype(np.float32))
    s = orc.f32_to_h16((0.05 * rng.random((K // G, N))).astype(np.float32))
    if awq:
        z = rng.integers(0, 16, size=(K // G, N), dtype=np.int32)
        qw, qz = orc.pack_cols(q, awq_order=True), orc.pack_cols(z, awq_order=True)
        kw = dict(layout=orc.W_AWQ, zp_mode=orc.ZP_AWQ)
        rp = lambda **k: ops.w4a16_repack(i32(qw), to_dev(s), i32(qz), G, _lib.W_AWQ, _lib.ZP_AWQ, **k)
    else:
        qw, qz, kw = orc.gptq_pack(q), None, {}
        rp = lambda **k: ops.w4a16_repack(i32(qw), to_dev(s), None, G, **k)
    w = rp()
    assert N // w.tile >= 128     # wide enough: not the split-K capable fallback
    ref = orc.w4a16_gemm(a, qw, s, qz, None, G=G, **kw)
    out = ops.w4a16_gemm(to_dev(a), w)
    np.testing.assert_allclose(f32(to_bits(out)), f32(ref), rtol=1e-2, atol=1e-2)
    assert (to_bits(out) == ref).mean() > 0.98
    assert np.array_equal(to_bits(ops.w4a16_dequantize(w)), orc.f32_to_h16(orc.w4a16_dequant(qw, s, qz, G=G, **kw)))
    if M <= 4:
        r = orc.f32_to_h16(rng.standard_normal((M, K)).astype(np.float32))
        nw = orc.f32_to_h16((1 + 0.1 * rng.standard_normal(K)).astype(np.float32))
        xn, rn = orc.fused_add_rms_norm(a, r, nw, 1e-5)
        y, res = ops.w4a16_gemm_add_rms_norm(to_dev(a), to_dev(r), to_dev(nw), 1e-5, w)
        assert np.array_equal(to_bits(res), rn)
        refn = orc.w4a16_gemm(xn, qw, s, qz, None, G=G, **kw)
        assert (to_bits(y) == refn).mean() > 0.98
        if ops.w4a16_gate_up_silu_supported(K, N, G):
            wp = rp(pair_halves=True)
            act, _ = ops.w4a16_gate_up_silu(to_dev(xn), None, None, 1e-5, wp, add_norm=False)
            assert (to_bits(act) == orc.silu_and_mul(refn)).mean() > 0.97


def test_w4a16_many_token_route_vs_oracle(monkeypatch):
    """Above the token threshold the weight is dequantised to 16-bit and fed to a dense GEMM
    (exllama reconstruct route, csrc/quantization/gptq/q_gemm.cu:1509-1522): compare with the
    oracle's rounded-weight GEMM at the reference test tolerance (test_triton_w4a16.py:162)."""
    monkeypatch.setattr(ops, "DEQUANT_MATMUL_MIN_TOKENS", 32)
    rng = np.random.default_rng(11)
    M, K, N, G = 96, 1024, 384, 128
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    a = orc.f32_to_h16(0.25 * rng.standard_normal((M, K)).astype(np.float32))
    s = orc.f32_to_h16((0.05 * rng.random((K // G, N))).astype(np.float32))
    bias = orc.f32_to_h16(rng.standard_normal(N).astype(np.float32))
    qw = orc.gptq_pack(q)
    ref = orc.w4a16_gemm(a, qw, s, None, bias, G=G, round_weight=True)
    w = ops.w4a16_repack(i32(qw), to_dev(s), None, G)
    out = ops.w4a16_gemm(to_dev(a), w, bias=to_dev(bias))
    np.testing.assert_allclose(f32(to_bits(out)), f32(ref), rtol=1e-2, atol=1e-2)
    # the opt-in resident expanded copy: expanded at first use, kept, same product
    assert w.dense is None
    monkeypatch.setattr(ops, "DENSE_RESIDENT", True)
    first = ops.w4a16_gemm(to_dev(a), w, bias=to_dev(bias))
    kept = w.dense
    assert kept is not None and kept.shape == (K, N)
    again = ops.w4a16_gemm(to_dev(a), w, bias=to_dev(bias))
    assert w.dense is kept and torch.equal(first, out) and torch.equal(again, out)


def test_w4a16_split_k_invariance_of_argmax():
    """Greedy token ids must not depend on the split chosen: compare splits 1, 4, 8 on an
    lm_head-like GEMM against the oracle's argmax."""
    rng = np.random.default_rng(7)
    M, K, N, G = 4, 2048, 1024, 128
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    a = orc.f32_to_h16(rng.standard_normal((M, K)).astype(np.float32))
    s = orc.f32_to_h16((0.02 * rng.random((K // G, N)) + 0.01).astype(np.float32))
    qw = orc.gptq_pack(q)
    ref_ids = orc.argmax_h16(orc.w4a16_gemm(a, qw, s, None, None, G=G))
    w = ops.w4a16_repack(i32(qw), to_dev(s), None, G)
    for split in (1, 4, 8):
        ids = ops.argmax(ops.w4a16_gemm(to_dev(a), w, split_k=split))
        assert ids.cpu().numpy().tolist() == ref_ids.tolist()


@pytest.mark.parametrize("M,K,N", [(1, 1024, 256), (7, 14336 // 4, 512), (16, 512, 128)])
def test_w4a16_fused_silu_mul(M, K, N):
    rng = np.random.default_rng(3)
    G = 128
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    x = orc.f32_to_h16(rng.standard_normal((M, 2 * K)).astype(np.float32))
    s = orc.f32_to_h16((0.05 * rng.random((K // G, N))).astype(np.float32))
    qw = orc.gptq_pack(q)
    act = orc.silu_and_mul(x)
    ref = orc.w4a16_gemm(act, qw, s, None, None, G=G)
    w = ops.w4a16_repack(i32(qw), to_dev(s), None, G)
    out = ops.w4a16_gemm(to_dev(x), w, fuse_silu_mul=True)
    np.testing.assert_allclose(f32(to_bits(out)), f32(ref), rtol=1e-2, atol=2e-2)
    # and equals the unfused HIP pipeline bit for bit up to fp32 sum order
    act_d = torch.empty((M, K), dtype=torch.bfloat16, device=DEV)
    ops.silu_and_mul(act_d, to_dev(x))
    np.testing.assert_allclose(f32(to_bits(act_d)), f32(act), rtol=1e-2, atol=1e-3)
    out2 = ops.w4a16_gemm(act_d, w)
    np.testing.assert_allclose(f32(to_bits(out)), f32(to_bits(out2)), rtol=1e-2, atol=1e-2)


def test_w4a16_repack_device_matches_host():
    rng = np.random.default_rng(11)
    K, N, G = 256, 128, 64
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    z = rng.integers(0, 16, size=(K // G, N), dtype=np.int32)
    s = torch.from_numpy(rng.random((K // G, N)).astype(np.float32)).to(torch.bfloat16)
    for layout, src, zmode, qz in [(_lib.W_GPTQ, orc.gptq_pack(q), _lib.ZP_GPTQ_V1, orc.pack_cols(z)),
                                   (_lib.W_AWQ, orc.pack_cols(q, True), _lib.ZP_AWQ, orc.pack_cols(z, True)),
                                   (_lib.W_NSEQ, orc.pack_cols(q), _lib.ZP_NSEQ, orc.pack_cols(z))]:
        host = ops.w4a16_repack(torch.from_numpy(src), s, torch.from_numpy(qz), G, layout, zmode)
        dev = ops.w4a16_repack(i32(src), s.to(DEV), i32(qz), G, layout, zmode)
        assert torch.equal(host.packed, dev.packed.cpu())
        assert torch.equal(host.zc, dev.zc.cpu())


def test_w4a16_rejects_bad_shapes():
    s = torch.zeros((1, 48), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ops.w4a16_repack(torch.zeros((4, 48), dtype=torch.int32, device=DEV), s, None, 32)
    w = ops.w4a16_repack(torch.zeros((4, 64), dtype=torch.int32, device=DEV),
                         torch.zeros((1, 64), dtype=torch.bfloat16, device=DEV), None, 32)
    with pytest.raises(RuntimeError, match="feature size"):
        ops.w4a16_gemm(torch.zeros((1, 64), dtype=torch.bfloat16, device=DEV), w)
    assert ops.w4a16_gemm(torch.zeros((0, 32), dtype=torch.bfloat16, device=DEV), w).shape == (0, 64)


# ---------------------------------------------------------------------------
# unquantized skinny linear
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("M,K,N", [(1, 4096, 1000), (3, 768, 517), (8, 4096, 256), (13, 1024, 129), (2, 8192, 64),
                                   (16, 4096, 2000), (12, 8200, 300), (5, 136, 48), (8, 4096, 128256),
                                   (32, 4096, 2000), (17, 4096, 131), (23, 2304, 517), (32, 256, 128), (31, 8192, 300),
                                   (40, 1024, 200), (20, 136, 48)])
def test_linear_skinny(M, K, N):
    rng = np.random.default_rng(M + N)
    a = orc.f32_to_h16(rng.standard_normal((M, K)).astype(np.float32))
    w = orc.f32_to_h16((0.05 * rng.standard_normal((N, K))).astype(np.float32))
    ref = orc.linear_h16(a, w)
    out = ops.linear_skinny(to_dev(a), to_dev(w))
    np.testing.assert_allclose(f32(to_bits(out)), f32(ref), rtol=1.6e-2, atol=1e-2)
    assert (to_bits(out) == ref).mean() > 0.97


# ---------------------------------------------------------------------------
# norms, rope, activation (golden + oracle)
# ---------------------------------------------------------------------------
def test_rmsnorm_golden(golden):
    g = golden("rmsnorm")
    for tag in g["names"]:
        tag = str(tag)
        x, r, w = to_dev(g[tag + "_x"]), to_dev(g[tag + "_r"]), to_dev(g[tag + "_w"])
        y = torch.empty_like(x)
        ops.rms_norm(y, x, w, 1e-5)
        assert (to_bits(y) == g[tag + "_kernel"]).mean() > 0.999, tag
        np.testing.assert_allclose(f32(to_bits(y)), f32(g[tag + "_native"]), rtol=1.6e-2, atol=1e-3)
        x2, r2 = x.clone(), r.clone()
        ops.fused_add_rms_norm(x2, r2, w, 1e-5)
        assert np.array_equal(to_bits(r2), g[tag + "_fused_kernel_res"]), tag
        assert (to_bits(x2) == g[tag + "_fused_kernel"]).mean() > 0.999, tag


def test_rmsnorm_edge_cases():
    x = torch.randn(5, 64, device=DEV).to(torch.float16)
    y = torch.empty_like(x)
    ops.rms_norm(y, x, None, 1e-6)
    ref = orc.rms_norm(to_bits(x), None, 1e-6, orc.FP16)
    assert (to_bits(y) == ref).mean() > 0.99
    # strided input rows (qkv slice), zero tokens
    big = torch.randn(4, 256, device=DEV).to(torch.bfloat16)
    xs = big[:, 64:192]
    ys = torch.empty((4, 128), dtype=torch.bfloat16, device=DEV)
    ops.rms_norm(ys, xs, None, 1e-5)
    ref = orc.rms_norm(to_bits(xs.contiguous()), None, 1e-5)
    assert (to_bits(ys) == ref).mean() > 0.99
    ops.rms_norm(torch.empty((0, 64), dtype=torch.bfloat16, device=DEV),
                 torch.empty((0, 64), dtype=torch.bfloat16, device=DEV), None, 1e-5)


def test_rope_golden(golden):
    g = golden("rope")
    for tag in g["names"]:
        tag = str(tag)
        head, rot, neox, nh, nkv = [int(v) for v in g[tag + "_meta"]]
        q, k = to_dev(g[tag + "_q"]), to_dev(g[tag + "_k"])
        pos = torch.from_numpy(g[tag + "_pos"]).to(DEV)
        ops.rotary_embedding(pos, q, k, head, to_dev(g[tag + "_cache"]), bool(neox))
        qo, ko = orc.rotary_embedding(g[tag + "_pos"], g[tag + "_q"], g[tag + "_k"], g[tag + "_cache"], head, neox)
        assert np.array_equal(to_bits(q), qo), tag  # same fp32 expression, one rounding
        assert np.array_equal(to_bits(k), ko), tag
        np.testing.assert_allclose(f32(to_bits(q)), f32(g[tag + "_qo"]), rtol=1.6e-2, atol=2e-2)


def test_silu_and_mul_pairs_equals_silu_and_mul_of_the_separated_halves():
    x = (torch.randn(5, 2 * 1024, device=DEV) * 3).to(torch.bfloat16)
    ref = torch.empty((5, 1024), dtype=torch.bfloat16, device=DEV)
    ops.silu_and_mul(ref, ops.unpair_halves(x).contiguous())
    out = torch.empty_like(ref)
    ops.silu_and_mul_pairs(out, x)
    assert torch.equal(out, ref)
    assert np.array_equal(to_bits(out), orc.silu_and_mul(to_bits(ops.unpair_halves(x).contiguous())))


def test_silu_and_mul():
    x = torch.randn(7, 2 * 1000, device=DEV).to(torch.bfloat16) * 3
    out = torch.empty((7, 1000), dtype=torch.bfloat16, device=DEV)
    ops.silu_and_mul(out, x)
    ref = orc.silu_and_mul(to_bits(x))
    np.testing.assert_allclose(f32(to_bits(out)), f32(ref), rtol=1.6e-2, atol=1e-3)
    assert (to_bits(out) == ref).mean() > 0.98


# ---------------------------------------------------------------------------
# KV cache + paged attention
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("fp8,per_head", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("hnd", [False, True])
@pytest.mark.parametrize("bs,head", [(16, 128), (32, 64), (16, 120)])
def test_reshape_and_cache_flash(fp8, per_head, hnd, bs, head):
    """ops.reshape_and_cache_flash (tests/kernels/attention/test_cache.py:184-330 of the reference: NHD and HND
    storage, fp8 with one scale or one per head, padded slots) bit for bit against the oracle's scatter; the key
    tensor carries more rows than slot_mapping, as under graph padding."""
    rng = np.random.default_rng(7 + bs + head)
    T, heads, nblk = 11, 3, 5
    key = orc.f32_to_h16(rng.standard_normal((T + 2, heads, head)).astype(np.float32))
    val = orc.f32_to_h16(rng.standard_normal((T + 2, heads, head)).astype(np.float32))
    slots = rng.permutation(nblk * bs)[:T].astype(np.int64)
    slots[4] = -1
    cdt, tdt = (np.uint8, torch.uint8) if fp8 else (np.uint16, torch.bfloat16)
    shape = (nblk, heads, bs, head) if hnd else (nblk, bs, heads, head)
    kc_ref, vc_ref = np.zeros(shape, dtype=cdt), np.zeros(shape, dtype=cdt)
    ks = rng.uniform(0.02, 0.08, heads if per_head else 1).astype(np.float32) if fp8 else np.ones(1, np.float32)
    vs = rng.uniform(0.02, 0.08, heads if per_head else 1).astype(np.float32) if fp8 else np.ones(1, np.float32)
    orc.reshape_and_cache_flash(key[:T], val[:T], kc_ref, vc_ref, slots, hnd=hnd, fp8=fp8, k_scale=ks, v_scale=vs)
    kc, vc = torch.zeros(shape, dtype=tdt, device=DEV), torch.zeros(shape, dtype=tdt, device=DEV)
    kview, vview = (kc.permute(0, 2, 1, 3), vc.permute(0, 2, 1, 3)) if hnd else (kc, vc)   # [blocks, bs, heads, head]
    ops.reshape_and_cache_flash(to_dev(key), to_dev(val), kview, vview, torch.from_numpy(slots).to(DEV),
                                "fp8" if fp8 else "auto", torch.from_numpy(ks).to(DEV), torch.from_numpy(vs).to(DEV))
    got_k = kc.cpu().numpy() if fp8 else to_bits(kc)
    got_v = vc.cpu().numpy() if fp8 else to_bits(vc)
    assert np.array_equal(got_k, kc_ref) and np.array_equal(got_v, vc_ref)
    assert kc_ref.any() and not np.array_equal(kc_ref, vc_ref)
    with pytest.raises(RuntimeError, match="shape \\[1\\] or \\[num_heads\\]"):
        ops.reshape_and_cache_flash(to_dev(key), to_dev(val), kview, vview, torch.from_numpy(slots).to(DEV), "fp8",
                                    torch.ones(2, device=DEV), torch.ones(2, device=DEV))


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("bs,head", [(16, 128), (32, 64)])
def test_reshape_and_cache(fp8, bs, head):
    rng = np.random.default_rng(5)
    T, kvh, nblk = 9, 2, 6
    x = 16 if fp8 else 8
    key = orc.f32_to_h16(rng.standard_normal((T, kvh, head)).astype(np.float32))
    val = orc.f32_to_h16(rng.standard_normal((T, kvh, head)).astype(np.float32))
    slots = rng.permutation(nblk * bs)[:T].astype(np.int64)
    slots[3] = -1
    cdt = np.uint8 if fp8 else np.uint16
    kc_ref = np.zeros((nblk, kvh, head // x, bs, x), dtype=cdt)
    vc_ref = np.zeros((nblk, kvh, head, bs), dtype=cdt)
    ks, vs = (0.05, 0.02) if fp8 else (1.0, 1.0)
    orc.reshape_and_cache(key, val, kc_ref, vc_ref, slots, fp8=fp8, k_scale=ks, v_scale=vs)
    tdt = torch.uint8 if fp8 else torch.bfloat16
    kc = torch.zeros((nblk, kvh, head // x, bs, x), dtype=tdt, device=DEV)
    vc = torch.zeros((nblk, kvh, head, bs), dtype=tdt, device=DEV)
    kst = torch.tensor(ks, dtype=torch.float32, device=DEV)
    vst = torch.tensor(vs, dtype=torch.float32, device=DEV)
    ops.reshape_and_cache(to_dev(key), to_dev(val), kc, vc, torch.from_numpy(slots).to(DEV),
                          "fp8" if fp8 else "auto", kst, vst)
    got_k = kc.cpu().numpy() if fp8 else to_bits(kc)
    got_v = vc.cpu().numpy() if fp8 else to_bits(vc)
    assert np.array_equal(got_k, kc_ref)
    assert np.array_equal(got_v, vc_ref)
    if fp8:  # convert_fp8 round trip as tests/kernels/attention/test_cache.py does
        back = torch.empty(kc.shape, dtype=torch.bfloat16, device=DEV)
        ops.convert_fp8(back, kc, ks)
        expect = orc.f32_to_h16(orc.e4m3_to_f32(kc_ref) * np.float32(ks))
        assert np.array_equal(to_bits(back), expect)


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("head,rot,neox,bs", [(128, 128, True, 16), (64, 64, False, 32), (128, 32, True, 16)])
def test_rope_and_cache_matches_the_oracle_sequence(fp8, head, rot, neox, bs):
    """One launch = rotary_embedding, then reshape_and_cache on the rotated key: checked bit for bit
    against the oracle's two steps (NeoX and GPT-J pairing, partial rotary, fp8 cache, a skipped slot)."""
    rng = np.random.default_rng(31)
    T, nh, kvh, nblk, max_pos = 11, 4, 2, 6, 64
    x = 16 if fp8 else 8
    q = orc.f32_to_h16(rng.standard_normal((T, nh * head)).astype(np.float32))
    k = orc.f32_to_h16(rng.standard_normal((T, kvh * head)).astype(np.float32))
    v = orc.f32_to_h16(rng.standard_normal((T, kvh * head)).astype(np.float32))
    pos = rng.integers(0, max_pos, size=T).astype(np.int64)
    inv = 1.0 / (10000.0 ** (np.arange(0, rot, 2, dtype=np.float32) / rot))
    fr = np.arange(max_pos, dtype=np.float32)[:, None] * inv[None, :]
    cache = orc.f32_to_h16(np.concatenate([np.cos(fr), np.sin(fr)], axis=1).astype(np.float32))
    slots = rng.permutation(nblk * bs)[:T].astype(np.int64)
    slots[4] = -1
    ks, vs = (0.05, 0.02) if fp8 else (1.0, 1.0)
    qo, ko = orc.rotary_embedding(pos, q, k, cache, head, int(neox))
    cdt = np.uint8 if fp8 else np.uint16
    kc_ref = np.zeros((nblk, kvh, head // x, bs, x), dtype=cdt)
    vc_ref = np.zeros((nblk, kvh, head, bs), dtype=cdt)
    orc.reshape_and_cache(ko.reshape(T, kvh, head), v.reshape(T, kvh, head), kc_ref, vc_ref, slots, fp8=fp8,
                          k_scale=ks, v_scale=vs)
    tdt = torch.uint8 if fp8 else torch.bfloat16
    kc = torch.zeros((nblk, kvh, head // x, bs, x), dtype=tdt, device=DEV)
    vc = torch.zeros((nblk, kvh, head, bs), dtype=tdt, device=DEV)
    # q, k, v as slices of one fused projection output, as the model passes them
    qkv = to_dev(np.concatenate([q, k, v], axis=1))
    qs, ksz = nh * head, kvh * head
    qd, kd, vd = qkv[:, :qs], qkv[:, qs:qs + ksz], qkv[:, qs + ksz:]
    ops.rope_and_cache(torch.from_numpy(pos).to(DEV), qd, kd, vd, head, to_dev(cache), neox, kc, vc,
                       torch.from_numpy(slots).to(DEV), "fp8" if fp8 else "auto",
                       torch.tensor(ks, dtype=torch.float32, device=DEV), torch.tensor(vs, dtype=torch.float32, device=DEV))
    assert np.array_equal(to_bits(qd), qo) and np.array_equal(to_bits(kd), ko)
    assert np.array_equal(to_bits(vd), v)
    assert np.array_equal(kc.cpu().numpy() if fp8 else to_bits(kc), kc_ref)
    assert np.array_equal(vc.cpu().numpy() if fp8 else to_bits(vc), vc_ref)


def _run_paged(g, tag, varlen_rows=None):
    H, KVH, D, bs, fp8 = [int(v) for v in g[tag + "_meta"]]
    scale = float(g[tag + "_scale"][0])
    q = to_dev(g[tag + "_q"])
    if fp8:
        kc = torch.from_numpy(g[tag + "_kc"]).to(DEV)
        vc = torch.from_numpy(g[tag + "_vc"]).to(DEV)
        ks, vs = [float(v) for v in g[tag + "_kv_scale"]]
    else:
        kc, vc = to_dev(g[tag + "_kc"]), to_dev(g[tag + "_vc"])
        ks = vs = 1.0
    bt = torch.from_numpy(g[tag + "_bt"]).to(DEV)
    sl = torch.from_numpy(g[tag + "_sl"]).to(DEV)
    alibi = torch.from_numpy(g[tag + "_alibi"]).to(DEV) if (tag + "_alibi") in g.files else None
    S = q.shape[0]
    max_len = int(g[tag + "_sl"].max())
    P = (max_len + 255) // 256
    out = torch.empty_like(q)
    tmp = torch.empty((S, H, P, D), dtype=q.dtype, device=DEV)
    es = torch.empty((S, H, P), dtype=torch.float32, device=DEV)
    ml = torch.empty_like(es)
    kst = torch.tensor(ks, dtype=torch.float32, device=DEV)
    vst = torch.tensor(vs, dtype=torch.float32, device=DEV)
    ops.paged_attention_rocm(out, es, ml, tmp, q, kc, vc, KVH, scale, bt, sl, None, bs, max_len, alibi,
                             "fp8" if fp8 else "auto", kst, vst)
    return out, fp8


def test_paged_attention_golden(golden):
    """Fixtures follow tests/kernels/attention/test_attention.py:124-291; tolerance of :288-291."""
    g = golden("paged_attn")
    for tag in g["names"]:
        tag = str(tag)
        out, fp8 = _run_paged(g, tag)
        atol = 1e-2 if fp8 else 1e-3
        np.testing.assert_allclose(f32(to_bits(out)), f32(g[tag + "_ref"]), rtol=1e-5, atol=atol, err_msg=tag)


@pytest.mark.parametrize("H,KVH,D,bs,fp8", [(32, 8, 128, 16, False), (64, 8, 128, 16, True), (40, 40, 64, 32, False),
                                            (16, 1, 128, 32, False)])
def test_paged_attention_vs_oracle(H, KVH, D, bs, fp8):
    rng = np.random.default_rng(H + D)
    lens = [1, 15, 16, 255, 256, 257, 1000]
    S = len(lens)
    nblk = 80
    x = 16 if fp8 else 8
    scale = 1.0 / D ** 0.5
    q = orc.f32_to_h16(rng.uniform(-1, 1, (S, H, D)).astype(np.float32))
    kf = rng.uniform(-1, 1, (nblk, KVH, D // x, bs, x)).astype(np.float32)
    vf = rng.uniform(-1, 1, (nblk, KVH, D, bs)).astype(np.float32)
    ks, vs = (1 / 64, 1 / 32) if fp8 else (1.0, 1.0)
    if fp8:
        kc, vc = orc.f32_to_e4m3(kf / ks), orc.f32_to_e4m3(vf / vs)
        kcd, vcd = torch.from_numpy(kc).to(DEV), torch.from_numpy(vc).to(DEV)
    else:
        kc, vc = orc.f32_to_h16(kf), orc.f32_to_h16(vf)
        kcd, vcd = to_dev(kc), to_dev(vc)
    maxb = (max(lens) + bs - 1) // bs
    bt = rng.integers(0, nblk, size=(S, maxb)).astype(np.int32)
    sl = np.array(lens, dtype=np.int32)
    ref = orc.paged_attention(q, kc, vc, bt, sl, scale, fp8=fp8, k_scale=ks, v_scale=vs)
    P = (max(lens) + 255) // 256
    qd = to_dev(q)
    out = torch.empty_like(qd)
    tmp = torch.empty((S, H, P, D), dtype=qd.dtype, device=DEV)
    es = torch.empty((S, H, P), dtype=torch.float32, device=DEV)
    ml = torch.empty_like(es)
    kst = torch.tensor(ks, dtype=torch.float32, device=DEV)
    vst = torch.tensor(vs, dtype=torch.float32, device=DEV)
    ops.paged_attention_rocm(out, es, ml, tmp, qd, kcd, vcd, KVH, scale, torch.from_numpy(bt).to(DEV),
                             torch.from_numpy(sl).to(DEV), None, bs, max(lens), None, "fp8" if fp8 else "auto", kst, vst)
    # tolerance of the reference's test of its kernel (tests/kernels/attention/test_attention.py:288-291)
    np.testing.assert_allclose(f32(to_bits(out)), f32(ref), rtol=1e-5, atol=1e-3 if not fp8 else 1e-2)


@pytest.mark.parametrize("fp8", [False, True])
def test_decode_step_attention_serving_batch_vs_oracle(fp8):
    """32 sequences of about 1024 tokens, Llama-3-8B heads (the (32, 1024) shape of the benchmark: 1280 workgroups,
    several dispatch rounds, every sequence's partitions merged in the kernel): the fused decode step against the
    oracle's attention over the cache it leaves behind, at the tolerance of the reference's kernel test."""
    rng = np.random.default_rng(32)
    H, KVH, D, bs, S = 32, 8, 128, 16, 32
    lens = [int(v) for v in rng.integers(990, 1060, size=S)]
    x = 16 if fp8 else 8
    per = (max(lens) + bs - 1) // bs
    nblk = S * per
    scale = 1.0 / D ** 0.5
    ks, vs = (1 / 64, 1 / 32) if fp8 else (1.0, 1.0)
    kst = torch.tensor(ks, dtype=torch.float32, device=DEV)
    vst = torch.tensor(vs, dtype=torch.float32, device=DEV)
    kf = rng.uniform(-1, 1, (nblk, KVH, D // x, bs, x)).astype(np.float32)
    vf = rng.uniform(-1, 1, (nblk, KVH, D, bs)).astype(np.float32)
    if fp8:
        kcd = torch.from_numpy(orc.f32_to_e4m3(kf / ks)).to(DEV)
        vcd = torch.from_numpy(orc.f32_to_e4m3(vf / vs)).to(DEV)
    else:
        kcd, vcd = to_dev(orc.f32_to_h16(kf)), to_dev(orc.f32_to_h16(vf))
    bt = rng.permutation(nblk).reshape(S, per).astype(np.int32)
    sl = np.array(lens, dtype=np.int32)
    qkv = to_dev(orc.f32_to_h16(rng.uniform(-1, 1, (S, (H + 2 * KVH) * D)).astype(np.float32)))
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float) / D))
    fr = torch.einsum("i,j->ij", torch.arange(2048, dtype=torch.float), inv)
    cache = torch.cat((fr.cos(), fr.sin()), -1).to(torch.bfloat16).to(DEV)
    pos = torch.from_numpy(sl.astype(np.int64) - 1).to(DEV)
    P = (max(lens) + 255) // 256
    out = torch.zeros((S, H, D), dtype=torch.bfloat16, device=DEV)
    es = torch.empty((S, H, P), dtype=torch.float32, device=DEV)
    ml = torch.empty_like(es)
    tmp = torch.empty((S, H, P, D), dtype=torch.bfloat16, device=DEV)
    ops.decode_attention_fused(out, es, ml, tmp, qkv, pos, cache, kcd, vcd, H, KVH, D, scale, torch.from_numpy(bt).to(DEV),
                               torch.from_numpy(sl).to(DEV), bs, max(lens), "fp8" if fp8 else "auto", kst, vst)
    # the oracle attends over the cache the step has left behind, with the rotated query
    q = qkv[:, :H * D].clone()
    k = qkv[:, H * D:(H + KVH) * D].clone()
    ops.rotary_embedding(pos, q, k, D, cache, True)
    kc_after = kcd.cpu().numpy() if fp8 else to_bits(kcd)
    vc_after = vcd.cpu().numpy() if fp8 else to_bits(vcd)
    ref = orc.paged_attention(to_bits(q.view(S, H, D)), kc_after, vc_after, bt, sl, scale, fp8=fp8, k_scale=ks, v_scale=vs)
    np.testing.assert_allclose(f32(to_bits(out)), f32(ref), rtol=1e-5, atol=1e-3 if not fp8 else 1e-2)


@pytest.mark.parametrize("fp8", [False, True])
def test_paged_attention_inline_merge_equals_merge_kernel(fp8, monkeypatch):
    """Partitions merged inside the attention kernel (last workgroup to arrive) give the output of
    the stand-alone merge kernel (to one rounding step), call after call on the same scratch buffers with
    changing queries and lengths (a stale read of another workgroup's partial would show)."""
    rng = np.random.default_rng(21)
    H, KVH, D, bs, S, nblk = 32, 8, 128, 16, 5, 160
    x = 16 if fp8 else 8
    scale = D ** -0.5
    kf = rng.uniform(-1, 1, (nblk, KVH, D // x, bs, x)).astype(np.float32)
    vf = rng.uniform(-1, 1, (nblk, KVH, D, bs)).astype(np.float32)
    ks, vs = (1 / 64, 1 / 32) if fp8 else (1.0, 1.0)
    if fp8:
        kc, vc = orc.f32_to_e4m3(kf / ks), orc.f32_to_e4m3(vf / vs)
        kcd, vcd = torch.from_numpy(kc).to(DEV), torch.from_numpy(vc).to(DEV)
    else:
        kc, vc = orc.f32_to_h16(kf), orc.f32_to_h16(vf)
        kcd, vcd = to_dev(kc), to_dev(vc)
    max_len = 2000
    maxb = (max_len + bs - 1) // bs
    P = (max_len + 255) // 256
    kst = torch.tensor(ks, dtype=torch.float32, device=DEV)
    vst = torch.tensor(vs, dtype=torch.float32, device=DEV)
    tmp = torch.empty((S, H, P, D), dtype=torch.bfloat16, device=DEV)
    es = torch.empty((S, H, P), dtype=torch.float32, device=DEV)
    ml = torch.empty_like(es)
    for it in range(4):
        lens = rng.integers(1, max_len + 1, size=S)
        lens[it % S] = [1, 256, 257, 2000][it]          # one partition / exact boundary / just past it / maximum
        q = orc.f32_to_h16(rng.uniform(-1, 1, (S, H, D)).astype(np.float32))
        bt = rng.integers(0, nblk, size=(S, maxb)).astype(np.int32)
        sl = np.array(lens, dtype=np.int32)
        qd = to_dev(q)
        outs = []
        for env in (None, "1"):
            _lib.set_option("MI355Q_PA_REDUCE_KERNEL", env)
            try:
                out = torch.full_like(qd, float("nan"))
                ops.paged_attention_rocm(out, es, ml, tmp, qd, kcd, vcd, KVH, scale, torch.from_numpy(bt).to(DEV),
                                         torch.from_numpy(sl).to(DEV), None, bs, max_len, None, "fp8" if fp8 else "auto", kst, vst)
            finally:
                _lib.set_option("MI355Q_PA_REDUCE_KERNEL", None)
            outs.append(out)
        # same formula in two kernels: the compiler may contract the multiply-adds differently, so a
        # handful of outputs can differ by one rounding step; anything more would be a stale read
        a, b = outs[0].float(), outs[1].float()
        assert not torch.isnan(a).any() and not torch.isnan(b).any()
        assert (a != b).float().mean() < 1e-3, f"iteration {it}"
        assert torch.allclose(a, b, rtol=2.0 ** -7, atol=1e-6), f"iteration {it}"
        ref = orc.paged_attention(q, kc, vc, bt, sl, scale, fp8=fp8, k_scale=ks, v_scale=vs)
        np.testing.assert_allclose(f32(to_bits(outs[0])), f32(ref), rtol=1e-5, atol=1e-3 if not fp8 else 1e-2)


def test_paged_attention_skips_prompt_rows_and_ignores_stale_cache():
    """Rows whose query length != 1 are left untouched (csrc/rocm/attention.cu:351-354), and
    cache slots past seq_len may hold NaN without leaking into the output."""
    rng = np.random.default_rng(9)
    H, KVH, D, bs = 8, 2, 128, 16
    S, nblk = 3, 8
    q = to_dev(orc.f32_to_h16(rng.uniform(-1, 1, (S + 2, H, D)).astype(np.float32)))
    kc = torch.full((nblk, KVH, D // 8, bs, 8), float("nan"), dtype=torch.bfloat16, device=DEV)
    vc = torch.full((nblk, KVH, D, bs), float("nan"), dtype=torch.bfloat16, device=DEV)
    lens = np.array([5, 20, 3], dtype=np.int32)
    bt = np.array([[0, 1], [2, 3], [4, 5]], dtype=np.int32)
    kf = rng.uniform(-1, 1, (nblk, KVH, D // 8, bs, 8)).astype(np.float32)
    vf = rng.uniform(-1, 1, (nblk, KVH, D, bs)).astype(np.float32)
    kcn, vcn = orc.f32_to_h16(kf), orc.f32_to_h16(vf)
    for s in range(S):
        for j in range(lens[s]):
            b, o = bt[s, j // bs], j % bs
            kc[b, :, :, o, :] = to_dev(kcn[b, :, :, o, :])
            vc[b, :, :, o] = to_dev(vcn[b, :, :, o])
    qsl = torch.tensor([0, 1, 4, 5], dtype=torch.int32, device=DEV)  # row 1 is a 3-token prompt row
    out = torch.full((S + 2, H, D), 7.0, dtype=torch.bfloat16, device=DEV)
    one = torch.ones((), dtype=torch.float32, device=DEV)
    ops.paged_attention_rocm(out, None, None, None, q, kc, vc, KVH, 0.1, torch.from_numpy(bt).to(DEV),
                             torch.from_numpy(lens).to(DEV), qsl, bs, 20, None, "auto", one, one)
    o = out.float().cpu()
    assert torch.all(o[1:4] == 7.0)
    assert torch.isfinite(o[0]).all() and torch.isfinite(o[4]).all()
    qrows = to_bits(q)[[0, 4]]
    ref = orc.paged_attention(qrows, kcn, vcn, bt[[0, 2]], lens[[0, 2]], 0.1)
    np.testing.assert_allclose(f32(to_bits(out[[0, 4]])), f32(ref), rtol=1e-5, atol=1e-3)


@pytest.fixture
def prefill_kernel(request):
    """Forces the partitioned or the running-softmax prompt kernel, the latter in its 64- ("running") or 32-token
    ("running32") form (the library picks by query length and batch)."""
    v = "100000000" if request.param == "partitioned" else "32"
    _lib.set_option("MI355Q_PREFILL_FLASH_MIN", v)
    _lib.set_option("MI355Q_PREFILL_FLASH_WGS", v)
    if request.param != "partitioned":
        _lib.set_option("MI355Q_PREFILL_FLASH_CTL", "1" if request.param == "running32" else "2")
    yield "running" if request.param == "running32" else request.param
    _lib.set_option("MI355Q_PREFILL_FLASH_MIN", None)
    _lib.set_option("MI355Q_PREFILL_FLASH_WGS", None)
    _lib.set_option("MI355Q_PREFILL_FLASH_CTL", None)


@pytest.mark.parametrize("prefill_kernel", ["partitioned", "running", "running32"], indirect=True)
@pytest.mark.parametrize("H,KVH,D,bs,fp8", [(8, 2, 128, 16, False), (4, 4, 64, 32, False), (16, 2, 128, 16, True),
                                           (16, 2, 128, 16, False), (4, 2, 128, 64, False), (6, 2, 64, 16, False),
                                           (32, 8, 128, 16, False),
                                           (8, 2, 128, 48, False)])   # not a power of two: the partitioned kernel serves it
def test_paged_prefill_attention_vs_oracle(H, KVH, D, bs, fp8, prefill_kernel):
    """Prompt rows: write K/V with reshape_and_cache, attend through the cache, compare with
    dense causal attention of the oracle (and a decode row mixed in)."""
    rng = np.random.default_rng(D + H)
    qlens = [37, 1, 300]
    ctx0 = [0, 40, 20]  # tokens already in the cache before this step
    S = len(qlens)
    x = 16 if fp8 else 8
    nblk = 64
    scale = 1.0 / D ** 0.5
    ks, vs = (1 / 32, 1 / 32) if fp8 else (1.0, 1.0)
    tdt = torch.uint8 if fp8 else torch.bfloat16
    kc = torch.zeros((nblk, KVH, D // x, bs, x), dtype=tdt, device=DEV)
    vc = torch.zeros((nblk, KVH, D, bs), dtype=tdt, device=DEV)
    kst = torch.tensor(ks, dtype=torch.float32, device=DEV)
    vst = torch.tensor(vs, dtype=torch.float32, device=DEV)
    seq_lens = [c + q for c, q in zip(ctx0, qlens)]
    maxb = (max(seq_lens) + bs - 1) // bs
    perm = rng.permutation(nblk)
    bt = perm[: S * maxb].reshape(S, maxb).astype(np.int32)
    allk, allv, allq, slots_new = [], [], [], []
    refs = []
    for s in range(S):
        L = seq_lens[s]
        k = orc.f32_to_h16(rng.uniform(-1, 1, (L, KVH, D)).astype(np.float32))
        v = orc.f32_to_h16(rng.uniform(-1, 1, (L, KVH, D)).astype(np.float32))
        q = orc.f32_to_h16(rng.uniform(-1, 1, (L, H, D)).astype(np.float32))
        slots = np.array([bt[s, j // bs] * bs + j % bs for j in range(L)], dtype=np.int64)
        ops.reshape_and_cache(to_dev(k), to_dev(v), kc, vc, torch.from_numpy(slots).to(DEV),
                              "fp8" if fp8 else "auto", kst, vst)
        if fp8:  # the oracle attends the quantised values the cache holds
            k = orc.f32_to_h16(orc.e4m3_to_f32(orc.f32_to_e4m3(orc.h16_to_f32(k) / np.float32(ks))) * np.float32(ks))
            v = orc.f32_to_h16(orc.e4m3_to_f32(orc.f32_to_e4m3(orc.h16_to_f32(v) / np.float32(vs))) * np.float32(vs))
        full = orc.prefill_attention(q, k, v, scale)
        refs.append(full[ctx0[s]:])
        allq.append(q[ctx0[s]:])
    qcat = np.concatenate(allq, 0)
    ref = np.concatenate(refs, 0)
    T = qcat.shape[0]
    qsl = torch.tensor(np.concatenate([[0], np.cumsum(qlens)]).astype(np.int32), device=DEV)
    P = (max(seq_lens) + 255) // 256
    qd = to_dev(qcat)
    out = torch.empty_like(qd)
    tmp = torch.empty((T, H, P, D), dtype=qd.dtype, device=DEV)
    es = torch.empty((T, H, P), dtype=torch.float32, device=DEV)
    ml = torch.empty_like(es)
    ops.paged_prefill_attention(out, es, ml, tmp, qd, kc, vc, KVH, scale, torch.from_numpy(bt).to(DEV),
                                torch.tensor(seq_lens, dtype=torch.int32, device=DEV), qsl, max(qlens), bs,
                                max(seq_lens), None, "fp8" if fp8 else "auto", kst, vst)
    got, want = f32(to_bits(out)), f32(ref)
    if fp8:
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-2)
    elif prefill_kernel == "partitioned":
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=3e-3)
    else:
        # The oracle rounds NORMALISED probabilities to bf16 (as the reference's test helper does, and the partitioned
        # kernel); the running-softmax kernel rounds them relative to the running maximum, as production prompt
        # kernels do.  The reference holds its own prompt kernel to atol 1e-4 on outputs of magnitude <= 1e-3
        # (tests/kernels/attention/test_prefix_prefill.py:159-163, 322-323), a tenth of the value range; here, with
        # values in (-1, 1): every output within ONE bf16 step at that range (2^-8), mean deviation under a tenth of it.
        np.testing.assert_allclose(got, want, rtol=0, atol=4e-3)
        assert np.abs(got - want).mean() < 4e-4


@pytest.mark.parametrize("narrow", [False, True])
@pytest.mark.parametrize("fp8", [False, True])
def test_paged_prefill_attention_long_prompts(fp8, narrow):
    """Llama-3-8B heads, a 1500-token prompt and a 700-token chunk behind 300 cached tokens: 24 context tiles per
    query block, ragged last tiles, in the kernel the library picks at these lengths (running softmax, 64 query
    tokens per workgroup) and in its 32-token form.  Same bar as test_paged_prefill_attention_vs_oracle."""
    _lib.set_option("MI355Q_PREFILL_FLASH_CTL", "1" if narrow else None)
    try:
        _long_prompts_case(fp8)
    finally:
        _lib.set_option("MI355Q_PREFILL_FLASH_CTL", None)


def _long_prompts_case(fp8):
    H, KVH, D, bs = 32, 8, 128, 16
    rng = np.random.default_rng(77)
    qlens, ctx0 = [1500, 700], [0, 300]
    ks, vs = (0.02, 0.015) if fp8 else (1.0, 1.0)
    tdt = torch.uint8 if fp8 else torch.bfloat16
    x = 16 if fp8 else 8
    seq_lens = [c + q for c, q in zip(ctx0, qlens)]
    maxb = (max(seq_lens) + bs - 1) // bs
    nblk = 2 * maxb + 3
    kc = torch.zeros((nblk, KVH, D // x, bs, x), dtype=tdt, device=DEV)
    vc = torch.zeros((nblk, KVH, D, bs), dtype=tdt, device=DEV)
    kst = torch.tensor(ks, dtype=torch.float32, device=DEV)
    vst = torch.tensor(vs, dtype=torch.float32, device=DEV)
    bt = rng.permutation(nblk)[: 2 * maxb].reshape(2, maxb).astype(np.int32)
    allq, refs = [], []
    for s_ in range(2):
        L = seq_lens[s_]
        k = orc.f32_to_h16(rng.uniform(-1, 1, (L, KVH, D)).astype(np.float32))
        v = orc.f32_to_h16(rng.uniform(-1, 1, (L, KVH, D)).astype(np.float32))
        q = orc.f32_to_h16(rng.uniform(-1, 1, (L, H, D)).astype(np.float32))
        slots = np.array([bt[s_, j // bs] * bs + j % bs for j in range(L)], dtype=np.int64)
        ops.reshape_and_cache(to_dev(k), to_dev(v), kc, vc, torch.from_numpy(slots).to(DEV), "fp8" if fp8 else "auto", kst, vst)
        if fp8:
            k = orc.f32_to_h16(orc.e4m3_to_f32(orc.f32_to_e4m3(orc.h16_to_f32(k) / np.float32(ks))) * np.float32(ks))
            v = orc.f32_to_h16(orc.e4m3_to_f32(orc.f32_to_e4m3(orc.h16_to_f32(v) / np.float32(vs))) * np.float32(vs))
        refs.append(orc.prefill_attention(q, k, v, D ** -0.5)[ctx0[s_]:])
        allq.append(q[ctx0[s_]:])
    qd = to_dev(np.concatenate(allq, 0))
    ref = np.concatenate(refs, 0)
    T = qd.shape[0]
    P = (max(seq_lens) + 255) // 256
    out = torch.empty_like(qd)
    tmp = torch.empty((T, H, P, D), dtype=qd.dtype, device=DEV)
    es = torch.empty((T, H, P), dtype=torch.float32, device=DEV)
    ml = torch.empty_like(es)
    qsl = torch.tensor(np.concatenate([[0], np.cumsum(qlens)]).astype(np.int32), device=DEV)
    ops.paged_prefill_attention(out, es, ml, tmp, qd, kc, vc, KVH, D ** -0.5, torch.from_numpy(bt).to(DEV),
                                torch.tensor(seq_lens, dtype=torch.int32, device=DEV), qsl, max(qlens), bs,
                                max(seq_lens), None, "fp8" if fp8 else "auto", kst, vst)
    got, want = f32(to_bits(out)), f32(ref)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-2 if fp8 else 4e-3)
    assert np.abs(got - want).mean() < (1e-3 if fp8 else 4e-4)


# ---------------------------------------------------------------------------
# sampler inner loop
# ---------------------------------------------------------------------------
def test_sampler_golden(golden):
    g = golden("sampler")
    logits = to_dev(g["logits"])
    assert ops.argmax(logits).cpu().numpy().tolist() == g["argmax"].tolist()
    assert ops.argmax(logits.float()).cpu().numpy().tolist() == g["argmax"].tolist()
    lg = logits.float().contiguous()
    ops.apply_repetition_penalties(lg, torch.from_numpy(g["prompt_mask"]).to(DEV),
                                   torch.from_numpy(g["output_mask"]).to(DEV), torch.from_numpy(g["penalties"]).to(DEV))
    ref = orc.apply_repetition_penalties(f32(g["logits"]), g["prompt_mask"], g["output_mask"], g["penalties"])
    assert np.array_equal(lg.cpu().numpy(), ref)
    np.testing.assert_allclose(lg.cpu().numpy(), g["penalized"], rtol=2e-7, atol=0)


@pytest.mark.parametrize("V,narrow", [(128256, False), (32000, False), (2048, True)])
def test_argmax_decode_step_advances_and_gathers(V, narrow):
    """Arg-max with the decode step's bookkeeping in the same launch: ids as ops.argmax gives them, positions and
    sequence lengths advanced by one, the sampled tokens' embedding rows gathered; wide (segmented) and narrow
    arg-max paths, under a captured graph replayed twice."""
    torch.manual_seed(V)
    B, H = 5, 256
    logits = torch.randn((B, V), device=DEV).to(torch.bfloat16)
    logits[1, 7] = logits[1, V - 3] = 9.0            # a tie: the first index wins
    embed = torch.randn((V, H), device=DEV).to(torch.bfloat16)
    ids = torch.zeros(B, dtype=torch.int64, device=DEV)
    pos = torch.arange(10, 10 + B, dtype=torch.int64, device=DEV)
    sl = torch.arange(11, 11 + B, dtype=torch.int32, device=DEV)
    hid = torch.zeros((B, H), dtype=torch.bfloat16, device=DEV)
    want = ops.argmax(logits).clone()
    assert int(want[1]) == 7 and torch.equal(want.cpu(), logits.float().argmax(-1).cpu())
    ops.argmax_decode_step(logits, ids, pos, sl, embed, hid)
    assert torch.equal(ids, want) and torch.equal(hid, embed[want])
    assert pos.tolist() == list(range(11, 11 + B)) and sl.tolist() == list(range(12, 12 + B))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.argmax_decode_step(logits, ids, pos, sl, embed, hid)
    logits[2, 5] = 50.0
    g.replay(); g.replay()
    torch.cuda.synchronize()
    assert int(ids[2]) == 5 and torch.equal(hid[2], embed[5])
    assert pos.tolist() == list(range(13, 13 + B)) and sl.tolist() == list(range(14, 14 + B))   # one call and two replays
    ops.argmax_decode_step(logits, ids, None, None)          # nothing but the arg-max
    assert pos.tolist() == list(range(13, 13 + B))
    with pytest.raises(RuntimeError, match="argmax_decode_step"):
        ops.argmax_decode_step(logits, ids, pos.to(torch.int32), sl)


def test_argmax_full_vocab_ties_and_strides():
    V = 128256
    rng = np.random.default_rng(0)
    x = rng.standard_normal((5, V)).astype(np.float32)
    x[0, 128255] = 100.0
    x[1, 0] = 100.0
    x[2, 777] = 50.0
    x[2, 90000] = 50.0
    x[3, :] = -np.inf
    bits = orc.f32_to_h16(x)
    ids = ops.argmax(to_dev(bits))
    assert ids.cpu().numpy().tolist() == orc.argmax_h16(bits).tolist()
    assert ids.cpu().numpy().tolist()[:4] == [128255, 0, 777, 0]
    wide = torch.zeros((5, V + 64), dtype=torch.float32, device=DEV)
    wide[:, :V] = torch.from_numpy(x).to(DEV)
    assert ops.argmax(wide[:, :V]).cpu().numpy().tolist() == orc.argmax_f32(x).tolist()


@pytest.mark.parametrize("M,K,N,split,with_res", [(1, 4096, 1024, 0, True), (1, 4096, 6144, 4, True), (3, 1024, 256, 1, True),
                                                  (4, 512, 128, 2, False), (9, 256, 64, 0, True)])
def test_w4a16_fused_add_rms_norm_prologue(M, K, N, split, with_res):
    """GEMM with fused_add_rms_norm folded in == the oracle's norm followed by its GEMM."""
    rng = np.random.default_rng(M + K)
    G = 128
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    x = orc.f32_to_h16(rng.standard_normal((M, K)).astype(np.float32))
    r = orc.f32_to_h16(rng.standard_normal((M, K)).astype(np.float32))
    wn = orc.f32_to_h16((1 + 0.1 * rng.standard_normal(K)).astype(np.float32))
    s = orc.f32_to_h16((0.05 * rng.random((K // G, N))).astype(np.float32))
    qw = orc.gptq_pack(q)
    if with_res:
        xn, res_ref = orc.fused_add_rms_norm(x, r, wn, 1e-5)
    else:
        xn, res_ref = orc.rms_norm(x, wn, 1e-5), x
    ref = orc.w4a16_gemm(xn, qw, s, None, None, G=G)
    w = ops.w4a16_repack(i32(qw), to_dev(s), None, G)
    xd, rd = to_dev(x), to_dev(r)
    out, res = ops.w4a16_gemm_add_rms_norm(xd, rd if with_res else None, to_dev(wn), 1e-5, w, split_k=split)
    assert np.array_equal(to_bits(res), res_ref)
    assert np.array_equal(to_bits(xd), x) and np.array_equal(to_bits(rd), r)  # inputs untouched
    np.testing.assert_allclose(f32(to_bits(out)), f32(ref), rtol=1e-2, atol=2e-2)
    assert (to_bits(out) == ref).mean() > 0.9


def test_w4a16_split_k_counters_are_left_clean():
    """Back-to-back split-K GEMMs of different shapes share one workspace; results stay equal to
    the single-split result because every launch leaves the arrival counters at zero."""
    rng = np.random.default_rng(21)
    outs = {}
    for rep in range(3):
        for (M, K, N) in [(2, 2048, 256), (1, 4096, 512), (5, 1024, 128)]:
            rs = np.random.default_rng(K + N)
            q = rs.integers(0, 16, size=(K, N), dtype=np.int32)
            a = orc.f32_to_h16(rs.standard_normal((M, K)).astype(np.float32))
            s = orc.f32_to_h16((0.05 * rs.random((K // 128, N))).astype(np.float32))
            w = ops.w4a16_repack(i32(orc.gptq_pack(q)), to_dev(s), None, 128)
            o = ops.w4a16_gemm(to_dev(a), w, split_k=4 + rep)
            o1 = ops.w4a16_gemm(to_dev(a), w, split_k=1)
            np.testing.assert_allclose(o.float().cpu().numpy(), o1.float().cpu().numpy(), rtol=1e-2, atol=1e-2)
            outs.setdefault((M, K, N), []).append(o.cpu())
    ws = ops._WS[torch.device(DEV)]
    assert int(ws[:4096].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize("H,KVH,D,bs,fp8", [(32, 8, 128, 16, False), (8, 2, 128, 32, True), (4, 4, 64, 16, False),
                                          (8, 1, 128, 16, False), (16, 1, 128, 64, True), (6, 2, 64, 32, False)])
def test_decode_attention_fused_equals_unfused(H, KVH, D, bs, fp8):
    """One launch (RoPE + cache write + attention) == rotary_embedding, reshape_and_cache,
    paged_attention_rocm back to back: outputs and cache contents bit for bit."""
    _fused_equals_unfused(H, KVH, D, bs, fp8, [1, 16, 130, 256, 257, 600])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_decode_attention_fused_equals_unfused_random_batches(seed):
    """Batches of random context lengths (one to eight partitions, Llama-3-8B heads): bit for bit again.  The
    kernels of this file repeat the same fp32 expressions and are built without implicit fused multiply-adds, so
    that the compiler cannot contract an expression in one of them and not in the other."""
    lens = [int(v) for v in np.random.default_rng(seed).integers(1, 2000, size=12)] + [512, 900, 1025]
    _fused_equals_unfused(32, 8, 128, 16, seed == 1, lens)


def _fused_equals_unfused(H, KVH, D, bs, fp8, lens):      # lens: seq_len including the new token
    rng = np.random.default_rng(H * D)
    S = len(lens)
    x = 16 if fp8 else 8
    nblk = S * ((max(lens) + bs - 1) // bs)
    maxb = nblk // S
    tdt = torch.uint8 if fp8 else torch.bfloat16
    scale = 1.0 / D ** 0.5
    ks, vs = (1 / 32, 1 / 16) if fp8 else (1.0, 1.0)
    kst = torch.tensor(ks, dtype=torch.float32, device=DEV)
    vst = torch.tensor(vs, dtype=torch.float32, device=DEV)
    bt = torch.from_numpy(rng.permutation(nblk).reshape(S, maxb).astype(np.int32)).to(DEV)
    sl = torch.tensor(lens, dtype=torch.int32, device=DEV)
    pos = (sl - 1).to(torch.int64)
    # a cache already holding the earlier tokens
    if fp8:
        kc0 = torch.from_numpy(orc.f32_to_e4m3(rng.uniform(-8, 8, (nblk, KVH, D // x, bs, x)).astype(np.float32))).to(DEV)
        vc0 = torch.from_numpy(orc.f32_to_e4m3(rng.uniform(-8, 8, (nblk, KVH, D, bs)).astype(np.float32))).to(DEV)
    else:
        kc0 = torch.from_numpy(rng.uniform(-1, 1, (nblk, KVH, D // x, bs, x)).astype(np.float32)).to(torch.bfloat16).to(DEV)
        vc0 = torch.from_numpy(rng.uniform(-1, 1, (nblk, KVH, D, bs)).astype(np.float32)).to(torch.bfloat16).to(DEV)
    qkv = torch.from_numpy(rng.uniform(-1, 1, (S, (H + 2 * KVH) * D)).astype(np.float32)).to(torch.bfloat16).to(DEV)
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float) / D))
    fr = torch.einsum("i,j->ij", torch.arange(2048, dtype=torch.float), inv)
    cache = torch.cat((fr.cos(), fr.sin()), -1).to(torch.bfloat16).to(DEV)
    P = (max(lens) + 255) // 256

    def bufs():
        return (torch.zeros((S, H, D), dtype=torch.bfloat16, device=DEV), torch.empty((S, H, P), dtype=torch.float32, device=DEV),
                torch.empty((S, H, P), dtype=torch.float32, device=DEV), torch.empty((S, H, P, D), dtype=torch.bfloat16, device=DEV))

    # unfused
    kc1, vc1 = kc0.clone(), vc0.clone()
    qkv1 = qkv.clone()
    q1, k1, v1 = qkv1[:, :H * D], qkv1[:, H * D:(H + KVH) * D], qkv1[:, (H + KVH) * D:]
    ops.rotary_embedding(pos, q1, k1, D, cache, True)
    slots = torch.stack([bt[s, (lens[s] - 1) // bs].to(torch.int64) * bs + (lens[s] - 1) % bs for s in range(S)])
    ops.reshape_and_cache(k1.view(S, KVH, D), v1.view(S, KVH, D), kc1, vc1, slots, "fp8" if fp8 else "auto", kst, vst)
    o1, es, ml, tmp = bufs()
    ops.paged_attention_rocm(o1, es, ml, tmp, q1.view(S, H, D), kc1, vc1, KVH, scale, bt, sl, None, bs, max(lens), None,
                             "fp8" if fp8 else "auto", kst, vst)
    # fused
    kc2, vc2 = kc0.clone(), vc0.clone()
    o2, es2, ml2, tmp2 = bufs()
    qkv_before = qkv.clone()
    ops.decode_attention_fused(o2, es2, ml2, tmp2, qkv, pos, cache, kc2, vc2, H, KVH, D, scale, bt, sl, bs, max(lens),
                               "fp8" if fp8 else "auto", kst, vst)
    assert torch.equal(qkv, qkv_before)
    assert torch.equal(kc1.view(torch.uint8), kc2.view(torch.uint8))
    assert torch.equal(vc1.view(torch.uint8), vc2.view(torch.uint8))
    assert torch.equal(o1, o2)


# ---------------------------------------------------------------------------
# fp8 weight-only
# ---------------------------------------------------------------------------
def test_w8a16_golden(golden):
    """Inputs/outputs of the reference's dequantise-then-linear route (fp8.py:433-454); the kernel
    keeps the weight in fp32, so it agrees to the W4A16 test tolerance (rtol = atol = 1e-2)."""
    g = golden("w8a16")
    for tag in g["names"]:
        tag = str(tag)
        w = torch.from_numpy(g[tag + "_w"]).to(DEV)
        N = w.shape[0]
        scale = torch.from_numpy(np.broadcast_to(g[tag + "_scale"], (N,)).astype(np.float32).copy()).to(DEV)
        bias = to_dev(g[tag + "_bias"]) if tag + "_bias" in g else None
        out = ops.w8a16_gemm(to_dev(g[tag + "_x"]), w, scale, None, bias)
        np.testing.assert_allclose(f32(to_bits(out)), f32(g[tag + "_ref"]), rtol=1e-2, atol=1e-2, err_msg=tag)


@pytest.mark.parametrize("M,N,K,block,dt", [(1, 4096, 4096, None, orc.BF16), (3, 1000, 512, None, orc.BF16),
                                            (4, 384, 1024, (128, 128), orc.BF16), (7, 200, 256, (64, 32), orc.FP16),
                                            (1, 8192, 14336, (128, 128), orc.BF16), (40, 256, 512, None, orc.BF16)])
def test_w8a16_vs_oracle(M, N, K, block, dt):
    rng = np.random.default_rng(N + K)
    w8 = orc.f32_to_e4m3((rng.standard_normal((N, K)) * 40).astype(np.float32))
    if block is None:
        s = (0.001 + 0.002 * rng.random(N)).astype(np.float32)
    else:
        s = (0.001 + 0.002 * rng.random(((N + block[0] - 1) // block[0], (K + block[1] - 1) // block[1]))).astype(np.float32)
    a = orc.f32_to_h16(rng.standard_normal((M, K)).astype(np.float32), dt)
    bias = orc.f32_to_h16(rng.standard_normal(N).astype(np.float32), dt)
    ref = orc.w8a16_gemm(a, w8, s, bias, block=block, dt=dt)
    out = ops.w8a16_gemm(to_dev(a, dt), torch.from_numpy(w8).to(DEV), torch.from_numpy(s).to(DEV), block, to_dev(bias, dt))
    np.testing.assert_allclose(f32(to_bits(out), dt), f32(ref, dt), rtol=1e-2, atol=1e-2)
    # fp32 products on both sides (the oracle sums in double): all but rounding ties are bit-equal
    assert (to_bits(out) == ref).mean() > (0.97 if dt == orc.BF16 else 0.9)
    deq = ops.w8a16_dequantize(torch.from_numpy(w8).to(DEV), torch.from_numpy(s).to(DEV), block, TDT[dt])
    assert np.array_equal(to_bits(deq), orc.f32_to_h16(orc.w8a16_dequant(w8, s, block=block, dt=dt), dt))


def test_w8a16_many_token_route(monkeypatch):
    monkeypatch.setattr(ops, "DEQUANT_MATMUL_MIN_TOKENS", 32)
    rng = np.random.default_rng(2)
    M, N, K = 64, 256, 512
    w8 = orc.f32_to_e4m3((rng.standard_normal((N, K)) * 40).astype(np.float32))
    s = (0.001 + 0.002 * rng.random(N)).astype(np.float32)
    a = orc.f32_to_h16(rng.standard_normal((M, K)).astype(np.float32))
    ref = orc.w8a16_gemm(a, w8, s, None, round_weight=2)   # weight rounded to bf16, scale kept in fp32
    out = ops.w8a16_gemm(to_dev(a), torch.from_numpy(w8).to(DEV), torch.from_numpy(s).to(DEV))
    np.testing.assert_allclose(f32(to_bits(out)), f32(ref), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("M,N,K,G,dt", [(1, 4096, 4096, 128, orc.BF16), (3, 1000, 512, 64, orc.BF16), (4, 256, 2048, -1, orc.FP16),
                                        (16, 512, 1024, 32, orc.BF16)])
def test_gptq_8bit_vs_oracle(M, N, K, G, dt, monkeypatch):
    """8-bit symmetric GPTQ: W = (q - 128) * s (uint8b128).  Kernel against the oracle's restatement;
    the many-token route (dequantise + dense GEMM) against the rounded-weight mode."""
    rng = np.random.default_rng(M + N + K)
    Ge = K if G == -1 else G
    q = rng.integers(0, 256, size=(K, N), dtype=np.int32)
    a = orc.f32_to_h16(0.25 * rng.standard_normal((M, K)).astype(np.float32), dt)
    s = orc.f32_to_h16((0.004 * rng.random((K // Ge, N)) + 0.001).astype(np.float32), dt)
    bias = orc.f32_to_h16(rng.standard_normal(N).astype(np.float32), dt)
    w, sc = ops.gptq8_repack(i32(orc.gptq8_pack(q)), to_dev(s, dt), G)
    assert w.shape == (N, K) and np.array_equal(w.cpu().numpy(), q.T.astype(np.uint8))
    assert np.array_equal(sc.cpu().numpy(), f32(s, dt).T)
    ref = orc.gptq8_gemm(a, q, s, Ge, None, dt)
    out = ops.wi8a16_gemm(to_dev(a, dt), w, sc, G)
    np.testing.assert_allclose(f32(to_bits(out), dt), f32(ref, dt), rtol=1e-2, atol=1e-2)
    assert (to_bits(out) == ref).mean() > (0.97 if dt == orc.BF16 else 0.9)
    deq = ops.wi8a16_dequantize(w, sc, G, TDT[dt])
    assert np.array_equal(to_bits(deq), orc.f32_to_h16(orc.gptq8_dequant(q, s, Ge, dt).T, dt))
    monkeypatch.setattr(ops, "DEQUANT_MATMUL_MIN_TOKENS", 0)
    many = ops.wi8a16_gemm(to_dev(a, dt), w, sc, G, bias=to_dev(bias, dt))
    ref_rw = orc.gptq8_gemm(a, q, s, Ge, None, dt, round_weight=True)
    want = f32(ref_rw, dt) + f32(bias, dt)[None, :]
    np.testing.assert_allclose(f32(to_bits(many), dt), want, rtol=2e-2, atol=2e-2)


def test_w8a16_bad_arguments():
    w = torch.zeros((64, 128), dtype=torch.uint8, device=DEV)
    x = torch.zeros((1, 128), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="per-channel scales must have 64"):
        ops.w8a16_gemm(x, w, torch.ones(3, device=DEV))
    with pytest.raises(RuntimeError, match="block scales shape"):
        ops.w8a16_gemm(x, w, torch.ones((1, 2), device=DEV), (64, 32))
    with pytest.raises(RuntimeError, match="multiple of 16"):
        ops.w8a16_gemm(x, w, torch.ones((1, 16), device=DEV), (64, 8))


# ---------------------------------------------------------------------------
# mixture of experts
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("T,E,topk,renorm,dt", [(1, 8, 2, True, torch.bfloat16), (5, 8, 2, False, torch.float32),
                                                (3, 64, 6, True, torch.float16), (2, 160, 8, False, torch.float32),
                                                (4, 4, 4, True, torch.bfloat16)])
def test_topk_softmax(T, E, topk, renorm, dt):
    torch.manual_seed(E + topk)
    g = torch.randn((T, E), dtype=torch.float32).to(dt)
    g[0, 1] = g[0, 0]                                   # a tie: the lower expert index wins
    w = torch.empty((T, topk), dtype=torch.float32, device=DEV)
    ids = torch.empty((T, topk), dtype=torch.int32, device=DEV)
    ops.topk_softmax(w, ids, None, g.to(DEV), renorm)
    p = torch.softmax(g.float(), dim=-1)
    order = torch.argsort(-p, dim=-1, stable=True)[:, :topk]
    ref_w = p.gather(1, order)
    if renorm:
        ref_w = ref_w / ref_w.sum(-1, keepdim=True)
    assert torch.equal(ids.cpu().long(), order)
    torch.testing.assert_close(w.cpu(), ref_w, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("T,E,H,topk,renorm,dt", [(1, 8, 4096, 2, True, torch.bfloat16), (3, 8, 4096, 2, True, torch.float16),
                                                   (2, 64, 1024, 6, False, torch.bfloat16), (16, 7, 6144, 3, True, torch.bfloat16),
                                                   (5, 60, 2048, 4, True, torch.bfloat16), (2, 3, 136, 1, False, torch.float16)])
def test_moe_router_topk_equals_the_three_launches(T, E, H, topk, renorm, dt):
    """One launch for add + RMSNorm, gate projection and softmax top-k: every output (residual, normalised row,
    logits, weights, ids) holds the bits of fused_add_rms_norm -> linear_skinny -> topk_softmax."""
    torch.manual_seed(T * E + H)
    x = torch.randn((T, H), device=DEV).to(dt)
    res = torch.randn((T, H), device=DEV).to(dt)
    nw = (1 + 0.1 * torch.randn(H, device=DEV)).to(dt)
    router = (torch.randn((E, H), device=DEV) / H ** 0.5).to(dt)
    router[1] = router[0]                                # tied logits: the lower expert index wins
    assert ops.moe_router_topk_supported(E, H)
    x0, r0 = x.clone(), res.clone()
    ops.fused_add_rms_norm(x0, r0, nw, 1e-5)
    lg0 = ops.linear_skinny(x0, router)
    w0 = torch.empty((T, topk), dtype=torch.float32, device=DEV)
    i0 = torch.empty((T, topk), dtype=torch.int32, device=DEV)
    ops.topk_softmax(w0, i0, None, lg0, renorm)
    x1, r1 = x.clone(), res.clone()
    w1, i1, lg1 = ops.moe_router_topk(x1, r1, nw, 1e-5, router, topk, renorm, return_logits=True)
    assert torch.equal(r1, r0) and torch.equal(x1, x0)
    assert torch.equal(lg1, lg0)
    assert torch.equal(i1, i0) and torch.equal(w1, w0)
    # and against the oracle's norm, for the stand-alone meaning of the outputs
    odt = orc.BF16 if dt == torch.bfloat16 else orc.FP16
    _, want_r = orc.fused_add_rms_norm(to_bits(x), to_bits(res), to_bits(nw), 1e-5, odt)
    np.testing.assert_array_equal(to_bits(r1), want_r)
    assert not ops.moe_router_topk_supported(65, 4096) and not ops.moe_router_topk_supported(8, 8200)
    with pytest.raises(RuntimeError, match="up to 64 experts"):
        ops.moe_router_topk(x1, r1, nw, 1e-5, torch.zeros((65, H), dtype=dt, device=DEV), 2, True)


def _moe_case(g, tag):
    m, n, k, e, topk, G, has_zp = [int(v) for v in g[tag + "_meta"]]
    w1 = (g[tag + "_w1_q"], g[tag + "_w1_s"], g[tag + "_w1_z"] if has_zp else None)
    w2 = (g[tag + "_w2_q"], g[tag + "_w2_s"], g[tag + "_w2_z"] if has_zp else None)
    dev = [ops.moe_w4a16_repack(torch.from_numpy(w[0]).to(DEV), to_dev(w[1]),
                                None if w[2] is None else torch.from_numpy(w[2]).to(DEV), G) for w in (w1, w2)]
    return (m, n, k, e, topk, G, has_zp), w1, w2, dev


def test_moe_golden_and_oracle(golden, monkeypatch):
    """The expert path against the reference's expected outputs (tests/kernels/moe/test_moe.py:698,
    atol 2e-2) and, tightly, against the oracle restatement with fp32 weights; the per-slot
    kernel route, the every-expert-over-all-tokens route and the grouped dense route for many tokens."""
    import moe_ref
    from mi355q.plugin import fused_experts, fused_topk, moe as moe_mod
    g = golden("moe")
    for tag in g["names"]:
        tag = str(tag)
        (m, n, k, e, topk, G, has_zp), w1, w2, (d13, d2) = _moe_case(g, tag)
        a = to_dev(g[tag + "_a"])
        tw, ids = fused_topk(to_dev(g[tag + "_score"]), topk, False)
        rw, rids = moe_ref.route(g[tag + "_score"], topk)
        assert np.array_equal(ids.cpu().numpy(), rids), tag
        np.testing.assert_allclose(tw.cpu().numpy(), rw, rtol=1e-5)
        ref = f32(g[tag + "_ref"])
        exact = f32(moe_ref.moe(g[tag + "_a"], w1, w2, rw, rids, G))
        for route in ("slots", "all_experts", "grouped", "grouped_host", None):
            monkeypatch.setattr(moe_mod, "FORCE_ROUTE", route)
            out = f32(to_bits(fused_experts(a, d13, d2, tw, ids)))
            np.testing.assert_allclose(out, ref, rtol=0, atol=2e-2, err_msg=tag)
            np.testing.assert_allclose(out, exact, rtol=1e-2, atol=2e-3, err_msg=tag)


def test_moe_align_block_size():
    """Token sort of the grouped expert GEMM (ops.moe_align_block_size of the reference,
    vllm/_custom_ops.py:2237-2255): every expert's run holds exactly its slots, runs are padded to whole
    tiles with the sentinel, tiles are labelled with their expert, slot_pos inverts the sort; slots of experts
    held elsewhere (negative ids) are left out."""
    rng = np.random.default_rng(3)
    for slots, E in [(1, 4), (130, 8), (4096, 8), (5000, 60), (700, 256)]:
        ids = rng.integers(-1 if E == 8 else 0, E, size=slots).astype(np.int32)
        block = _lib.lib().mi355q_moe_grouped_block_rows()
        sorted_ids, tile_expert, num_tiles, slot_pos = (t.cpu().numpy() for t in ops.moe_align_block_size(
            torch.from_numpy(ids).to(DEV), E))
        assert block == 64 and sorted_ids.size == tile_expert.size * block
        pos, nt = 0, 0
        for e in range(E):
            mine = np.nonzero(ids == e)[0]
            tiles = -(-mine.size // block)
            run = sorted_ids[pos:pos + tiles * block]
            assert np.array_equal(np.sort(run[:mine.size]), mine) and np.all(run[mine.size:] == slots), (slots, E, e)
            assert np.all(tile_expert[nt:nt + tiles] == e)
            pos, nt = pos + tiles * block, nt + tiles
        assert int(num_tiles[0]) == nt and np.all(tile_expert[nt:] == -1) and np.all(sorted_ids[pos:] == slots)
        held = ids >= 0
        assert np.all(slot_pos[~held] == -1) and np.array_equal(sorted_ids[slot_pos[held]], np.nonzero(held)[0])


@pytest.mark.parametrize("awq,pair", [(False, True), (True, False), (True, True)])
@pytest.mark.parametrize("T", [65, 300])
def test_moe_grouped_route_for_prompt_batches(T, awq, pair, monkeypatch):
    """More tokens than the every-expert route takes: token-sorted rows through the grouped expert GEMM (one
    launch per projection, no host round trip), to the same bar as the other routes -- fp32 scales, one
    rounding -- with and without zero points, the SiluAndMul epilogue, and an expert map; the route is the
    default choice at this size and can be captured."""
    import moe_ref
    from mi355q.plugin import fused_experts, moe as moe_mod
    rng = np.random.default_rng(41 + T + awq)
    E, H, I, G, topk = 6, 1024, 1024, 128, 2
    def mk(n, k):
        z = rng.integers(0, 256, size=(E, n // 2, k // G), dtype=np.uint8) if awq else None
        return (rng.integers(0, 256, size=(E, n, k // 2), dtype=np.uint8),
                orc.f32_to_h16((0.01 * rng.random((E, n, k // G)) + 0.002).astype(np.float32)), z)
    w1, w2 = mk(2 * I, H), mk(H, I)
    emap = np.array([0, 1, -1, 2, 3, 4], dtype=np.int32)
    local = [0, 1, 3, 4, 5]
    sub = lambda w, idx: (w[0][idx], w[1][idx], None if w[2] is None else w[2][idx])
    def dev(w, idx, pair_halves=False):
        w = sub(w, idx)
        return ops.moe_w4a16_repack(torch.from_numpy(w[0]).to(DEV), to_dev(w[1]),
                                    None if w[2] is None else torch.from_numpy(w[2]).to(DEV), G, pair_halves=pair_halves)
    a = orc.f32_to_h16((0.5 * rng.standard_normal((T, H))).astype(np.float32))
    ids = np.stack([rng.permutation(E)[:topk] for _ in range(T)]).astype(np.int32)
    ids[: T // 3] = [4, 1]                                    # a crowded expert: several tiles in one run
    tw = rng.random((T, topk)).astype(np.float32)
    for use_map in (False, True):
        idx = local if use_map else list(range(E))
        d13, d2 = dev(w1, idx, pair), dev(w2, idx)
        assert d13.pair_halves == pair and ops.moe_grouped_mm_supported(d13) and ops.moe_grouped_mm_supported(d2)
        ref = moe_ref.moe(a, sub(w1, idx), sub(w2, idx), tw, ids, G, expert_map=emap if use_map else None)
        scale_mag = max(1.0, float(np.abs(f32(ref)).max()) / 4)
        args = (to_dev(a), d13, d2, torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV),
                torch.from_numpy(emap).to(DEV) if use_map else None)
        monkeypatch.setattr(moe_mod, "FORCE_ROUTE", None)
        out = fused_experts(*args)
        np.testing.assert_allclose(f32(to_bits(out)), f32(ref), rtol=1e-2, atol=5e-3 * scale_mag)
        monkeypatch.setattr(moe_mod, "FORCE_ROUTE", "all_experts")
        other = fused_experts(*args)
        assert (to_bits(out) == to_bits(other)).mean() > 0.97     # same arithmetic, other summation order at most
        monkeypatch.setattr(moe_mod, "FORCE_ROUTE", None)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cap = fused_experts(*args)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(cap, out)


def test_moe_expert_map_and_sum(monkeypatch):
    """Slots whose expert lives on another rank contribute nothing; moe_sum without weights is a
    plain fp32 sum."""
    import moe_ref
    from mi355q.plugin import fused_experts
    rng = np.random.default_rng(12)
    E, H, I, G, T, topk = 4, 128, 64, 32, 3, 2
    w1 = (rng.integers(0, 256, size=(E, 2 * I, H // 2), dtype=np.uint8),
          orc.f32_to_h16((0.02 * rng.random((E, 2 * I, H // G)) + 0.005).astype(np.float32)), None)
    w2 = (rng.integers(0, 256, size=(E, H, I // 2), dtype=np.uint8),
          orc.f32_to_h16((0.02 * rng.random((E, H, I // G)) + 0.005).astype(np.float32)), None)
    emap = np.array([-1, 0, -1, 1], dtype=np.int32)           # this rank holds global experts 1 and 3
    local = [1, 3]
    d13 = ops.moe_w4a16_repack(torch.from_numpy(w1[0][local]).to(DEV), to_dev(w1[1][local]), None, G)
    d2 = ops.moe_w4a16_repack(torch.from_numpy(w2[0][local]).to(DEV), to_dev(w2[1][local]), None, G)
    a = orc.f32_to_h16(rng.standard_normal((T, H)).astype(np.float32))
    ids = np.array([[0, 1], [3, 2], [1, 3]], dtype=np.int32)
    tw = rng.random((T, topk)).astype(np.float32)
    from mi355q.plugin import moe as moe_mod
    ref = moe_ref.moe(a, (w1[0][local], w1[1][local], None), (w2[0][local], w2[1][local], None), tw, ids, G,
                      expert_map=emap)
    for route in ("slots", "all_experts", None):
        monkeypatch.setattr(moe_mod, "FORCE_ROUTE", route)
        out = fused_experts(to_dev(a), d13, d2, torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV),
                            torch.from_numpy(emap).to(DEV))
        np.testing.assert_allclose(f32(to_bits(out)), f32(ref), rtol=1e-2, atol=2e-3, err_msg=str(route))
    y = torch.randn((T, topk, H), device=DEV).to(torch.bfloat16)
    s = torch.empty((T, H), dtype=torch.bfloat16, device=DEV)
    ops.moe_sum(y, s)
    assert torch.equal(s, y.float().sum(1).to(torch.bfloat16))


def test_moe_grouped_route_when_no_selected_expert_is_local(monkeypatch):
    """Expert parallelism, a batch none of whose selected experts this rank holds: the grouped route writes no
    tile, so every row it gathers back is memory no kernel wrote.  The output must be exact zeros (what the rank
    adds to the reduce-scatter), even when that memory holds NaN."""
    from mi355q.plugin import fused_experts, moe as moe_mod
    rng = np.random.default_rng(5)
    E, H, I, G, T, topk = 6, 1024, 1024, 128, 70, 2
    def mk(n, k):
        return (rng.integers(0, 256, size=(2, n, k // 2), dtype=np.uint8),
                orc.f32_to_h16((0.01 * rng.random((2, n, k // G)) + 0.002).astype(np.float32)))
    w1, w2 = mk(2 * I, H), mk(H, I)
    d13 = ops.moe_w4a16_repack(torch.from_numpy(w1[0]).to(DEV), to_dev(w1[1]), None, G, pair_halves=True)
    d2 = ops.moe_w4a16_repack(torch.from_numpy(w2[0]).to(DEV), to_dev(w2[1]), None, G)
    emap = torch.tensor([0, 1, -1, -1, -1, -1], dtype=torch.int32, device=DEV)   # this rank holds experts 0 and 1
    ids = torch.from_numpy(np.stack([2 + rng.permutation(4)[:topk] for _ in range(T)]).astype(np.int32)).to(DEV)
    tw = torch.from_numpy(rng.random((T, topk)).astype(np.float32)).to(DEV)
    a = to_dev(orc.f32_to_h16((0.5 * rng.standard_normal((T, H))).astype(np.float32)))
    monkeypatch.setattr(moe_mod, "FORCE_ROUTE", "grouped")
    # poison the allocator's free blocks: the route's torch.empty buffers then start out as NaN
    for _ in range(2):
        junk = [torch.full((n,), float("nan"), dtype=torch.bfloat16, device=DEV) for n in (1 << 24, 1 << 22, 1 << 20, 1 << 18)]
        del junk
        out = fused_experts(a, d13, d2, tw, ids, emap)
        assert out.shape == (T, H) and not out.isnan().any() and not out.any()


@pytest.mark.parametrize("awq,pair", [(False, False), (True, False), (False, True), (True, True)])
def test_moe_decode_slots_on_the_burst_kernel(awq, pair, monkeypatch):
    """Expert layers wide enough to occupy the chip (>= 128 strips per slot) take the one-workgroup-
    per-strip decode kernel with the expert indirection: same bar as the dense GEMV, including slots
    whose expert is held elsewhere (zero rows) and zero points."""
    import moe_ref
    from mi355q.plugin import fused_experts
    rng = np.random.default_rng(31 + awq)
    E, H, I, G, T, topk = 4, 1024, 2048, 128, 2, 2
    def mk(n, k):
        z = rng.integers(0, 256, size=(E, n // 2, k // G), dtype=np.uint8) if awq else None
        return (rng.integers(0, 256, size=(E, n, k // 2), dtype=np.uint8),
                orc.f32_to_h16((0.02 * rng.random((E, n, k // G)) + 0.005).astype(np.float32)), z)
    w1, w2 = mk(2 * I, H), mk(H, I)
    emap = np.array([0, 1, -1, 2], dtype=np.int32)
    local = [0, 1, 3]
    def dev(w, pair_halves=False):
        return ops.moe_w4a16_repack(torch.from_numpy(w[0][local]).to(DEV), to_dev(w[1][local]),
                                    None if w[2] is None else torch.from_numpy(w[2][local]).to(DEV), G,
                                    pair_halves=pair_halves)
    d13, d2 = dev(w1, pair), dev(w2)
    assert d13.pair_halves == pair
    assert d13.N // d13.tile >= 128 and d2.N // d2.tile * T * topk >= 128
    a = orc.f32_to_h16((0.5 * rng.standard_normal((T, H))).astype(np.float32))
    ids = np.array([[0, 2], [3, 1]], dtype=np.int32)
    tw = rng.random((T, topk)).astype(np.float32)
    from mi355q.plugin import moe as moe_mod
    sub = lambda w: (w[0][local], w[1][local], None if w[2] is None else w[2][local])
    ref = moe_ref.moe(a, sub(w1), sub(w2), tw, ids, G, expert_map=emap)
    ref_rw = moe_ref.moe(a, sub(w1), sub(w2), tw, ids, G, expert_map=emap, round_weight=True)
    scale_mag = max(1.0, float(np.abs(f32(ref)).max()) / 4)   # zero points un-centre the weights: larger outputs
    assert ops.moe_grouped_mm_supported(d13) and ops.moe_grouped_mm_supported(d2)
    for route in ("slots", "all_experts", "grouped", "grouped_host"):
        monkeypatch.setattr(moe_mod, "FORCE_ROUTE", route)
        out = fused_experts(to_dev(a), d13, d2, torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV),
                            torch.from_numpy(emap).to(DEV))
        # the host-grouped route multiplies with weights rounded to 16 bits (the reference's many-token
        # arithmetic): compared with the oracle in that mode
        want = ref_rw if route == "grouped_host" else ref
        np.testing.assert_allclose(f32(to_bits(out)), f32(want), rtol=1e-2, atol=5e-3 * scale_mag, err_msg=route)
    # the slot rows themselves: expert GEMV against the oracle, zero row for the absent expert
    lids = torch.from_numpy(emap[ids.reshape(-1)]).to(DEV)
    r0 = moe_ref.expert_gemm(a[:1], w1[0][0], w1[1][0], None if w1[2] is None else w1[2][0], G)
    if pair:
        h = ops.moe_w4a16_gemm(to_dev(a), d13, lids, topk, silu_pairs=True)
        assert h.shape == (T * topk, I) and not h[1].any()
        assert (to_bits(h[:1]) == orc.silu_and_mul(r0)).mean() > 0.97
    else:
        h = ops.moe_w4a16_gemm(to_dev(a), d13, lids, topk)
        assert not h[1].any()
        assert (to_bits(h[:1]) == r0).mean() > 0.98


# ---------------------------------------------------------------------------
# gate_up projection with SiluAndMul in its epilogue (pair_halves layout)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("M,K,I,awq,norm", [(1, 4096, 14336, False, True), (2, 1024, 2048, True, True),
                                           (5, 512, 1024, False, False), (1, 8192, 3584, False, True)])
def test_w4a16_gate_up_silu(M, K, I, awq, norm):
    """= fused_add_rms_norm -> gate_up GEMM -> silu_and_mul of the oracle, with the two halves of
    the checkpoint's output channels interleaved at repack time."""
    rng = np.random.default_rng(K + I)
    G, N = 128, 2 * I
    assert ops.w4a16_gate_up_silu_supported(K, N, G)
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    s = orc.f32_to_h16((0.03 * rng.random((K // G, N)) + 0.005).astype(np.float32))
    x = orc.f32_to_h16(rng.standard_normal((M, K)).astype(np.float32))
    r = orc.f32_to_h16(rng.standard_normal((M, K)).astype(np.float32))
    nw = orc.f32_to_h16((1 + 0.1 * rng.standard_normal(K)).astype(np.float32))
    if awq:
        z = rng.integers(0, 16, size=(K // G, N), dtype=np.int32)
        qw, qz = orc.pack_cols(q, awq_order=True), orc.pack_cols(z, awq_order=True)
        w = ops.w4a16_repack(i32(qw), to_dev(s), i32(qz), G, _lib.W_AWQ, _lib.ZP_AWQ, pair_halves=True)
        kw = dict(layout=orc.W_AWQ, zp_mode=orc.ZP_AWQ)
    else:
        qw, qz = orc.gptq_pack(q), None
        w = ops.w4a16_repack(i32(qw), to_dev(s), None, G, pair_halves=True)
        kw = {}
    if norm:
        xn, rn = orc.fused_add_rms_norm(x, r, nw, 1e-5)
    else:
        xn, rn = x, None
    y = orc.w4a16_gemm(xn, qw, s, qz, None, G=G, **kw)
    ref = orc.silu_and_mul(y)
    act, res = ops.w4a16_gate_up_silu(to_dev(x), to_dev(r) if norm else None, to_dev(nw) if norm else None, 1e-5, w,
                                      add_norm=norm)
    assert act.shape == (M, I)
    np.testing.assert_allclose(f32(to_bits(act)), f32(ref), rtol=2e-2, atol=1e-2)
    assert (to_bits(act) == ref).mean() > 0.97
    if norm:
        assert np.array_equal(to_bits(res), rn)
    # the same weight through the plain GEMM gives the interleaved projection output
    plain = ops.unpair_halves(ops.w4a16_gemm(to_dev(xn), w))
    assert (to_bits(plain) == y).mean() > 0.98
    deq = ops.unpair_halves(ops.w4a16_dequantize(w))
    assert np.array_equal(to_bits(deq), orc.f32_to_h16(orc.w4a16_dequant(qw, s, qz, G=G, **kw)))


# ---- random sampling with top-k / top-p ---------------------------------------------------------
def _kept(masked_row):
    return ~np.isneginf(masked_row)


@pytest.mark.gpu
def test_sample_topk_topp_vs_reference_golden(golden):
    """The HIP sampler against the outputs of the reference's apply_top_k_top_p_pytorch + random_sample
    (tests/golden/sampler_topk_topp.npz, topk_topp_sampler.py:367-407, 441-466): same token for every
    row (the golden records how decisive each draw was: runner-up / best score, all far from 1)."""
    d = golden("sampler_topk_topp")
    assert d["runner_up_ratio"].max() < 0.999
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ids = ops.sample_topk_topp(t(d["logits"]), t(d["temperature"]), t(d["top_k"]), t(d["top_p"]), t(d["noise"]))
    assert np.array_equal(ids.cpu().numpy(), d["ids"])
    # and through the Sampler module (temperature, masks and draw in one call)
    from mi355q.plugin.sampler import Sampler, SamplingMetadata
    B = d["logits"].shape[0]
    md = SamplingMetadata(temperature=t(d["temperature"]), all_greedy=False, all_random=True, top_p=t(d["top_p"]),
                          top_k=t(d["top_k"]), generators={i: torch.Generator(device=dev).manual_seed(5 + i) for i in range(B)})
    out = Sampler()(t(d["logits"]), md).sampled_token_ids.view(-1).cpu().numpy()
    scaled = d["logits"] / d["temperature"][:, None]
    masked = orc.top_k_top_p_mask(scaled, d["top_k"], d["top_p"])
    assert all(_kept(masked[b])[out[b]] for b in range(B))   # whatever the GPU generators drew, the token survives both masks


@pytest.mark.gpu
@pytest.mark.parametrize("V,B", [(128256, 4), (1000, 3), (32000, 16)])
def test_sample_topk_topp_vs_oracle(V, B):
    """Full vocabulary sizes against the CPU restatement: identical tokens unless the two best scores are
    within 1e-5 of each other; masks checked through draws that would expose a wrongly kept token
    (noise tiny on one dropped token)."""
    rng = np.random.default_rng(V + B)
    logits = (rng.standard_normal((B, V)) * 3).astype(np.float32)
    logits[0, : V // 7] = -np.inf
    temp = rng.uniform(0.3, 2.0, B).astype(np.float32)
    k = rng.integers(1, min(V, 2000), B).astype(np.int32)
    k[-1] = V                                   # no top-k on the last row
    p = rng.uniform(0.05, 1.0, B).astype(np.float32)
    p[0] = 1.0                                  # no top-p on the first
    noise = rng.exponential(1.0, (B, V)).astype(np.float32) + 1e-10
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(a).to(dev)
    got = ops.sample_topk_topp(t(logits), t(temp), t(k), t(p), t(noise)).cpu().numpy()
    want = orc.sample_topk_topp(logits, temp, k, p, noise)
    scaled = logits / temp[:, None]
    masked = orc.top_k_top_p_mask(scaled, k, p)
    e = np.exp(masked - masked.max(axis=1, keepdims=True))
    score = e / e.sum(axis=1, keepdims=True) / noise
    for b in range(B):
        if got[b] != want[b]:
            assert score[b, got[b]] >= score[b, want[b]] * (1 - 1e-5), (b, got[b], want[b])
    # a dropped token with vanishing noise would win the arg-max if a mask let it through
    for b in range(B):
        dropped = np.flatnonzero(~_kept(masked[b]) & np.isfinite(scaled[b]))
        if dropped.size == 0:
            continue
        j = dropped[np.argmax(scaled[b][dropped])]      # the largest dropped logit: the first a loose threshold would admit
        n2 = noise.copy()
        n2[b, j] = 1e-30
        got2 = ops.sample_topk_topp(t(logits), t(temp), t(k), t(p), t(n2)).cpu().numpy()
        assert got2[b] != j, (b, j)
    # only the top-k: k = 1 is the arg-max whatever the noise
    one = ops.sample_topk_topp(t(logits), None, t(np.ones(B, dtype=np.int32)), None, t(noise)).cpu().numpy()
    assert np.array_equal(one, np.argmax(logits, axis=1))


# ---- 17..32 rows (up to 64 on request), one streaming workgroup per strip ------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,G,tile,awq,dt,silu", [
    (32, 4096, 28672, 128, 112, False, orc.BF16, False),   # gate_up at a 32-row decode batch
    (32, 4096, 28672, 128, 112, False, orc.BF16, True),    # ... with the SiLU-pair epilogue the decode layer uses
    (24, 4096, 6144, 128, 32, True, orc.BF16, False),      # qkv strips, zero points, ragged row tile
    (17, 14336, 4096, 128, 16, False, orc.BF16, False),    # down: 7 chunks in flight per wave
    (64, 4096, 4096, 128, 16, False, orc.BF16, False),     # two 32-row passes
    (33, 1024, 4096, 256, 64, True, orc.FP16, False),      # group of two chunks, fp16, one row in the second pass
    (20, 8192, 7168, 128, 32, False, orc.BF16, False),     # a 70B TP=8 shard shape
    (32, 4096, 6144, 128, 32, True, orc.BF16, False),      # activation rows fetched whole, fragments through LDS (32-wide strips: from 17 rows on)
    (27, 2048, 6144, 128, 32, False, orc.FP16, False),     # ... ragged second row tile, fp16
    (29, 2048, 7168, 128, 112, True, orc.BF16, False),     # ... 112-wide strips with zero points (two subtile passes)
    (14, 2048, 7168, 128, 112, False, orc.BF16, True),     # 112-wide strips are streamed from 13 rows on (one real row tile)
    (13, 4096, 7168, 128, 112, True, orc.FP16, False),
    (32, 4096, 4096, 128, 16, False, orc.BF16, False),     # 16-wide strips on a short K stream too (o_proj)
    (26, 1024, 8192, 128, 16, True, orc.FP16, False),
    (19, 4096, 6144, 128, 32, False, orc.BF16, False),     # ragged second row tile on the whole-row fetch
    (13, 2048, 4096, 128, 16, True, orc.BF16, False),      # 16-wide strips, short K: streamed from 13 rows on
])
def test_w4a16_stream_rows_vs_oracle(M, K, N, G, tile, awq, dt, silu):
    """Decode batches beyond the 16-row kernels: every strip streamed once by one workgroup, activations
    read from global memory per lane (or whole rows, through LDS), per-wave partial sums in LDS or registers.  Same arithmetic as the other
    kernels, so the same bar (reference tolerance rtol = atol = 1e-2, test_triton_w4a16.py:162; nearly
    all outputs bit-equal to the fp32-product oracle)."""
    rng = np.random.default_rng(M + K + N)
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    a = orc.f32_to_h16(0.25 * rng.standard_normal((M, K)).astype(np.float32), dt)
    s = orc.f32_to_h16((0.05 * rng.random((K // G, N))).astype(np.float32), dt)
    bias = None if silu else orc.f32_to_h16(rng.standard_normal(N).astype(np.float32), dt)
    if awq:
        z = rng.integers(0, 16, size=(K // G, N), dtype=np.int32)
        qw, qz = orc.pack_cols(q, awq_order=True), orc.pack_cols(z, awq_order=True)
        ref = orc.w4a16_gemm(a, qw, s, qz, bias, G=G, dt=dt, layout=orc.W_AWQ, zp_mode=orc.ZP_AWQ)
        w = ops.w4a16_repack(i32(qw), to_dev(s, dt), i32(qz), G, _lib.W_AWQ, _lib.ZP_AWQ, tile=tile, pair_halves=silu)
    else:
        qw = orc.gptq_pack(q)
        ref = orc.w4a16_gemm(a, qw, s, None, bias, G=G, dt=dt)
        w = ops.w4a16_repack(i32(qw), to_dev(s, dt), None, G, tile=tile, pair_halves=silu)
    assert w.tile == tile
    # beyond 32 rows the library prefers the tiled kernel; the two-pass form of this one is asked for by option
    _lib.set_option("MI355Q_STREAM_ROWS_MAX", "64")
    try:
        out = ops.w4a16_mm(to_dev(a, dt), w, bias=None if bias is None else to_dev(bias, dt), silu_pairs=silu)
        again = ops.w4a16_mm(to_dev(a, dt), w, bias=None if bias is None else to_dev(bias, dt), silu_pairs=silu)
    finally:
        _lib.set_option("MI355Q_STREAM_ROWS_MAX", None)
    if silu:   # the checkpoint's (gate | up) halves: SiluAndMul of the rounded projection
        ref = orc.silu_and_mul(ref, dt)
    got, want = f32(to_bits(out), dt), f32(ref, dt)
    bad = ~np.isclose(got, want, rtol=1e-2, atol=1e-2)
    assert bad.mean() < 1e-5
    assert (to_bits(out) == ref).mean() > (0.9 if dt == orc.FP16 or silu else 0.97)
    assert torch.equal(out, again)
    if 17 <= M <= 32 and tile != 16:   # whole-row fetch through LDS and per-lane gather of fragments: the same bits
        # (16-wide strips have no gathered form to compare with: without the whole-row fetch they take the tiled kernel)
        _lib.set_option("MI355Q_STREAM_ROWS_XLDS_MIN", "99")
        try:
            gathered = ops.w4a16_mm(to_dev(a, dt), w, bias=None if bias is None else to_dev(bias, dt), silu_pairs=silu)
        finally:
            _lib.set_option("MI355Q_STREAM_ROWS_XLDS_MIN", None)
        assert torch.equal(out, gathered)
    if M > 32:   # the default route at these row counts (tiled kernel) against the same oracle
        dflt = ops.w4a16_mm(to_dev(a, dt), w, bias=None if bias is None else to_dev(bias, dt), silu_pairs=silu)
        assert (~np.isclose(f32(to_bits(dflt), dt), want, rtol=1e-2, atol=1e-2)).mean() < 1e-5


# ---- 2- and 3-bit GPTQ ---------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("bits,sym,v2,M,K,N,G", [(2, True, False, 1, 4096, 4096, 128), (3, True, False, 3, 2048, 6144, 128),
                                                 (3, False, False, 1, 4096, 4096, 64), (2, False, True, 20, 1024, 4096, 128),
                                                 (3, False, True, 200, 1024, 2048, 128)])
def test_gptq_2_and_3_bit_checkpoints(bits, sym, v2, M, K, N, G):
    """2- / 3-bit AutoGPTQ tensors widened to the 4-bit layout at load time (mi355q_gptq_expand_bits) and
    run on the W4A16 kernels: bit-exact widening against the oracle's bit-string unpacker, and the GEMM
    against the oracle on the same values -- W[k, n] = (q - z) * s with z = 2^(bits-1) for symmetric
    checkpoints, stored as z - 1 unless use_v2_format (q_gemm.cu:202,323-682)."""
    rng = np.random.default_rng(bits * 1000 + K + N + M)
    q = rng.integers(0, 1 << bits, size=(K, N), dtype=np.int32)
    s = orc.f32_to_h16((0.05 * rng.random((K // G, N)) + 0.01).astype(np.float32))
    a = orc.f32_to_h16(0.25 * rng.standard_normal((M, K)).astype(np.float32))
    qw = orc.gptq_pack_bits(q, bits)
    wide = ops.gptq_expand_bits(i32(qw), bits, add=3, axis=0)
    assert np.array_equal(wide.cpu().numpy(), orc.gptq_pack(q + 3))
    if sym:
        w = ops.gptq_repack(i32(qw), to_dev(s), None, G, bits=bits)
        ref = orc.w4a16_gemm(a, orc.gptq_pack(q + 8 - (1 << (bits - 1))), s, None, None, G=G)
    else:
        z = rng.integers(1, 1 << bits, size=(K // G, N), dtype=np.int32)          # true zero points
        stored = z if v2 else z - 1
        qz = orc.gptq_pack_bits(stored, bits, axis=1)
        assert np.array_equal(ops.gptq_expand_bits(i32(qz), bits, add=0, axis=1).cpu().numpy(), orc.pack_cols(stored))
        w = ops.gptq_repack(i32(qw), to_dev(s), i32(qz), G, bits=bits, use_v2_format=v2)
        ref = orc.w4a16_gemm(a, orc.gptq_pack(q), s, orc.pack_cols(z), None, G=G, zp_mode=orc.ZP_NSEQ)
    out = ops.gptq_gemm(to_dev(a), w, None, None, None, True, v2, bits)
    got, want = f32(to_bits(out)), f32(ref)
    assert np.allclose(got, want, rtol=1e-2, atol=1e-2)
    assert (to_bits(out) == ref).mean() > 0.97


@pytest.mark.parametrize("K,N,awq,dt", [(4096, 6144, True, torch.bfloat16), (4096, 6144, False, torch.bfloat16),
                                        (4096, 28672, False, torch.float16), (2048, 4096, True, torch.bfloat16)])
def test_w4a16_gemm_moe_sum_add_rms_norm_equals_the_three_launches(K, N, awq, dt):
    """The combine of a top-2 routed layer, the residual add, the norm and the projection as one launch: output and new
    residual hold the bits of moe_sum -> w4a16_gemm_add_rms_norm."""
    odt = orc.BF16 if dt == torch.bfloat16 else orc.FP16
    rng = np.random.default_rng(K + N)
    G = 128
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    s = orc.f32_to_h16((0.05 * rng.random((K // G, N))).astype(np.float32), odt)
    if awq:
        z = rng.integers(0, 16, size=(K // G, N), dtype=np.int32)
        w = ops.w4a16_repack(i32(orc.pack_cols(q, awq_order=True)), to_dev(s, odt), i32(orc.pack_cols(z, awq_order=True)), G,
                             _lib.W_AWQ, _lib.ZP_AWQ)
    else:
        w = ops.w4a16_repack(i32(orc.gptq_pack(q)), to_dev(s, odt), None, G)
    assert ops.w4a16_gemm_moe_sum_add_rms_norm_supported(1, 2, w)
    assert not ops.w4a16_gemm_moe_sum_add_rms_norm_supported(2, 2, w) and not ops.w4a16_gemm_moe_sum_add_rms_norm_supported(1, 3, w)
    torch.manual_seed(K)
    y = torch.randn((1, 2, K), device=DEV).to(dt)
    y[0, 0, :8] = 0.0                                     # signed zeros through the sum
    y[0, 1, :4] = -0.0
    tw = torch.tensor([[0.64, 0.36]], dtype=torch.float32, device=DEV)
    res = torch.randn((1, K), device=DEV).to(dt)
    nw = (1 + 0.1 * torch.randn(K, device=DEV)).to(dt)
    x = torch.empty((1, K), dtype=dt, device=DEV)
    ops.moe_sum(y, x, tw)
    want, want_res = ops.w4a16_gemm_add_rms_norm(x, res, nw, 1e-5, w)
    got, got_res = ops.w4a16_gemm_moe_sum_add_rms_norm(y, tw, res, nw, 1e-5, w)
    assert torch.equal(got_res.view(torch.int16), want_res.view(torch.int16))
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    with pytest.raises(RuntimeError, match="one token, two selected experts"):
        ops.w4a16_gemm_moe_sum_add_rms_norm(torch.zeros((1, 3, K), dtype=dt, device=DEV),
                                            torch.zeros((1, 3), dtype=torch.float32, device=DEV), res, nw, 1e-5, w)
