"""fp8 activation quantisers, fused norm / activation quantisers and the fp8 x fp8 scaled product (HIP,
through the C ABI) against the CPU oracle and the golden fixtures of the reference's own test oracles.
Byte outputs without a reduction in front of them are held bit-exact; behind a sum of squares the
reference's own tests allow one e4m3 step (tests/kernels/core/test_fused_quant_layernorm.py:346-360,
test_layernorm.py:209-225), and so do these.  Needs a real MI355X: run with `-m gpu`."""
import numpy as np
import pytest
import torch

import oracle as orc
from mi355q import ops

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TDT = {orc.BF16: torch.bfloat16, orc.FP16: torch.float16, orc.FP32: torch.float32}


@pytest.fixture(autouse=True)
def _kernel_route(monkeypatch):
    """These tests check this library's kernels at every row count; the library-GEMM route for many rows
    has its own test."""
    monkeypatch.setattr(ops, "FP8_MM_BLAS_MIN_TOKENS", 1 << 30)


def to_dev(a, dt=orc.BF16):
    if dt == orc.FP32:
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(TDT[dt]).to(DEV)


def bits16(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def bits8(t):
    return t.detach().cpu().contiguous().view(torch.uint8).numpy()


def fp8_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint8)).to(DEV).view(torch.float8_e4m3fn)


def host_bits(t):
    """numpy view the oracle takes: fp32 array or 16-bit patterns."""
    t = t.detach().cpu().contiguous()
    return t.numpy() if t.dtype == torch.float32 else t.view(torch.int16).numpy().view(np.uint16)


def odt(dtype):
    return {torch.bfloat16: orc.BF16, torch.float16: orc.FP16, torch.float32: orc.FP32}[dtype]


def ulps(a, b):
    def line(x):
        x = x.astype(np.int32)
        return np.where(x & 0x80, -(x & 0x7f), x & 0x7f)
    return np.abs(line(a) - line(b))


def assert_fp8_close(got, want, what, frac=0.002):
    d = ulps(got, want)
    assert d.max() <= 1 and (d > 0).mean() <= frac, (what, int(d.max()), float((d > 0).mean()))


# ---------------------------------------------------------------------------
# stand-alone quantisers: bit-exact
# ---------------------------------------------------------------------------
def test_dynamic_quantisers_golden(golden):
    """The reference's test oracles for dynamic_per_token_scaled_fp8_quant / dynamic_scaled_fp8_quant
    (tests/kernels/quant_utils.py:22-88 via tests/kernels/quantization/test_fp8_quant.py:57-108): equal scales
    and equal fp8 values."""
    g = golden("fp8_w8a8")
    for tag in g["dyn_names"]:
        x = g[tag + "_x"]
        dt = orc.FP32 if x.dtype == np.float32 else orc.BF16
        xd = to_dev(x, dt)
        ub = torch.from_numpy(g[tag + "_ub"]).to(DEV) if tag + "_ub" in g else None
        q, s = ops.scaled_fp8_quant(xd, scale_ub=ub, use_per_token_if_dynamic=True)
        assert np.array_equal(s.cpu().numpy(), g[tag + "_s_tok"]), tag
        assert np.array_equal(bits8(q), g[tag + "_q_tok"]), tag
        q, s = ops.scaled_fp8_quant(xd)
        assert np.array_equal(s.cpu().numpy(), g[tag + "_s_ten"]), tag
        assert np.array_equal(bits8(q), g[tag + "_q_ten"]), tag


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("T,H", [(1, 17), (7, 1024), (7, 1025), (3, 5137), (4096, 1026), (2, 8193), (33, 4096)])
def test_dynamic_quantisers_vs_oracle(T, H, dtype):
    """Shapes of test_fp8_quant.py:21-22 (odd widths take the scalar path), every input dtype, strided rows."""
    torch.manual_seed(T * 131 + H)
    wide = (torch.randn(T, H + 24, dtype=torch.float32) * 3).to(dtype).to(DEV)
    x = wide[:, 8:8 + H] if H % 8 == 0 else wide[:, :H]         # row stride != H; 16-byte aligned only for even cases
    xh = host_bits(x)
    ub = x.float().abs().mean().reshape(1)
    for use_ub in (False, True):
        q, s = ops.scaled_fp8_quant(x, scale_ub=ub if use_ub else None, use_per_token_if_dynamic=True)
        wq, ws = orc.dynamic_per_token_scaled_fp8_quant(xh, float(ub) if use_ub else None, odt(dtype))
        assert np.array_equal(s.cpu().numpy(), ws) and np.array_equal(bits8(q), wq)
    q, s = ops.scaled_fp8_quant(x)
    wq, ws = orc.dynamic_scaled_fp8_quant(xh, odt(dtype))
    assert np.array_equal(s.cpu().numpy(), ws) and np.array_equal(bits8(q), wq)


def test_static_quantiser_golden_and_oracle(golden):
    """Group shapes of test_fp8_quant.py:136-146: bit-exact against the oracle, within one e4m3 step of
    quant_utils.scaled_quantize (which divides where the kernel multiplies by the reciprocal; the reference
    test allows rtol 0.12 for it); 1-D scales with an explicit group shape read the same values."""
    g = golden("fp8_w8a8")
    for tag in g["static_names"]:
        x = g[tag + "_x"]
        dt = orc.FP32 if x.dtype == np.float32 else orc.BF16
        gs, scale = tuple(int(v) for v in g[tag + "_gs"]), g[tag + "_scale"]
        xd, sd = to_dev(x, dt), torch.from_numpy(scale).to(DEV)
        q, s_out = ops.scaled_fp8_quant(xd, scale=sd, group_shape=gs)
        assert s_out is sd
        assert np.array_equal(bits8(q), orc.static_scaled_fp8_quant(x, scale, gs, dt)), tag
        assert_fp8_close(bits8(q), g[tag + "_q"], tag, frac=0.02)
        if scale.shape[0] == 1 or scale.shape[1] == 1:
            q1, _ = ops.scaled_fp8_quant(xd, scale=sd.reshape(-1), group_shape=gs)
            assert torch.equal(q1.view(torch.uint8), q.view(torch.uint8)), tag


def test_static_quantiser_errors():
    x = torch.zeros(4, 64, dtype=torch.bfloat16, device=DEV)
    s = torch.ones(4, dtype=torch.float32, device=DEV)
    with pytest.raises(RuntimeError, match="1D scale requires explicit group_shape"):
        ops.scaled_fp8_quant(x, scale=s)
    with pytest.raises(RuntimeError, match="does not match"):
        ops.scaled_fp8_quant(x, scale=s, group_shape=(2, -1))
    with pytest.raises(RuntimeError, match="does not match inferred"):
        ops.scaled_fp8_quant(x, scale=torch.ones(4, 2, dtype=torch.float32, device=DEV), group_shape=(1, 16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.scaled_fp8_quant(x.cpu())
    q, sc = ops.scaled_fp8_quant(x, num_token_padding=16)       # padded rows exist, are not written
    assert q.shape == (16, 64) and sc.shape == (1,)


def test_saturation_and_special_values():
    """Values beyond +-448 * scale saturate; a NaN saturates to +448 (fminf drops it), as in
    csrc/quantization/w8a8/fp8/common.cuh:57-77."""
    x = torch.tensor([[1e6, -1e6, float("nan"), 0.0, -0.0, 448.0, 464.0, 1e-9]], dtype=torch.float32, device=DEV)
    one = torch.ones(1, dtype=torch.float32, device=DEV)
    q, _ = ops.scaled_fp8_quant(x, scale=one)
    assert np.array_equal(bits8(q), orc.static_scaled_fp8_quant(x.cpu().numpy(), np.ones(1, np.float32), None, orc.FP32))
    assert bits8(q)[0, :3].tolist() == [0x7e, 0xfe, 0x7e]


# ---------------------------------------------------------------------------
# fused norm / activation quantisers
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [orc.BF16, orc.FP16])
@pytest.mark.parametrize("T,H", [(1, 64), (7, 769), (83, 4096), (300, 5120), (2, 8192)])
@pytest.mark.parametrize("add_residual", [False, True])
def test_rms_norm_static_fp8_quant(T, H, add_residual, dt):
    """Shapes and scales of tests/kernels/core/test_layernorm.py:134-230: against the oracle, and against the
    unfused sequence (fused_add_)rms_norm -> static_scaled_fp8_quant with that test's outlier budget."""
    torch.manual_seed(H + T)
    dtype = TDT[dt]
    w = torch.empty(H, dtype=dtype).normal_(mean=1.0, std=0.1).to(DEV)
    xb = (torch.randn(T, 2 * H, dtype=dtype) * (1 / (2 * H))).to(DEV)
    x = xb[:, :H]                                                    # strided rows
    res = (torch.randn(T, H, dtype=dtype) * (1 / (2 * H))).to(DEV) if add_residual else None
    for qs in (0.01, 1.0, 10.0):
        scale = torch.tensor([qs * (1.0 / (2 * H)) if qs != 1.0 else 1.0 / 64], dtype=torch.float32, device=DEV)
        out = torch.empty(T, H, dtype=torch.float8_e4m3fn, device=DEV)
        r_f = res.clone() if add_residual else None
        if add_residual:
            ops.fused_add_rms_norm_static_fp8_quant(out, x, r_f, w, scale, 1e-6)
        else:
            ops.rms_norm_static_fp8_quant(out, x, w, scale, 1e-6)
        wq, wres = orc.rms_norm_static_fp8_quant(bits16(x), bits16(w), float(scale), 1e-6,
                                                 bits16(res) if add_residual else None, dt)
        assert_fp8_close(bits8(out), wq, ("oracle", qs))
        # unfused sequence of HIP kernels
        if add_residual:
            xu, r_u = x.contiguous().clone(), res.clone()
            ops.fused_add_rms_norm(xu, r_u, w, 1e-6)
            assert torch.equal(r_u.view(torch.int16), r_f.view(torch.int16))
            assert np.array_equal(bits16(r_f), wres)
        else:
            xu = torch.empty(T, H, dtype=dtype, device=DEV)
            ops.rms_norm(xu, x, w, 1e-6)
        q_u, _ = ops.scaled_fp8_quant(xu, scale=scale)
        d = ulps(bits8(out), bits8(q_u))
        assert int((d > 0).sum()) <= d.size // 100_000 + 8 and d.max() <= 1, (qs, int((d > 0).sum()))


def test_rms_norm_dynamic_per_token_quant_golden(golden):
    """The reference test's composite (native RMSNorm, then the per-token test oracle) at its tolerances:
    scales rtol 1e-5, fp8 within one step (test_fused_quant_layernorm.py:346-360); residual bit-exact."""
    g = golden("fp8_w8a8")
    for tag in g["normq_names"]:
        x, w = to_dev(g[tag + "_x"]), to_dev(g[tag + "_w"])
        ub = torch.from_numpy(g[tag + "_ub"]).to(DEV) if tag + "_ub" in g else None
        res = to_dev(g[tag + "_res"]) if tag + "_res" in g else None
        q, s = ops.rms_norm_dynamic_per_token_quant(x, w, 1e-6, scale_ub=ub, residual=res)
        np.testing.assert_allclose(s.cpu().numpy(), g[tag + "_s"], rtol=1e-5, atol=1e-8, err_msg=tag)
        assert_fp8_close(bits8(q), g[tag + "_q"], tag, frac=0.05)
        if res is not None:
            assert np.array_equal(bits16(res), g[tag + "_new_res"]), tag


@pytest.mark.parametrize("dt", [orc.BF16, orc.FP16])
@pytest.mark.parametrize("T,H", [(1, 128), (1, 1025), (5, 5137), (2048, 1024), (3, 8192)])
@pytest.mark.parametrize("add_residual", [False, True])
def test_rms_norm_dynamic_per_token_quant_vs_oracle(T, H, add_residual, dt):
    torch.manual_seed(H * 3 + T)
    dtype = TDT[dt]
    w = torch.empty(H, dtype=dtype).normal_(mean=1.0, std=0.1).to(DEV)
    xb = (torch.randn(T, 2 * H, dtype=dtype) * (1 / H)).to(DEV)
    x = xb[:, :H]
    res = (torch.randn(T, H, dtype=dtype) * (1 / H)).to(DEV) if add_residual else None
    for use_ub in (False, True):
        r = res.clone() if add_residual else None
        ub = torch.tensor([0.5], dtype=torch.float32, device=DEV) if use_ub else None
        q, s = ops.rms_norm_dynamic_per_token_quant(x, w, 1e-6, scale_ub=ub, residual=r)
        wq, ws, wres = orc.rms_norm_dynamic_per_token_quant(bits16(x), bits16(w), 1e-6, 0.5 if use_ub else None,
                                                            bits16(res) if add_residual else None, dt)
        np.testing.assert_allclose(s.cpu().numpy(), ws, rtol=1e-5, atol=1e-8)
        assert_fp8_close(bits8(q), wq, (T, H, use_ub), frac=0.02)
        if add_residual:
            assert np.array_equal(bits16(r), wres)


@pytest.mark.parametrize("dt", [orc.BF16, orc.FP16])
@pytest.mark.parametrize("T,d", [(1, 64), (7, 1023), (83, 2048), (2, 14336), (2048, 512)])
def test_silu_and_mul_quant(T, d, dt):
    """expf differs in the last place between device and host; after two roundings to the activation dtype
    almost every byte agrees, none is further than one step."""
    torch.manual_seed(d + T)
    x = torch.randn(T, 2 * d, dtype=TDT[dt]).to(DEV)
    scale = torch.tensor([0.05], dtype=torch.float32, device=DEV)
    out = torch.empty(T, d, dtype=torch.float8_e4m3fn, device=DEV)
    ops.silu_and_mul_quant(out, x, scale)
    assert_fp8_close(bits8(out), orc.silu_and_mul_quant(bits16(x), 0.05, dt), (T, d), frac=0.001)
    # the unfused sequence silu_and_mul -> static_scaled_fp8_quant (whose exponential is the fast one) agrees
    # to the same bar
    y = torch.empty(T, d, dtype=TDT[dt], device=DEV)
    ops.silu_and_mul(y, x)
    q_u, _ = ops.scaled_fp8_quant(y, scale=scale)
    assert_fp8_close(bits8(q_u), bits8(out), ("unfused", T, d), frac=0.001)


# ---------------------------------------------------------------------------
# scaled product
# ---------------------------------------------------------------------------
def test_fp8_scaled_mm_golden(golden):
    """torch._scaled_mm outputs on the inputs of test_rocm_skinny_gemms.py:250-279 (xnorm branch: atol 1e-3)."""
    g = golden("fp8_w8a8")
    for tag in g["mm_names"]:
        dt = int(g[tag + "_dt"][0])
        bias = to_dev(g[tag + "_bias"], dt) if tag + "_bias" in g else None
        sa, sb = torch.from_numpy(g[tag + "_sa"]).to(DEV), torch.from_numpy(g[tag + "_sb"]).to(DEV)
        a, b = fp8_dev(g[tag + "_a"]), fp8_dev(g[tag + "_b"])
        out = ops.fp8_scaled_mm(a, b, TDT[dt], sa, sb, bias)
        np.testing.assert_allclose(orc.h16_to_f32(bits16(out), dt), orc.h16_to_f32(g[tag + "_ref"], dt), atol=1e-3,
                                   rtol=1e-2, err_msg=tag)
        out2 = ops.wvSplitKQ(b, a, TDT[dt], sa, sb, 256, bias)
        assert torch.equal(out2.view(torch.int16), out.view(torch.int16)), tag


def _mm_case(M, N, K, dt, per_token, per_channel, biased, seed=0):
    g = torch.Generator().manual_seed(seed + M * 7 + N * 3 + K)
    xav = (2.0 / K) ** 0.5
    a = ((torch.rand(M, K, generator=g) * 2 - 1) * xav * 200).clamp(-448, 448).to(torch.float8_e4m3fn)
    w = ((torch.rand(N, K, generator=g) * 2 - 1) * xav * 200).clamp(-448, 448).to(torch.float8_e4m3fn)
    sa = (torch.rand(M if per_token else 1, generator=g) * 0.01 + 0.002).float()
    sb = (torch.rand(N if per_channel else 1, generator=g) * 0.01 + 0.002).float()
    bias = (torch.rand(N, generator=g) * 2 - 1).to(TDT[dt]) if biased else None
    want = orc.fp8_scaled_mm(bits8(a), bits8(w), sa.numpy(), sb.numpy(), None if bias is None else bits16(bias), dt)
    got = ops.fp8_scaled_mm(a.to(DEV), w.to(DEV), TDT[dt], sa.to(DEV), sb.to(DEV), None if bias is None else bias.to(DEV))
    return orc.h16_to_f32(bits16(got), dt), orc.h16_to_f32(want, dt)


@pytest.mark.parametrize("M", [1, 2, 4, 5, 16, 17, 32, 33, 64])
@pytest.mark.parametrize("N,K", [(16, 16), (80, 80), (4096, 4096), (6144, 4096), (4112, 2064), (28672, 4096), (4096, 14336)])
def test_fp8_scaled_mm_decode_rows_vs_oracle(M, N, K):
    """Up to 64 rows (the weight-streaming kernel: every R x T instantiation), ragged N and K
    (test_rocm_skinny_gemms.py:76-107 has K, N = 16 n + 16), per-tensor scales as the ROCm kernel takes
    them.  Exact products, fp32 sums: 2e-3 of the output scale."""
    if N * K > 30_000_000 and M not in (1, 16, 64):
        pytest.skip("large shapes at three row counts")
    got, want = _mm_case(M, N, K, orc.BF16, False, False, M % 2 == 0)
    np.testing.assert_allclose(got, want, rtol=1e-2, atol=2e-3 * max(1.0, float(np.abs(want).max())))


@pytest.mark.parametrize("dt", [orc.BF16, orc.FP16])
@pytest.mark.parametrize("M,N,K", [(65, 128, 128), (128, 4096, 4096), (200, 1040, 528), (513, 6144, 4096), (2048, 3584, 4096),
                                   (1024, 4096, 14336), (300, 4112, 2064)])
def test_fp8_scaled_mm_prompt_rows_vs_oracle(M, N, K, dt):
    """The tiled kernel: partial tiles in M and N, a K tail that is not a multiple of the 128-byte step,
    per-token and per-channel scales, bias."""
    if dt == orc.FP16 and M * N > 3_000_000:
        pytest.skip("large shapes in one dtype")
    got, want = _mm_case(M, N, K, dt, True, True, True)
    np.testing.assert_allclose(got, want, rtol=1e-2, atol=2e-3 * max(1.0, float(np.abs(want).max())))


def test_fp8_scaled_mm_many_rows_route(monkeypatch):
    """Per-tensor scales and 128+ rows go to the library GEMM (the reference's own call above four rows):
    same oracle, same tolerance, and the same values as the tiled kernel to that tolerance."""
    monkeypatch.setattr(ops, "FP8_MM_BLAS_MIN_TOKENS", 128)
    got, want = _mm_case(256, 1024, 512, orc.BF16, False, False, True)
    np.testing.assert_allclose(got, want, rtol=1e-2, atol=2e-3 * max(1.0, float(np.abs(want).max())))
    got, want = _mm_case(256, 1024, 512, orc.BF16, True, True, True)      # row scales stay on this library's kernel
    np.testing.assert_allclose(got, want, rtol=1e-2, atol=2e-3 * max(1.0, float(np.abs(want).max())))


def test_fp8_scaled_mm_identity_and_layout():
    """Asymmetric exact-integer operands: catches a transposed output or a permuted K inside the MFMA
    operands, which random data with a tolerance could blur."""
    for M, N, K in [(3, 48, 64), (40, 160, 192), (130, 256, 384)]:
        a = torch.zeros(M, K)
        w = torch.zeros(N, K)
        for m in range(M):
            a[m, (5 * m + 1) % K] = 1.0 + (m % 3)
        for n in range(N):
            w[n, :] = torch.arange(K) % 7 - 3
            w[n, n % K] += 1 + n % 2
        want = (a @ w.t()).to(torch.bfloat16)
        one = torch.ones(1, dtype=torch.float32, device=DEV)
        got = ops.fp8_scaled_mm(a.to(torch.float8_e4m3fn).to(DEV), w.to(torch.float8_e4m3fn).to(DEV), torch.bfloat16, one, one)
        assert torch.equal(got.cpu(), want), (M, N, K)


def test_fp8_scaled_mm_errors():
    a = torch.zeros(4, 24, device=DEV).to(torch.float8_e4m3fn)
    w = torch.zeros(8, 24, device=DEV).to(torch.float8_e4m3fn)
    one = torch.ones(1, dtype=torch.float32, device=DEV)
    with pytest.raises(RuntimeError, match="multiples of 16"):
        ops.fp8_scaled_mm(a, w, torch.bfloat16, one, one)
    a, w = a[:, :16].contiguous(), w[:, :16].contiguous()
    with pytest.raises(RuntimeError, match="scale_a has 3 elements"):
        ops.fp8_scaled_mm(a, w, torch.bfloat16, torch.ones(3, dtype=torch.float32, device=DEV), one)
    with pytest.raises(RuntimeError, match="float8_e4m3fn"):
        ops.fp8_scaled_mm(a.view(torch.uint8), w, torch.bfloat16, one, one)
    assert ops.fp8_scaled_mm(a[:0], w, torch.bfloat16, one, one).shape == (0, 8)


# ---------------------------------------------------------------------------
# block-scaled product: per-token-group activation scales x per-block weight scales
# ---------------------------------------------------------------------------
def test_per_token_group_quant_golden(golden):
    """The reference test's restatement (tests/kernels/quant_utils.py:157-180) on its inputs: bytes and scales bit
    for bit -- a maximum is the only reduction in front of them (the reference's own ROCm bar allows one e4m3 step on
    0.05 % of the elements against its Triton kernel, test_block_fp8.py:96-110)."""
    g = golden("fp8_w8a8")
    for tag in g["ptg_names"]:
        gs = int(g[tag + "_gs"][0])
        q, sc = ops.per_token_group_quant_fp8(to_dev(g[tag + "_x"]), gs)
        assert np.array_equal(sc.cpu().numpy(), g[tag + "_s"]), tag
        assert np.array_equal(bits8(q), g[tag + "_q"]), tag
        q2, sc2 = ops.per_token_group_quant_fp8(to_dev(g[tag + "_x"]), gs, column_major_scales=True)
        assert sc2.stride(-2) == 1 and sc2.shape == sc.shape, tag
        assert torch.equal(sc2.contiguous(), sc) and torch.equal(q2.view(torch.uint8), q.view(torch.uint8)), tag


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("T,H,gs", [(1, 128, 128), (7, 4096, 128), (2050, 512, 64), (5, 13824, 512), (3, 1024, 32), (9, 960, 24)])
def test_per_token_group_quant_vs_oracle(T, H, gs, dtype):
    """Every input dtype, group sizes of test_block_fp8.py:46-48 and two that take the element-wise form, an
    all-zero group (scale = eps / 448), a row stride wider than the row."""
    gen = torch.Generator().manual_seed(T + H + gs)
    x = ((torch.rand(T, H + 64, generator=gen) - 0.3) * 8).to(dtype)
    x[0, :gs] = 0
    xs = x[:, :H]
    want_q, want_s = orc.per_token_group_quant_fp8(host_bits(xs.contiguous()), gs, dt=odt(dtype))
    q, sc = ops.per_token_group_quant_fp8(x.to(DEV)[:, :H], gs)
    assert np.array_equal(sc.cpu().numpy(), want_s)
    assert np.array_equal(bits8(q), want_q)
    assert float(sc[0, 0]) == np.float32(1e-10) / np.float32(448.0)


def test_block_scaled_mm_golden(golden):
    """native_w8a8_block_matmul outputs (tests/kernels/quant_utils.py:91-154) on the inputs of test_block_fp8.py:125-153,
    at that test's bar: mean |diff| / mean |ref| < 1e-3."""
    g = golden("fp8_w8a8")
    for tag in g["blockmm_names"]:
        a, b = fp8_dev(g[tag + "_a"]), fp8_dev(g[tag + "_b"])
        As, Bs = torch.from_numpy(g[tag + "_as"]).to(DEV), torch.from_numpy(g[tag + "_bs"]).to(DEV)
        out = ops.w8a8_block_scaled_mm(a, b, As, Bs, [128, 128], torch.bfloat16)
        got, ref = orc.h16_to_f32(bits16(out)), orc.h16_to_f32(g[tag + "_ref"])
        assert np.abs(got - ref).mean() / np.abs(ref).mean() < 1e-3, tag
        np.testing.assert_allclose(got, ref, rtol=2e-2, atol=2e-2 * float(np.abs(ref).max()), err_msg=tag)


def _block_case(M, N, K, dt, bn=128, biased=False, seed=0):
    g = torch.Generator().manual_seed(seed + M * 7 + N * 3 + K)
    a = ((torch.rand(M, K, generator=g) - 0.5) * 2 * 448).clamp(-448, 448).to(torch.float8_e4m3fn)
    w = ((torch.rand(N, K, generator=g) - 0.5) * 2 * 448).clamp(-448, 448).to(torch.float8_e4m3fn)
    kt, nt = -(-K // 128), -(-N // bn)
    As = (torch.rand(M, kt, generator=g) * 1e-2).float()
    Bs = (torch.rand(nt, kt, generator=g) * 1e-2).float()
    bias = (torch.rand(N, generator=g) * 2 - 1).to(TDT[dt]) if biased else None
    want = orc.h16_to_f32(orc.w8a8_block_scaled_mm(bits8(a), bits8(w), As.numpy(), Bs.numpy(), [bn, 128], dt,
                                                   None if bias is None else bits16(bias)), dt)
    got = ops.w8a8_block_scaled_mm(a.to(DEV), w.to(DEV), As.to(DEV), Bs.to(DEV), [bn, 128], TDT[dt],
                                   None if bias is None else bias.to(DEV))
    return orc.h16_to_f32(bits16(got), dt), want


@pytest.mark.parametrize("M", [1, 7, 8, 16, 17, 33, 64])
@pytest.mark.parametrize("N,K", [(128, 256), (512, 4096), (576, 3884), (7168, 4096), (4608, 7168), (1040, 528)])
def test_block_scaled_mm_decode_rows_vs_oracle(M, N, K):
    """Up to 64 rows on the weight-streaming kernel (every R x T instantiation): the shapes of test_block_fp8.py:50-52
    (K = 3884 goes through zero-padded copies), a ragged last block in N and K."""
    if N * K > 20_000_000 and M not in (1, 16, 64):
        pytest.skip("large shapes at three row counts")
    got, want = _block_case(M, N, K, orc.BF16)
    assert np.abs(got - want).mean() / np.abs(want).mean() < 1e-3
    np.testing.assert_allclose(got, want, rtol=2e-2, atol=1e-2 * float(np.abs(want).max()))


@pytest.mark.parametrize("dt", [orc.BF16, orc.FP16])
@pytest.mark.parametrize("M,N,K,bn", [(65, 128, 128, 128), (83, 576, 4096, 128), (200, 1040, 528, 128), (513, 7168, 4096, 128),
                                      (4096, 512, 3888, 128), (130, 512, 384, 64), (300, 4608, 7168, 128)])
def test_block_scaled_mm_prompt_rows_vs_oracle(M, N, K, bn, dt):
    """The tiled kernel: partial tiles in M and N, a ragged last scale block, 64-channel weight blocks, bias."""
    if dt == orc.FP16 and M * N > 1_000_000:
        pytest.skip("large shapes in one dtype")
    got, want = _block_case(M, N, K, dt, bn=bn, biased=(M % 2 == 0))
    assert np.abs(got - want).mean() / np.abs(want).mean() < 1e-3
    np.testing.assert_allclose(got, want, rtol=2e-2, atol=1e-2 * float(np.abs(want).max()))


def test_block_scaled_mm_equals_dequantised_product():
    """Quantise a bf16 activation and weight the way a block-quantised checkpoint and W8A8BlockFp8LinearOp do, and
    compare the product with the fp32 product of the dequantised operands (fp8_utils.py:886-910's upcast form)."""
    g = torch.Generator().manual_seed(5)
    M, N, K = 5, 512, 1024
    x = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16)
    w = torch.randn(N, K, generator=g) * 0.05
    wb = w.view(N // 128, 128, K // 128, 128)
    ws = wb.abs().amax(dim=(1, 3)) / 448.0
    wq = (wb / ws[:, None, :, None]).clamp(-448, 448).to(torch.float8_e4m3fn).view(N, K)
    xq, xs = ops.per_token_group_quant_fp8(x.to(DEV), 128)
    out = ops.w8a8_block_scaled_mm(xq, wq.to(DEV), xs, ws.to(DEV), [128, 128], torch.bfloat16)
    deq_x = xq.float().cpu() * xs.cpu().repeat_interleave(128, dim=1)
    deq_w = wq.float() * ws.repeat_interleave(128, dim=0).repeat_interleave(128, dim=1)
    ref = deq_x @ deq_w.t()
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), rtol=1e-2, atol=1e-2 * float(ref.abs().max()))
    # and the quantisation error itself stays where fp8 puts it
    exact = x.float() @ w.t()
    assert float((out.float().cpu() - exact).abs().mean() / exact.abs().mean()) < 0.05


def test_block_scaled_mm_errors():
    a = torch.zeros((4, 256), dtype=torch.float8_e4m3fn, device=DEV)
    w = torch.zeros((128, 256), dtype=torch.float8_e4m3fn, device=DEV)
    As, Bs = torch.ones((4, 2), device=DEV), torch.ones((1, 2), device=DEV)
    with pytest.raises(RuntimeError, match="block_k"):
        ops.w8a8_block_scaled_mm(a, w, torch.ones((4, 4), device=DEV), torch.ones((1, 4), device=DEV), [128, 64])
    with pytest.raises(RuntimeError, match="scale shapes"):
        ops.w8a8_block_scaled_mm(a, w, As, torch.ones((2, 2), device=DEV), [128, 128])
    with pytest.raises(RuntimeError, match="divisible"):
        ops.per_token_group_quant_fp8(torch.zeros((2, 100), dtype=torch.bfloat16, device=DEV), 64)
    assert ops.w8a8_block_scaled_mm(a, w, As, Bs, [128, 128]).shape == (4, 128)
