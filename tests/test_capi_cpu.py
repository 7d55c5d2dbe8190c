"""CPU-only checks of the C-ABI library: it loads, exports every symbol the header
declares, rejects bad arguments with a message, and its host-side repack agrees with
the oracle's unpackers bit for bit.  No kernel is launched here."""
import ctypes

import numpy as np
import pytest

import oracle as orc
from mi355q import _lib


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    declared = _lib.declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/mi355q.h but not exported"
        assert name in _lib._SIGS, f"{name} has no ctypes signature"
    assert L.mi355q_version() >= 1


def _repack_host(qw, qz, K, N, G, layout, zp_mode, act, tile=64):
    L = _lib.lib()
    packed = np.zeros(K * N // 8, dtype=np.int32)
    zc = np.zeros((K // G, N), dtype=np.uint16) if zp_mode != _lib.ZP_CONST else None
    st = L.mi355q_w4a16_repack_host(qw.ctypes.data, None if qz is None else qz.ctypes.data, packed.ctypes.data,
                                    None if zc is None else zc.ctypes.data, K, N, G, tile, layout, zp_mode, act)
    _lib.check(st)
    return packed, zc


@pytest.mark.parametrize("tile", [64, 32, 16])
@pytest.mark.parametrize("layout", [_lib.W_GPTQ, _lib.W_NSEQ, _lib.W_AWQ])
def test_repack_roundtrip_bit_exact(layout, tile):
    rng = np.random.default_rng(0)
    K, N = 128, 192
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    if layout == _lib.W_GPTQ:
        src = orc.gptq_pack(q)
    else:
        src = orc.pack_cols(q, awq_order=(layout == _lib.W_AWQ))
    packed, _ = _repack_host(src, None, K, N, K, layout, _lib.ZP_CONST, _lib.BF16, tile)
    back = np.empty((K, N), dtype=np.int32)
    _lib.check(_lib.lib().mi355q_w4a16_unpack_host(packed.ctypes.data, back.ctypes.data, K, N, tile))
    assert np.array_equal(back, q)


@pytest.mark.parametrize("tile", [64, 32, 16])
def test_repack_tile_layout_is_the_documented_one(tile):
    """dword i = ks*NS + sub of lane l in tile t of a strip holds
    n = SW strip + (SW/4) ((l&15)>>2) + 4 sub + (l&3), k = 32 (KT t + ks) + 8 (l>>4) + j,
    nibble j=2p at bit 4p and j=2p+1 at bit 16+4p."""
    rng = np.random.default_rng(1)
    K, N = 256, 128
    ns = tile // 16
    kt = 4 // ns
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    packed, _ = _repack_host(orc.gptq_pack(q), None, K, N, K, _lib.W_GPTQ, _lib.ZP_CONST, _lib.BF16, tile)
    packed = packed.view(np.uint32).reshape(N // tile, K // (32 * kt), 64, 4)
    for strip, t, l, i in [(0, 0, 0, 0), (1, 1, 37, 2), (0, 1, 63, 3), (1, 0, 18, 1)]:
        sub, ks = i % ns, i // ns
        n = tile * strip + (tile // 4) * ((l & 15) >> 2) + 4 * sub + (l & 3)
        w = int(packed[strip, t, l, i])
        for j in range(8):
            k = 32 * (kt * t + ks) + 8 * (l >> 4) + j
            shift = 16 * (j & 1) + 4 * (j >> 1)
            assert (w >> shift) & 0xF == q[k, n]


def test_choose_tile():
    L = _lib.lib()
    assert L.mi355q_w4a16_choose_tile(4096, 28672, 128) == 64
    assert L.mi355q_w4a16_choose_tile(4096, 6144, 128) == 32
    assert L.mi355q_w4a16_choose_tile(4096, 4096, 128) == 16
    assert L.mi355q_w4a16_choose_tile(14336, 4096, 128) == 16
    assert L.mi355q_w4a16_choose_tile(4096, 4096, 32) == 64   # a tile may not straddle a scale group
    assert L.mi355q_w4a16_choose_tile(4096, 4096, 64) == 32


@pytest.mark.parametrize("zp_mode,awq", [(_lib.ZP_NSEQ, False), (_lib.ZP_AWQ, True), (_lib.ZP_GPTQ_V1, False)])
def test_repack_zero_points(zp_mode, awq):
    rng = np.random.default_rng(2)
    K, N, G = 128, 64, 32
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    z = rng.integers(0, 15, size=(K // G, N), dtype=np.int32)
    qz = orc.pack_cols(z, awq_order=awq)
    for act, magic in [(_lib.BF16, 0x4300), (_lib.FP16, 0x6400)]:
        _, zc = _repack_host(orc.gptq_pack(q), qz, K, N, G, _lib.W_GPTQ, zp_mode, act)
        expect = z + (1 if zp_mode == _lib.ZP_GPTQ_V1 else 0) + magic
        assert np.array_equal(zc.astype(np.int32), expect)
        # the stored 16-bit pattern is the float (bias + z)
        bias = 128.0 if act == _lib.BF16 else 1024.0
        assert np.array_equal(orc.h16_to_f32(zc, act), (expect - magic + bias).astype(np.float32))


def test_bad_arguments_return_einval_with_message():
    L = _lib.lib()
    buf = np.zeros(1024, dtype=np.int32)
    st = L.mi355q_w4a16_repack_host(buf.ctypes.data, None, buf.ctypes.data, None, 100, 64, 32, 64, 0, 0, 0)
    assert st == _lib.EINVAL
    assert b"multiple of 32" in L.mi355q_last_error()
    st = L.mi355q_w4a16_repack_host(buf.ctypes.data, None, buf.ctypes.data, None, 64, 60, 32, 64, 0, 0, 0)
    assert st == _lib.EINVAL and b"multiple of 64" in L.mi355q_last_error()
    st = L.mi355q_w4a16_repack_host(buf.ctypes.data, None, buf.ctypes.data, None, 64, 64, 32, 16, 0, 0, 0)
    assert st == _lib.EINVAL and b"tile width" in L.mi355q_last_error()
    st = L.mi355q_w4a16_repack_host(buf.ctypes.data, None, buf.ctypes.data, None, 64, 64, 32, 64, 0, _lib.ZP_AWQ, 0)
    assert st == _lib.EINVAL and b"qzeros" in L.mi355q_last_error()
    with pytest.raises(RuntimeError, match="mi355q error"):
        _lib.check(st)
    assert L.mi355q_w4a16_packed_bytes(4096, 4096) == 4096 * 4096 // 2
    assert L.mi355q_w4a16_workspace_bytes(1, 4096, 4096, 128) >= 8 * 4096 * 4


def test_ops_refuse_cpu_tensors():
    import torch
    from mi355q import ops
    x = torch.zeros(1, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.rms_norm(torch.empty_like(x), x, None, 1e-5)
