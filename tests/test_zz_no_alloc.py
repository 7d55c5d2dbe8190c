"""The entry point that allocates nothing (mi355q_paged_attention_noalloc) against the entry point it
is a variant of (mi355q_paged_attention with caller-allocated scratch): same kernels, so the outputs
must be bit-identical.  Two calls in a row, the second with a smaller input, because every call on a
GPU shares one scratch array that keeps the previous call's contents.  Collected last on purpose."""
import numpy as np
import pytest
import torch

import oracle as orc
from mi355q import _lib, ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def _dev16(bits):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(torch.bfloat16).to(DEV)


def _case(rng, S, H, KVH, D, bs, lens, fp8, nblk=96):
    x = 16 if fp8 else 8
    q = orc.f32_to_h16(rng.uniform(-1, 1, (S, H, D)).astype(np.float32))
    kf = rng.uniform(-1, 1, (nblk, KVH, D // x, bs, x)).astype(np.float32)
    vf = rng.uniform(-1, 1, (nblk, KVH, D, bs)).astype(np.float32)
    ks, vs = (1 / 64, 1 / 32) if fp8 else (1.0, 1.0)
    if fp8:
        kc, vc = orc.f32_to_e4m3(kf / ks), orc.f32_to_e4m3(vf / vs)
        kcd, vcd = torch.from_numpy(kc).to(DEV), torch.from_numpy(vc).to(DEV)
    else:
        kc, vc = orc.f32_to_h16(kf), orc.f32_to_h16(vf)
        kcd, vcd = _dev16(kc), _dev16(vc)
    max_len = int(max(lens))
    bt = rng.integers(0, nblk, size=(S, (max_len + bs - 1) // bs)).astype(np.int32)
    sl = np.array(lens, dtype=np.int32)
    return dict(q=q, kc=kc, vc=vc, bt=bt, sl=sl, ks=ks, vs=vs, qd=_dev16(q), kcd=kcd, vcd=vcd, max_len=max_len,
                btd=torch.from_numpy(bt).to(DEV), sld=torch.from_numpy(sl).to(DEV), H=H, KVH=KVH, D=D, bs=bs, fp8=fp8,
                kst=torch.tensor(ks, dtype=torch.float32, device=DEV), vst=torch.tensor(vs, dtype=torch.float32, device=DEV))


def _both(c):
    S, H, D = c["qd"].shape
    P = (c["max_len"] + 255) // 256
    scale = 1.0 / D ** 0.5
    kv = "fp8" if c["fp8"] else "auto"
    want = torch.full_like(c["qd"], float("nan"))
    tmp = torch.empty((S, H, P, D), dtype=want.dtype, device=DEV)
    es = torch.empty((S, H, P), dtype=torch.float32, device=DEV)
    ops.paged_attention_rocm(want, es, torch.empty_like(es), tmp, c["qd"], c["kcd"], c["vcd"], c["KVH"], scale, c["btd"],
                             c["sld"], None, c["bs"], c["max_len"], None, kv, c["kst"], c["vst"])
    got = torch.full_like(c["qd"], float("nan"))
    ops.paged_attention_noalloc(got, c["qd"], c["kcd"], c["vcd"], c["KVH"], scale, c["btd"], c["sld"], None, c["bs"],
                                c["max_len"], None, kv, c["kst"], c["vst"])
    torch.cuda.synchronize()
    return got, want, scale


@pytest.mark.parametrize("fp8", [False, True])
def test_noalloc_attention_equals_the_caller_scratch_entry_point(fp8):
    rng = np.random.default_rng(11 + fp8)
    # first call: long contexts (8 partitions); second: fewer, shorter sequences over the same scratch
    big = _case(rng, 7, 32, 8, 128, 16, [1, 255, 256, 257, 1000, 2000, 2048], fp8)
    small = _case(rng, 3, 32, 8, 128, 16, [300, 17, 513], fp8)
    for c in (big, small):
        got, want, scale = _both(c)
        assert not torch.isnan(got).any()
        assert np.array_equal(_bits(got), _bits(want))            # same kernels, same scratch layout: bit-exact
        ref = orc.paged_attention(c["q"], c["kc"], c["vc"], c["bt"], c["sl"], scale, fp8=fp8, k_scale=c["ks"], v_scale=c["vs"])
        np.testing.assert_allclose(orc.h16_to_f32(_bits(got), orc.BF16), orc.h16_to_f32(ref, orc.BF16), rtol=1e-5,
                                   atol=1e-2 if fp8 else 2e-3)


def test_noalloc_attention_refuses_what_its_scratch_cannot_hold():
    rng = np.random.default_rng(3)
    c = _case(rng, 5, 32, 8, 128, 16, [100, 200, 300, 400, 500], False)
    out = torch.empty_like(c["qd"])
    with pytest.raises(RuntimeError, match="exceeds the built-in scratch"):
        # 5 sequences x 32 heads x 128 partitions (a 32768-token context limit) = 20480 > 16384
        ops.paged_attention_noalloc(out, c["qd"], c["kcd"], c["vcd"], 8, 0.1, c["btd"], c["sld"], None, 16, 32768, None,
                                    "auto", c["kst"], c["vst"])
