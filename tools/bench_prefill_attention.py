"""Prompt attention, partitioned kernel against the running-softmax kernel: us per call (Llama-3-8B heads).
python tools/bench_prefill_attention.py SEQS LEN [SEQS LEN ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from mi355q import ops, _lib
DEV = "cuda:0"
H, KVH, D, bs = 32, 8, 128, 16
args = [int(a) for a in sys.argv[1:]] or [1, 128, 1, 512, 1, 2048]
for S, L in zip(args[0::2], args[1::2]):
    nblk = S * ((L + bs - 1) // bs) + 1
    kc = torch.randn((nblk, KVH, D // 8, bs, 8), device=DEV).to(torch.bfloat16)
    vc = torch.randn((nblk, KVH, D, bs), device=DEV).to(torch.bfloat16)
    maxb = (L + bs - 1) // bs
    bt = torch.arange(S * maxb, dtype=torch.int32, device=DEV).reshape(S, maxb)
    q = torch.randn((S * L, H, D), device=DEV).to(torch.bfloat16)
    out = torch.empty_like(q)
    P = (L + 255) // 256
    tmp = torch.empty((S * L, H, P, D), dtype=q.dtype, device=DEV); es = torch.empty((S * L, H, P), dtype=torch.float32, device=DEV); ml = torch.empty_like(es)
    sl = torch.full((S,), L, dtype=torch.int32, device=DEV)
    qsl = torch.arange(0, S * L + 1, L, dtype=torch.int32, device=DEV)
    one = torch.ones(1, device=DEV)
    res = []
    for mn in ("100000000", "1"):
        _lib.set_option("MI355Q_PREFILL_FLASH_MIN", mn)
        _lib.set_option("MI355Q_PREFILL_FLASH_WGS", mn)
        f = lambda: ops.paged_prefill_attention(out, es, ml, tmp, q, kc, vc, KVH, D ** -0.5, bt, sl, qsl, L, bs, L, None, "auto", one, one)
        for _ in range(3): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): f()
        b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) * 100)
    print(f"{S:3d} x {L:5d} tokens: partitioned {res[0]:9.1f} us   running softmax {res[1]:9.1f} us", flush=True)
