"""Mixtral-8x7B-shaped routed-expert layer on prompt batches (T tokens, top-2 of 8 experts, int4 g128):
the token-sorted grouped route against the per-expert host loop (dequantise + dense GEMM), us per layer
and PFLOP/s of the 2 * 2 * T * (3 H I) useful flops."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd")); sys.path.insert(0, ROOT)
import torch
from mi355q import ops
from mi355q.plugin import moe as moe_mod
dev = "cuda:0"
E, H, I, G, topk = 8, 4096, 14336, 128, 2
g = torch.Generator(device=dev).manual_seed(0)
qw13 = torch.randint(0, 256, (E, 2 * I, H // 2), dtype=torch.uint8, device=dev, generator=g)
s13 = (torch.rand((E, 2 * I, H // G), device=dev, generator=g) * 0.02).to(torch.bfloat16)
qw2 = torch.randint(0, 256, (E, H, I // 2), dtype=torch.uint8, device=dev, generator=g)
s2 = (torch.rand((E, H, I // G), device=dev, generator=g) * 0.02).to(torch.bfloat16)
w13, w2 = ops.moe_w4a16_repack(qw13, s13, None, G, pair_halves=True), ops.moe_w4a16_repack(qw2, s2, None, G)
del qw13, qw2
for T in [int(a) for a in sys.argv[1:]] or [128, 512, 2048]:
    x = (torch.randn((T, H), device=dev, generator=g) * 0.25).to(torch.bfloat16)
    logits = torch.randn((T, E), device=dev, generator=g)
    tw = torch.empty((T, topk), dtype=torch.float32, device=dev); ti = torch.empty((T, topk), dtype=torch.int32, device=dev)
    ops.topk_softmax(tw, ti, None, logits, True)
    res = {}
    for route in ("grouped", "grouped_host", "all_experts"):
        if route == "all_experts" and T > 512:
            continue
        moe_mod.FORCE_ROUTE = route
        for _ in range(2): moe_mod.fused_experts(x, w13, w2, tw, ti)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): moe_mod.fused_experts(x, w13, w2, tw, ti)
        b.record(); torch.cuda.synchronize()
        res[route] = a.elapsed_time(b) * 1e3 / 5
    fl = 2.0 * T * topk * 3 * H * I
    print(f"T={T}: " + "  ".join(f"{r} {us:9.1f} us ({fl/us/1e9:5.3f} PF)" for r, us in res.items()), flush=True)
