"""Mixtral-8x7B-shaped routed-expert layer at decode (T tokens, top-2 of 8 experts, int4 g128): us per layer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd")); sys.path.insert(0, ROOT)
import torch
from mi355q import ops
from mi355q.plugin.moe import fused_experts
dev = "cuda:0"
E, H, I, G, topk = 8, 4096, 14336, 128, 2
g = torch.Generator(device=dev).manual_seed(0)
layers = []
for _ in range(3):    # 3 layers x 8 experts x 88 MB: nothing stays cache-resident
    qw13 = torch.randint(0, 256, (E, 2 * I, H // 2), dtype=torch.uint8, device=dev, generator=g)
    s13 = (torch.rand((E, 2 * I, H // G), device=dev, generator=g) * 0.02).to(torch.bfloat16)
    qw2 = torch.randint(0, 256, (E, H, I // 2), dtype=torch.uint8, device=dev, generator=g)
    s2 = (torch.rand((E, H, I // G), device=dev, generator=g) * 0.02).to(torch.bfloat16)
    layers.append((ops.moe_w4a16_repack(qw13, s13, None, G, pair_halves=True), ops.moe_w4a16_repack(qw2, s2, None, G)))
    del qw13, qw2
for T in [int(a) for a in sys.argv[1:]] or [1, 4]:
    x = (torch.randn((T, H), device=dev, generator=g) * 0.25).to(torch.bfloat16)
    logits = torch.randn((T, E), device=dev, generator=g)
    tw = torch.empty((T, topk), dtype=torch.float32, device=dev); ti = torch.empty((T, topk), dtype=torch.int32, device=dev)
    def step():
        for w13, w2 in layers:
            ops.topk_softmax(tw, ti, None, logits, True)
            fused_experts(x, w13, w2, tw, ti)
    step(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(4): step()
    gr.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / (4 * len(layers))
    nbytes = min(T * topk, E) * (2 * I * H // 2 + H * I // 2) * 1.03
    print(f"T={T}: {us:7.1f} us per MoE layer; <= {nbytes/1e6:.0f} MB of expert weights -> {nbytes/us/1e3:.0f} GB/s", flush=True)
