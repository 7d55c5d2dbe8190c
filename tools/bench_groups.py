"""Decode GEMV (M = 1) of the 8B layer shapes at scale-group sizes 128, 64 and 32."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd")); sys.path.insert(0, ROOT)
import torch
from mi355q import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
shapes = [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096)]
for G in (128, 64, 32):
    line, tot = [], 0.0
    for name, K, N in shapes:
        ws = []
        for _ in range(max(2, int(600e6 // (K * N // 2)))):
            qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev, generator=g)
            sc = (torch.rand((K // G, N), device=dev, generator=g) * 0.05).to(torch.bfloat16)
            ws.append(ops.w4a16_repack(qw, sc, None, G))
        x = (torch.randn((1, K), device=dev, generator=g) * 0.25).to(torch.bfloat16)
        for w in ws: ops.w4a16_gemm(x, w)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(4):
                for w in ws: ops.w4a16_gemm(x, w)
        gr.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / (4 * len(ws)); tot += us
        line.append(f"{name} tile {ws[0].tile}: {us:5.1f}")
        del ws
    print(f"G={G}: " + " | ".join(line) + f" | sum {tot:.1f} us", flush=True)
