"""Decode GEMV of 8-bit weight-only layers (fp8 and GPTQ 8-bit) on the 8B shapes: us and GB/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd")); sys.path.insert(0, ROOT)
import torch
from mi355q import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for M in (1, 4):
    for name, K, N in [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096)]:
        n = max(2, int(700e6 // (K * N)))
        ws = [torch.randint(0, 256, (N, K), dtype=torch.uint8, device=dev, generator=g) for _ in range(n)]
        sg = torch.rand((N, K // 128), device=dev, generator=g) * 0.004
        sc = torch.rand((N,), device=dev, generator=g) * 0.004
        x = (torch.randn((M, K), device=dev, generator=g) * 0.25).to(torch.bfloat16)
        res = []
        for fn in (lambda w: ops.wi8a16_gemm(x, w, sg, 128), lambda w: ops.w8a16_gemm(x, w.view(torch.float8_e4m3fn), sc)):
            for w in ws: fn(w)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(4):
                    for w in ws: fn(w)
            gr.replay(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
            res.append(a.elapsed_time(b) * 1e3 / (4 * n))
        print(f"M={M} {name:8s} {K}x{N}: int8 g128 {res[0]:6.1f} us ({K*N/res[0]/1e3:5.0f} GB/s)   fp8 per-channel {res[1]:6.1f} us ({K*N/res[1]/1e3:5.0f} GB/s)", flush=True)
        del ws
