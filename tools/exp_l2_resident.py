"""Does a decode GEMV run faster when its weights were just read (L2 / Infinity-Cache resident)?  Same weight
launched back to back vs a rotation of weights larger than the caches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd")); sys.path.insert(0, ROOT)
import torch
from mi355q import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for name, K, N in [("o", 4096, 4096), ("qkv", 4096, 6144), ("down", 14336, 4096), ("gate_up", 4096, 28672)]:
    n = max(2, int(700e6 // (K * N // 2)))
    ws = []
    for _ in range(n):
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev, generator=g)
        sc = (torch.rand((K // 128, N), device=dev, generator=g) * 0.05).to(torch.bfloat16)
        ws.append(ops.w4a16_repack(qw, sc, None, 128))
    x = (torch.randn((1, K), device=dev, generator=g) * 0.25).to(torch.bfloat16)
    res = []
    for seq in (ws, [ws[0]] * n):
        for w in seq: ops.w4a16_gemm(x, w)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(4):
                for w in seq: ops.w4a16_gemm(x, w)
        gr.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) * 1e3 / (4 * n))
    print(f"{name:8s} {K*N/2/1e6:5.1f} MB: rotating {res[0]:5.1f} us   same weight {res[1]:5.1f} us", flush=True)
    del ws
