"""Microbenchmark of the tiled W4A16 kernel on the 8B layer shapes: us per call and TFLOP/s, next to the
dense bf16 GEMM of the same shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd")); sys.path.insert(0, ROOT)
import torch
from mi355q import ops, _lib

dev = "cuda:0"
Ms = [int(a) for a in sys.argv[1:]] or [128, 1024]
shapes = [("gate_up", 4096, 28672), ("down", 14336, 4096), ("qkv", 4096, 6144), ("o", 4096, 4096)]
g = torch.Generator(device=dev).manual_seed(0)
for name, K, N in shapes:
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev, generator=g)
    sc = (torch.rand((K // 128, N), device=dev, generator=g) * 0.05).to(torch.bfloat16)
    w = ops.w4a16_repack(qw, sc, None, 128)
    wd = ops.w4a16_dequantize(w)
    for M in Ms:
        x = (torch.randn((M, K), device=dev, generator=g) * 0.25).to(torch.bfloat16)
        res = []
        for fn in (lambda: ops.w4a16_mm(x, w), lambda: torch.matmul(x, wd)):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): fn()
            b.record(); torch.cuda.synchronize()
            res.append(a.elapsed_time(b) / 20 * 1e3)
        fl = 2.0 * M * K * N
        print(f"{name:8s} tile {w.tile:3d} M={M:5d}  tiled {res[0]:8.1f} us ({fl/res[0]/1e6:6.1f} TF)   dense bf16 {res[1]:8.1f} us ({fl/res[1]/1e6:6.1f} TF)", flush=True)
