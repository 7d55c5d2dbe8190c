"""One W4A16 prompt-GEMM shape, a few calls (for counter passes): python tools/bench_mm_one.py M K N [tile]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd")); sys.path.insert(0, ROOT)
import torch
from mi355q import ops
M, K, N = (int(a) for a in sys.argv[1:4])
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev, generator=g)
sc = (torch.rand((K // 128, N), device=dev, generator=g) * 0.05).to(torch.bfloat16)
w = ops.w4a16_repack(qw, sc, None, 128)
x = (torch.randn((M, K), device=dev, generator=g) * 0.25).to(torch.bfloat16)
for _ in range(5):
    ops.w4a16_mm(x, w)
torch.cuda.synchronize()
