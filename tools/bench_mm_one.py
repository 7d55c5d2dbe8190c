"""One shape of the tiled kernel, a few launches (for counter passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd")); sys.path.insert(0, ROOT)
import torch
from mi355q import ops
dev = "cuda:0"
M, K, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = torch.Generator(device=dev).manual_seed(0)
qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev, generator=g)
sc = (torch.rand((K // 128, N), device=dev, generator=g) * 0.05).to(torch.bfloat16)
w = ops.w4a16_repack(qw, sc, None, 128)
x = (torch.randn((M, K), device=dev, generator=g) * 0.25).to(torch.bfloat16)
for _ in range(4):
    ops.w4a16_mm(x, w)
torch.cuda.synchronize()
