"""bf16 lm_head (128256 x 4096) at 1..32 rows: us per call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd")); sys.path.insert(0, ROOT)
import torch
from mi355q import ops
dev = "cuda:0"
w = (torch.randn((128256, 4096), device=dev) * 0.02).to(torch.bfloat16)
for M in (1, 2, 4, 8, 16, 17, 24, 32):
    x = torch.randn((M, 4096), device=dev).to(torch.bfloat16)
    for _ in range(3): ops.linear_skinny(x, w)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): ops.linear_skinny(x, w)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 100
    print(f"M={M:2d}: {us:7.1f} us  ({w.numel()*2/us/1e3:5.0f} GB/s)", flush=True)
x = torch.randn((32, 4096), device=dev).to(torch.bfloat16)
for _ in range(3): torch.nn.functional.linear(x, w)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): torch.nn.functional.linear(x, w)
b.record(); torch.cuda.synchronize()
print(f"M=32 through torch.nn.functional.linear: {a.elapsed_time(b) * 100:7.1f} us", flush=True)
