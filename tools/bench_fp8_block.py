"""Block-scaled fp8 product (per-token-group activation scales x [128, 128] weight blocks) beside the per-tensor
scaled product on the same shapes: us per launch and weight bytes per second, 16 weight sets in rotation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd"))
import torch
from mi355q import ops
ops.FP8_MM_BLAS_MIN_TOKENS = 1 << 30
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
def timeit(fn, n):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(2 * n): fn(i)
    gr.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / (2 * n))
    return best
for N, K in [(28672, 4096), (4096, 14336), (7168, 4096), (6144, 4096)]:
    n = 16
    ws = [torch.randint(0, 255, (N, K), dtype=torch.uint8, device=dev, generator=g).clamp_(max=0x7e).view(torch.float8_e4m3fn) for _ in range(n)]
    Bs = torch.rand((N // 128, K // 128), device=dev, generator=g) * 1e-2
    sb = torch.rand((1,), device=dev, generator=g) * 1e-2
    for M in (1, 16, 64, 512):
        x = (torch.randn((M, K), device=dev, generator=g) * 0.3).to(torch.bfloat16)
        xq, xs = ops.per_token_group_quant_fp8(x, 128)
        xq1, s1 = ops.scaled_fp8_quant(x)
        tb = timeit(lambda i: ops.w8a8_block_scaled_mm(xq, ws[i % n], xs, Bs, [128, 128], torch.bfloat16), n)
        tt = timeit(lambda i: ops.fp8_scaled_mm(xq1, ws[i % n], torch.bfloat16, s1, sb), n)
        tq = timeit(lambda i: ops.per_token_group_quant_fp8(x, 128), n)
        print(f"N {N:5d} K {K:5d} M {M:3d}: block {tb:7.2f} us ({N * K / tb / 1e6:5.2f} TB/s) | per-tensor {tt:7.2f} us | group quant {tq:5.2f} us", flush=True)
