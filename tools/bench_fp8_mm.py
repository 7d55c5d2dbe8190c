"""fp8 x fp8 scaled product on the 8B layer shapes: us per call, TB/s of weight bytes (decode rows) or
PFLOP/s (prompt rows), beside torch._scaled_mm (hipBLASLt, the route the reference takes above four rows)
and the fp8 weight-only kernel.  Weights rotate through ~0.7 GB so no call finds them in the Infinity Cache."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd")); sys.path.insert(0, ROOT)
import torch
from mi355q import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
rows = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 4, 16, 32, 64, 128, 512, 2048]


def timed(fn, ws, reps):
    for w in ws: fn(w)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            for w in ws: fn(w)
    gr.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / (reps * len(ws)))
    return best


for M in rows:
    for name, K, N in [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096)]:
        n = max(2, int(700e6 // (K * N)))
        ws = [torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev, generator=g).view(torch.float8_e4m3fn) for _ in range(n)]
        a8 = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev, generator=g).view(torch.float8_e4m3fn)
        x16 = (torch.randn((M, K), device=dev, generator=g) * 0.25).to(torch.bfloat16)
        sa, sb = torch.full((1,), 0.01, device=dev), torch.full((1,), 0.02, device=dev)
        sc = torch.rand((N,), device=dev, generator=g) * 0.004
        reps = 4 if M <= 64 else 1
        t_mine = timed(lambda w: ops.fp8_scaled_mm(a8, w, torch.bfloat16, sa, sb), ws, reps)
        try:
            t_blas = timed(lambda w: torch._scaled_mm(a8, w.t(), out_dtype=torch.bfloat16, scale_a=sa, scale_b=sb), ws, reps)
        except Exception as e:   # noqa: BLE001
            t_blas = float("nan")
        t_w8a16 = timed(lambda w: ops.w8a16_gemm(x16, w, sc), ws, reps) if M <= 16 else float("nan")
        fl = 2.0 * M * N * K
        print(f"M={M:5d} {name:8s} {K}x{N}: w8a8 {t_mine:8.1f} us ({K*N/t_mine/1e6:5.2f} TB/s, {fl/t_mine/1e9:6.3f} PF)   "
              f"_scaled_mm {t_blas:8.1f} us ({fl/t_blas/1e9:6.3f} PF)   w8a16 {t_w8a16:7.1f} us", flush=True)
        del ws
