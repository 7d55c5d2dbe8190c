"""TP=8 per-rank decode shapes of Llama-3-70B at M = 1: time of w4a16_gemm for every K-split the library offers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd")); sys.path.insert(0, ROOT)
import torch
from mi355q import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for name, K, N in (("qkv", 8192, 1280), ("o", 1024, 8192), ("down", 3584, 8192), ("gate_up", 8192, 7168)):
    for tile in (None, 16, 32):
        ws = []
        try:
            for _ in range(24):
                qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev, generator=g)
                sc = (torch.rand((K // 128, N), device=dev, generator=g) * 0.05).to(torch.bfloat16)
                ws.append(ops.w4a16_repack(qw, sc, None, 128) if tile is None else ops.w4a16_repack(qw, sc, None, 128, tile=tile))
        except Exception as e:
            print(name, tile, "repack failed:", str(e)[:80]); continue
        x = (torch.randn((1, K), device=dev, generator=g) * 0.25).to(torch.bfloat16)
        for sk in (0, 2, 3, 4, 8):
            try:
                fn = lambda w: ops.w4a16_gemm(x, w, split_k=sk)
                for w in ws: fn(w)
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    for _ in range(4):
                        for w in ws: fn(w)
                gr.replay(); torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
                print(f"{name} {K}x{N} tile {ws[0].tile} split_k {sk}: {a.elapsed_time(b) * 1e3 / (4 * len(ws)):6.2f} us", flush=True)
            except Exception as e:
                print(name, tile, sk, "failed:", str(e)[:100])
