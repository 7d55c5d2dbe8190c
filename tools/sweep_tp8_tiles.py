"""Decode GEMV time of the Llama-3-70B TP=8 shard shapes at M = 1 for every strip width the kernel takes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd")); sys.path.insert(0, ROOT)
import torch
from mi355q import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
H, I, QKV, tp = 8192, 28672, (64 + 16) * 128, 8
for name, K, N in [("qkv", H, QKV // tp), ("o", H // tp, H), ("gate_up", H, 2 * I // tp), ("down", I // tp, H)]:
    for tile in (0, 16, 32, 64, 112):
        if tile and N % tile:
            continue
        ws = []
        try:
            for _ in range(24):
                qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev, generator=g)
                sc = (torch.rand((K // 128, N), device=dev, generator=g) * 0.05).to(torch.bfloat16)
                ws.append(ops.w4a16_repack(qw, sc, None, 128, tile=tile, pair_halves=(name == "gate_up")))
            x = (torch.randn((1, K), device=dev, generator=g) * 0.25).to(torch.bfloat16)
            fn = (lambda w: ops.w4a16_gate_up_silu(x, None, None, 1e-5, w, add_norm=False)) if name == "gate_up" else (lambda w: ops.w4a16_gemm(x, w))
            for w in ws: fn(w)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(4):
                    for w in ws: fn(w)
            gr.replay(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b) * 1e3 / (4 * len(ws))
            print(f"{name} {K}x{N} tile {tile} -> {ws[0].tile}: {us:6.2f} us", flush=True)
        except RuntimeError as e:
            print(f"{name} {K}x{N} tile {tile}: {str(e)[:100]}", flush=True)
        del ws
