"""Experiment: strips per wave and K-slices of the tiled W4A16 kernel at prompt sizes (us per call incl. slice merge)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd")); sys.path.insert(0, ROOT)
import torch
from mi355q import ops, _lib
dev = "cuda:0"
Ms = [int(a) for a in sys.argv[1:]] or [128, 512]
shapes = [("qkv", 4096, 6144, [2, 1]), ("o", 4096, 4096, [4, 2, 1]), ("down", 14336, 4096, [4, 2, 1]), ("gate_up", 4096, 28672, [1])]
g = torch.Generator(device=dev).manual_seed(0)
for name, K, N, spws in shapes:
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev, generator=g)
    sc = (torch.rand((K // 128, N), device=dev, generator=g) * 0.05).to(torch.bfloat16)
    w = ops.w4a16_repack(qw, sc, None, 128)
    for M in Ms:
        x = (torch.randn((M, K), device=dev, generator=g) * 0.25).to(torch.bfloat16)
        row = []
        for spw in spws:
            for s in [0, 1, 2, 4, 7, 8, 14, 16]:
                if s and (K // 128) % s: continue
                _lib.set_option("MI355Q_MM_SPW", str(spw)); _lib.set_option("MI355Q_MM_SPLITS", str(s) if s else None)
                try:
                    for _ in range(3): ops.w4a16_mm(x, w)
                except RuntimeError as e:
                    continue
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    for _ in range(20): ops.w4a16_mm(x, w)
                gr.replay(); torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
                row.append(f"spw{spw}/s{s if s else 'auto'}:{a.elapsed_time(b) / 20 * 1e3:6.1f}")
        print(f"{name:8s} M={M:5d} " + "  ".join(row), flush=True)
