import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd"))
import torch
from mi355q import _lib, ops
dev = "cuda:0"
K = 4096
for N in (28672, 28672 * 2, 28672 * 4, 28672 * 8):
    for tile, waves in ((64, 8), (64, 4), (32, 8)):
        _lib.set_option("MI355Q_WAVES", str(waves))
        _lib.set_option("MI355Q_SPLITK", "1")
        ws = []
        for _ in range(3):
            qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
            sc = (torch.rand((K // 128, N), device=dev) * 0.01).to(torch.bfloat16)
            ws.append(ops.w4a16_repack(qw, sc, None, 128, tile=tile))
        x = torch.randn((1, K), device=dev).to(torch.bfloat16)
        for w in ws: ops.w4a16_gemm(x, w)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for rep in range(4):
                for w in ws: ops.w4a16_gemm(x, w)
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): g.replay()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 60
        nbytes = K * N // 2 + (K // 128) * N * 2
        print(f"N={N:7d} tile={tile} waves={waves}: {us:8.2f} us {nbytes/us/1e3:7.0f} GB/s", flush=True)
        del ws
