"""How much faster is the W4A16 GEMV when its weights sit in the Infinity Cache (MALL)?
Times each Llama-3-8B shape with NL distinct weight copies cycled through: NL = 1 keeps the
weights cache-resident, large NL streams from HBM."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd"))
import torch
from mi355q import ops

dev = "cuda:0"
shapes = {"qkv": (4096, 6144, "norm"), "o": (4096, 4096, "none"), "gate_up": (4096, 28672, "norm"), "down": (14336, 4096, "silu")}

def run(K, N, pro, NL):
    ws = []
    for _ in range(NL):
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
        sc = (torch.rand((K // 128, N), device=dev) * 0.01).to(torch.bfloat16)
        ws.append(ops.w4a16_repack(qw, sc, None, 128))
    x = torch.randn((1, K * (2 if pro == "silu" else 1)), device=dev).to(torch.bfloat16)
    res = torch.randn((1, K), device=dev).to(torch.bfloat16)
    nw = torch.ones(K, device=dev, dtype=torch.bfloat16)
    def call(w):
        if pro == "norm": ops.w4a16_gemm_add_rms_norm(x, res, nw, 1e-5, w)
        elif pro == "silu": ops.w4a16_gemm(x, w, fuse_silu_mul=True)
        else: ops.w4a16_gemm(x, w)
    for w in ws: call(w)
    torch.cuda.synchronize()
    reps = max(1, 64 // NL)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            for w in ws: call(w)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (5 * reps * NL)

for name, (K, N, pro) in shapes.items():
    nbytes = K * N // 2 + (K // 128) * N * 2
    row = []
    for NL in (1, 2, 4, 16, 40):
        us = run(K, N, pro, NL)
        row.append(f"NL={NL}: {us:6.2f} us {nbytes / us / 1e3:6.0f} GB/s")
        torch.cuda.empty_cache()
    print(f"{name:8s} {nbytes / 1e6:6.1f} MB  " + " | ".join(row), flush=True)
