"""Times the W4A16 GEMV shapes of Llama-3-8B under different tile widths / wave counts."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd"))
import torch
from mi355q import ops, _lib

dev = "cuda:0"
M = int(os.environ.get("M", "1"))
shapes = {"qkv": (4096, 6144, "norm"), "o": (4096, 4096, "none"), "gate_up": (4096, 28672, "norm"), "down": (14336, 4096, "silu")}
if os.environ.get("PRO"):
    shapes = {k: (v[0], v[1], os.environ["PRO"]) for k, v in shapes.items()}
    shapes["down"] = (14336, 4096, "none" if os.environ["PRO"] == "none" else "silu")
NL = 12  # distinct weight copies so that launches do not hit the Infinity Cache

def make(K, N, tile):
    ws = []
    for _ in range(NL):
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
        sc = (torch.rand((K // 128, N), device=dev) * 0.01).to(torch.bfloat16)
        ws.append(ops.w4a16_repack(qw, sc, None, 128, tile=tile))
    return ws

def run(name, K, N, pro, ws):
    if pro == "norm" and K != 4096 and False: return 0, 0
    x = torch.randn((M, K * (2 if pro == "silu" else 1)), device=dev).to(torch.bfloat16)
    res = torch.randn((M, K), device=dev).to(torch.bfloat16)
    nw = torch.ones(K, device=dev, dtype=torch.bfloat16)
    def call(w):
        if pro == "norm":
            ops.w4a16_gemm_add_rms_norm(x, res, nw, 1e-5, w)
        elif pro == "silu":
            ops.w4a16_gemm(x, w, fuse_silu_mul=True)
        else:
            ops.w4a16_gemm(x, w)
    for w in ws: call(w)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for rep in range(4):
            for w in ws: call(w)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / (5 * 4 * NL)
    nbytes = K * N // 2 + (K // 128) * N * 2
    return us, nbytes / us / 1e3

for name, (K, N, pro) in shapes.items():
    if len(sys.argv) > 1 and name not in sys.argv[1:]: continue
    for tile in (64, 32, 16):
        ws = make(K, N, tile)
        for waves in (4, 8, 16):
            if tile == 64 and waves == 16: continue
            _lib.set_option("MI355Q_WAVES", str(waves))
            for split in (1,):
                _lib.set_option("MI355Q_SPLITK", str(split))
                us, gbs = run(name, K, N, pro, ws)
                print(f"{name:8s} M={M} tile={tile:2d} waves={waves:2d} split={split} : {us:7.2f} us  {gbs:7.0f} GB/s", flush=True)
        del ws
        torch.cuda.empty_cache()
