"""Per-launch time of the four decode GEMVs of a Llama-3-8B layer as the runner launches them (batch 1), each
measured as a graph of back-to-back launches over 32 distinct weight sets (no launch finds its weights cached)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd")); sys.path.insert(0, ROOT)
import torch
from mi355q import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
H, I = 4096, 14336
def mk(K, N, pair=False, n=32):
    ws = []
    for _ in range(n):
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev, generator=g)
        sc = (torch.rand((K // 128, N), device=dev, generator=g) * 0.01).to(torch.bfloat16)
        ws.append(ops.w4a16_repack(qw, sc, None, 128, pair_halves=pair))
    return ws
x = (torch.randn((1, H), device=dev, generator=g) * 0.25).to(torch.bfloat16)
res = (torch.randn((1, H), device=dev, generator=g) * 0.25).to(torch.bfloat16)
act = (torch.randn((1, I), device=dev, generator=g) * 0.25).to(torch.bfloat16)
nw = torch.ones(H, device=dev, dtype=torch.bfloat16)
cases = [("qkv+norm", mk(H, 6144), lambda w: ops.w4a16_gemm_add_rms_norm(x, res, nw, 1e-5, w)),
         ("o", mk(H, H), lambda w: ops.w4a16_gemm(x, w)),
         ("gate_up+norm+silu", mk(H, 2 * I, True), lambda w: ops.w4a16_gate_up_silu(x, res, nw, 1e-5, w)),
         ("down", mk(I, H), lambda w: ops.w4a16_gemm(act, w))]
out = []
for name, ws, fn in cases:
    for w in ws[:3]: fn(w)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(4):
            for w in ws: fn(w)
    gr.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / (4 * len(ws)))
    out.append(f"{name} {min(ts):.2f}")
print(" | ".join(out), flush=True)
