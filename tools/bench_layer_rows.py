"""Per-launch time of the four projections of a Llama-3-8B layer at a decode batch of M rows (default 32), called the
way the runner calls them from two rows on (from 17 rows: qkv and gate_up (+SiLU pairs) on the tiled / streaming kernels, o and
down with the residual add and the next norm fused behind their K-slice merge where a merge exists).  Each is a graph of back-to-back
launches over 32 distinct weight sets, so that no launch finds its weights cached.

    python tools/bench_layer_rows.py [M ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd"))
import torch  # noqa: E402
from mi355q import ops  # noqa: E402

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


W = {"qkv": mk(H, 6144), "o": mk(H, H), "gate_up": mk(H, 2 * I, True), "down": mk(I, H)}
nw = torch.ones(H, device=dev, dtype=torch.bfloat16)


def run(M):
    x = (torch.randn((M, H), device=dev, generator=g) * 0.25).to(torch.bfloat16)
    res = (torch.randn((M, H), device=dev, generator=g) * 0.25).to(torch.bfloat16)
    act = (torch.randn((M, I), device=dev, generator=g) * 0.25).to(torch.bfloat16)
    def proj_norm(a, w):   # as the runner: one fused op where the K-slice merge can take the norm, else two launches
        if ops.w4a16_mm_add_rms_norm_preferred(M, w):
            return ops.w4a16_mm_add_rms_norm(a, w, res, nw, 1e-5)
        o = ops.w4a16_gemm(a, w)
        ops.fused_add_rms_norm(o, res, nw, 1e-5)
        return o

    def gate_up(w):
        if M >= ops.mm_min_tokens(w):
            return ops.w4a16_mm(x, w, silu_pairs=True)
        return ops.w4a16_gate_up_silu(x, None, None, 1e-5, w, add_norm=False)[0]

    cases = [("qkv", W["qkv"], lambda w: ops.w4a16_gemm(x, w)),
             ("o+norm", W["o"], lambda w: proj_norm(x, w)),
             ("gate_up+silu", W["gate_up"], gate_up),
             ("down+norm", W["down"], lambda w: proj_norm(act, w))]
    out, total = [], 0.0
    for name, ws, fn in cases:
        for w in ws[:3]:
            fn(w)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(2):
                for w in ws:
                    fn(w)
        gr.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            gr.replay()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / (2 * len(ws)))
        out.append(f"{name} {min(ts):.2f}")
        total += min(ts)
    print(f"M={M}: " + " | ".join(out) + f" | sum {total:.2f}", flush=True)


if __name__ == "__main__":
    for m in ([int(a) for a in sys.argv[1:]] or [32]):
        run(m)
