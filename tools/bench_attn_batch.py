"""Decode attention alone at serving batch sizes: us per launch and cache bytes per second.

    python tools/bench_attn_batch.py [batch ctx [kv_dtype]] ...   (default: a sweep around 32 x 1060)

The block tables hand every sequence blocks scattered over the pool, the way a long-running server's
allocator leaves them.  Launches run back to back inside one captured graph.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd"))
import torch  # noqa: E402
from mi355q import ops  # noqa: E402

H, KVH, D, BS = 32, 8, 128, 16
dev = "cuda:0"


def run(batch, ctx, kv="auto", iters=40):
    torch.manual_seed(0)
    fp8 = kv == "fp8"
    cdt = torch.uint8 if fp8 else torch.bfloat16
    x = 16 if fp8 else 8
    max_len = (ctx + 16 + 255) // 256 * 256
    per = max_len // BS
    nblk = batch * per
    layers = 4   # rotate over several caches so that no launch finds its predecessor's lines on-die
    kc = [torch.randn((nblk, KVH, D // x, BS, x), device=dev).mul_(0.5).to(torch.bfloat16) for _ in range(layers)]
    vc = [torch.randn((nblk, KVH, D, BS), device=dev).mul_(0.5).to(torch.bfloat16) for _ in range(layers)]
    if fp8:
        kc = [t.to(torch.float8_e4m3fn).view(torch.uint8) for t in kc]
        vc = [t.to(torch.float8_e4m3fn).view(torch.uint8) for t in vc]
    bt = torch.randperm(nblk, device=dev).to(torch.int32).view(batch, per)
    seq_lens = torch.full((batch,), ctx, dtype=torch.int32, device=dev)
    positions = torch.full((batch,), ctx - 1, dtype=torch.int64, device=dev)
    cos_sin = torch.randn((max_len + 8, D), device=dev).to(torch.bfloat16)
    qkv = torch.randn((batch, (H + 2 * KVH) * D), device=dev).to(torch.bfloat16)
    out = torch.empty((batch, H, D), dtype=torch.bfloat16, device=dev)
    parts = max_len // 256
    es = torch.empty((batch, H, parts), dtype=torch.float32, device=dev)
    ml = torch.empty_like(es)
    tmp = torch.empty((batch, H, parts, D), dtype=torch.bfloat16, device=dev)
    one = torch.ones((1,), dtype=torch.float32, device=dev)

    def go(i):
        ops.decode_attention_fused(out, es, ml, tmp, qkv, positions, cos_sin, kc[i % layers], vc[i % layers], H, KVH, D,
                                   D ** -0.5, bt, seq_lens, BS, max_len, kv, one, one)

    for i in range(3):
        go(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            go(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (3 * iters)
    nbytes = batch * ctx * KVH * D * 2 * (1 if fp8 else 2)
    print(f"batch {batch:3d} ctx {ctx:5d} kv {kv:4s}: {us:7.2f} us  {nbytes / us / 1e6:6.2f} TB/s of cache", flush=True)


def stamps(batch, ctx):
    """Phase times of a sample of workgroups (one per (kv head, partition): the sequences share the slot, the last to
    write wins) of one launch at full load."""
    import ctypes
    from mi355q import _lib
    L = _lib.lib()
    L.mi355q_debug_attention_stamps.argtypes = [ctypes.c_void_p]
    L.mi355q_debug_attention_stamps.restype = None
    st = torch.zeros((KVH * 16, 10), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    L.mi355q_debug_attention_stamps(st.data_ptr())
    run(batch, ctx, iters=1)
    L.mi355q_debug_attention_stamps(None)
    parts = (ctx + 16 + 255) // 256
    t = st[:KVH * parts, [0, 6, 8, 7, 1, 2, 3, 4, 5]].cpu().double()
    t = (t - t[:, 0].min()) / 100.0
    print("us after the first entry: entry | seq_len known | requests issued | q roped | patched | QK | softmax | PV | stored")
    for i, row in enumerate(t.tolist()):
        print(f"head {i // parts} part {i % parts}: " + " ".join(f"{v:6.2f}" for v in row))


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "stamps":
        stamps(int(a[1]), int(a[2]))
    elif a:
        run(int(a[0]), int(a[1]), a[2] if len(a) > 2 else "auto")
    else:
        for b, c in [(32, 1024), (32, 1060), (32, 1280), (16, 1060), (64, 1060), (32, 2048), (8, 1060), (32, 512)]:
            run(b, c)
        run(32, 1060, "fp8")
