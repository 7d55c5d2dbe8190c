"""Which (sequence, head) outputs of the fused decode step differ from rotary_embedding + reshape_and_cache +
paged_attention_rocm (debugging aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd"))
import torch
from mi355q import ops
H, KVH, D, BS = 32, 8, 128, 16
dev = "cuda:0"
import random
random.seed(1)
cases = [[1, 700, 16, 130, 256, 1025, 257, 600, 512, 3], [512], [900]] + [[random.randint(1, 2000) for _ in range(12)] for _ in range(30)]
bad = 0
for lens in cases:
    torch.manual_seed(0)
    S = len(lens)
    per = (max(lens) + BS - 1) // BS
    nblk = S * per
    kc0 = torch.rand((nblk, KVH, D // 8, BS, 8), device=dev).mul_(2).sub_(1).to(torch.bfloat16)
    vc0 = torch.rand((nblk, KVH, D, BS), device=dev).mul_(2).sub_(1).to(torch.bfloat16)
    bt = torch.randperm(nblk, device=dev).to(torch.int32).view(S, per)
    sl = torch.tensor(lens, dtype=torch.int32, device=dev)
    pos = (sl - 1).to(torch.int64)
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float) / D))
    fr = torch.einsum("i,j->ij", torch.arange(2048, dtype=torch.float), inv)
    cache = torch.cat((fr.cos(), fr.sin()), -1).to(torch.bfloat16).to(dev)
    qkv = torch.rand((S, (H + 2 * KVH) * D), device=dev).mul_(2).sub_(1).to(torch.bfloat16)
    P = (max(lens) + 255) // 256
    one = torch.ones((1,), dtype=torch.float32, device=dev)
    def bufs():
        return (torch.zeros((S, H, D), dtype=torch.bfloat16, device=dev), torch.zeros((S, H, P), dtype=torch.float32, device=dev),
                torch.zeros((S, H, P), dtype=torch.float32, device=dev), torch.zeros((S, H, P, D), dtype=torch.bfloat16, device=dev))
    kc1, vc1 = kc0.clone(), vc0.clone()
    qkv1 = qkv.clone()
    q1, k1, v1 = qkv1[:, :H * D], qkv1[:, H * D:(H + KVH) * D], qkv1[:, (H + KVH) * D:]
    ops.rotary_embedding(pos, q1, k1, D, cache, True)
    slots = torch.stack([bt[s, (lens[s] - 1) // BS].to(torch.int64) * BS + (lens[s] - 1) % BS for s in range(S)])
    ops.reshape_and_cache(k1.view(S, KVH, D), v1.view(S, KVH, D), kc1, vc1, slots, "auto", one, one)
    o1, es1, ml1, tmp1 = bufs()
    ops.paged_attention_rocm(o1, es1, ml1, tmp1, q1.view(S, H, D), kc1, vc1, KVH, D ** -0.5, bt, sl, None, BS, max(lens), None, "auto", one, one)
    kc2, vc2 = kc0.clone(), vc0.clone()
    o2, es2, ml2, tmp2 = bufs()
    ops.decode_attention_fused(o2, es2, ml2, tmp2, qkv, pos, cache, kc2, vc2, H, KVH, D, D ** -0.5, bt, sl, BS, max(lens), "auto", one, one)
    torch.cuda.synchronize()
    d = (o1 != o2).any(-1).nonzero()
    dt = (tmp1 != tmp2).any(-1).nonzero()
    bad += int(d.shape[0] > 0 or dt.shape[0] > 0)
    if d.shape[0] or dt.shape[0] or len(lens) < 11: print(lens, "out differs at (seq, head):", d[:10].tolist(), "count", d.shape[0], "| tmp differs:", dt[:6].tolist(), dt.shape[0],
          "| max abs", float((o1.float() - o2.float()).abs().max()), "| caches equal", torch.equal(kc1, kc2), torch.equal(vc1, vc2), flush=True)
print("cases with a difference:", bad, "of", len(cases))
