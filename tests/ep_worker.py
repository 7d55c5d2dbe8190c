"""Worker of the expert-parallel tests: one rank of an EP group over gloo (launched by
`python -m torch.distributed.run`).

mode "cpu": the exchange alone -- dispatch (all_gatherv) / combine (reduce_scatterv) around dense fp32
experts computed in torch, ragged token counts per rank; every rank checks its rows against the
single-rank computation over all experts.
mode "gpu": the quantized FusedMoE layer with its HIP kernels (all ranks share GPU 0; collectives staged
through the host): an expert-parallel layer (expert_map_for_rank) against the same checkpoint loaded
whole on the same rank."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "vllm-fork_amd")]

import oracle as orc  # noqa: E402
from mi355q.plugin import parallel  # noqa: E402
from mi355q.plugin.moe import AgRsAll2AllManager, expert_map_for_rank  # noqa: E402


def dense_experts(x, w13, w2, weights, ids, expert_map):
    """sum over a token's slots of weight * w2_e(silu(x w1_e) * (x w3_e)); slots whose expert lives
    elsewhere (expert_map < 0) contribute nothing."""
    out = torch.zeros_like(x)
    inter = w2.shape[1]
    for t in range(x.shape[0]):
        for s in range(ids.shape[1]):
            e = int(ids[t, s]) if expert_map is None else int(expert_map[int(ids[t, s])])
            if e < 0:
                continue
            h = x[t] @ w13[e]
            out[t] += weights[t, s] * ((torch.nn.functional.silu(h[:inter]) * h[inter:]) @ w2[e])
    return out


def run_cpu(rank, size, outdir):
    E, H, I, topk = 8, 32, 48, 2
    sizes = [3 + 2 * r for r in range(size)]            # ragged: 3, 5, 7, ... tokens
    g = torch.Generator().manual_seed(0)                # the same "checkpoint" and inputs on every rank
    w13 = torch.randn(E, H, 2 * I, generator=g) * 0.2
    w2 = torch.randn(E, I, H, generator=g) * 0.2
    xs = [torch.randn(n, H, generator=g) for n in sizes]
    logits = [torch.randn(n, E, generator=g) for n in sizes]
    x, lg = xs[rank], logits[rank]
    wts, ids = torch.topk(torch.softmax(lg, dim=-1), topk, dim=-1)
    wts = wts / wts.sum(dim=-1, keepdim=True)
    emap = expert_map_for_rank(E, rank, size)
    local = [int(v) for v in torch.nonzero(emap >= 0).view(-1)]
    a2a = AgRsAll2AllManager(sizes)
    hs, w_all, ids_all = a2a.dispatch(x, wts, ids.to(torch.int32))
    assert hs.shape[0] == sum(sizes) and torch.equal(hs[sum(sizes[:rank]):sum(sizes[:rank + 1])], x)
    part = dense_experts(hs, w13[local], w2[local], w_all, ids_all, emap)
    mine = a2a.combine(part)
    want = dense_experts(x, w13, w2, wts, ids, None)
    res = {"rank": rank, "size": size, "rows": list(mine.shape), "local_experts": local,
           "max_err": float((mine - want).abs().max()), "scale": float(want.abs().max())}
    # equal token counts take the reduce-scatter branch on backends that have one; here: same answer
    even = AgRsAll2AllManager(None)
    hs2, w2_all, ids2_all = even.dispatch(xs[0], *[t for t in (torch.ones(sizes[0], topk), torch.zeros(sizes[0], topk, dtype=torch.int32))])
    res["even_rows"] = list(even.combine(hs2).shape)
    json.dump(res, open(os.path.join(outdir, f"ep_rank{rank}.json"), "w"))


def run_gpu(rank, size, outdir):
    from mi355q.plugin import FusedMoE, MoeWNA16Config
    dev = "cuda:0"
    rng = np.random.default_rng(21)
    E, H, I, G, topk = 8, 256, 128, 128, 2
    sizes = [2 + r for r in range(size)]
    cfg = MoeWNA16Config.from_config({"quant_method": "gptq", "bits": 4, "group_size": G, "sym": True, "desc_act": False})

    def bf(b):
        return torch.from_numpy(np.ascontiguousarray(b).view(np.int16)).view(torch.bfloat16)

    ck = {}
    for e in range(E):
        for sid, (K, N) in (("w1", (H, I)), ("w3", (H, I)), ("w2", (I, H))):
            ck[e, sid] = (rng.integers(0, 16, size=(K, N), dtype=np.int32),
                          orc.f32_to_h16((0.02 * rng.random((K // G, N)) + 0.005).astype(np.float32)))
    xs = [orc.f32_to_h16(rng.standard_normal((n, H)).astype(np.float32)) for n in sizes]
    lgs = [orc.f32_to_h16(rng.standard_normal((n, E)).astype(np.float32)) for n in sizes]

    def build(expert_map):
        with torch.device(dev):
            moe = FusedMoE(E, topk, H, I, params_dtype=torch.bfloat16, renormalize=True, quant_config=cfg,
                           prefix="model.layers.0.block_sparse_moe.experts", expert_map=expert_map)
        for e in range(E):
            for sid in ("w1", "w3", "w2"):
                q, s = ck[e, sid]
                target = "w2" if sid == "w2" else "w13"
                name = f"experts.{e}.{sid}"
                moe.weight_loader(getattr(moe, target + "_qweight"), torch.from_numpy(orc.gptq_pack(q)), name + ".qweight", sid, e)
                moe.weight_loader(getattr(moe, target + "_scales"), bf(s), name + ".scales", sid, e)
        moe.process_weights_after_loading()
        return moe

    x, lg = bf(xs[rank]).to(dev), bf(lgs[rank]).to(dev)
    parallel.init_expert_parallel(0, 1)                         # the whole layer on this rank
    whole = build(None)(x, lg).float().cpu()
    parallel.init_expert_parallel()                             # the default process group
    emap = expert_map_for_rank(E, rank, size).to(dev)
    sharded = build(emap)(x, lg, tokens_per_rank=sizes).float().cpu()
    err = float((sharded - whole).abs().max())
    json.dump({"rank": rank, "size": size, "rows": list(sharded.shape), "max_err": err, "scale": float(whole.abs().max())},
              open(os.path.join(outdir, f"ep_rank{rank}.json"), "w"))


def main():
    dist.init_process_group("gloo")
    parallel.init_expert_parallel()
    rank, size = parallel.get_ep_rank(), parallel.get_ep_world_size()
    (run_gpu if os.environ.get("EP_MODE") == "gpu" else run_cpu)(rank, size, os.environ["EP_OUT"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
