"""Worker of test_tp_gloo.py: one tensor-parallel rank on CPU (gloo).  Launched with
`python -m torch.distributed.run --nproc-per-node N`, as the benchmark is on a GPU node.

The GPU kernel cannot run here, so MI355XW4A16LinearKernel.apply_weights is replaced by the CPU
oracle reading the rank's repacked shard: what is under test is the host logic around it --
shard loading, load-time repack, rank-0-only bias, all-reduce after row-parallel layers and
all-gather of column-parallel outputs."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "vllm-fork_amd")]

import oracle as orc  # noqa: E402
from mi355q import _lib, ops  # noqa: E402
from mi355q.plugin import AutoGPTQConfig, MI355XW4A16LinearKernel, parallel  # noqa: E402
from mi355q.plugin.linear import (ColumnParallelLinear, MergedColumnParallelLinear,  # noqa: E402
                                  RowParallelLinear)


def bits(t):
    return t.detach().contiguous().view(torch.int16).numpy().view(np.uint16)


def bf(b):
    return torch.from_numpy(np.ascontiguousarray(b).view(np.int16)).view(torch.bfloat16)


def oracle_apply(self, layer, x, bias=None):
    w = layer.mi355q_weight
    q = np.empty((w.K, w.N), dtype=np.int32)
    _lib.check(_lib.lib().mi355q_w4a16_unpack_host(w.packed.data_ptr(), q.ctypes.data, w.K, w.N, w.tile))
    out = orc.w4a16_gemm(bits(x.reshape(-1, w.K)), orc.gptq_pack(q), bits(ops.scales_row_major(w)), None,
                         None if bias is None else bits(bias), G=w.G)
    return bf(out).reshape(x.shape[:-1] + (w.N,))


def main():
    dist.init_process_group("gloo")
    # asking for the exchange over peer memory where no rank has a GPU: construction fails on every rank, all ranks
    # agree on that, and the group's own collectives serve
    parallel.init_tensor_parallel(peer_all_reduce=True)
    peer_missing = parallel.get_peer_all_reduce() is None
    rank, size = parallel.get_tensor_model_parallel_rank(), parallel.get_tensor_model_parallel_world_size()
    MI355XW4A16LinearKernel.apply_weights = oracle_apply

    rng = np.random.default_rng(0)            # same checkpoint on every rank
    hidden, inter, G, T = 128, 256 * size, 64, 3
    cfg = AutoGPTQConfig.from_config({"bits": 4, "group_size": G, "desc_act": False, "sym": True})
    ck = {}
    for name, (K, N) in {"gate": (hidden, inter), "up": (hidden, inter), "down": (inter, hidden)}.items():
        ck[name] = (rng.integers(0, 16, size=(K, N), dtype=np.int32),
                    orc.f32_to_h16((0.02 * rng.random((K // G, N)) + 0.005).astype(np.float32)))
    down_bias = orc.f32_to_h16(rng.standard_normal(hidden).astype(np.float32))
    x = orc.f32_to_h16(rng.standard_normal((T, hidden)).astype(np.float32))

    gate_up = MergedColumnParallelLinear(hidden, [inter, inter], bias=False, params_dtype=torch.bfloat16,
                                         quant_config=cfg, prefix="mlp.gate_up_proj", return_bias=False)
    down = RowParallelLinear(inter, hidden, bias=True, params_dtype=torch.bfloat16, quant_config=cfg,
                             prefix="mlp.down_proj", return_bias=False)
    for i, n in enumerate(("gate", "up")):
        gate_up.weight_loader(gate_up.qweight, torch.from_numpy(orc.gptq_pack(ck[n][0])), i)
        gate_up.weight_loader(gate_up.scales, bf(ck[n][1]), i)
    down.weight_loader(down.qweight, torch.from_numpy(orc.gptq_pack(ck["down"][0])))
    down.weight_loader(down.scales, bf(ck["down"][1]))
    down.weight_loader(down.bias, bf(down_bias))
    gate_up.process_weights_after_loading()
    down.process_weights_after_loading()

    gu = gate_up(bf(x))
    act = bf(orc.silu_and_mul(bits(gu)))
    y = down(act)                              # all-reduced inside

    # expected, from the full checkpoint: every rank's partial rounded to bf16 (bias on rank 0), summed in bf16
    ni = inter // size
    total = None
    for r in range(size):
        sl = slice(r * ni, (r + 1) * ni)
        g = orc.w4a16_gemm(x, orc.gptq_pack(np.ascontiguousarray(ck["gate"][0][:, sl])),
                           np.ascontiguousarray(ck["gate"][1][:, sl]), None, None, G=G)
        u = orc.w4a16_gemm(x, orc.gptq_pack(np.ascontiguousarray(ck["up"][0][:, sl])),
                           np.ascontiguousarray(ck["up"][1][:, sl]), None, None, G=G)
        a = orc.silu_and_mul(np.concatenate([g, u], axis=1))
        gpr = ni // G
        p = orc.w4a16_gemm(a, orc.gptq_pack(np.ascontiguousarray(ck["down"][0][sl])),
                           np.ascontiguousarray(ck["down"][1][r * gpr:(r + 1) * gpr]), None,
                           down_bias if r == 0 else None, G=G)
        total = bf(p) if total is None else total + bf(p)
    # unsharded model, for closeness
    g = orc.w4a16_gemm(x, orc.gptq_pack(ck["gate"][0]), ck["gate"][1], None, None, G=G)
    u = orc.w4a16_gemm(x, orc.gptq_pack(ck["up"][0]), ck["up"][1], None, None, G=G)
    full = orc.w4a16_gemm(orc.silu_and_mul(np.concatenate([g, u], axis=1)), orc.gptq_pack(ck["down"][0]),
                          ck["down"][1], None, down_bias, G=G)

    # column-parallel 16-bit layer with gathered output (the lm_head pattern)
    w16 = torch.from_numpy(rng.standard_normal((64 * size, hidden)).astype(np.float32)).to(torch.bfloat16)
    head = ColumnParallelLinear(hidden, 64 * size, bias=False, gather_output=True, params_dtype=torch.bfloat16,
                                prefix="lm_head", return_bias=False)
    head.weight_loader(head.weight, w16)
    logits = head(bf(x))
    ref_logits = torch.nn.functional.linear(bf(x), w16)

    res = {
        "rank": rank, "size": size, "peer_exchange_absent": peer_missing,
        "tp_exact": bool(torch.equal(y, total)),
        # with more than two ranks the collective's summation order is its own: one bf16 ulp of slack
        "tp_partials_close": bool(torch.allclose(y.float(), total.float(), rtol=2 ** -7, atol=2 ** -7)),
        "tp_close": bool(torch.allclose(y.float(), bf(full).float(), rtol=3e-2, atol=3e-2)),
        "bias_once": bool(abs(float((y.float() - bf(full).float()).mean())) < 0.02),
        "gather_shape": list(logits.shape),
        "gather_close": bool(torch.allclose(logits.float(), ref_logits.float(), rtol=2e-2, atol=2e-2)),
    }
    with open(os.path.join(os.environ["TP_OUT"], f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
