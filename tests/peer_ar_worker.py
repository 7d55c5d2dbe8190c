"""Worker of the peer all-reduce test: one rank of a gloo group (launched by `python -m torch.distributed.run`);
all ranks share GPU 0, so every rank's region is peer memory of the others through the same IPC path that
crosses xGMI between GPUs.  Builds the communicator (its own check against the group's all-reduce included),
then times it and checks a longer randomised series, eager and as a replayed graph."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "vllm-fork_amd")]

from mi355q.plugin.peer_all_reduce import PeerAllReduce  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ar = PeerAllReduce(device=dev, max_bytes=64 * 1024)
    res = {"rank": rank, "world": world, "disabled": ar.disabled, "reason": ar.reason}
    if not ar.disabled:
        gen = torch.Generator(device=dev).manual_seed(7 + rank)
        worst = 0.0
        for it in range(40):
            n = [8192, 2, 4096 * 3 + 2, 32768][it % 4]
            x = (torch.randn(n, device=dev, generator=gen) * (1 + it % 3)).to(torch.bfloat16 if it % 2 else torch.float16)
            want = x.float().cpu()
            dist.all_reduce(want)
            got = ar.all_reduce(x.clone()).float().cpu()
            worst = max(worst, float((got - want).abs().max() / (want.abs().max() + 1e-6)))
        res["worst_rel_err"] = worst
        # exact: integers small enough that every partial sum is exact in 16 bits
        xi = torch.randint(-8, 9, (8192,), device=dev, generator=gen).to(torch.bfloat16)
        want = xi.float().cpu()
        dist.all_reduce(want)
        res["exact"] = bool(torch.equal(ar.all_reduce(xi.clone()).float().cpu(), want))
        # ranks out of step: no host synchronisation between calls, every rank stalls its stream at different calls,
        # sizes change from call to call; inputs are functions of (call, rank), so every rank knows the exact sum
        base = torch.arange(16384, device=dev)
        bad = torch.zeros((), dtype=torch.int64, device=dev)
        for it in range(400):
            n = (2, 8192, 16384, 770)[it % 4]
            if (it * 7 + rank * 3) % 5 == 0:
                torch.cuda._sleep(150000 + 50000 * rank)
            x = (((base[:n] + it * 7 + rank * 13) % 17) - 8).to(torch.bfloat16)
            want = sum((((base[:n] + it * 7 + r * 13) % 17) - 8) for r in range(world)).to(torch.bfloat16)
            bad += (ar.all_reduce(x) != want).sum()
        res["skewed_series_mismatches"] = int(bad.item())
        # a replayed graph of 32 dependent reductions (what a decode step issues), timed
        x = torch.ones(8192, device=dev, dtype=torch.bfloat16) * (1.0 / 64)
        y = torch.empty_like(x)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            ar.all_reduce(x, out=y)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(32):
                    ar.all_reduce(x, out=y)
            g.replay()
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(10):
                g.replay()
            torch.cuda.synchronize()
            res["us_per_all_reduce_16k"] = (time.perf_counter() - t0) / 320 * 1e6
        res["graph_value_ok"] = bool(torch.equal(y.float().cpu(), torch.full((8192,), world / 64.0)))
        res["timeouts"] = ar.timeouts()
    # the plugin's TP collectives prefer the exchange over peer memory (a second communicator on the same group)
    from mi355q.plugin import parallel
    parallel.init_tensor_parallel()
    peer = parallel.get_peer_all_reduce()
    res["plugin_peer"] = peer is not None
    if peer is not None:
        x = torch.full((2, 4096), float(rank + 1), device=dev, dtype=torch.bfloat16)
        got = parallel.tensor_model_parallel_all_reduce(x.clone())
        res["plugin_sum_ok"] = bool(torch.equal(got.float().cpu(), torch.full((2, 4096), world * (world + 1) / 2.0)))
        cat = parallel.tensor_model_parallel_all_gather(x[:, :6])
        want = torch.cat([torch.full((2, 6), float(r + 1)) for r in range(world)], dim=-1)
        res["plugin_gather_ok"] = bool(torch.equal(cat.float().cpu(), want))
        big = torch.ones((512, 4096), device=dev, dtype=torch.bfloat16)       # beyond the region: the group's own path
        res["plugin_large_goes_to_group"] = not peer.should_use(big)
        res["plugin_timeouts"] = peer.timeouts()
        parallel.init_tensor_parallel(peer_all_reduce=False)                  # closes it
    with open(os.path.join(os.environ["PEER_AR_OUT"], f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    ar.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
