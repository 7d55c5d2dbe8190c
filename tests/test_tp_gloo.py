"""Tensor-parallel host path on CPU: world_size 2 and 4 over gloo, launched the way the
benchmark is launched on a GPU node (torch.distributed.run, 127.0.0.1 rendezvous)."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4])
def test_row_and_column_parallel_over_gloo(world, tmp_path):
    env = dict(os.environ, TP_OUT=str(tmp_path), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "tp_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    for r in range(world):
        res = json.load(open(tmp_path / f"rank{r}.json"))
        assert res["size"] == world
        assert res["peer_exchange_absent"] or __import__("torch").cuda.is_available(), res
        # all-reduce of bf16 partials, bias added on rank 0 only; two addends have one possible sum
        assert res["tp_partials_close"] and (res["tp_exact"] or world > 2), res
        assert res["tp_close"] and res["bias_once"], res
        assert res["gather_shape"] == [3, 64 * world] and res["gather_close"], res


@pytest.mark.parametrize("world", [2, 3])
def test_expert_parallel_dispatch_and_combine_over_gloo(world, tmp_path):
    """all_gatherv dispatch + reduce_scatterv combine around the local experts (the reference's
    AgRsAll2AllManager, device_communicators/all2all.py:44-153): ragged token counts, experts placed by
    expert_map_for_rank; every rank's rows equal the single-rank computation over all experts."""
    env = dict(os.environ, EP_OUT=str(tmp_path), EP_MODE="cpu", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "ep_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    seen = []
    for r in range(world):
        res = json.load(open(tmp_path / f"ep_rank{r}.json"))
        assert res["rows"] == [3 + 2 * r, 32] and res["even_rows"] == [3, 32], res
        assert res["max_err"] <= 1e-5 * max(res["scale"], 1.0), res
        seen += res["local_experts"]
    assert sorted(seen) == list(range(8))       # every expert lives on exactly one rank
