"""The HIP path under tensor parallelism, on one GPU: bench.py is launched the way the driver launches
it for N = 2 and 4 (torch.distributed.run, one rank per process), the ranks share GPU 0 and exchange
through gloo, and the greedy tokens must equal the single-rank run of the same synthetic checkpoint.
The `small` model has 2 KV heads, so N = 4 also covers KV-head replication across ranks
(vllm/model_executor/layers/linear.py QKVParallelLinear; parameter.py:178-230)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_tp2_and_tp4_decode_the_tokens_of_tp1(tmp_path):
    env = dict(os.environ, PYTHONUNBUFFERED="1")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "tp_rehearsal.sh"), str(tmp_path), "small"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    sys.stdout.write(r.stdout[-2000:])
    sys.stderr.write(r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "REHEARSAL OK" in r.stdout
    assert "12/12 leading decode steps" in r.stdout.split("TP=2:")[1].split("\n")[0]


@pytest.mark.gpu
def test_expert_parallel_fused_moe_two_ranks_share_the_gpu(tmp_path):
    """The quantized FusedMoE layer (HIP expert kernels) under expert parallelism: two ranks, experts split
    by expert_map_for_rank, tokens exchanged by all_gatherv / reduce_scatterv (gloo, staged through the
    host because both ranks use GPU 0).  Each rank's rows equal the whole layer's on the same tokens up to
    the extra bf16 rounding of the per-rank partial sums."""
    import json
    env = dict(os.environ, EP_OUT=str(tmp_path), EP_MODE="gpu", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "tests", "ep_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    for rank in range(2):
        res = json.load(open(tmp_path / f"ep_rank{rank}.json"))
        assert res["rows"] == [2 + rank, 256], res
        assert res["max_err"] <= 2e-2 * res["scale"] + 2e-3, res
