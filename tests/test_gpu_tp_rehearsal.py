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


@pytest.mark.gpu
def test_rccl_collectives_run_and_capture_at_world_size_one(tmp_path):
    """The RCCL path every rank of an N > 1 run executes, on the one GPU: bench.py under
    torch.distributed.run with backend nccl and MI355Q_FORCE_TP_PATH=1 issues the two all-reduces per layer
    (vllm/model_executor/layers/linear.py:1763-1767) and the logits all-gather
    (vllm/distributed/device_communicators/cuda_communicator.py:275-390) into the process group, captures them
    into the decode graph, and must decode the tokens of the plain run; the 16 KiB all-reduce is timed eager and
    captured."""
    import json
    common = ["bench.py", "--gpus", "1", "--model", "small", "--steps", "12", "--warmup", "2", "--skip-cpu", "--skip-bf16",
              "--skip-shapes"]
    env = dict(os.environ, PYTHONUNBUFFERED="1")
    env.pop("MI355Q_FORCE_TP_PATH", None)
    plain = subprocess.run([sys.executable] + common + ["--tokens-out", str(tmp_path / "plain.json")], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=400)
    assert plain.returncode == 0, plain.stderr[-2000:]
    env["MI355Q_FORCE_TP_PATH"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29547"] + common + ["--tokens-out", str(tmp_path / "rccl.json")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["graph"] is True and line["graph_error"] is None, line
    ar = line["allreduce_us_16k"]
    assert ar["eager"] > 0 and ar["captured"] is not None and ar["captured"] > 0, ar
    assert json.load(open(tmp_path / "rccl.json")) == json.load(open(tmp_path / "plain.json"))
