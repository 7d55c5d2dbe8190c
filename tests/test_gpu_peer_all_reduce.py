"""The one-shot all-reduce over peer memory (csrc/allreduce.hip, plugin/peer_all_reduce.py) with 2 and 4
ranks that share the one GPU: IPC handles exchanged over gloo, every rank pushes into the others' regions and
sweeps its own.  Against the group's own all-reduce (fp32 sums on the host), bit-exact where every partial
sum is representable, eager and as a replayed graph; no wait may run into its time limit."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("world,port", [(2, 29561), (4, 29563)])
def test_peer_all_reduce_ranks_share_the_gpu(world, port, tmp_path):
    env = dict(os.environ, PEER_AR_OUT=str(tmp_path), OMP_NUM_THREADS="2", PYTHONUNBUFFERED="1",
               MI355Q_PEER_AR_TIMEOUT_MS="3000")
    env.pop("MI355Q_PEER_AR", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "peer_ar_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    for rank in range(world):
        res = json.load(open(tmp_path / f"rank{rank}.json"))
        assert not res["disabled"], res
        assert res["timeouts"] == 0 and res["exact"] and res["graph_value_ok"], res
        assert res["worst_rel_err"] < 1e-2 and res["skewed_series_mismatches"] == 0, res
        assert res["plugin_peer"] and res["plugin_sum_ok"] and res["plugin_gather_ok"], res
        assert res["plugin_large_goes_to_group"] and res["plugin_timeouts"] == 0, res
        print(f"world {world} rank {rank}: {res['us_per_all_reduce_16k']:.2f} us per 16 KiB all-reduce (ranks share one GPU)")


def test_peer_all_reduce_argument_errors():
    import ctypes
    from mi355q import _lib
    L = _lib.lib()
    comm = ctypes.c_void_p()
    assert L.mi355q_peer_ar_init(0, 9, 65536, ctypes.byref(comm)) == _lib.EINVAL
    assert L.mi355q_peer_ar_init(3, 2, 65536, ctypes.byref(comm)) == _lib.EINVAL
    assert L.mi355q_peer_ar_init(0, 2, 24, ctypes.byref(comm)) == _lib.EINVAL
    assert L.mi355q_peer_ar_all_reduce(None, None, None, 8, _lib.BF16, None) == _lib.EINVAL
    assert L.mi355q_peer_ar_handle_bytes() == 64


@pytest.mark.gpu
def test_peer_exchange_under_an_rccl_process_group_of_one_rank(tmp_path):
    """The host side as an N > 1 run executes it under the nccl (RCCL) backend -- handles gathered on the device, the
    construction check against RCCL's own all-reduce and all-gather, a capture while the group's watchdog thread is
    alive -- on the one GPU: a group of one rank that exchanges with itself (MI355Q_PEER_AR_LOOPBACK=1), inside
    bench.py with the TP path forced.  The decode graph then holds the peer kernels, and the tokens are those of the
    plain run."""
    common = ["bench.py", "--gpus", "1", "--model", "small", "--steps", "12", "--warmup", "2", "--skip-cpu", "--skip-bf16",
              "--skip-shapes"]
    env = dict(os.environ, PYTHONUNBUFFERED="1")
    for k in ("MI355Q_FORCE_TP_PATH", "MI355Q_PEER_AR_LOOPBACK", "MI355Q_PEER_AR"):
        env.pop(k, None)
    plain = subprocess.run([sys.executable] + common + ["--tokens-out", str(tmp_path / "plain.json")], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=400)
    assert plain.returncode == 0, plain.stderr[-2000:]
    env.update(MI355Q_FORCE_TP_PATH="1", MI355Q_PEER_AR_LOOPBACK="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29549"] + common + ["--tokens-out", str(tmp_path / "peer.json")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["graph"] is True and line["graph_error"] is None, line
    assert line["tp_exchange"] == "peer memory, one-shot", line
    pa = line["peer_allreduce_us_16k"]
    assert pa["timeouts"] == 0 and pa["captured"] is not None and pa["captured"] > 0, pa
    assert json.load(open(tmp_path / "peer.json")) == json.load(open(tmp_path / "plain.json"))
