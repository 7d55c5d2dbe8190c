"""Tensor-parallel state of this process: one process per GPU, collectives through
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
Stands in for get_tensor_model_parallel_{rank,world_size} and
tensor_model_parallel_all_{reduce,gather} (vllm/distributed/parallel_state.py:662-690,
communication_op.py)."""
import torch
import torch.distributed as dist

_STATE = {"rank": 0, "size": 1, "group": None}


def init_tensor_parallel(rank: int | None = None, size: int | None = None, group=None) -> None:
    """Adopt the default process group (or `group`) as the TP group.  With no process
    group initialised the state stays at rank 0 of 1."""
    if rank is None or size is None:
        if dist.is_available() and dist.is_initialized():
            rank, size = dist.get_rank(group), dist.get_world_size(group)
        else:
            rank, size = 0, 1
    if not 0 <= rank < size:
        raise ValueError(f"bad tensor-parallel rank {rank} of {size}")
    _STATE.update(rank=rank, size=size, group=group)


def get_tensor_model_parallel_rank() -> int:
    return _STATE["rank"]


def get_tensor_model_parallel_world_size() -> int:
    return _STATE["size"]


def tensor_model_parallel_all_reduce(x: torch.Tensor) -> torch.Tensor:
    """Sum over the TP group.  In place on x (the reference returns a new tensor from its
    custom all-reduce; callers here never reuse the input)."""
    if _STATE["size"] > 1:
        dist.all_reduce(x, group=_STATE["group"])
    return x


def tensor_model_parallel_all_gather(x: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """Concatenate the ranks' shards along `dim` (logits before sampling)."""
    n = _STATE["size"]
    if n == 1:
        return x
    parts = [torch.empty_like(x) for _ in range(n)]
    dist.all_gather(parts, x.contiguous(), group=_STATE["group"])
    return torch.cat(parts, dim=dim)


def divide(a: int, b: int) -> int:
    if a % b:
        raise ValueError(f"{a} is not divisible by {b}")
    return a // b
