"""Tensor-parallel state of this process: one process per GPU, collectives through
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
Stands in for get_tensor_model_parallel_{rank,world_size} and
tensor_model_parallel_all_{reduce,gather} (vllm/distributed/parallel_state.py:662-690,
communication_op.py)."""
import torch
import torch.distributed as dist

_STATE = {"rank": 0, "size": 1, "group": None, "peer": None}


def init_tensor_parallel(rank: int | None = None, size: int | None = None, group=None,
                         peer_all_reduce: bool | None = None) -> None:
    """Adopt the default process group (or `group`) as the TP group.  With no process
    group initialised the state stays at rank 0 of 1.  peer_all_reduce: build the one-shot exchange over
    peer memory for decode-sized tensors (peer_all_reduce.PeerAllReduce; every rank must pass the same
    value; default: when the ranks have GPUs), which the two collectives below then prefer -- what
    CudaCommunicator does with its custom / quick all-reduce (cuda_communicator.py:275-340)."""
    if rank is None or size is None:
        if dist.is_available() and dist.is_initialized():
            rank, size = dist.get_rank(group), dist.get_world_size(group)
        else:
            rank, size = 0, 1
    if not 0 <= rank < size:
        raise ValueError(f"bad tensor-parallel rank {rank} of {size}")
    if _STATE["peer"] is not None:
        _STATE["peer"].close()
    _STATE.update(rank=rank, size=size, group=group, peer=None)
    if peer_all_reduce is None:
        peer_all_reduce = size > 1 and torch.cuda.is_available()
    if peer_all_reduce and size > 1:
        from .peer_all_reduce import PeerAllReduce
        ar = PeerAllReduce(group=group)
        _STATE["peer"] = None if ar.disabled else ar


def get_peer_all_reduce():
    """The group's exchange over peer memory, or None where it could not be set up."""
    return _STATE["peer"]


def get_tensor_model_parallel_rank() -> int:
    return _STATE["rank"]


def get_tensor_model_parallel_world_size() -> int:
    return _STATE["size"]


def tensor_model_parallel_all_reduce(x: torch.Tensor) -> torch.Tensor:
    """Sum over the TP group.  In place on x (the reference returns a new tensor from its
    custom all-reduce; callers here never reuse the input)."""
    if _STATE["size"] > 1:
        peer = _STATE["peer"]
        if peer is not None and peer.should_use(x):
            return peer.all_reduce(x)
        dist.all_reduce(x, group=_STATE["group"])
    return x


def tensor_model_parallel_all_gather(x: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """Concatenate the ranks' shards along `dim` (logits before sampling)."""
    n = _STATE["size"]
    if n == 1:
        return x
    peer = _STATE["peer"]
    if peer is not None and x.dim() == 2 and dim in (-1, 1) and peer.should_gather(x):
        return peer.all_gather(x)
    parts = [torch.empty_like(x) for _ in range(n)]
    dist.all_gather(parts, x.contiguous(), group=_STATE["group"])
    return torch.cat(parts, dim=dim)


# ---- expert parallelism ---------------------------------------------------------------------------
# Ranks of the expert-parallel group each hold their own tokens (data-parallel attention) and a slice of
# the experts.  The exchange is the reference's all-gather / reduce-scatter pair
# (vllm/distributed/device_communicators/all2all.py:44-153 AgRsAll2AllManager, over
# GroupCoordinator.all_gatherv / reduce_scatterv, parallel_state.py): tokens of every rank are gathered,
# each rank runs its experts over all of them, the partial outputs are summed and every rank keeps the
# rows of its own tokens.  Row counts may differ per rank.
_EP = {"rank": 0, "size": 1, "group": None}


def init_expert_parallel(rank: int | None = None, size: int | None = None, group=None) -> None:
    if rank is None or size is None:
        if dist.is_available() and dist.is_initialized():
            rank, size = dist.get_rank(group), dist.get_world_size(group)
        else:
            rank, size = 0, 1
    if not 0 <= rank < size:
        raise ValueError(f"bad expert-parallel rank {rank} of {size}")
    _EP.update(rank=rank, size=size, group=group)


def get_ep_rank() -> int:
    return _EP["rank"]


def get_ep_world_size() -> int:
    return _EP["size"]


def _host_staged(group) -> bool:
    return dist.get_backend(group) == "gloo"


def all_gatherv(tensors, sizes: list[int], dim: int = 0, group=None, size: int | None = None):
    """Concatenate every rank's tensor along dim 0; rank r contributes sizes[r] rows.  A list of tensors
    is gathered entry by entry (one collective each, like the reference's loop over its input list)."""
    if dim != 0:
        raise NotImplementedError("all_gatherv gathers along dim 0")
    group = _EP["group"] if group is None else group
    n = _EP["size"] if size is None else size
    single = isinstance(tensors, torch.Tensor)
    outs = []
    for t in ([tensors] if single else tensors):
        if n == 1:
            outs.append(t)
            continue
        if len(sizes) != n:
            raise ValueError(f"all_gatherv: {len(sizes)} sizes for {n} ranks")
        pad = max(sizes)
        send = t if t.shape[0] == pad else torch.cat((t, t.new_zeros((pad - t.shape[0],) + t.shape[1:])))
        staged = _host_staged(group) and send.is_cuda
        send = send.cpu() if staged else send.contiguous()
        parts = [torch.empty_like(send) for _ in range(n)]
        dist.all_gather(parts, send, group=group)
        out = torch.cat([parts[r][: sizes[r]] for r in range(n)])
        outs.append(out.to(t.device) if staged else out)
    return outs[0] if single else outs


def reduce_scatterv(t: torch.Tensor, sizes: list[int], dim: int = 0, group=None, rank: int | None = None,
                    size: int | None = None) -> torch.Tensor:
    """Sum t over the ranks and keep this rank's sizes[rank] rows (rows are laid out rank after rank, as
    all_gatherv produced them)."""
    if dim != 0:
        raise NotImplementedError("reduce_scatterv scatters along dim 0")
    group = _EP["group"] if group is None else group
    n = _EP["size"] if size is None else size
    r = _EP["rank"] if rank is None else rank
    if n == 1:
        return t
    if t.shape[0] != sum(sizes):
        raise ValueError(f"reduce_scatterv: {t.shape[0]} rows, sizes sum to {sum(sizes)}")
    start = sum(sizes[:r])
    if _host_staged(group) or len(set(sizes)) != 1:
        # gloo has no reduce-scatter, and ragged chunks would need padding to the largest: sum everything,
        # keep the own rows (the row counts that differ are small decode batches)
        buf = t.cpu() if (_host_staged(group) and t.is_cuda) else t.contiguous()
        dist.all_reduce(buf, group=group)
        return buf[start:start + sizes[r]].to(t.device)
    out = torch.empty((sizes[r],) + t.shape[1:], dtype=t.dtype, device=t.device)
    dist.reduce_scatter_tensor(out, t.contiguous(), group=group)
    return out


def divide(a: int, b: int) -> int:
    if a % b:
        raise ValueError(f"{a} is not divisible by {b}")
    return a // b
