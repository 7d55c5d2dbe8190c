"""Checkpoint-tensor parameters that know how to take their tensor-parallel shard.

Same class names, constructor keywords and loader methods as vllm/model_executor/parameter.py
(BasevLLMParameter :28-126, column/row loaders :129-230, packed variants :313-395) so the
linear methods can be written exactly as in the reference.  The implementation is one class
that carries optional `input_dim` / `output_dim` / `packed_dim` / `packed_factor` attributes;
the named subclasses only fix which of them are required.

Packing rule (parameter.py:344-350,388-394): when a fused layer's shard offset/size is given in
unpacked output channels and the parameter is packed along that dimension, both are divided by
`packed_factor` (8 nibbles per int32).
"""
from fractions import Fraction

import torch
from torch.nn import Parameter

from .parallel import get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size


class BasevLLMParameter(Parameter):
    _needs = ()

    def __new__(cls, data: torch.Tensor, **kwargs):
        return super().__new__(cls, data=data, requires_grad=False)

    def __init__(self, data: torch.Tensor, weight_loader=None, input_dim: int | None = None,
                 output_dim: int | None = None, packed_dim: int | None = None,
                 packed_factor: int | Fraction | None = None, marlin_tile_size: int | None = None):
        for name in self._needs:
            if locals()[name] is None:
                raise TypeError(f"{type(self).__name__} requires {name}")
        self.weight_loader = weight_loader
        self.input_dim, self.output_dim = input_dim, output_dim
        self.packed_dim, self.packed_factor = packed_dim, packed_factor
        self.marlin_tile_size = marlin_tile_size
        self.tp_rank = get_tensor_model_parallel_rank()
        self.tp_size = get_tensor_model_parallel_world_size()

    # torch.nn.Parameter subclasses must not be re-wrapped by torch functions
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        return super().__torch_function__(func, types, args, kwargs or {})

    # -- helpers -----------------------------------------------------------
    def _store(self, dst: torch.Tensor, src: torch.Tensor) -> None:
        if src.dim() == 0:
            src = src.reshape(1)
        if dst.shape != src.shape:
            raise ValueError(f"{type(self).__name__}: checkpoint shard {tuple(src.shape)} does not fit "
                             f"parameter slice {tuple(dst.shape)}")
        dst.copy_(src)

    def adjust_shard_indexes_for_packing(self, shard_size: int, shard_offset: int):
        f = self.packed_factor
        return shard_size // f, shard_offset // f

    def _packed_on_output(self) -> bool:
        return self.packed_dim is not None and self.packed_dim == self.output_dim

    # -- loaders called by the linear layers --------------------------------
    def load_column_parallel_weight(self, loaded_weight: torch.Tensor) -> None:
        if self.output_dim is None:
            return self._store(self.data, loaded_weight)
        n = self.data.shape[self.output_dim]
        self._store(self.data, loaded_weight.narrow(self.output_dim, self.tp_rank * n, n))

    def load_row_parallel_weight(self, loaded_weight: torch.Tensor) -> None:
        if self.input_dim is None:
            return self._store(self.data, loaded_weight)
        n = self.data.shape[self.input_dim]
        self._store(self.data, loaded_weight.narrow(self.input_dim, self.tp_rank * n, n))

    def load_merged_column_weight(self, loaded_weight: torch.Tensor, shard_offset: int | None = None,
                                  shard_size: int | None = None, **_) -> None:
        if self.output_dim is None or shard_offset is None:
            return self._store(self.data, loaded_weight)
        if self._packed_on_output():
            shard_size, shard_offset = self.adjust_shard_indexes_for_packing(shard_size, shard_offset)
        dst = self.data.narrow(self.output_dim, shard_offset, shard_size)
        self._store(dst, loaded_weight.narrow(self.output_dim, self.tp_rank * shard_size, shard_size))

    def load_qkv_weight(self, loaded_weight: torch.Tensor, shard_offset: int | None = None,
                        shard_size: int | None = None, shard_id: str | None = None, num_heads: int = 1,
                        **_) -> None:
        """`num_heads` is the number of ranks that share one KV head (parameter.py:194):
        q shards are indexed by rank, k/v shards by rank // replicas."""
        if self.output_dim is None or shard_offset is None:
            return self._store(self.data, loaded_weight)
        if self._packed_on_output():
            shard_size, shard_offset = self.adjust_shard_indexes_for_packing(shard_size, shard_offset)
        src_index = self.tp_rank if shard_id == "q" else self.tp_rank // num_heads
        dst = self.data.narrow(self.output_dim, shard_offset, shard_size)
        self._store(dst, loaded_weight.narrow(self.output_dim, src_index * shard_size, shard_size))


class RowvLLMParameter(BasevLLMParameter):
    _needs = ("input_dim",)


class ChannelQuantScaleParameter(BasevLLMParameter):
    _needs = ("output_dim",)


class ModelWeightParameter(BasevLLMParameter):
    _needs = ("input_dim", "output_dim")


class GroupQuantScaleParameter(ModelWeightParameter):
    pass


class PackedColumnParameter(BasevLLMParameter):
    _needs = ("output_dim", "packed_dim", "packed_factor")


class PackedvLLMParameter(ModelWeightParameter):
    _needs = ("input_dim", "output_dim", "packed_dim", "packed_factor")


class BlockQuantScaleParameter(ModelWeightParameter):
    """One scale per (block_n x block_k) tile of the weight (parameter.py:397-403).  The linear
    layers convert shard offsets and sizes from channels to tiles before calling the loaders."""


def adjust_block_scale_shard(weight_block_size, shard_size: int, shard_offset: int) -> tuple[int, int]:
    """Channels -> scale tiles along the output dimension, rounding up (linear.py:85-96)."""
    if weight_block_size is None:
        raise ValueError("a block-quantised layer must set weight_block_size")
    bn = weight_block_size[0]
    return (shard_size + bn - 1) // bn, (shard_offset + bn - 1) // bn


class PerTensorScaleParameter(BasevLLMParameter):
    """One scale per logical matrix of a fused layer (parameter.py:260-310): shard ids
    'q'/'k'/'v' or 0/1/... select the slot."""

    def _slot(self, shard_id) -> int:
        return shard_id if isinstance(shard_id, int) else {"q": 0, "k": 1, "v": 2}[shard_id]

    def _load_slot(self, loaded_weight: torch.Tensor, shard_id=None, **_) -> None:
        if shard_id is None:
            return self._store(self.data, loaded_weight)
        if loaded_weight.dim() != 0:
            if loaded_weight.shape[0] != 1:
                raise ValueError("per-tensor scale must hold one value")
            loaded_weight = loaded_weight[0]
        self.data[self._slot(shard_id)].copy_(loaded_weight)

    load_merged_column_weight = _load_slot
    load_qkv_weight = _load_slot

    def load_column_parallel_weight(self, loaded_weight):
        self._store(self.data, loaded_weight)

    load_row_parallel_weight = load_column_parallel_weight


def permute_param_layout_(param: BasevLLMParameter, input_dim: int, output_dim: int, **kwargs) -> BasevLLMParameter:
    """Bring a 2-D parameter to the requested (input_dim, output_dim[, packed_dim]) order by
    transposing when needed (parameter.py:544-600).  Returns the same object."""
    cur_in, cur_out = param.input_dim, param.output_dim
    if cur_in is None or cur_out is None:
        raise ValueError("permute_param_layout_ needs a parameter with input_dim and output_dim")
    if (cur_in, cur_out) != (input_dim, output_dim):
        if {cur_in, cur_out} != {input_dim, output_dim} or param.data.dim() != 2:
            raise ValueError("only a 2-D transpose is supported")
        param.data = param.data.t().contiguous()
        if param.packed_dim is not None:
            param.packed_dim = input_dim if param.packed_dim == cur_in else output_dim
        param.input_dim, param.output_dim = input_dim, output_dim
    if "packed_dim" in kwargs and param.packed_dim != kwargs["packed_dim"]:
        raise ValueError(f"packed_dim {param.packed_dim} != requested {kwargs['packed_dim']}")
    return param
