"""Tensor-parallel linear layers that call into a quantization method.

Mirrors the caller side of the hot path, vllm/model_executor/layers/linear.py: LinearMethodBase
:133-181, UnquantizedLinearMethod :184-230, LinearBase :233-310, ColumnParallelLinear :419-616,
MergedColumnParallelLinear :661-1020, QKVParallelLinear :1022-1426, RowParallelLinear
:1613-1782, and of the output embedding, vocab_parallel_embedding.py: UnquantizedEmbeddingMethod :35-84,
ParallelLMHead :521-583 with the vocabulary-sharded loader :445-485.  Only the v2 weight-loading protocol is provided (every method on this path is in
WEIGHT_LOADER_V2_SUPPORTED, :49-66): checkpoint tensors arrive through
`layer.weight_loader(param, tensor[, shard_id])` and the parameter object narrows its own
tensor-parallel shard.
"""
from abc import abstractmethod

import torch
from torch import nn

from .. import ops
from .parallel import (divide, get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                       tensor_model_parallel_all_gather, tensor_model_parallel_all_reduce)
from .parameter import (BasevLLMParameter, BlockQuantScaleParameter, ModelWeightParameter,
                        PerTensorScaleParameter, adjust_block_scale_shard)
from .registry import QuantizationConfig, QuantizeMethodBase


class LinearMethodBase(QuantizeMethodBase):
    @abstractmethod
    def create_weights(self, layer: nn.Module, input_size_per_partition: int, output_partition_sizes: list[int],
                       input_size: int, output_size: int, params_dtype: torch.dtype, **extra_weight_attrs): ...

    @abstractmethod
    def apply(self, layer: nn.Module, x: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor: ...


class UnquantizedLinearMethod(LinearMethodBase):
    """16-bit weight [N, K].  Few tokens go to the skinny GEMV kernel, many to the BLAS GEMM
    (the split rocm_unquantized_gemm makes, vllm/model_executor/layers/utils.py:116-186)."""

    SKINNY_MAX_TOKENS = 8

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs):
        w = ModelWeightParameter(data=torch.empty(sum(output_partition_sizes), input_size_per_partition,
                                                  dtype=params_dtype),
                                 input_dim=1, output_dim=0, weight_loader=extra_weight_attrs.get("weight_loader"))
        layer.register_parameter("weight", w)

    def apply(self, layer, x, bias=None):
        tokens = x.numel() // x.shape[-1]
        if x.is_cuda and tokens <= self.SKINNY_MAX_TOKENS and x.dtype in (torch.bfloat16, torch.float16):
            return ops.linear_skinny(x, layer.weight, bias)
        return torch.nn.functional.linear(x, layer.weight, bias)


class LinearBase(nn.Module):
    def __init__(self, input_size: int, output_size: int, skip_bias_add: bool = False,
                 params_dtype: torch.dtype | None = None, quant_config: QuantizationConfig | None = None,
                 prefix: str = "", *, return_bias: bool = True):
        super().__init__()
        self.input_size, self.output_size = input_size, output_size
        self.skip_bias_add, self.return_bias, self.prefix = skip_bias_add, return_bias, prefix
        self.params_dtype = params_dtype or torch.get_default_dtype()
        self.quant_config = quant_config
        self.tp_rank = get_tensor_model_parallel_rank()
        self.tp_size = get_tensor_model_parallel_world_size()
        method = quant_config.get_quant_method(self, prefix=prefix) if quant_config is not None else None
        self.quant_method: QuantizeMethodBase = method or UnquantizedLinearMethod()

    def _make_bias(self, n: int, bias: bool, output_dim_sharded: bool):
        if not bias:
            self.register_parameter("bias", None)
            return
        b = BasevLLMParameter(data=torch.zeros(n, dtype=self.params_dtype), weight_loader=self.weight_loader,
                              output_dim=0 if output_dim_sharded else None)
        self.register_parameter("bias", b)

    def _ret(self, out, bias_used):
        if not self.return_bias:
            return out
        return out, (self.bias if self.skip_bias_add else None)

    def process_weights_after_loading(self) -> None:
        self.quant_method.process_weights_after_loading(self)


class ColumnParallelLinear(LinearBase):
    """Y = X W with W's output channels split over ranks (linear.py:419-616)."""

    def __init__(self, input_size: int, output_size: int, bias: bool = True, gather_output: bool = False,
                 skip_bias_add: bool = False, params_dtype=None, quant_config=None,
                 output_sizes: list[int] | None = None, prefix: str = "", *, return_bias: bool = True):
        super().__init__(input_size, output_size, skip_bias_add, params_dtype, quant_config, prefix,
                         return_bias=return_bias)
        self.gather_output = gather_output
        self.input_size_per_partition = input_size
        self.output_size_per_partition = divide(output_size, self.tp_size)
        self.output_partition_sizes = [self.output_size_per_partition]
        if hasattr(self, "output_sizes"):
            self.output_partition_sizes = self._partition_sizes()
        self.quant_method.create_weights(layer=self, input_size_per_partition=self.input_size_per_partition,
                                         output_partition_sizes=self.output_partition_sizes,
                                         input_size=self.input_size, output_size=self.output_size,
                                         params_dtype=self.params_dtype, weight_loader=self.weight_loader)
        self._make_bias(self.output_size_per_partition, bias, True)

    def _partition_sizes(self) -> list[int]:
        return [divide(n, self.tp_size) for n in self.output_sizes]

    def weight_loader(self, param: BasevLLMParameter, loaded_weight: torch.Tensor, loaded_shard_id=None):
        if loaded_weight.dim() == 0:
            loaded_weight = loaded_weight.reshape(1)
        param.load_column_parallel_weight(loaded_weight=loaded_weight)

    weight_loader_v2 = weight_loader

    def forward(self, input_):
        bias = self.bias if not self.skip_bias_add else None
        out = self.quant_method.apply(self, input_, bias)
        if self.gather_output and self.tp_size > 1:
            out = tensor_model_parallel_all_gather(out)
        return self._ret(out, bias)


class MergedColumnParallelLinear(ColumnParallelLinear):
    """Several column-parallel matrices sharing one input, stored side by side along the
    output dimension (gate_up_proj).  `loaded_shard_id` i selects output_sizes[i]."""

    def __init__(self, input_size: int, output_sizes: list[int], bias: bool = True, gather_output: bool = False,
                 skip_bias_add: bool = False, params_dtype=None, quant_config=None, prefix: str = "", *,
                 return_bias: bool = True):
        self.output_sizes = list(output_sizes)
        super().__init__(input_size, sum(output_sizes), bias, gather_output, skip_bias_add, params_dtype,
                         quant_config, prefix=prefix, return_bias=return_bias)

    def weight_loader(self, param: BasevLLMParameter, loaded_weight: torch.Tensor, loaded_shard_id: int | None = None):
        if loaded_shard_id is None:
            # tensor already fused on disk: split it into the logical matrices
            if isinstance(param, PerTensorScaleParameter) or param.output_dim is None:
                for i in range(len(self.output_sizes) if isinstance(param, PerTensorScaleParameter) else 1):
                    param.load_merged_column_weight(loaded_weight=loaded_weight,
                                                    shard_id=i if isinstance(param, PerTensorScaleParameter) else None)
                return
            off = 0
            for i, n in enumerate(self.output_sizes):
                size, start = n, off
                if isinstance(param, BlockQuantScaleParameter):
                    size, start = adjust_block_scale_shard(getattr(self, "weight_block_size", None), n, off)
                elif param.packed_dim is not None and param.packed_dim == param.output_dim:
                    size, start = param.adjust_shard_indexes_for_packing(n, off)
                self.weight_loader(param, loaded_weight.narrow(param.output_dim, start, size), i)
                off += n
            return
        if not 0 <= loaded_shard_id < len(self.output_sizes):
            raise ValueError(f"shard id {loaded_shard_id} out of range for {len(self.output_sizes)} matrices")
        offset = sum(self.output_sizes[:loaded_shard_id]) // self.tp_size
        size = self.output_sizes[loaded_shard_id] // self.tp_size
        if isinstance(param, BlockQuantScaleParameter):
            size, offset = adjust_block_scale_shard(getattr(self, "weight_block_size", None), size, offset)
        param.load_merged_column_weight(loaded_weight=loaded_weight, shard_id=loaded_shard_id,
                                        shard_offset=offset, shard_size=size)

    weight_loader_v2 = weight_loader


class QKVParallelLinear(ColumnParallelLinear):
    """Fused q|k|v projection.  Query heads are split over ranks; KV heads are split when
    there are at least tp_size of them and replicated otherwise (linear.py:1060-1100)."""

    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int, total_num_kv_heads: int | None = None,
                 bias: bool = True, skip_bias_add: bool = False, params_dtype=None, quant_config=None,
                 prefix: str = "", *, return_bias: bool = True):
        tp = get_tensor_model_parallel_world_size()
        self.hidden_size, self.head_size, self.v_head_size = hidden_size, head_size, head_size
        self.total_num_heads = total_num_heads
        self.total_num_kv_heads = total_num_kv_heads if total_num_kv_heads is not None else total_num_heads
        self.num_heads = divide(total_num_heads, tp)
        if tp >= self.total_num_kv_heads:
            self.num_kv_heads, self.num_kv_head_replicas = 1, divide(tp, self.total_num_kv_heads)
        else:
            self.num_kv_heads, self.num_kv_head_replicas = divide(self.total_num_kv_heads, tp), 1
        # sizes as if every rank held its own KV heads, so that size / tp is the per-rank width
        self.output_sizes = [self.num_heads * head_size * tp, self.num_kv_heads * head_size * tp,
                             self.num_kv_heads * head_size * tp]
        super().__init__(hidden_size, sum(self.output_sizes), bias, False, skip_bias_add, params_dtype,
                         quant_config, prefix=prefix, return_bias=return_bias)

    def _shard(self, shard_id: str) -> tuple[int, int]:
        q, kv = self.num_heads * self.head_size, self.num_kv_heads * self.head_size
        return {"q": (0, q), "k": (q, kv), "v": (q + kv, kv)}[shard_id]

    def weight_loader(self, param: BasevLLMParameter, loaded_weight: torch.Tensor, loaded_shard_id: str | None = None):
        if loaded_shard_id is None:
            if isinstance(param, PerTensorScaleParameter):
                for i in range(param.data.shape[0]):
                    param.load_qkv_weight(loaded_weight=loaded_weight, shard_id=i)
                return
            if param.output_dim is None:
                return param.load_qkv_weight(loaded_weight=loaded_weight)
            off = 0
            for sid, n in (("q", self.total_num_heads * self.head_size),
                           ("k", self.total_num_kv_heads * self.head_size),
                           ("v", self.total_num_kv_heads * self.head_size)):
                size, start = n, off
                if isinstance(param, BlockQuantScaleParameter):
                    size, start = adjust_block_scale_shard(getattr(self, "weight_block_size", None), n, off)
                elif param.packed_dim is not None and param.packed_dim == param.output_dim:
                    size, start = param.adjust_shard_indexes_for_packing(n, off)
                self.weight_loader(param, loaded_weight.narrow(param.output_dim, start, size), sid)
                off += n
            return
        if loaded_shard_id not in ("q", "k", "v"):
            raise ValueError(f"Shard id for QKVParallelLinear should be 'q', 'k', or 'v', got {loaded_shard_id}.")
        offset, size = self._shard(loaded_shard_id)
        if isinstance(param, BlockQuantScaleParameter):
            size, offset = adjust_block_scale_shard(getattr(self, "weight_block_size", None), size, offset)
        param.load_qkv_weight(loaded_weight=loaded_weight, num_heads=self.num_kv_head_replicas,
                              shard_id=loaded_shard_id, shard_offset=offset, shard_size=size)

    weight_loader_v2 = weight_loader


class RowParallelLinear(LinearBase):
    """Y = X W with W's input channels split over ranks; partial sums are all-reduced.
    The bias is added on rank 0 only, before the reduction (linear.py:1760-1767)."""

    def __init__(self, input_size: int, output_size: int, bias: bool = True, input_is_parallel: bool = True,
                 skip_bias_add: bool = False, params_dtype=None, reduce_results: bool = True, quant_config=None,
                 prefix: str = "", *, return_bias: bool = True):
        super().__init__(input_size, output_size, skip_bias_add, params_dtype, quant_config, prefix,
                         return_bias=return_bias)
        self.input_is_parallel, self.reduce_results = input_is_parallel, reduce_results
        self.input_size_per_partition = divide(input_size, self.tp_size)
        self.output_size_per_partition = output_size
        self.output_partition_sizes = [output_size]
        if not reduce_results and bias and not skip_bias_add:
            raise ValueError("When not reduce the results, adding bias to the results can lead to incorrect results")
        self.quant_method.create_weights(layer=self, input_size_per_partition=self.input_size_per_partition,
                                         output_partition_sizes=self.output_partition_sizes,
                                         input_size=self.input_size, output_size=self.output_size,
                                         params_dtype=self.params_dtype, weight_loader=self.weight_loader)
        self._make_bias(output_size, bias, False)

    def weight_loader(self, param: BasevLLMParameter, loaded_weight: torch.Tensor, loaded_shard_id=None):
        if loaded_weight.dim() == 0:
            loaded_weight = loaded_weight.reshape(1)
        param.load_row_parallel_weight(loaded_weight=loaded_weight)

    weight_loader_v2 = weight_loader

    def forward(self, input_):
        if not self.input_is_parallel and self.tp_size > 1:
            k = self.input_size_per_partition
            input_ = input_.narrow(-1, self.tp_rank * k, k).contiguous()
        bias = None if (self.tp_rank > 0 or self.skip_bias_add) else self.bias
        out = self.quant_method.apply(self, input_, bias)
        if self.reduce_results and self.tp_size > 1:
            out = tensor_model_parallel_all_reduce(out)
        return self._ret(out, bias)


DEFAULT_VOCAB_PADDING_SIZE = 64


def pad_vocab_size(vocab_size: int, pad_to: int = DEFAULT_VOCAB_PADDING_SIZE) -> int:
    return ((vocab_size + pad_to - 1) // pad_to) * pad_to


class UnquantizedEmbeddingMethod(QuantizeMethodBase):
    """16-bit [vocab / tp, hidden] weight of an embedding / output head (vocab_parallel_embedding.py:35-84)."""

    def create_weights(self, layer, input_size_per_partition, output_partition_sizes, input_size, output_size,
                       params_dtype, **extra_weight_attrs):
        w = ModelWeightParameter(data=torch.zeros(sum(output_partition_sizes), input_size_per_partition,
                                                  dtype=params_dtype),
                                 input_dim=1, output_dim=0, weight_loader=extra_weight_attrs.get("weight_loader"))
        layer.register_parameter("weight", w)

    def apply(self, layer, x, bias=None):
        return UnquantizedLinearMethod.apply(self, layer, x, bias)

    SKINNY_MAX_TOKENS = UnquantizedLinearMethod.SKINNY_MAX_TOKENS

    def embedding(self, layer, input_):
        return torch.nn.functional.embedding(input_, layer.weight)


class ParallelLMHead(nn.Module):
    """Output embedding split over ranks along the vocabulary (padded to a multiple of 64 so that every
    rank holds the same number of rows).  Its weights are used by the logits computation through
    `quant_method.apply(lm_head, hidden)`; a quantization config that says `lm_head: true` gives it a
    quantized linear method, whose packed tensors are sharded along their output dimension with the
    packing factor taken into account (vocab_parallel_embedding.py:445-485)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, bias: bool = False, params_dtype=None,
                 org_num_embeddings: int | None = None, padding_size: int = DEFAULT_VOCAB_PADDING_SIZE,
                 quant_config: QuantizationConfig | None = None, prefix: str = ""):
        super().__init__()
        self.tp_rank, self.tp_size = get_tensor_model_parallel_rank(), get_tensor_model_parallel_world_size()
        self.num_embeddings, self.embedding_dim, self.prefix = num_embeddings, embedding_dim, prefix
        self.org_vocab_size = org_num_embeddings or num_embeddings
        if self.org_vocab_size != num_embeddings:
            raise NotImplementedError("added (LoRA) vocabulary is not supported")
        self.num_embeddings_padded = pad_vocab_size(num_embeddings, padding_size * self.tp_size)
        self.num_embeddings_per_partition = self.num_embeddings_padded // self.tp_size
        self.vocab_start = min(self.tp_rank * self.num_embeddings_per_partition, self.org_vocab_size)
        self.vocab_end = min(self.vocab_start + self.num_embeddings_per_partition, self.org_vocab_size)
        self.params_dtype = params_dtype or torch.get_default_dtype()
        method = quant_config.get_quant_method(self, prefix=prefix) if quant_config is not None else None
        self.quant_method: QuantizeMethodBase = method or UnquantizedEmbeddingMethod()
        self.quant_method.create_weights(self, embedding_dim, [self.num_embeddings_per_partition], embedding_dim,
                                         self.num_embeddings_padded, params_dtype=self.params_dtype,
                                         weight_loader=self.weight_loader)
        if bias:
            self.bias = nn.Parameter(torch.zeros(self.num_embeddings_per_partition, dtype=self.params_dtype),
                                     requires_grad=False)
            self.bias.output_dim, self.bias.weight_loader = 0, self.weight_loader
        else:
            self.register_parameter("bias", None)

    def weight_loader(self, param, loaded_weight: torch.Tensor, loaded_shard_id=None) -> None:
        output_dim = getattr(param, "output_dim", None)
        if output_dim is None:      # replicated on every rank (g_idx of an activation-ordered GPTQ head)
            if loaded_weight.dim() == 0:
                loaded_weight = loaded_weight.reshape(1)
            if param.data.shape != loaded_weight.shape:
                raise ValueError(f"{self.prefix}: replicated tensor of shape {tuple(loaded_weight.shape)} for a "
                                 f"parameter of shape {tuple(param.data.shape)}")
            param.data.copy_(loaded_weight)
            return
        start, size, full = self.vocab_start, self.vocab_end - self.vocab_start, self.org_vocab_size
        if getattr(param, "packed_dim", None) == output_dim:
            f = param.packed_factor
            start, size, full = start // f, size // f, full // f
        if loaded_weight.shape[output_dim] != full:
            raise ValueError(f"{self.prefix}: checkpoint tensor has {loaded_weight.shape[output_dim]} entries along the "
                             f"vocabulary, expected {full}")
        shard = loaded_weight.narrow(output_dim, start, size)
        param.data.narrow(output_dim, 0, size).copy_(shard)
        rest = param.data.shape[output_dim] - size
        if rest:      # padding rows of the last rank
            param.data.narrow(output_dim, size, rest).zero_()

    weight_loader_v2 = weight_loader

    def process_weights_after_loading(self) -> None:
        self.quant_method.process_weights_after_loading(self)

    def forward(self, input_):
        raise RuntimeError("LMHead's weights should be used in the sampler.")
