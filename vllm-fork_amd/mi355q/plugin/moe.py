"""Mixture-of-experts layers with 4-bit experts (MoE WNA16).

Mirrors vllm/model_executor/layers/quantization/moe_wna16.py:47-197 (MoeWNA16Config: GPTQ or AWQ
checkpoints routed here because the Marlin MoE kernels do not exist on ROCm), :204-360 (weight
shapes: w13 uint8 [E, 2I, H/2], w2 uint8 [E, H, I/2], two 4-bit values per byte along the input
dimension with the even index in the low nibble; scales [E, out, in/G]; zeros uint8
[E, out/2, in/G]; the group size is halved until it divides both H and I), :516-638 (the loader's
AWQ / GPTQ tensor conversions) and the expert computation of fused_moe.py:1650-1850 as
tests/kernels/utils.py:912-1078 states it: per (token, selected expert) slot
y = (silu(x W1^T) * (x W3^T)) W2^T, then out = sum_j weight_j * y_j in fp32.
"""
from typing import Any

import torch

from .. import ops
from .awq import AutoAWQConfig
from .gptq import AutoGPTQConfig
from .linear import LinearBase, UnquantizedLinearMethod
from . import parallel
from .parallel import (get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                       tensor_model_parallel_all_reduce)
from .registry import QuantizationConfig, QuantizeMethodBase, register_quantization_config

# Routes for the routed experts:
#   slots (tokens x topk) <= local experts: every slot streams its own expert's weights (decode GEMV
#     with the expert indirection); no token sort, capturable;
#   more slots, up to ALL_EXPERTS_MAX_TOKENS tokens: every local expert runs over ALL tokens (its
#     weights are read once, and at these sizes a projection costs what its weights cost to
#     stream, so rows an expert was not selected for are free); the combine picks each token's
#     selected rows.  No data-dependent shapes: capturable;
#   larger batches ("grouped"): slots sorted by expert on the device, every expert's run padded to whole
#     row tiles, one grouped GEMM launch per projection over all experts (the tiled W4A16 kernel with a
#     tile -> expert indirection); no host round trip, capturable;
#   "grouped_host", only for layer shapes the tiled kernel does not take, outside graph capture: tokens
#     grouped per expert on the host, dequantise + dense GEMM per expert.
ALL_EXPERTS_MAX_TOKENS = 8      # measured on a Mixtral-sized layer: 332 us at 9 tokens against 350-370 us token-sorted
ALL_EXPERTS_MAX_TOKENS_UNSORTED = 64   # layers the token-sorted kernel does not take
FORCE_ROUTE: str | None = None   # tests: "slots" | "all_experts" | "grouped" | "grouped_host"


def convert_awq_tensor(tensor: torch.Tensor, tensor_type: str) -> torch.Tensor:
    """AWQ int32 [a, b] (nibble order [0,2,4,6,1,3,5,7] along the packed output dimension) ->
    uint8 output-major: qweight [8b, a/2], qzeros [4b, a] (moe_wna16.py:517-553)."""
    rows = tensor.size(0)
    t = tensor.contiguous().view(torch.uint8)
    t = torch.stack((t & 0xF, t >> 4), dim=-1)                      # [a, 4b, 2] nibbles, low first
    t = t.view(-1, 8)[:, [0, 4, 1, 5, 2, 6, 3, 7]].reshape(rows, -1)  # logical output order
    t = t.t().contiguous()                                          # [8b, a]
    if tensor_type == "qweight":
        return t[:, 1::2] * 16 + t[:, ::2]
    if tensor_type == "qzeros":
        return t[1::2, :] * 16 + t[::2, :]
    raise ValueError(tensor_type)


def convert_gptq_int4_qzeros(tensor: torch.Tensor) -> torch.Tensor:
    """GPTQ v1 zero points store z - 1: add one per nibble (moe_wna16.py:555-561)."""
    t = tensor.contiguous().view(torch.uint8)
    lo, hi = (t & 0xF) + 1, (t >> 4) + 1
    return lo + hi * 16


@register_quantization_config("moe_wna16")
class MoeWNA16Config(QuantizationConfig):
    def __init__(self, linear_quant_method: str, weight_bits: int, group_size: int, has_zp: bool,
                 lm_head_quantized: bool, modules_to_not_convert: list[str] | None, full_config: dict[str, Any]) -> None:
        super().__init__()
        if linear_quant_method not in ("gptq", "awq", "awq_marlin"):
            raise ValueError("moe_wna16 only support gptq and awq.")
        self.linear_quant_method = linear_quant_method
        self.weight_bits, self.group_size, self.has_zp = weight_bits, group_size, has_zp
        self.bit8_pack_factor = 8 // weight_bits
        self.lm_head_quantized = lm_head_quantized
        self.modules_to_not_convert = modules_to_not_convert or []
        self.full_config = full_config

    @classmethod
    def get_name(cls) -> str:
        return "moe_wna16"

    @classmethod
    def get_supported_act_dtypes(cls) -> list[torch.dtype]:
        return [torch.bfloat16, torch.half]

    @classmethod
    def get_min_capability(cls) -> int:
        return 70

    @classmethod
    def get_config_filenames(cls) -> list[str]:
        return ["quantize_config.json"]

    @classmethod
    def from_config(cls, config: dict[str, Any]) -> "MoeWNA16Config":
        method = cls.get_from_keys(config, ["quant_method"])
        bits = cls.get_from_keys(config, ["bits"])
        group_size = cls.get_from_keys(config, ["group_size"])
        lm_head = cls.get_from_keys_or(config, ["lm_head"], default=False)
        if method == "gptq":
            has_zp, skip = not cls.get_from_keys(config, ["sym"]), []
        elif method in ("awq", "awq_marlin"):
            has_zp = cls.get_from_keys(config, ["zero_point"])
            skip = cls.get_from_keys_or(config, ["modules_to_not_convert"], None)
        else:
            raise ValueError("moe_wna16 only support gptq and awq.")
        return cls(method, bits, group_size, has_zp, lm_head, skip, config)

    @classmethod
    def is_moe_wna16_compatible(cls, quant_config: dict[str, Any]) -> bool:
        method = str(quant_config.get("quant_method", "")).lower()
        bits = quant_config.get("bits")
        return (method == "gptq" and not quant_config.get("desc_act") and bits in (4, 8)) or \
            (method == "awq" and bits == 4)

    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant, hf_config=None) -> str | None:
        if cls.is_moe_wna16_compatible(hf_quant_cfg) and user_quant == "moe_wna16":
            return cls.get_name()
        return None

    def get_quant_method(self, layer: torch.nn.Module, prefix: str):
        if any(m in prefix for m in self.modules_to_not_convert):
            return None if isinstance(layer, FusedMoE) else UnquantizedLinearMethod()
        if isinstance(layer, LinearBase):
            inner = AutoGPTQConfig if self.linear_quant_method == "gptq" else AutoAWQConfig
            return inner.from_config(self.full_config).get_quant_method(layer, prefix)
        if isinstance(layer, FusedMoE):
            return MoeWNA16Method(self)
        return None


class MoeWNA16Method(QuantizeMethodBase):
    def __init__(self, quant_config: MoeWNA16Config):
        if quant_config.weight_bits != 4:
            raise ValueError("MoeWNA16Method on MI355X supports int4 experts only")
        self.quant_config = quant_config

    def create_weights(self, layer, num_experts: int, hidden_size: int, intermediate_size_per_partition: int,
                       params_dtype: torch.dtype, **extra_weight_attrs):
        qc = self.quant_config
        pf = qc.bit8_pack_factor
        G, div = qc.group_size if qc.group_size != -1 else hidden_size, 1
        while intermediate_size_per_partition % G or hidden_size % G:
            G //= 2
            div *= 2
            if G < 32:
                raise ValueError("group size cannot be made to divide the hidden and intermediate sizes")
        layer.group_size, layer.group_size_div_factor = G, div
        I, H, E = intermediate_size_per_partition, hidden_size, num_experts

        def reg(name, t):
            p = torch.nn.Parameter(t, requires_grad=False)
            p.weight_loader = self.get_weight_loader(layer)
            layer.register_parameter(name, p)

        reg("w13_qweight", torch.empty(E, 2 * I, H // pf, dtype=torch.uint8))
        reg("w2_qweight", torch.empty(E, H, I // pf, dtype=torch.uint8))
        reg("w13_scales", torch.zeros(E, 2 * I, H // G, dtype=params_dtype))
        reg("w2_scales", torch.zeros(E, H, I // G, dtype=params_dtype))
        if qc.has_zp:
            reg("w13_qzeros", torch.zeros(E, 2 * I // pf, H // G, dtype=torch.uint8))
            reg("w2_qzeros", torch.zeros(E, H // pf, I // G, dtype=torch.uint8))

    def get_weight_loader(self, layer):
        qc = self.quant_config

        def loader(param, loaded_weight: torch.Tensor, weight_name: str, shard_id: str, expert_id: int,
                   return_success: bool = False):
            """shard_id: "w1" (gate), "w3" (up) or "w2" (down); tensors arrive in checkpoint layout."""
            if "g_idx" in weight_name or (not qc.has_zp and "qzeros" in weight_name):
                return False if return_success else None
            is_w, is_z = "qweight" in weight_name, "qzeros" in weight_name
            t = loaded_weight
            if qc.linear_quant_method in ("awq", "awq_marlin"):
                t = convert_awq_tensor(t, "qweight") if is_w else convert_awq_tensor(t, "qzeros") if is_z else t.t()
            else:
                if is_w:
                    t = t.t().contiguous().view(torch.uint8)
                elif is_z:
                    t = convert_gptq_int4_qzeros(t).t()
                else:
                    t = t.t()
            if layer.group_size_div_factor > 1 and not is_w:
                t = t.repeat_interleave(layer.group_size_div_factor, 1)
            rank, size = layer.tp_rank, layer.tp_size
            dst = param.data[expert_id]
            if shard_id in ("w1", "w3"):           # column parallel: shard the output rows
                rows = dst.shape[0] // 2
                src = t.reshape(size, -1, t.shape[1])[rank]
                dst[:rows].copy_(src) if shard_id == "w1" else dst[rows:].copy_(src)
            elif shard_id == "w2":                 # row parallel: shard the input columns
                dst.copy_(t.reshape(t.shape[0], size, -1)[:, rank])
            else:
                raise ValueError(f"unknown expert shard id {shard_id}")
            return True if return_success else None

        return loader

    def process_weights_after_loading(self, layer) -> None:
        z13 = layer.w13_qzeros.data if self.quant_config.has_zp else None
        z2 = layer.w2_qzeros.data if self.quant_config.has_zp else None
        slots = int(getattr(layer, "top_k", 1))
        layer.w13_packed = ops.moe_w4a16_repack(layer.w13_qweight.data, layer.w13_scales.data, z13, layer.group_size,
                                                pair_halves=True, decode_slots=slots)
        layer.w2_packed = ops.moe_w4a16_repack(layer.w2_qweight.data, layer.w2_scales.data, z2, layer.group_size,
                                               decode_slots=slots)
        for name in ("w13_qweight", "w2_qweight", "w13_scales", "w2_scales", "w13_qzeros", "w2_qzeros"):
            if name in layer._parameters:
                del layer._parameters[name]

    def apply(self, layer, x: torch.Tensor, topk_weights: torch.Tensor, topk_ids: torch.Tensor) -> torch.Tensor:
        return fused_experts(x, layer.w13_packed, layer.w2_packed, topk_weights, topk_ids, layer.expert_map)


def fused_topk(gating_output: torch.Tensor, topk: int, renormalize: bool):
    """softmax + top-k routing (fused_moe `fused_topk`): (fp32 weights [T, topk], int32 ids [T, topk])."""
    T = gating_output.shape[0]
    w = torch.empty((T, topk), dtype=torch.float32, device=gating_output.device)
    ids = torch.empty((T, topk), dtype=torch.int32, device=gating_output.device)
    ops.topk_softmax(w, ids, None, gating_output.contiguous(), renormalize)
    return w, ids


def fused_experts(x: torch.Tensor, w13: ops.PackedMoeW4, w2: ops.PackedMoeW4, topk_weights: torch.Tensor,
                  topk_ids: torch.Tensor, expert_map: torch.Tensor | None = None, combine: bool = True) -> torch.Tensor:
    """combine=False: the selected experts' outputs [T, topk, H] without the weighted sum (a caller that folds the
    sum into its next kernel)."""
    T, H = x.shape
    topk = topk_ids.shape[1]
    ids = topk_ids.reshape(-1)
    if expert_map is not None:             # global -> local expert index, -1 when held elsewhere
        ids = expert_map[ids.long()].to(torch.int32)
    ids = ids.contiguous()
    slots = T * topk
    capturing = torch.cuda.is_current_stream_capturing()
    if FORCE_ROUTE is not None:
        route = FORCE_ROUTE
    elif slots <= w13.E:
        route = "slots"
    elif T <= ALL_EXPERTS_MAX_TOKENS:
        route = "all_experts"
    else:
        route = "grouped"
    if route == "grouped" and not (ops.moe_grouped_mm_supported(w13) and ops.moe_grouped_mm_supported(w2)):
        # layer shapes the tiled kernel does not take
        route = "all_experts" if capturing or T <= ALL_EXPERTS_MAX_TOKENS_UNSORTED else "grouped_host"
    if route == "all_experts":
        E = w13.E
        ys = torch.empty((E + 1, T, H), dtype=x.dtype, device=x.device)
        ys[E].zero_()                                   # what a slot of an absent expert contributes
        act = torch.empty((T, w2.K), dtype=x.dtype, device=x.device)
        for e in range(E):
            w13_e = w13.expert(e)
            if not w13.pair_halves:
                ops.silu_and_mul(act, ops.w4a16_gemm(x, w13_e))
            elif T < ops.mm_min_tokens(w13_e) or not ops._mm_supported(w13_e):   # SiluAndMul in the projection's epilogue
                act, _ = ops.w4a16_gate_up_silu(x, None, None, 0.0, w13_e, add_norm=False)
            else:
                act = ops.w4a16_mm(x, w13_e, silu_pairs=True)
            ops.w4a16_gemm(act, w2.expert(e), out=ys[e])
        sel = ids.view(T, topk).long()
        sel = torch.where(sel < 0, torch.full_like(sel, E), sel)
        y = ys[sel, torch.arange(T, device=x.device)[:, None]].contiguous()       # [T, topk, H]
    elif route == "slots" and w13.pair_halves:
        act = ops.moe_w4a16_gemm(x, w13, ids, topk, silu_pairs=True)   # SiluAndMul in the epilogue
        y = ops.moe_w4a16_gemm(act, w2, ids, 1)
    elif route == "slots":
        h = ops.moe_w4a16_gemm(x, w13, ids, topk)
        if w2.G % 128 == 0:      # SiluAndMul runs in the prologue of the second GEMM
            y = ops.moe_w4a16_gemm(h, w2, ids, 1, fuse_silu_mul=True)
        else:                    # small scale groups: the prologue is not built for them
            act = torch.empty((slots, w2.K), dtype=x.dtype, device=x.device)
            ops.silu_and_mul(act, h)
            y = ops.moe_w4a16_gemm(act, w2, ids, 1)
    elif route == "grouped":
        # token-sorted rows, one launch per projection over all experts (fused_experts_impl,
        # fused_moe.py:1650-1850: moe_align_block_size, two grouped GEMMs, activation between them); no host
        # round trip, so the route can be captured
        sorted_ids, tile_expert, num_tiles, slot_pos = ops.moe_align_block_size(ids, w13.E)
        src = torch.clamp(sorted_ids, max=slots - 1).long() // topk        # padding rows read some valid token
        xs = x.index_select(0, src)
        if w13.pair_halves:
            act = ops.moe_w4a16_grouped_mm(xs, w13, tile_expert, num_tiles, silu_pairs=True)
        else:
            h = ops.moe_w4a16_grouped_mm(xs, w13, tile_expert, num_tiles)
            act = torch.empty((h.shape[0], h.shape[1] // 2), dtype=x.dtype, device=x.device)
            ops.silu_and_mul(act, h)
        ys = ops.moe_w4a16_grouped_mm(act, w2, tile_expert, num_tiles)
        held = slot_pos >= 0
        y = ys.index_select(0, torch.clamp(slot_pos, min=0).long())        # back to slot order
        if expert_map is not None:
            # slots of experts held elsewhere add nothing.  Selected, not multiplied: their rows were gathered
            # from row 0 of `ys`, which no tile wrote when this rank holds none of the selected experts, and
            # uninitialised memory times zero may be NaN
            y = torch.where(held.view(-1, 1), y, torch.zeros((), dtype=y.dtype, device=y.device))
    else:
        y = torch.zeros((slots, H), dtype=x.dtype, device=x.device)
        for e in ids.unique().tolist():
            if e < 0:
                continue
            rows = (ids == e).nonzero().view(-1)
            h = torch.matmul(x.index_select(0, rows // topk), ops.w4a16_dequantize(w13.expert(e)))
            if w13.pair_halves:
                h = ops.unpair_halves(h)
            act = torch.empty((h.shape[0], h.shape[1] // 2), dtype=x.dtype, device=x.device)
            ops.silu_and_mul(act, h)
            y.index_copy_(0, rows, torch.matmul(act, ops.w4a16_dequantize(w2.expert(e))))
    if not combine:
        return y.view(T, topk, H)
    out = torch.empty((T, H), dtype=x.dtype, device=x.device)
    ops.moe_sum(y.view(T, topk, H), out, topk_weights.contiguous())
    return out


class AgRsAll2AllManager:
    """dispatch = all-gather of (hidden_states, topk_weights, topk_ids) over the expert-parallel ranks,
    combine = reduce-scatter of the partial expert outputs
    (vllm/distributed/device_communicators/all2all.py:44-153)."""

    def __init__(self, sizes: list[int] | None = None):
        self.sizes = sizes    # tokens per rank; None: every rank holds as many as this one

    def _sizes(self, local_tokens: int) -> list[int]:
        return self.sizes if self.sizes is not None else [local_tokens] * parallel.get_ep_world_size()

    def dispatch(self, hidden_states, topk_weights, topk_ids):
        sizes = self._sizes(hidden_states.shape[0])
        if sizes[parallel.get_ep_rank()] != hidden_states.shape[0]:
            raise ValueError("dispatch: this rank's token count differs from its entry in sizes")
        return tuple(parallel.all_gatherv([hidden_states, topk_weights, topk_ids], sizes))

    def combine(self, hidden_states):
        n = parallel.get_ep_world_size()
        sizes = self.sizes if self.sizes is not None else [hidden_states.shape[0] // n] * n
        return parallel.reduce_scatterv(hidden_states, sizes)


def expert_map_for_rank(num_experts: int, ep_rank: int, ep_size: int) -> torch.Tensor:
    """Linear placement (fused_moe/layer.py determine_expert_map): rank r holds a contiguous run of
    experts, the first num_experts % ep_size ranks one more; others map to -1."""
    base, extra = divmod(num_experts, ep_size)
    count = base + (1 if ep_rank < extra else 0)
    start = ep_rank * base + min(ep_rank, extra)
    m = torch.full((num_experts,), -1, dtype=torch.int32)
    m[start:start + count] = torch.arange(count, dtype=torch.int32)
    return m


class FusedMoE(torch.nn.Module):
    """Routed experts of one decoder layer (fused_moe/layer.py `FusedMoE`): weights sharded over
    the intermediate dimension under tensor parallelism, optional expert_map for expert
    parallelism, softmax top-k routing; forward(hidden_states, router_logits)."""

    def __init__(self, num_experts: int, top_k: int, hidden_size: int, intermediate_size: int,
                 params_dtype: torch.dtype | None = None, reduce_results: bool = False, renormalize: bool = True,
                 quant_config: QuantizationConfig | None = None, prefix: str = "",
                 expert_map: torch.Tensor | None = None):
        super().__init__()
        self.tp_rank, self.tp_size = get_tensor_model_parallel_rank(), get_tensor_model_parallel_world_size()
        if intermediate_size % self.tp_size:
            raise ValueError("intermediate size must be divisible by the tensor-parallel size")
        self.global_num_experts = num_experts
        self.local_num_experts = num_experts if expert_map is None else int((expert_map >= 0).sum())
        self.top_k, self.hidden_size = top_k, hidden_size
        self.intermediate_size_per_partition = intermediate_size // self.tp_size
        self.reduce_results, self.renormalize, self.expert_map = reduce_results, renormalize, expert_map
        self.params_dtype = params_dtype or torch.get_default_dtype()
        method = quant_config.get_quant_method(self, prefix) if quant_config is not None else None
        if method is None:
            raise ValueError("FusedMoE on this path needs a quantization config that provides an expert method")
        self.quant_method = method
        method.create_weights(self, self.local_num_experts, hidden_size, self.intermediate_size_per_partition,
                              self.params_dtype)

    def weight_loader(self, param, loaded_weight, weight_name: str, shard_id: str, expert_id: int,
                      return_success: bool = False):
        if self.expert_map is not None:
            expert_id = int(self.expert_map[expert_id])
            if expert_id < 0:
                return False if return_success else None
        return param.weight_loader(param, loaded_weight, weight_name, shard_id, expert_id, return_success)

    def process_weights_after_loading(self) -> None:
        self.quant_method.process_weights_after_loading(self)

    def forward(self, hidden_states: torch.Tensor, router_logits: torch.Tensor,
                tokens_per_rank: list[int] | None = None) -> torch.Tensor:
        """With expert parallelism (parallel.init_expert_parallel, an expert_map from expert_map_for_rank)
        the routed tokens of all ranks are exchanged around the local experts; `tokens_per_rank` gives the
        ranks' token counts when they differ."""
        w, ids = fused_topk(router_logits, self.top_k, self.renormalize)
        if parallel.get_ep_world_size() > 1 and self.expert_map is not None:
            a2a = AgRsAll2AllManager(tokens_per_rank)
            hs, w_all, ids_all = a2a.dispatch(hidden_states, w, ids)
            return a2a.combine(self.quant_method.apply(self, hs, w_all, ids_all))
        out = self.quant_method.apply(self, hidden_states, w, ids)
        if self.reduce_results and self.tp_size > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out
