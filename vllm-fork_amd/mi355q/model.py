"""Llama / Mixtral-architecture decode/prefill runner built only from mi355q ops.

This is the harness the benchmark and the end-to-end tests drive: the layer order of
vllm/model_executor/models/llama.py:311-328 (RMSNorm -> qkv_proj -> RoPE -> KV-cache
write -> attention -> o_proj -> fused add+RMSNorm -> gate_up_proj -> SiluAndMul ->
down_proj), GPTQ/AWQ 4-bit linears, a 16-bit lm_head and greedy sampling, with the
column/row tensor-parallel split of vllm/model_executor/layers/linear.py (all-reduce
after o_proj and down_proj, linear.py:1763-1767).  Configurations with experts replace the MLP by the
sparse block of vllm/model_executor/models/mixtral.py:126-130 (16-bit router, softmax top-k with
renormalisation, 4-bit experts in the MoE WNA16 layout of
vllm/model_executor/layers/quantization/moe_wna16.py:204-360, weighted sum).  Weights are synthetic.
"""
import os
from dataclasses import dataclass

import torch

from . import _lib, ops


@dataclass
class LlamaConfig:
    name: str
    hidden: int
    inter: int
    layers: int
    heads: int
    kv_heads: int
    head_dim: int
    vocab: int
    rope_theta: float = 500000.0
    rms_eps: float = 1e-5
    max_pos: int = 8192
    num_experts: int = 0       # > 0: the MLP is a routed-experts block (Mixtral)
    top_k: int = 0


# token counts up to which add+RMSNorm runs in the prologue of the GEMM that consumes it: every
# workgroup of that GEMM repeats the norm, which beats the separate launch for one row (1.71 vs
# 1.79 ms per step) and loses from two rows on (batch 4: 2.15 vs 2.03 ms)
FUSE_NORM_MAX_TOKENS = int(os.environ.get("MI355Q_FUSE_NORM_MAX", "1"))
# up to this many tokens the MoE router runs norm, gate projection and top-k as one launch (one workgroup per token)
ROUTER_FUSED_MAX_TOKENS = int(os.environ.get("MI355Q_ROUTER_FUSED_MAX", "16"))
# a routed layer's weighted sum in the prologue of the next layer's qkv projection (one token, top-2)
DEFER_MOE_SUM = os.environ.get("MI355Q_DEFER_MOE_SUM", "1") != "0"

CONFIGS = {
    "llama3-8b": LlamaConfig("llama3-8b", 4096, 14336, 32, 32, 8, 128, 128256),
    "llama3-70b": LlamaConfig("llama3-70b", 8192, 28672, 80, 64, 8, 128, 128256),
    "tiny": LlamaConfig("tiny", 256, 512, 2, 4, 2, 64, 1024, max_pos=512),
    "small": LlamaConfig("small", 1024, 2048, 2, 8, 2, 128, 2048, max_pos=512),
    "mixtral-8x7b": LlamaConfig("mixtral-8x7b", 4096, 14336, 32, 32, 8, 128, 32000, rope_theta=1e6, max_pos=8192,
                                num_experts=8, top_k=2),
    "tiny-moe": LlamaConfig("tiny-moe", 256, 512, 2, 4, 2, 64, 1024, max_pos=512, num_experts=4, top_k=2),
    "small-moe": LlamaConfig("small-moe", 1024, 2048, 2, 8, 2, 128, 2048, max_pos=512, num_experts=8, top_k=2),
    # Mixtral's widths on two layers: the expert kernels at full size in an end-to-end test
    "mixtral-2l": LlamaConfig("mixtral-2l", 4096, 14336, 2, 32, 8, 128, 32000, rope_theta=1e6, max_pos=2048,
                              num_experts=8, top_k=2),
}


def linear_shapes(cfg: LlamaConfig, tp: int = 1):
    """Per-rank [K, N] of the four quantized projections (benchmarks/kernels/weight_shapes.py:28-33,58-63)."""
    h, kvh = cfg.heads // tp, max(cfg.kv_heads // tp, 1)
    return {
        "qkv": (cfg.hidden, (h + 2 * kvh) * cfg.head_dim),
        "o": (h * cfg.head_dim, cfg.hidden),
        "gate_up": (cfg.hidden, 2 * cfg.inter // tp),
        "down": (cfg.inter // tp, cfg.hidden),
    }


def w4a16_bytes(M: int, K: int, N: int, G: int, zeros: bool) -> int:
    """Algorithmic bytes of one W4A16 GEMM: int4 weights + 16-bit scales (+ int4 zeros) + activations in and out."""
    return K * N // 2 + (K // G) * N * 2 + ((K // G) * N // 2 if zeros else 0) + M * K * 2 + M * N * 2


def decode_bytes_per_step(cfg: LlamaConfig, batch: int, ctx: int, G: int, zeros: bool, kv_bytes: int, tp: int = 1) -> int:
    """Algorithmic HBM bytes of one decode step on one rank: quantized linears, 16-bit lm_head,
    and the KV cache read at context `ctx` (norm/activation traffic is < 1 MB and left out)."""
    total = 0
    for name, (K, N) in linear_shapes(cfg, tp).items():
        one = K * N // 2 + (K // G) * N * 2 + ((K // G) * N // 2 if zeros else 0)
        if cfg.num_experts and name in ("gate_up", "down"):
            # experts a step touches: every (token, selected expert) slot reads its expert, at most all of them
            one *= min(cfg.num_experts, max(batch, 1) * cfg.top_k)
        total += one
    if cfg.num_experts:
        total += cfg.num_experts * cfg.hidden * 2          # 16-bit router
    total *= cfg.layers
    total += cfg.vocab // tp * cfg.hidden * 2
    kvh = max(cfg.kv_heads // tp, 1)
    total += 2 * cfg.layers * kvh * cfg.head_dim * ctx * kv_bytes * batch
    return total


def _shard(kind: str, full: torch.Tensor, rows_dim: int, cols_dim: int, packed_dim, pack: int, tp_rank: int, tp_size: int,
           parts, kv_replicas: int = 1) -> torch.Tensor:
    """This rank's shard of one FULL checkpoint tensor, cut by the parameter loaders the linear layers use
    (plugin/parameter.py, the reference's parameter.py:148-230): `row` narrows the input dimension,
    `column` the output dimension, `merged` / `qkv` the output dimension per logical sub-matrix.
    `parts`: [(shard_id, full_width)] of the logical sub-matrices along the output dimension."""
    from .plugin import parallel
    from .plugin.parameter import PackedvLLMParameter
    saved = dict(parallel._STATE)
    parallel._STATE.update(rank=tp_rank, size=tp_size)
    try:
        shape = list(full.shape)
        if kind == "row":
            shape[rows_dim] //= tp_size
        else:
            shape[cols_dim] = sum((w // (tp_size // kv_replicas if sid in ("k", "v") and kv_replicas > 1 else tp_size))
                                  for sid, w in parts) // (pack if packed_dim == cols_dim else 1)
        prm = PackedvLLMParameter(torch.empty(shape, dtype=full.dtype, device=full.device), input_dim=rows_dim,
                                  output_dim=cols_dim, packed_dim=packed_dim if packed_dim is not None else cols_dim,
                                  packed_factor=pack if packed_dim is not None else 1)
        if kind == "row":
            prm.load_row_parallel_weight(full)
        elif kind == "column":
            prm.load_column_parallel_weight(full)
        else:
            src_off = dst_off = 0
            for sid, width in parts:
                fw = width // (pack if packed_dim == cols_dim else 1)
                piece = full.narrow(cols_dim, src_off, fw)
                local = width // (tp_size // kv_replicas if sid in ("k", "v") and kv_replicas > 1 else tp_size)
                if kind == "qkv":
                    prm.load_qkv_weight(piece, shard_offset=dst_off, shard_size=local, shard_id=sid, num_heads=kv_replicas)
                else:
                    prm.load_merged_column_weight(piece, shard_offset=dst_off, shard_size=local)
                src_off += fw
                dst_off += local
        return prm.data
    finally:
        parallel._STATE.update(saved)


class Linear4:
    """One quantized projection: synthetic FULL checkpoint tensors (the same on every rank) -> this rank's
    shard -> kernel layout -> apply."""

    def __init__(self, K, N, G, dtype, device, gen, asym: bool, gain: float = 1.0, keep_checkpoint: bool = False,
                 pair_halves: bool = False, shard=("replicated", None), tp_rank: int = 0, tp_size: int = 1,
                 kv_replicas: int = 1):
        """K, N: the FULL layer; shard = (kind, parts) as `_shard` takes them."""
        self.checkpoint = None
        kind, parts = shard
        qweight = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=device, generator=gen)
        # (q - 8) has variance 21.25: keep the projection roughly norm-preserving
        s = gain / (21.25 * K) ** 0.5
        scales = (s * (0.5 + torch.rand((K // G, N), device=device, generator=gen))).to(dtype)
        qzeros = None
        if asym:  # AWQ checkpoint tensors: N-packed AWQ order + packed zero points
            qweight = qweight.view(-1)[: K * N // 8].view(K, N // 8)
            qzeros = torch.randint(-2**31, 2**31 - 1, (K // G, N // 8), dtype=torch.int32, device=device, generator=gen)
        if tp_size > 1 and kind != "replicated":
            cut = lambda t, packed_dim, pack: _shard(kind, t, 0, 1, packed_dim, pack, tp_rank, tp_size, parts, kv_replicas)
            if asym:    # [K, N/8], [K/G, N], [K/G, N/8]: packed along the output dimension
                qweight, scales, qzeros = cut(qweight, 1, 8), cut(scales, None, 1), cut(qzeros, 1, 8)
            else:       # [K/8, N] packed along the input dimension, [K/G, N]
                qweight, scales = cut(qweight, 0, 8), cut(scales, None, 1)
        if asym:
            self.w = ops.w4a16_repack(qweight, scales, qzeros, G, _lib.W_AWQ, _lib.ZP_AWQ, pair_halves=pair_halves)
            if keep_checkpoint:
                self.checkpoint = ("awq", qweight.cpu(), scales.cpu(), qzeros.cpu(), G)
        else:
            self.w = ops.w4a16_repack(qweight, scales, None, G, _lib.W_GPTQ, _lib.ZP_CONST, pair_halves=pair_halves)
            if keep_checkpoint:
                self.checkpoint = ("gptq", qweight.cpu(), scales.cpu(), None, G)

    def __call__(self, x, fuse_silu_mul=False):
        y = ops.w4a16_gemm(x, self.w, fuse_silu_mul=fuse_silu_mul)
        return ops.unpair_halves(y) if self.w.pair_halves else y


class Linear16:
    """Unquantized projection (the bf16 comparator model); weight [N, K] cut like the quantized ones."""

    def __init__(self, K, N, dtype, device, gen, gain: float = 1.0, shard=("replicated", None), tp_rank: int = 0,
                 tp_size: int = 1, kv_replicas: int = 1):
        w = (torch.randn((N, K), device=device, generator=gen) * (gain / K ** 0.5)).to(dtype)
        kind, parts = shard
        if tp_size > 1 and kind != "replicated":
            # the loaders see [K, N] orientation: input dimension 0, output dimension 1
            w = _shard(kind, w.t().contiguous(), 0, 1, None, 1, tp_rank, tp_size, parts, kv_replicas).t().contiguous()
        self.weight = w

    def __call__(self, x, fuse_silu_mul=False):
        if fuse_silu_mul:
            d = x.shape[-1] // 2
            act = torch.empty(x.shape[:-1] + (d,), dtype=x.dtype, device=x.device)
            ops.silu_and_mul(act, x)
            x = act
        if x.numel() // x.shape[-1] <= 16:
            return ops.linear_skinny(x, self.weight)
        return torch.nn.functional.linear(x, self.weight)


class ExpertRows:
    """A routed layer's output before its combine: the selected experts' rows [T, topk, H] and their fp32 weights.
    The next layer's qkv projection takes the weighted sum in its prologue (one token, top-2); anything else
    asks for the sum."""

    def __init__(self, rows: torch.Tensor, weights: torch.Tensor):
        self.rows, self.weights = rows, weights
        self.shape = (rows.shape[0], rows.shape[2])
        self.dtype, self.device = rows.dtype, rows.device

    def sum(self) -> torch.Tensor:
        out = torch.empty(self.shape, dtype=self.dtype, device=self.device)
        ops.moe_sum(self.rows, out, self.weights)
        return out


class MoeBlock:
    """Routed experts of one decoder layer: a 16-bit router [E, H] and 4-bit experts generated directly in
    the layout the MoE WNA16 loader leaves them in (moe_wna16.py:516-638): w13 uint8 [E, 2I, H/2] (gate rows,
    then up rows; two values per byte along the input dimension, even index low), scales [E, 2I, H/G], zero
    points uint8 [E, I, H/G] for AWQ; w2 likewise [E, H, I/2].  Under tensor parallelism the intermediate
    dimension is cut as the loader cuts it (w1 / w3 rows, w2 input columns) and the block's output is
    all-reduced by the caller."""

    def __init__(self, cfg: LlamaConfig, G: int, dtype, device, gen, asym: bool, tp_rank: int = 0, tp_size: int = 1,
                 keep_checkpoint: bool = False):
        E, H, I = cfg.num_experts, cfg.hidden, cfg.inter
        assert I % tp_size == 0 and (I // tp_size) % G == 0 and H % G == 0, "expert shards must hold whole scale groups"
        self.top_k, self.G = cfg.top_k, G
        self.router = (torch.randn((E, H), device=device, generator=gen) * (1.0 / H ** 0.5)).to(dtype)
        Il = I // tp_size

        def tensors(n, k, gain):
            qw = torch.randint(0, 256, (E, n, k // 2), dtype=torch.uint8, device=device, generator=gen)
            sc = ((gain / (21.25 * k) ** 0.5) * (0.5 + torch.rand((E, n, k // G), device=device, generator=gen))).to(dtype)
            qz = torch.randint(0, 256, (E, n // 2, k // G), dtype=torch.uint8, device=device, generator=gen) if asym else None
            return qw, sc, qz

        w13, w2 = tensors(2 * I, H, 1.0), tensors(H, I, 0.5)
        if tp_size > 1:
            rows = lambda t, per: torch.cat((t[:, tp_rank * per:(tp_rank + 1) * per],
                                             t[:, t.shape[1] // 2 + tp_rank * per:t.shape[1] // 2 + (tp_rank + 1) * per]), 1)
            w13 = (rows(w13[0], Il), rows(w13[1], Il), None if w13[2] is None else rows(w13[2], Il // 2))
            cols = lambda t, per: t[:, :, tp_rank * per:(tp_rank + 1) * per].contiguous()
            w2 = (cols(w2[0], Il // 2), cols(w2[1], Il // G), None if w2[2] is None else cols(w2[2], Il // G))
        self.w13 = ops.moe_w4a16_repack(w13[0], w13[1], w13[2], G, pair_halves=True, decode_slots=cfg.top_k)
        self.w2 = ops.moe_w4a16_repack(w2[0], w2[1], w2[2], G, decode_slots=cfg.top_k)
        self.checkpoint = None
        if keep_checkpoint:
            cpu = lambda w: tuple(None if t is None else t.cpu() for t in w)
            self.checkpoint = (cpu(w13), cpu(w2))

    def __call__(self, x: torch.Tensor, residual: torch.Tensor | None = None, norm_weight: torch.Tensor | None = None,
                 eps: float = 0.0, combine: bool = True):
        """With `residual`: x is the attention block's output, and the post-attention add + norm runs here (in
        place on both, inside the router's launch at decode sizes).  combine=False: ExpertRows instead of the sum."""
        from .plugin.moe import fused_experts, fused_topk
        T = x.shape[0]
        if residual is not None and T <= ROUTER_FUSED_MAX_TOKENS and ops.moe_router_topk_supported(*self.router.shape):
            w, ids = ops.moe_router_topk(x, residual, norm_weight, eps, self.router, self.top_k, True)
            if not combine:
                return ExpertRows(fused_experts(x, self.w13, self.w2, w, ids, combine=False), w)
            return fused_experts(x, self.w13, self.w2, w, ids)
        if residual is not None:
            ops.fused_add_rms_norm(x, residual, norm_weight, eps)
        logits = ops.linear_skinny(x, self.router) if T <= 16 else torch.nn.functional.linear(x, self.router)
        w, ids = fused_topk(logits, self.top_k, True)      # Mixtral renormalises the selected weights
        return fused_experts(x, self.w13, self.w2, w, ids)


class LlamaRunner:
    def __init__(self, cfg: LlamaConfig, quant: str = "gptq", kv_cache_dtype: str = "auto", group_size: int = 128,
                 dtype=torch.bfloat16, device="cuda:0", tp_rank: int = 0, tp_size: int = 1, tp_group=None,
                 max_batch: int = 1, max_seq_len: int = 512, block_size: int = 16, seed: int = 0,
                 keep_checkpoint: bool = False):
        assert quant in ("gptq", "awq", "none")
        self.cfg, self.quant, self.dtype, self.device = cfg, quant, dtype, torch.device(device)
        self.tp_rank, self.tp_size, self.tp_group = tp_rank, tp_size, tp_group
        # MI355Q_FORCE_TP_PATH=1 with an initialised process group: issue the two all-reduces per layer and the
        # logits all-gather even at world size 1, so that the RCCL calls (and their capture into the decode
        # graph) execute on a one-GPU box exactly as a TP rank issues them
        self.tp_collectives = tp_size > 1 or (os.environ.get("MI355Q_FORCE_TP_PATH") == "1"
                                              and torch.distributed.is_available() and torch.distributed.is_initialized())
        # decode-sized sums and the logits gather go over peer memory where the ranks can map each other's regions
        # (its construction checks itself against the group's collectives and is all-or-nothing over the ranks)
        self.peer_ar = None
        if self.tp_collectives and (tp_size > 1 or os.environ.get("MI355Q_PEER_AR_LOOPBACK") == "1"):
            from .plugin.peer_all_reduce import PeerAllReduce
            ar = PeerAllReduce(group=tp_group, device=self.device)
            self.peer_ar = None if ar.disabled else ar
            self.peer_ar_note = ar.reason
        self.G = group_size
        self.kv_cache_dtype = kv_cache_dtype
        self.fp8 = kv_cache_dtype.startswith("fp8")
        self.block_size, self.max_batch = block_size, max_batch
        self.max_seq_len = max_seq_len
        # ONE synthetic checkpoint: every rank draws the same full-size tensors (same seed, same order) and
        # keeps its own shard, so the TP = N model is the TP = 1 model
        gen = torch.Generator(device=self.device)
        gen.manual_seed(seed)
        shapes = linear_shapes(cfg, tp_size)       # per rank: what the kernels see
        full = linear_shapes(cfg, 1)               # the checkpoint's
        assert cfg.heads % tp_size == 0 and (cfg.kv_heads % tp_size == 0 or tp_size % cfg.kv_heads == 0)
        self.heads = cfg.heads // tp_size
        self.kv_heads = max(cfg.kv_heads // tp_size, 1)
        kv_replicas = max(tp_size // cfg.kv_heads, 1)   # ranks that share one KV head (linear.py QKVParallelLinear)
        D = cfg.head_dim
        cuts = {
            "qkv": ("qkv", [("q", cfg.heads * D), ("k", cfg.kv_heads * D), ("v", cfg.kv_heads * D)]),
            "o": ("row", None),
            "gate_up": ("merged", [("gate", cfg.inter), ("up", cfg.inter)]),
            "down": ("row", None),
        }

        def make(name, gain=1.0, pair_halves=False):
            K, N = full[name]
            kw = dict(shard=cuts[name], tp_rank=tp_rank, tp_size=tp_size, kv_replicas=kv_replicas)
            if quant == "none":
                return Linear16(K, N, dtype, self.device, gen, gain, **kw)
            return Linear4(K, N, self.G, dtype, self.device, gen, quant == "awq", gain, keep_checkpoint, pair_halves, **kw)

        # gate_i next to up_i lets the gate_up kernel finish SiluAndMul itself (decode path)
        pair = quant != "none" and ops.w4a16_gate_up_silu_supported(*shapes["gate_up"], self.G)

        self.layers = []
        for _ in range(cfg.layers):
            L = {
                "ln1": torch.ones(cfg.hidden, dtype=dtype, device=self.device),
                "ln2": torch.ones(cfg.hidden, dtype=dtype, device=self.device),
                "qkv": make("qkv"),
                "o": make("o", gain=0.5),
            }
            if cfg.num_experts:
                assert quant != "none", "the routed-experts block is built for 4-bit experts"
                L["moe"] = MoeBlock(cfg, self.G, dtype, self.device, gen, quant == "awq", tp_rank, tp_size, keep_checkpoint)
            else:
                L["gate_up"] = make("gate_up", pair_halves=pair)
                L["down"] = make("down", gain=0.5)
            self.layers.append(L)
        self.norm = torch.ones(cfg.hidden, dtype=dtype, device=self.device)
        vshard = cfg.vocab // tp_size
        # embedding replicated, lm_head vocab-parallel (rows of one full matrix)
        g2 = torch.Generator(device=self.device)
        g2.manual_seed(seed + 7)
        self.embed = (torch.randn((cfg.vocab, cfg.hidden), device=self.device, generator=g2) * 0.02).to(dtype)
        lm_full = (torch.randn((cfg.vocab, cfg.hidden), device=self.device, generator=gen) * 0.02).to(dtype)
        self.lm_head = lm_full[tp_rank * vshard:(tp_rank + 1) * vshard].clone() if tp_size > 1 else lm_full
        del lm_full

        # RoPE cache (rotary_embedding/base.py:80-103), cast to the activation dtype (:105-131)
        inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, D, 2, dtype=torch.float) / D))
        t = torch.arange(cfg.max_pos, dtype=torch.float)
        freqs = torch.einsum("i,j -> ij", t, inv_freq)
        self.cos_sin = torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(dtype).to(self.device)

        # paged KV cache, ROCM_ATTN layout (v1/attention/backends/rocm_attn.py:246-256, ops/paged_attn.py:16-29)
        self.blocks_per_seq = (max_seq_len + block_size - 1) // block_size
        nblk = self.blocks_per_seq * max_batch
        cdt = torch.uint8 if self.fp8 else dtype
        x = 16 // (1 if self.fp8 else 2)
        self.kc = [torch.zeros((nblk, self.kv_heads, D // x, block_size, x), dtype=cdt, device=self.device)
                   for _ in range(cfg.layers)]
        self.vc = [torch.zeros((nblk, self.kv_heads, D, block_size), dtype=cdt, device=self.device)
                   for _ in range(cfg.layers)]
        self.k_scale = torch.full((), 1.0 / 16 if self.fp8 else 1.0, dtype=torch.float32, device=self.device)
        self.v_scale = torch.full((), 1.0 / 16 if self.fp8 else 1.0, dtype=torch.float32, device=self.device)
        self.block_tables = torch.arange(nblk, dtype=torch.int32, device=self.device).view(max_batch, self.blocks_per_seq)
        self.scale = 1.0 / D ** 0.5
        self.P = (max_seq_len + _lib.PA_PARTITION - 1) // _lib.PA_PARTITION

        # static decode state (graph inputs)
        B = max_batch
        self.tokens = torch.zeros(B, dtype=torch.int64, device=self.device)
        self.positions = torch.zeros(B, dtype=torch.int64, device=self.device)
        self.seq_lens = torch.zeros(B, dtype=torch.int32, device=self.device)
        self.slot_base = (self.block_tables[:, 0].to(torch.int64) * block_size)  # sequences own contiguous blocks
        self.tmp_out = torch.empty((B, self.heads, self.P, D), dtype=dtype, device=self.device)
        self.exp_sums = torch.empty((B, self.heads, self.P), dtype=torch.float32, device=self.device)
        self.max_logits = torch.empty_like(self.exp_sums)
        self.graph = None
        # the next decode step's first hidden state: the embedding row of the token sampled last, fetched by
        # the step's last launch (ops.argmax_decode_step) together with the position / length increments
        self.h0 = torch.zeros((B, cfg.hidden), dtype=dtype, device=self.device)
        self.step_tail = os.environ.get("MI355Q_STEP_TAIL", "1") == "1" and dtype in (torch.bfloat16, torch.float16)

    # ------------------------------------------------------------------
    def weight_bytes(self) -> int:
        return decode_bytes_per_step(self.cfg, 0, 0, self.G, self.quant == "awq", 0, self.tp_size) if self.quant != "none" \
            else sum(K * N * 2 for K, N in linear_shapes(self.cfg, self.tp_size).values()) * self.cfg.layers + \
            self.cfg.vocab // self.tp_size * self.cfg.hidden * 2

    def _host_staged(self) -> bool:
        """gloo (CPU rehearsal of the multi-GPU path): collectives go through host memory."""
        return torch.distributed.get_backend(self.tp_group) == "gloo"

    def _all_reduce(self, x):
        if self.tp_collectives:
            if self.peer_ar is not None and self.peer_ar.should_use(x):
                return self.peer_ar.all_reduce(x)      # one kernel over peer memory (decode-sized sums)
            if self._host_staged():
                t = x.cpu()
                torch.distributed.all_reduce(t, group=self.tp_group)
                x.copy_(t)
            else:
                torch.distributed.all_reduce(x, group=self.tp_group)
        return x

    def _tail_norm_ok(self, T, lin) -> bool:
        """A row-parallel projection whose K-slice sums, residual add and the following RMSNorm can run
        as one kernel: the token range of the tiled kernel (or of the K-sliced rows kernel), whole
        projection on this rank."""
        return (self.quant != "none" and not self.tp_collectives and T <= ops.MM_MAX_TOKENS
                and ops.w4a16_mm_add_rms_norm_preferred(T, lin.w))

    def _layer(self, L, li, h, residual, positions, slot_mapping, attend, normed=False, next_norm=None):
        """One decoder layer.  `normed`: h already is RMSNorm(residual) (the previous layer's down
        projection finished it).  `next_norm`: weight of the norm that follows this layer; when the
        down projection can finish that norm too, the third return value is True."""
        cfg, D = self.cfg, self.cfg.head_dim
        T = h.shape[0]
        # small token counts: the add+RMSNorm runs in the prologue of the GEMM that consumes it
        fuse_norm = self.quant != "none" and T <= FUSE_NORM_MAX_TOKENS
        mixed = None
        if isinstance(h, ExpertRows):   # the previous layer's experts, not combined yet
            if fuse_norm and attend is None and ops.w4a16_gemm_moe_sum_add_rms_norm_supported(T, h.rows.shape[1], L["qkv"].w):
                mixed = h
            else:
                h = h.sum()
        # prompt tokens on the tiled kernel: projection, RoPE and cache write as one op
        qkv_tail = (attend is not None and self.quant != "none" and ops.MM_MIN_TOKENS <= T <= ops.MM_MAX_TOKENS
                    and ops._mm_supported(L["qkv"].w))

        def qkv_proj(x):
            if qkv_tail:
                return ops.w4a16_mm_rope_and_cache(x, L["qkv"].w, positions, D, self.heads, self.kv_heads, self.cos_sin,
                                                   True, self.kc[li], self.vc[li], slot_mapping, self.kv_cache_dtype,
                                                   self.k_scale, self.v_scale)
            return L["qkv"](x)

        if mixed is not None:   # combine, residual add, norm and projection in one launch
            qkv, residual = ops.w4a16_gemm_moe_sum_add_rms_norm(mixed.rows, mixed.weights, residual, L["ln1"], cfg.rms_eps,
                                                                L["qkv"].w)
        elif normed:
            qkv = qkv_proj(h)
        elif fuse_norm:
            qkv, residual = ops.w4a16_gemm_add_rms_norm(h, residual, L["ln1"], cfg.rms_eps, L["qkv"].w)
        else:
            if residual is None:
                residual = h
                x = torch.empty_like(h)
                ops.rms_norm(x, h, L["ln1"], cfg.rms_eps)
            else:
                ops.fused_add_rms_norm(h, residual, L["ln1"], cfg.rms_eps)
                x = h
            qkv = qkv_proj(x)
        qs, ks = self.heads * D, self.kv_heads * D
        attn = torch.empty((T, self.heads, D), dtype=h.dtype, device=h.device)
        if attend is None:
            # decode: RoPE + cache write + attention in one launch
            ops.decode_attention_fused(attn, self.exp_sums, self.max_logits, self.tmp_out, qkv, positions,
                                       self.cos_sin, self.kc[li], self.vc[li], self.heads, self.kv_heads, D,
                                       self.scale, self.block_tables, self.seq_lens, self.block_size,
                                       self.max_seq_len, self.kv_cache_dtype, self.k_scale, self.v_scale)
        else:
            q, k, v = qkv[:, :qs], qkv[:, qs:qs + ks], qkv[:, qs + ks:]
            if not qkv_tail:
                ops.rope_and_cache(positions, q, k, v, D, self.cos_sin, True, self.kc[li], self.vc[li], slot_mapping,
                                   self.kv_cache_dtype, self.k_scale, self.v_scale)
            attend(li, q.view(T, self.heads, D), attn)
        o_normed = self._tail_norm_ok(T, L["o"]) if self.quant != "none" else False
        if o_normed:   # o_proj, residual add and the post-attention norm
            o = ops.w4a16_mm_add_rms_norm(attn.view(T, qs), L["o"].w, residual, L["ln2"], cfg.rms_eps)
        else:
            o = self._all_reduce(L["o"](attn.view(T, qs)))
        if "moe" in L:   # router, top-k, experts, weighted sum (mixtral.py:126-130); all-reduce of the partial sums
            if o_normed:
                return self._all_reduce(L["moe"](o)), residual, False
            # one token, no tensor parallelism: the combine is left to the next layer's qkv projection
            defer = fuse_norm and attend is None and not self.tp_collectives and next_norm is not None and DEFER_MOE_SUM
            out = L["moe"](o, residual, L["ln2"], cfg.rms_eps, combine=not defer)
            return (out if isinstance(out, ExpertRows) else self._all_reduce(out)), residual, False
        tail = next_norm is not None and self._tail_norm_ok(T, L["down"]) if self.quant != "none" else False

        def down(act, **kw):
            if tail:   # down_proj, residual add and the norm that follows this layer
                return ops.w4a16_mm_add_rms_norm(act, L["down"].w, residual, next_norm, cfg.rms_eps), residual, True
            return self._all_reduce(L["down"](act, **kw)), residual, False

        if self.quant != "none" and L["gate_up"].w.pair_halves:
            gu_w = L["gate_up"].w
            tiled = ops.mm_min_tokens(gu_w) <= T <= ops.MM_MAX_TOKENS and ops._mm_supported(gu_w)
            if T < ops.MM_MIN_TOKENS or tiled:
                # SiluAndMul in the gate_up epilogue (and, for a few tokens, the norm in its prologue)
                if fuse_norm:
                    act, residual = ops.w4a16_gate_up_silu(o, residual, L["ln2"], cfg.rms_eps, gu_w)
                else:
                    if not o_normed:
                        ops.fused_add_rms_norm(o, residual, L["ln2"], cfg.rms_eps)
                    if tiled:
                        act = ops.w4a16_mm(o, gu_w, silu_pairs=True)
                    else:
                        act, _ = ops.w4a16_gate_up_silu(o, None, None, cfg.rms_eps, gu_w, add_norm=False)
                return down(act)
        if fuse_norm:
            gu, residual = ops.w4a16_gemm_add_rms_norm(o, residual, L["ln2"], cfg.rms_eps, L["gate_up"].w)
        else:
            if not o_normed:
                ops.fused_add_rms_norm(o, residual, L["ln2"], cfg.rms_eps)
            if self.quant != "none" and L["gate_up"].w.pair_halves:
                # long prompts (dequantise + dense GEMM): SiluAndMul straight on the pair-interleaved output
                raw = ops.w4a16_gemm(o, L["gate_up"].w)
                act = torch.empty((T, raw.shape[1] // 2), dtype=raw.dtype, device=raw.device)
                ops.silu_and_mul_pairs(act, raw)
                return down(act)
            gu = L["gate_up"](o)
        if tail:   # the fused tail takes the activation itself
            act = torch.empty((T, gu.shape[1] // 2), dtype=gu.dtype, device=gu.device)
            ops.silu_and_mul(act, gu)
            return down(act)
        return down(gu, fuse_silu_mul=True)

    def _next_norm(self, li):
        return self.layers[li + 1]["ln1"] if li + 1 < len(self.layers) else self.norm

    def _logits(self, h, residual, normed=False):
        if isinstance(h, ExpertRows):
            h = h.sum()
        if not normed:
            ops.fused_add_rms_norm(h, residual, self.norm, self.cfg.rms_eps)
        M = h.shape[0]
        # beyond 16 rows the BLAS GEMM is faster (lm_head of Llama-3-8B at 32 rows: 202 us against 231)
        logits = ops.linear_skinny(h, self.lm_head) if M <= 16 else torch.nn.functional.linear(h, self.lm_head)
        if self.tp_collectives and self.peer_ar is not None and self.peer_ar.should_gather(logits):
            return self.peer_ar.all_gather(logits)
        if self.tp_collectives:
            src = logits.cpu() if self._host_staged() else logits
            parts = [torch.empty_like(src) for _ in range(self.tp_size)]
            torch.distributed.all_gather(parts, src, group=self.tp_group)
            logits = torch.cat(parts, dim=-1).to(logits.device)
        return logits

    # ------------------------------------------------------------------
    def prefill(self, prompt: torch.Tensor) -> torch.Tensor:
        """prompt [B, T] token ids -> first sampled token per sequence [B]; fills the KV cache."""
        B, T = prompt.shape
        assert B <= self.max_batch and T <= self.max_seq_len
        assert int(prompt.min()) >= 0 and int(prompt.max()) < self.cfg.vocab, "token id out of range"
        dev = self.device
        tokens = prompt.reshape(-1).to(dev)
        positions = torch.arange(T, dtype=torch.int64, device=dev).repeat(B)
        slot_mapping = (self.slot_base[:B, None] + torch.arange(T, dtype=torch.int64, device=dev)[None, :]).reshape(-1)
        qsl = torch.arange(0, (B + 1) * T, T, dtype=torch.int32, device=dev)
        seq_lens = torch.full((B,), T, dtype=torch.int32, device=dev)
        P = (T + _lib.PA_PARTITION - 1) // _lib.PA_PARTITION
        tmp = torch.empty((B * T, self.heads, P, self.cfg.head_dim), dtype=self.dtype, device=dev)
        es = torch.empty((B * T, self.heads, P), dtype=torch.float32, device=dev)
        ml = torch.empty_like(es)

        def attend(li, q, out):
            ops.paged_prefill_attention(out, es, ml, tmp, q, self.kc[li], self.vc[li], self.kv_heads, self.scale,
                                        self.block_tables[:B], seq_lens, qsl, T, self.block_size, T, None,
                                        self.kv_cache_dtype, self.k_scale, self.v_scale)

        h = self.embed.index_select(0, tokens)
        residual = None
        normed = False
        for li, L in enumerate(self.layers):
            h, residual, normed = self._layer(L, li, h, residual, positions, slot_mapping, attend, normed,
                                              self._next_norm(li))
        last = torch.arange(T - 1, B * T, T, device=dev)
        logits = self._logits(h.index_select(0, last), residual.index_select(0, last), normed)
        ids = ops.argmax(logits)
        self.tokens[:B] = ids
        self.positions[:B] = T
        self.seq_lens[:B] = T + 1
        self.h0[:B] = self.embed.index_select(0, ids)
        return ids

    def _decode_body(self):
        """One decode step for the whole static batch: consumes self.tokens/positions/seq_lens,
        leaves the sampled ids in self.tokens and advances the positions."""
        B = self.max_batch
        attend = slot_mapping = None   # the fused attention step derives the slot from the block table
        if self.fp8 and self.cfg.head_dim != 128:  # the fused step needs head 128 with an fp8 cache
            slot_mapping = self.slot_base + self.positions

            def attend(li, q, out):
                ops.paged_attention_rocm(out, self.exp_sums, self.max_logits, self.tmp_out, q, self.kc[li], self.vc[li],
                                         self.kv_heads, self.scale, self.block_tables, self.seq_lens, None,
                                         self.block_size, self.max_seq_len, None, self.kv_cache_dtype, self.k_scale,
                                         self.v_scale)

        h = self.h0 if self.step_tail else self.embed.index_select(0, self.tokens)
        residual = None
        normed = False
        for li, L in enumerate(self.layers):
            h, residual, normed = self._layer(L, li, h, residual, self.positions, slot_mapping, attend, normed,
                                              self._next_norm(li))
        logits = self._logits(h, residual, normed)
        if self.step_tail:   # arg-max, position / length increments and the next step's embedding row in one launch
            ops.argmax_decode_step(logits, self.tokens, self.positions, self.seq_lens, self.embed, self.h0)
        else:
            ops.argmax(logits, out=self.tokens)
            self.positions.add_(1)
            self.seq_lens.add_(1)

    def capture(self):
        """Capture the decode step into a HIP graph (the reference captures the whole model forward,
        v1/worker/gpu_model_runner.py:6834; every op here is capture-safe)."""
        # warm up on a side stream so allocator state and the split-K workspace exist before capture
        state = (self.tokens.clone(), self.positions.clone(), self.seq_lens.clone(), self.h0.clone())
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._decode_body()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.tokens.copy_(state[0]); self.positions.copy_(state[1]); self.seq_lens.copy_(state[2]); self.h0.copy_(state[3])
        g = torch.cuda.CUDAGraph()
        # with a process group alive its watchdog thread queries events while this thread captures: only this
        # thread's calls may invalidate the capture (the default mode faults the whole process for them)
        mode = "thread_local" if self.tp_collectives else "global"
        with torch.cuda.graph(g, capture_error_mode=mode):
            self._decode_body()
        self.tokens.copy_(state[0]); self.positions.copy_(state[1]); self.seq_lens.copy_(state[2]); self.h0.copy_(state[3])
        self.graph = g

    def decode_step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self._decode_body()

    def reset(self, ctx: int):
        """Pretend `ctx` tokens are already cached for every sequence (cache content stays as is)."""
        self.tokens.fill_(1)
        self.positions.fill_(ctx)
        self.seq_lens.fill_(ctx + 1)
        self.h0.copy_(self.embed.index_select(0, self.tokens))
