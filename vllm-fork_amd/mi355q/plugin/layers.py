"""RMSNorm, rotary embedding and SiLU-and-mul layer objects over the HIP ops.

Mirrors vllm/model_executor/layers/layernorm.py:37-130 (RMSNorm: with a residual the call
returns (normed, residual) and both are updated in place), rotary_embedding/base.py:80-131,
221-252 (cos/sin cache = cat(cos, sin) of t * base^(-2i/d), cast to the activation dtype;
forward rotates q and k in place), rotary_embedding/llama3_rope.py (Llama-3.1 frequency
rescaling) and activation.py `SiluAndMul`.
"""
import math

import torch

from .. import ops


class RMSNorm(torch.nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6, has_weight: bool = True, dtype: torch.dtype | None = None):
        super().__init__()
        self.hidden_size, self.variance_epsilon, self.has_weight = hidden_size, eps, has_weight
        self.weight = torch.nn.Parameter(torch.ones(hidden_size, dtype=dtype or torch.get_default_dtype()),
                                         requires_grad=False)

    def forward(self, x: torch.Tensor, residual: torch.Tensor | None = None):
        w = self.weight.data if self.has_weight else None
        if residual is not None:
            ops.fused_add_rms_norm(x, residual, w, self.variance_epsilon)
            return x, residual
        out = torch.empty_like(x)
        ops.rms_norm(out, x, w, self.variance_epsilon)
        return out

    forward_cuda = forward_hip = forward


class RotaryEmbedding(torch.nn.Module):
    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: float,
                 is_neox_style: bool, dtype: torch.dtype):
        super().__init__()
        self.head_size, self.rotary_dim = head_size, rotary_dim
        self.max_position_embeddings, self.base, self.is_neox_style, self.dtype = \
            max_position_embeddings, base, is_neox_style, dtype
        self.register_buffer("cos_sin_cache", self._compute_cos_sin_cache().to(dtype), persistent=False)

    def _compute_inv_freq(self, base: float) -> torch.Tensor:
        return 1.0 / (base ** (torch.arange(0, self.rotary_dim, 2, dtype=torch.float) / self.rotary_dim))

    def _compute_cos_sin_cache(self) -> torch.Tensor:
        inv_freq = self._compute_inv_freq(self.base)
        t = torch.arange(self.max_position_embeddings, dtype=torch.float)
        freqs = torch.einsum("i,j -> ij", t, inv_freq)
        return torch.cat((freqs.cos(), freqs.sin()), dim=-1)

    def forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor | None = None):
        if self.cos_sin_cache.device != query.device:
            self.cos_sin_cache = self.cos_sin_cache.to(query.device)
        ops.rotary_embedding(positions, query, key, self.head_size, self.cos_sin_cache, self.is_neox_style)
        return query, key

    forward_cuda = forward_hip = forward


class Llama3RotaryEmbedding(RotaryEmbedding):
    """Frequencies below the low-frequency wavelength are divided by `scaling_factor`, those
    above the high-frequency wavelength kept, and the band between interpolated."""

    def __init__(self, head_size, rotary_dim, max_position_embeddings, base, is_neox_style, dtype,
                 scaling_factor: float, low_freq_factor: float, high_freq_factor: float, orig_max_position: int):
        self.scaling_factor, self.low_freq_factor = scaling_factor, low_freq_factor
        self.high_freq_factor, self.orig_max_position = high_freq_factor, orig_max_position
        super().__init__(head_size, rotary_dim, max_position_embeddings, base, is_neox_style, dtype)

    def _compute_inv_freq(self, base: float) -> torch.Tensor:
        inv = super()._compute_inv_freq(base)
        wave = 2 * math.pi / inv
        lo, hi = self.orig_max_position / self.low_freq_factor, self.orig_max_position / self.high_freq_factor
        if self.low_freq_factor != self.high_freq_factor:
            smooth = (self.orig_max_position / wave - self.low_freq_factor) / (self.high_freq_factor - self.low_freq_factor)
        else:
            smooth = torch.zeros_like(inv)
        mid = (1 - smooth) * inv / self.scaling_factor + smooth * inv
        return torch.where(wave < hi, inv, torch.where(wave > lo, inv / self.scaling_factor, mid))


def get_rope(head_size: int, max_position: int, is_neox_style: bool = True, rope_parameters: dict | None = None,
             dtype: torch.dtype | None = None) -> RotaryEmbedding:
    rp = dict(rope_parameters or {})
    base = rp.get("rope_theta", 10000.0)
    rotary_dim = int(head_size * rp.get("partial_rotary_factor", 1.0))
    dtype = dtype or torch.get_default_dtype()
    kind = rp.get("rope_type", "default")
    if kind == "default":
        return RotaryEmbedding(head_size, rotary_dim, max_position, base, is_neox_style, dtype)
    if kind == "llama3":
        return Llama3RotaryEmbedding(head_size, rotary_dim, max_position, base, is_neox_style, dtype, rp["factor"],
                                     rp["low_freq_factor"], rp["high_freq_factor"],
                                     rp["original_max_position_embeddings"])
    raise ValueError(f"Unknown RoPE scaling type {kind}")


class SiluAndMul(torch.nn.Module):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        d = x.shape[-1] // 2
        out = torch.empty(x.shape[:-1] + (d,), dtype=x.dtype, device=x.device)
        ops.silu_and_mul(out, x if x.is_contiguous() else x.contiguous())
        return out

    forward_cuda = forward_hip = forward
