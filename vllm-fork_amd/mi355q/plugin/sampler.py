"""Token sampler.

Mirrors vllm/v1/sample/sampler.py:72-149,228-302: logits are promoted to fp32, repetition
penalties applied (vllm/_custom_ops.py:309-357), greedy rows take the first arg-max, random rows
are temperature-scaled, top-k/top-p masked (ops/topk_topp_sampler.py:349-472: keep the k largest;
then drop the lowest-probability tokens whose cumulative mass stays <= 1 - p, always keeping the
top one) and sampled as argmax(softmax / Exp(1) noise) (:77-212).  Arg-max, the penalties and the
random rows (temperature, both masks, softmax / noise arg-max: ops.sample_topk_topp) run in HIP.  There
is no second formulation: CPU tensors are refused by the op layer.
"""
from dataclasses import dataclass, field

import torch

from .. import ops

_SAMPLING_EPS = 1e-5


@dataclass
class SamplingMetadata:
    temperature: torch.Tensor | None = None        # [B] fp32; < eps means greedy
    all_greedy: bool = True
    all_random: bool = False
    top_p: torch.Tensor | None = None              # [B] fp32
    top_k: torch.Tensor | None = None              # [B] int
    generators: dict[int, torch.Generator] = field(default_factory=dict)
    repetition_penalties: torch.Tensor | None = None   # [B]
    prompt_mask: torch.Tensor | None = None        # bool [B, V]: token occurs in the prompt
    output_mask: torch.Tensor | None = None        # bool [B, V]: token occurs in the output so far
    no_penalties: bool = True


@dataclass
class SamplerOutput:
    sampled_token_ids: torch.Tensor   # int32 [B, 1]


def exponential_noise(like: torch.Tensor, generators: dict[int, torch.Generator]) -> torch.Tensor:
    """Exp(1) draws, rows with their own generator overwritten from it (topk_topp_sampler.py:452-465)."""
    q = torch.empty_like(like, dtype=torch.float32)
    if len(generators) != like.shape[0]:
        q.exponential_()
    for i, g in generators.items():
        q[i].exponential_(generator=g)
    return q


class Sampler(torch.nn.Module):
    def forward(self, logits: torch.Tensor, sampling_metadata: SamplingMetadata) -> SamplerOutput:
        md = sampling_metadata
        if md.all_greedy and md.no_penalties:
            # arg-max of the 16-bit logits equals arg-max of their fp32 promotion
            ids = ops.argmax(logits if logits.is_contiguous() else logits.contiguous())
            return SamplerOutput(ids.to(torch.int32).unsqueeze(-1))
        logits = logits.to(torch.float32)
        if not md.no_penalties:
            ops.apply_repetition_penalties(logits, md.prompt_mask, md.output_mask, md.repetition_penalties)
        greedy = None if md.all_random else ops.argmax(logits)
        if md.all_greedy:
            return SamplerOutput(greedy.to(torch.int32).unsqueeze(-1))
        temp = md.temperature
        if not md.all_random:
            temp = torch.where(temp < _SAMPLING_EPS, 1.0, temp)
        # one HIP kernel: temperature, both masks by radix selection, softmax / noise arg-max
        sampled = ops.sample_topk_topp(logits.contiguous(), temp, md.top_k, md.top_p, exponential_noise(logits, md.generators))
        if greedy is not None:
            sampled = torch.where(md.temperature < _SAMPLING_EPS, greedy, sampled)
        return SamplerOutput(sampled.to(torch.int32).unsqueeze(-1))
