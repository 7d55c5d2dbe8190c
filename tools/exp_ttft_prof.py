"""One prefill per prompt length under rocprofv3 (kernel trace): per-kernel time of the tiled W4A16 kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd")); sys.path.insert(0, ROOT)
import torch
from mi355q import ops
from mi355q.model import CONFIGS, LlamaRunner
ops.MM_MAX_TOKENS = 1 << 30
cfg = CONFIGS["llama3-8b"]
r = LlamaRunner(cfg, quant="gptq", group_size=128, device="cuda:0", max_batch=1, max_seq_len=2304)
for T in [int(a) for a in sys.argv[1:]]:
    p = torch.randint(0, 10000, (1, T), generator=torch.Generator().manual_seed(0))
    for i in range(3):
        r.prefill(p)
    torch.cuda.synchronize()
