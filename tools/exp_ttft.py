"""TTFT of the 8B GPTQ harness against prompt length, tiled W4A16 kernel vs dequantise + dense GEMM."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd")); sys.path.insert(0, ROOT)
import torch
from mi355q import ops
from mi355q.model import CONFIGS, LlamaRunner

cfg = CONFIGS["llama3-8b"]
r = LlamaRunner(cfg, quant="gptq", group_size=128, device="cuda:0", max_batch=1, max_seq_len=4352)
for T in (32, 64, 128, 256, 512, 1024, 2048, 4096):
    p = torch.randint(0, 10000, (1, T), generator=torch.Generator().manual_seed(0))
    res = []
    for mm_max in (1 << 30, 0):
        ops.MM_MAX_TOKENS = mm_max
        ts = []
        for i in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r.prefill(p); torch.cuda.synchronize()
            if i >= 2: ts.append(time.perf_counter() - t0)
        res.append(statistics.median(ts) * 1e3)
    print(f"T={T:5d}  tiled {res[0]:8.2f} ms   dequant+blas {res[1]:8.2f} ms", flush=True)
