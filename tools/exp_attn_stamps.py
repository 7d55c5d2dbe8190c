import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd"))
import torch
from mi355q import _lib
from mi355q.model import CONFIGS, LlamaRunner
L = _lib.lib()
L.mi355q_debug_attention_stamps.argtypes = [ctypes.c_void_p]; L.mi355q_debug_attention_stamps.restype = None
cfg = CONFIGS["llama3-8b"]
import dataclasses
cfg = dataclasses.replace(cfg, layers=4)
KV = sys.argv[1] if len(sys.argv) > 1 else "auto"     # "auto" (16-bit cache) or "fp8"
r = LlamaRunner(cfg, quant="gptq", kv_cache_dtype=KV, device="cuda:0", max_batch=1, max_seq_len=256)
r.reset(200)
for _ in range(3): r._decode_body()
st = torch.zeros((64, 10), dtype=torch.int64, device="cuda:0")
torch.cuda.synchronize()
L.mi355q_debug_attention_stamps(st.data_ptr())
r._decode_body()          # the last layer's launch overwrites: that is the one kept
torch.cuda.synchronize()
L.mi355q_debug_attention_stamps(None)
t = st[:8, [0, 6, 8, 7, 1, 2, 3, 4, 5]].cpu().double()
t = (t - t[:, 0].min()) / 100.0
print("per kv-head workgroup, us after first entry: entry | seq_len known | requests issued | q roped | patched | QK | softmax | PV | stored")
for row in t.tolist(): print(" ".join(f"{v:6.2f}" for v in row))

# same launch repeated back to back: instruction cache and K/V lines warm
from mi355q import ops
import types
li = 3
def attn_only():
    qkv = torch.randn((1, (r.heads + 2 * r.kv_heads) * cfg.head_dim), device="cuda:0").to(torch.bfloat16)
    out = torch.empty((1, r.heads, cfg.head_dim), dtype=torch.bfloat16, device="cuda:0")
    def go():
        ops.decode_attention_fused(out, r.exp_sums, r.max_logits, r.tmp_out, qkv, r.positions, r.cos_sin, r.kc[li], r.vc[li],
                                   r.heads, r.kv_heads, cfg.head_dim, r.scale, r.block_tables, r.seq_lens, r.block_size,
                                   r.max_seq_len, r.kv_cache_dtype, r.k_scale, r.v_scale)
    return go
go = attn_only()
for _ in range(5): go()
torch.cuda.synchronize()
L.mi355q_debug_attention_stamps(st.data_ptr())
go()
torch.cuda.synchronize()
L.mi355q_debug_attention_stamps(None)
t = st[:8, [0, 6, 8, 7, 1, 2, 3, 4, 5]].cpu().double()
t = (t - t[:, 0].min()) / 100.0
print("repeated launch (warm):")
for row in t.tolist()[:3]: print(" ".join(f"{v:6.2f}" for v in row))
raw = st[:8].cpu().double()
print("shader clock during the kernel, GHz:", [round(float((raw[i, 9] - raw[i, 8]) / ((raw[i, 5] - raw[i, 0]) * 10.0) / 1e0) / 1000, 3) for i in range(3)])
