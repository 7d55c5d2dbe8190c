"""Phase timeline of the burst GEMV from in-kernel 100 MHz timestamps."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd"))
import torch
from mi355q import ops, _lib
dev = "cuda:0"
L = _lib.lib()
L.mi355q_debug_burst_stamps.argtypes = [ctypes.c_void_p]
L.mi355q_debug_burst_stamps.restype = None
def run(name, K, N, pro):
    ws = []
    for _ in range(6):
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
        sc = (torch.rand((K // 128, N), device=dev) * 0.01).to(torch.bfloat16)
        ws.append(ops.w4a16_repack(qw, sc, None, 128))
    x = torch.randn((1, K * (2 if pro == "silu" else 1)), device=dev).to(torch.bfloat16)
    res = torch.randn((1, K), device=dev).to(torch.bfloat16)
    nw = torch.ones(K, device=dev, dtype=torch.bfloat16)
    def call(w):
        if pro == "norm": ops.w4a16_gemm_add_rms_norm(x, res, nw, 1e-5, w)
        elif pro == "silu": ops.w4a16_gemm(x, w, fuse_silu_mul=True)
        else: ops.w4a16_gemm(x, w)
    for w in ws: call(w)
    blocks = N // ws[0].tile
    st = torch.zeros((blocks, 8), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    L.mi355q_debug_burst_stamps(st.data_ptr())
    call(ws[3])      # cold weights
    torch.cuda.synchronize()
    L.mi355q_debug_burst_stamps(None)
    t = st.cpu().double()
    t0 = t[:, 0].min()
    t = (t - t0) / 100.0   # us
    q = lambda v: f"{v.min():5.2f}/{v.median():5.2f}/{v.max():5.2f}"
    print(f"{name:10s} tile {ws[0].tile:3d} blocks {blocks}: start {q(t[:,0])} | scales in LDS (thread 0) {q(t[:,4])} | x staged (thread 0) {q(t[:,5])} | own sum {q(t[:,6])} | all sums {q(t[:,7])} | staged {q(t[:,1])} | consumed {q(t[:,2])} | done {q(t[:,3])}  (us after first start, min/med/max)")
shapes = [("gate_up", 4096, 28672, "norm"), ("gate_up_n", 4096, 28672, "none"), ("down", 14336, 4096, "silu"),
                        ("down_n", 14336, 4096, "none"), ("qkv", 4096, 6144, "norm"), ("o", 4096, 4096, "none")]
if len(sys.argv) > 1 and sys.argv[1] == "tp8":   # Llama-3-70B per-rank shapes at TP = 8
    shapes = [("qkv", 8192, 1280, "none"), ("qkv", 8192, 1280, "norm"), ("o", 1024, 8192, "none"), ("gate_up", 8192, 7168, "norm"),
              ("down", 3584, 8192, "none")]
for name, K, N, pro in shapes:
    run(name, K, N, pro)
