"""Random shapes through ops.w4a16_mm (every form of the prompt kernel the planner picks) against the oracle.
python tools/fuzz_mm.py [cases] [seed] [batches]
With a third argument: decode batches of 13 to 32 rows on layers wide enough for the streaming rows kernel (gathered
and whole-row activation fetch, one- and two-chunk rings) and the 32-row tiles with their K-slices."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import oracle as orc
from mi355q import ops, _lib
DEV = "cuda:0"


def run(cases, seed, verbose=True, decode_batches=False):
    rng = np.random.default_rng(seed)
    _print = print if verbose else (lambda *a, **k: None)
    bad = 0
    for it in range(cases):
        tile = int(rng.choice([16, 32, 64, 112]))
        bw = {16: 256, 32: 256, 64: 256, 112: 448}[tile]
        N = bw * int(rng.integers(1, 5))
        K = 128 * int(rng.choice([1, 2, 3, 4, 6, 8, 12, 16, 28]))
        G = int(rng.choice([128, K])) if K % 128 == 0 else 128
        M = int(rng.choice([17, 31, 33, 40, 48, 64, 65, 100, 128, 129, 200, 256, 300, 511, 640]))
        if decode_batches:
            N = bw * int(rng.choice([4, 8, 16, 24]))
            K = 128 * int(rng.choice([8, 16, 24, 32, 56, 112]))
            G = int(rng.choice([128, 128, 256, K]))
            M = int(rng.choice([13, 14, 16, 17, 20, 24, 25, 27, 29, 31, 32]))
        awq = bool(rng.integers(0, 2))
        dt = orc.BF16 if rng.integers(0, 3) else orc.FP16
        tdt = torch.bfloat16 if dt == orc.BF16 else torch.float16
        dev = lambda b: torch.from_numpy(np.ascontiguousarray(b).view(np.int16)).view(tdt).to(DEV)
        q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
        a = orc.f32_to_h16(0.25 * rng.standard_normal((M, K)).astype(np.float32), dt)
        s = orc.f32_to_h16((0.05 * rng.random((K // G, N))).astype(np.float32), dt)
        i32 = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
        if awq:
            z = rng.integers(0, 16, size=(K // G, N), dtype=np.int32)
            qw, qz = orc.pack_cols(q, awq_order=True), orc.pack_cols(z, awq_order=True)
            ref = orc.w4a16_gemm(a, qw, s, qz, None, G=G, dt=dt, layout=orc.W_AWQ, zp_mode=orc.ZP_AWQ)
            w = ops.w4a16_repack(i32(qw), dev(s), i32(qz), G, _lib.W_AWQ, _lib.ZP_AWQ, tile=tile)
        else:
            qw = orc.gptq_pack(q)
            ref = orc.w4a16_gemm(a, qw, s, None, None, G=G, dt=dt)
            w = ops.w4a16_repack(i32(qw), dev(s), None, G, tile=tile)
        if not ops._mm_supported(w):
            _print(f"case {it}: tile {tile} M {M} K {K} N {N} G {G}: not served by the tiled kernel"); continue
        out = ops.w4a16_mm(dev(a), w)
        torch.cuda.synchronize()
        got = orc.h16_to_f32(out.cpu().view(torch.int16).numpy().view(np.uint16), dt); want = orc.h16_to_f32(ref, dt)
        ok = np.allclose(got, want, rtol=1e-2, atol=1e-2)
        eq = float((out.cpu().view(torch.int16).numpy().view(np.uint16) == ref).mean())
        bad += not ok
        _print(f"case {it}: tile {tile:3d} M {M:4d} K {K:5d} N {N:5d} G {G:5d} awq {int(awq)} {'bf16' if dt == orc.BF16 else 'fp16'}: {'ok ' if ok else 'BAD'} bit-equal {eq:.4f}", flush=True)
    return bad


if __name__ == "__main__":
    n_bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0,
                decode_batches=len(sys.argv) > 3)
    print("FAILURES", n_bad)
    sys.exit(1 if n_bad else 0)
