"""Random head configurations, block sizes, query / context lengths through both prompt attention kernels
against the oracle.  python tools/fuzz_prefill_attention.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import oracle as orc
from mi355q import ops, _lib
DEV = "cuda:0"


def run(cases, seed, verbose=True):
    rng = np.random.default_rng(seed)
    _print = print if verbose else (lambda *a, **k: None)
    def dev(b): return torch.from_numpy(np.ascontiguousarray(b).view(np.int16)).view(torch.bfloat16).to(DEV)
    bad = 0
    for it in range(cases):
        KVH = int(rng.choice([1, 2, 4, 8])); g = int(rng.choice([1, 2, 3, 4, 8])); H = KVH * g
        fp8 = bool(rng.integers(0, 2))
        D = 128 if fp8 else int(rng.choice([64, 128]))
        bs = int(rng.choice([16, 32, 64]))
        S = int(rng.integers(1, 4))
        qlens = [int(rng.choice([1, 5, 33, 64, 65, 130, 257, 400])) for _ in range(S)]
        ctx0 = [int(rng.choice([0, 0, 7, 64, 200])) for _ in range(S)]
        ks, vs = (0.02, 0.015) if fp8 else (1.0, 1.0)
        x = 16 if fp8 else 8
        seq_lens = [c + q for c, q in zip(ctx0, qlens)]
        maxb = (max(seq_lens) + bs - 1) // bs
        nblk = S * maxb + 2
        kc = torch.zeros((nblk, KVH, D // x, bs, x), dtype=torch.uint8 if fp8 else torch.bfloat16, device=DEV)
        vc = torch.zeros((nblk, KVH, D, bs), dtype=torch.uint8 if fp8 else torch.bfloat16, device=DEV)
        kst = torch.tensor(ks, dtype=torch.float32, device=DEV); vst = torch.tensor(vs, dtype=torch.float32, device=DEV)
        bt = rng.permutation(nblk)[: S * maxb].reshape(S, maxb).astype(np.int32)
        allq, refs = [], []
        for s_ in range(S):
            L = seq_lens[s_]
            k = orc.f32_to_h16(rng.uniform(-1, 1, (L, KVH, D)).astype(np.float32)); v = orc.f32_to_h16(rng.uniform(-1, 1, (L, KVH, D)).astype(np.float32))
            q = orc.f32_to_h16(rng.uniform(-1, 1, (L, H, D)).astype(np.float32))
            slots = np.array([bt[s_, j // bs] * bs + j % bs for j in range(L)], dtype=np.int64)
            ops.reshape_and_cache(dev(k), dev(v), kc, vc, torch.from_numpy(slots).to(DEV), "fp8" if fp8 else "auto", kst, vst)
            if fp8:
                k = orc.f32_to_h16(orc.e4m3_to_f32(orc.f32_to_e4m3(orc.h16_to_f32(k) / np.float32(ks))) * np.float32(ks))
                v = orc.f32_to_h16(orc.e4m3_to_f32(orc.f32_to_e4m3(orc.h16_to_f32(v) / np.float32(vs))) * np.float32(vs))
            refs.append(orc.prefill_attention(q, k, v, D ** -0.5)[ctx0[s_]:]); allq.append(q[ctx0[s_]:])
        qd = dev(np.concatenate(allq, 0)); ref = orc.h16_to_f32(np.concatenate(refs, 0))
        T = qd.shape[0]; P = (max(seq_lens) + 255) // 256
        qsl = torch.tensor(np.concatenate([[0], np.cumsum(qlens)]).astype(np.int32), device=DEV)
        line = f"case {it}: H {H:2d} KVH {KVH} D {D:3d} bs {bs:2d} fp8 {int(fp8)} q {qlens} ctx0 {ctx0}:"
        for name, v_, ctl in (("partitioned", "100000000", None), ("running", "32", "2"), ("running32", "32", "1")):
            if name != "partitioned" and fp8 and D != 128: continue
            _lib.set_option("MI355Q_PREFILL_FLASH_MIN", v_); _lib.set_option("MI355Q_PREFILL_FLASH_WGS", v_)
            _lib.set_option("MI355Q_PREFILL_FLASH_CTL", ctl)
            out = torch.full_like(qd, 7.0)
            tmp = torch.empty((T, H, P, D), dtype=qd.dtype, device=DEV); es = torch.empty((T, H, P), dtype=torch.float32, device=DEV); ml = torch.empty_like(es)
            ops.paged_prefill_attention(out, es, ml, tmp, qd, kc, vc, KVH, D ** -0.5, torch.from_numpy(bt).to(DEV),
                                        torch.tensor(seq_lens, dtype=torch.int32, device=DEV), qsl, max(qlens), bs, max(seq_lens), None,
                                        "fp8" if fp8 else "auto", kst, vst)
            torch.cuda.synchronize()
            d = np.abs(out.float().cpu().numpy() - ref)
            tol = 1e-2 if fp8 else (3e-3 if name == "partitioned" else 4e-3)
            ok = bool(d.max() <= tol) and bool(np.isfinite(d).all())
            bad += not ok
            line += f"  {name} max |d| {d.max():.4f} {'ok' if ok else 'BAD'}"
        _print(line, flush=True)
    return bad


if __name__ == "__main__":
    n_bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print("FAILURES", n_bad)
    sys.exit(1 if n_bad else 0)
