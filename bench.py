#!/usr/bin/env python3
"""Decode benchmark of the MI355X quantized-inference path.

    python bench.py --gpus N --steps K --warmup W

N = 1: Llama-3-8B GPTQ int4 g128, TP=1, bf16 activations, synthetic weights and prompts (the
       configuration the headline metric is quoted on).  A "step" is one decode step of the whole batch.
N > 1: Llama-3-70B GPTQ int4 g128 with TP = N; launched by torch.distributed.run, one rank per
       GPU, all-reduce over RCCL.

Prints ONE JSON line on rank 0: value = output tokens per second of the whole job (inputs resident in
HBM), ttft_ms_p50 = prompt + first token, roofline = the dominant kernel (gate_up GEMV) timed with HIP
events on its launch stream against the 8 TB/s HBM peak, cpu_baseline = the CPU restatement of the
reference's cpu_gemm_wna16 path timed on this host, other_shapes = the latency benchmark's other
(batch, input_len) cases.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "vllm-fork_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (spec)
STREAM_PEAK_GBS = 6290.0  # best sustained read rate of a pure streaming kernel measured on this chip


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--input-len", type=int, default=128)
    ap.add_argument("--model", default=None, help="llama3-8b | llama3-70b | mixtral-8x7b | tiny | small ... (default: by --gpus)")
    ap.add_argument("--skip-configs", action="store_true", help="skip the other benchmark configurations (AWQ + fp8 KV, Mixtral)")
    ap.add_argument("--quant", default="gptq", choices=["gptq", "awq", "none"])
    ap.add_argument("--kv-dtype", default="auto", choices=["auto", "fp8"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--roofline-only", action="store_true", help="run only the dominant-kernel timing pass")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-bf16", action="store_true")
    ap.add_argument("--skip-ttft", action="store_true")
    ap.add_argument("--skip-shapes", action="store_true", help="skip the other (batch, input_len) shapes")
    ap.add_argument("--tokens-out", default=None, help="rank 0 writes every decoded token id here (JSON), for TP comparisons")
    return ap.parse_args()


def time_dominant_kernel(runner, batch, reps=10):
    """Per-launch duration of the gate_up W4A16 GEMM (the kernel that moves most bytes of a
    decode step), measured with an event pair on the launch stream around graph replays of
    back-to-back launches.  The launches walk over all layers' weights so no launch finds its
    weights in the 256 MiB Infinity Cache."""
    from mi355q import ops
    from mi355q.model import linear_shapes, w4a16_bytes
    if runner.cfg.num_experts:
        return time_dominant_kernel_moe(runner, batch, reps)
    K, N = linear_shapes(runner.cfg, runner.tp_size)["gate_up"]
    x = (torch.randn((batch, K), device=runner.device) * 0.25).to(runner.dtype)
    layers = runner.layers
    residual = (torch.randn((batch, K), device=runner.device) * 0.25).to(runner.dtype)
    fused = runner.quant != "none" and batch <= 4

    def launch(L):   # the gate_up launch of a decode step, as LlamaRunner._layer issues it
        if not fused:
            if runner.quant != "none" and L["gate_up"].w.pair_halves:
                return ops.w4a16_gate_up_silu(x, None, None, runner.cfg.rms_eps, L["gate_up"].w, add_norm=False)
            return L["gate_up"](x)
        if L["gate_up"].w.pair_halves:
            return ops.w4a16_gate_up_silu(x, residual, L["ln2"], runner.cfg.rms_eps, L["gate_up"].w)
        return ops.w4a16_gemm_add_rms_norm(x, residual, L["ln2"], runner.cfg.rms_eps, L["gate_up"].w)

    for L in layers[:4]:
        launch(L)
    torch.cuda.synchronize()
    # One graph holds `passes` walks over all layers' launches, so the event pair around a replay
    # brackets back-to-back kernels only (an event pair around every single launch adds the events'
    # own barrier packets, several microseconds, to each sample).
    passes = 4
    stream = torch.cuda.Stream(device=runner.device)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        for L in layers[:2]:
            launch(L)
        stream.synchronize()
        with torch.cuda.graph(graph, stream=stream):
            for _ in range(passes):
                for L in layers:
                    launch(L)
        graph.replay()
        stream.synchronize()
        times = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            graph.replay()
            b.record(stream)
            stream.synchronize()
            times.append(a.elapsed_time(b) * 1e-3 / (passes * len(layers)))
    avg = sum(times) / len(times)
    nbytes = w4a16_bytes(batch, K, N, runner.G, runner.quant == "awq")
    if fused:   # + residual in/out and norm weight of the prologue
        nbytes += 3 * batch * K * 2 + K * 2
    if runner.quant != "none" and layers[0]["gate_up"].w.pair_halves:   # the SiLU epilogue writes N/2 per row
        nbytes -= batch * (N // 2) * 2
    traffic, traffic_source = pmc_traffic(runner, batch, fused)
    return {"bound": "hbm", "achieved": nbytes / avg / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": nbytes / avg / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
            "frac_of_measured_stream_peak": nbytes / avg / 1e9 / STREAM_PEAK_GBS,
            "kernel": "w4a16_burst_kernel (gate_up_proj %dx%d%s + SiLU epilogue, M=%d)"
                      % (K, N, " + add/RMSNorm prologue" if fused else "", batch),
            "bytes_per_launch": nbytes, "avg_launch_us": avg * 1e6, "min_launch_us": min(times) * 1e6,  # best replay, per launch
            "launches": len(times) * passes * len(layers)}


def time_dominant_kernel_moe(runner, batch, reps=10):
    """The same measurement for a routed-experts model: the merged gate / up projection of the selected
    experts (one launch over the batch x top_k slots, SiluAndMul in its epilogue), every layer's experts in
    turn so that no launch finds its weights cached."""
    from mi355q import ops
    cfg = runner.cfg
    blk0 = runner.layers[0]["moe"]
    w13 = blk0.w13
    x = (torch.randn((batch, w13.K), device=runner.device) * 0.25).to(runner.dtype)
    slots = batch * cfg.top_k
    ids = (torch.arange(slots, device=runner.device, dtype=torch.int32) * 3) % w13.E   # distinct experts while slots <= E

    def launch(L):
        return ops.moe_w4a16_gemm(x, L["moe"].w13, ids, cfg.top_k, silu_pairs=L["moe"].w13.pair_halves)

    for L in runner.layers[:2]:
        launch(L)
    torch.cuda.synchronize()
    passes = 4
    stream = torch.cuda.Stream(device=runner.device)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        launch(runner.layers[0])
        stream.synchronize()
        with torch.cuda.graph(graph, stream=stream):
            for _ in range(passes):
                for L in runner.layers:
                    launch(L)
        graph.replay()
        stream.synchronize()
        times = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            graph.replay()
            b.record(stream)
            stream.synchronize()
            times.append(a.elapsed_time(b) * 1e-3 / (passes * len(runner.layers)))
    avg = sum(times) / len(times)
    G = w13.G
    per_expert = w13.K * w13.N // 2 + (w13.K // G) * w13.N * 2 + ((w13.K // G) * w13.N // 2 if w13.zc is not None else 0)
    nbytes = min(slots, w13.E) * per_expert + batch * w13.K * 2 + slots * (w13.N // 2 if w13.pair_halves else w13.N) * 2
    return {"bound": "hbm", "achieved": nbytes / avg / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": nbytes / avg / 1e9 / HBM_PEAK_GBS, "traffic": None, "traffic_source": "no counter passes for this workload",
            "frac_of_measured_stream_peak": nbytes / avg / 1e9 / STREAM_PEAK_GBS,
            "kernel": "w4a16_burst_kernel with the expert indirection (experts' gate_up %dx%d + SiLU epilogue, %d slots)"
                      % (w13.K, w13.N, slots),
            "bytes_per_launch": nbytes, "avg_launch_us": avg * 1e6, "min_launch_us": min(times) * 1e6,
            "launches": len(times) * passes * len(runner.layers)}


PMC_FILE = os.path.join(ROOT, "profiles", "r3", "decode_8b_gptq_pmc_fetch_write.json")
KERNEL_SOURCES = ("vllm-fork_amd/csrc/w4a16.hip", "vllm-fork_amd/csrc/common.h")


def kernel_source_digest():
    """sha256 over the sources the dominant kernel is built from (what the counter passes are tied to)."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def pmc_traffic(runner, batch, fused):
    """HBM bytes per launch of the dominant kernel from counter passes taken OUTSIDE this run (rocprofv3
    cannot attach to its own process): 2 x FETCH_SIZE (gfx950 tallies the 128-byte requests of
    16-byte-per-lane streaming reads at 64 bytes) + WRITE_SIZE, both KiB per dispatch.  The file records
    the digest of the kernel sources and the kernel's template name at the time of the passes; if either
    differs from this tree, or the workload is not the one the passes ran (Llama-3-8B GPTQ, TP=1,
    batch 1), there is no measured traffic for this run and the result is (None, reason)."""
    if not (fused and batch == 1 and runner.tp_size == 1 and runner.cfg.name == "llama3-8b" and runner.quant == "gptq"):
        return None, "no counter passes for this workload"
    if not os.path.exists(PMC_FILE):
        return None, "no counter passes committed"
    with open(PMC_FILE) as fh:
        rec = json.load(fh)
    if rec.get("kernel_source_sha256") != kernel_source_digest():
        return None, "kernel sources changed since the counter passes (%s)" % os.path.relpath(PMC_FILE, ROOT)
    row = rec["kernels"].get(rec["dominant_kernel"])
    if row is None:
        return None, "dominant kernel missing from the counter passes"
    return (2.0 * row["fetch_size_kb_avg"] + row["write_size_kb_avg"]) * 1024.0, \
        "%s: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this kernel build (%s), %d dispatches" % (
            os.path.relpath(PMC_FILE, ROOT), rec["dominant_kernel"], row["dispatches"])


def cpu_baseline(cfg, G, batch, budget_s=15.0):
    """The CPU restatement of the fork's CPU backend (oracle.cpu_gemm_wna16, the cpu_wna16.cpp
    algorithm on its 16-channel block layout; bf16 lm_head) timed on this box's host cores.
    Sample: whole decode steps of the linear layers (4 projections x all layers + lm_head, batch
    `batch`), cycling through 4 distinct weight sets so nothing stays cache-resident, repeated
    until about `budget_s` seconds have been spent."""
    import numpy as np
    import oracle as orc
    from mi355q.model import linear_shapes
    cores = len(os.sched_getaffinity(0))
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    rng = np.random.default_rng(0)
    shapes = linear_shapes(cfg, 1)
    nsets = 4
    sets = []
    for _ in range(nsets):
        mats = {}
        for name, (K, N) in shapes.items():
            qw = rng.integers(-2**31, 2**31 - 1, size=(K // 8, N), dtype=np.int32)
            mats[name] = (orc.cpu_repack_gptq(qw), rng.integers(0x3000, 0x3c00, size=(K // G, N), dtype=np.uint16),
                          orc.f32_to_h16(rng.standard_normal((batch, K)).astype(np.float32)))
        sets.append(mats)
    lm = rng.integers(0x3000, 0x3c00, size=(cfg.vocab, cfg.hidden), dtype=np.uint16)   # positive bf16 in [2^-31, 1)
    xh = orc.f32_to_h16(rng.standard_normal((batch, cfg.hidden)).astype(np.float32))
    for name, (qwb, s, a) in sets[0].items():  # warm-up (thread pool, page-in)
        orc.cpu_gemm_wna16(a, qwb, s, None, G=G)
    steps, t0 = 0, time.perf_counter()
    while True:
        for li in range(cfg.layers):
            for name, (qwb, s, a) in sets[li % nsets].items():
                orc.cpu_gemm_wna16(a, qwb, s, None, G=G)
        orc.linear_h16(xh, lm)
        steps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or el + el / steps > 2 * budget_s:
            break
    return {"value": batch * steps / el, "unit": "tok/s", "cores": cores, "kind": "port",
            "sample": "%d whole decode steps of the linear layers on CPU: oracle cpu_gemm_wna16 (cpu_wna16.cpp "
                      "restatement, 16-channel block layout) for the 4 projections of all %d layers + the bf16 "
                      "lm_head, batch %d; attention/norms excluded" % (steps, cfg.layers, batch),
            "sample_seconds": el}


def cpu_baseline_opt125m(budget_s=6.0):
    """OPT-125m-shaped bf16 greedy decode (the reference's CPU-backend plumbing case: LayerNorm, learned
    positions, ReLU MLP, tied lm_head) in the oracle, random-init weights, one sequence, prompt of 32
    tokens; decode steps are timed until about `budget_s` seconds have been spent."""
    import numpy as np
    from oracle import opt125m
    cores = len(os.sched_getaffinity(0))
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    w = opt125m.make_weights(0)
    d = opt125m.OptDecoder(w, max_len=2048)
    tok = d.forward(np.arange(5, 37, dtype=np.int64))
    steps, t0 = 0, time.perf_counter()
    while True:
        tok = d.forward(np.asarray([tok], dtype=np.int64))
        steps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or steps >= 1900:
            break
    return {"value": steps / el, "unit": "tok/s", "cores": cores, "kind": "port",
            "sample": "%d greedy decode steps of an OPT-125m-shaped bf16 model (12 layers, hidden 768, vocab 50272) in "
                      "oracle/opt125m.py, batch 1, prompt 32" % steps, "sample_seconds": el}


def time_allreduce_16k(dev, reps=200, fn=None):
    """The TP exchange of a 70B decode step: sum all-reduce of one [1, 8192] bf16 row (16 KiB) on the default
    process group.  Eager: `reps` back-to-back calls between two events; captured: the same calls inside one
    HIP graph, replayed.  Microseconds per all-reduce, (eager, captured or None, capture error or None)."""
    fn = fn or torch.distributed.all_reduce      # or the peer-memory exchange (same calls on every rank)
    x = torch.ones(8192, dtype=torch.bfloat16, device=dev)
    for _ in range(10):
        fn(x)
        x.fill_(1.0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn(x)
    b.record()
    torch.cuda.synchronize()
    eager = a.elapsed_time(b) * 1e3 / reps
    captured, err = None, None
    try:
        x.fill_(1.0)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):   # the group's watchdog thread keeps querying events
                for _ in range(50):
                    fn(x)
            g.replay()
            s.synchronize()
            x.fill_(1.0)
            a.record(s)
            for _ in range(4):
                g.replay()
            b.record(s)
            s.synchronize()
        captured = a.elapsed_time(b) * 1e3 / 200
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"[:200]
    return eager, captured, err


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box: MI355Q_DIST_BACKEND=gloo MI355Q_SHARE_GPU=1 lets all ranks use GPU 0
    share = os.environ.get("MI355Q_SHARE_GPU") == "1"
    dev_index = 0 if (world == 1 or share) else local_rank
    torch.cuda.set_device(dev_index)
    # MI355Q_FORCE_TP_PATH=1: a process group (RCCL) even at world size 1, and the runner issues its TP
    # collectives into it -- the one-GPU rehearsal of what every rank of an N > 1 run executes
    dist_on = world > 1 or os.environ.get("MI355Q_FORCE_TP_PATH") == "1"
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29541"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
        backend = os.environ.get("MI355Q_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            torch.distributed.init_process_group(backend)
        warm = torch.ones(8, device=f"cuda:{dev_index}" if backend == "nccl" else "cpu")
        torch.distributed.all_reduce(warm)       # communicator set-up happens outside any graph capture
        torch.cuda.synchronize()
    n_gpus = max(args.gpus, world)
    assert world == n_gpus or world == 1 and n_gpus == 1, "launch with torch.distributed.run for --gpus > 1"

    from mi355q.model import CONFIGS, LlamaRunner, decode_bytes_per_step

    model = args.model or ("llama3-8b" if n_gpus == 1 else "llama3-70b")
    cfg = CONFIGS[model]
    G = 128
    B, in_len = args.batch, args.input_len
    max_len = in_len + args.warmup + args.steps + 8
    max_len = (max_len + 255) // 256 * 256
    dev = f"cuda:{dev_index}"
    runner = LlamaRunner(cfg, quant=args.quant, kv_cache_dtype=args.kv_dtype, group_size=G, device=dev,
                         tp_rank=rank, tp_size=world, max_batch=B, max_seq_len=max_len)
    torch.cuda.synchronize()

    if args.roofline_only:
        print(json.dumps({"roofline": time_dominant_kernel(runner, B)}))
        return

    # ---- TTFT: prefill of the synthetic prompt + first sampled token ------------------------
    rng = torch.Generator().manual_seed(0)
    prompt = torch.randint(0, min(10000, cfg.vocab), (B, in_len), generator=rng)  # vllm/benchmarks/latency.py:104
    ttft = None
    if not args.skip_ttft:
        ts = []
        for i in range(20):      # 5 warm-ups + 15 timed iterations (.buildkite/performance-benchmarks/tests/latency-tests.json:8-9)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            runner.prefill(prompt)
            torch.cuda.synchronize()
            if i >= 5:
                ts.append(time.perf_counter() - t0)
        ttft = statistics.median(ts) * 1e3
    else:
        runner.reset(in_len)

    # ---- decode ------------------------------------------------------------------------------
    # collectives staged through the host (gloo) cannot be captured; with the exchange over peer memory the step has none
    graph = not args.no_graph and not (dist_on and torch.distributed.get_backend() == "gloo" and runner.peer_ar is None)
    graph_error = None
    if graph:
        try:
            runner.capture()
        except Exception as e:  # noqa: BLE001
            graph_error = f"{type(e).__name__}: {e}"[:300]
            if rank == 0:
                print(f"[bench] graph capture failed ({graph_error}); running eagerly", file=sys.stderr)
            graph = False
            runner.graph = None
        if dist_on:   # every rank must take the same path: one eager rank would deadlock the others' replays
            flag = torch.tensor([1 if graph else 0], device=dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            if int(flag.item()) == 0 and graph:
                graph, graph_error = False, "another rank failed to capture"
                runner.graph = None
    for _ in range(args.warmup):
        runner.decode_step()

    def barrier():
        if dist_on:
            torch.distributed.barrier()

    trace = []
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.decode_step()
        if args.tokens_out:     # untimed use only: the copy synchronises every step
            trace.append(runner.tokens.cpu().tolist())
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if args.tokens_out and rank == 0:
        with open(args.tokens_out, "w") as fh:
            json.dump(trace, fh)
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    step_s = elapsed / args.steps
    tok_s = B / step_s
    last_ids = runner.tokens.cpu().tolist()

    ctx_mid = in_len + args.warmup + args.steps // 2
    kv_bytes = 1 if args.kv_dtype == "fp8" else 2
    if args.quant == "none":
        from mi355q.model import linear_shapes
        step_bytes = sum(K * N * 2 for K, N in linear_shapes(cfg, world).values()) * cfg.layers + \
            cfg.vocab // world * cfg.hidden * 2 + 2 * cfg.layers * max(cfg.kv_heads // world, 1) * cfg.head_dim * ctx_mid * kv_bytes * B
    else:
        step_bytes = decode_bytes_per_step(cfg, B, ctx_mid, G, args.quant == "awq", kv_bytes, world)

    out = {
        "metric": "decode output tok/s (Llama-3-8B GPTQ-int4 TP=1 / 70B int4 TP=N), p50 TTFT in ttft_ms_p50",
        "value": tok_s, "unit": "tok/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "strong" if n_gpus > 1 else "weak",
        "vs_baseline": None, "dtype": "int4 weights x bf16 activations, fp32 accumulate", "data": "synthetic",
        "config": {"workload": "%s %s-int4 g128 decode, TP=%d, batch %d, prompt %d, %s KV, %s" % (
            model, args.quant, world, B, in_len, "fp8-e4m3" if args.kv_dtype == "fp8" else "bf16",
            "hipGraph" if graph else "eager"), "parallelism": "tp%d" % world, "batch": B, "input_len": in_len},
        "ttft_ms_p50": ttft,
        "graph": bool(graph),          # false: capture (of the RCCL all-reduce at N > 1) failed or was disabled; eager steps timed
        "graph_error": graph_error,
        "step_bytes_per_gpu": step_bytes,
        "step_hbm_gbs_per_gpu": step_bytes / step_s / 1e9,
        "step_hbm_frac": step_bytes / step_s / 1e9 / HBM_PEAK_GBS,
        "step_hbm_frac_of_measured_stream_peak": step_bytes / step_s / 1e9 / STREAM_PEAK_GBS,
        "last_token_ids": last_ids[:4],
    }
    if dist_on and torch.distributed.get_backend() == "nccl":
        # every rank takes part; latency of the 16 KiB exchange a 70B decode step makes 160 times
        ar_eager, ar_graph, ar_err = time_allreduce_16k(dev)
        out["allreduce_us_16k"] = {"eager": ar_eager, "captured": ar_graph, "capture_error": ar_err, "world": world,
                                   "backend": "nccl (RCCL)"}
    if world > 1 or runner.peer_ar is not None:
        # which exchange the decode step used: one kernel over peer memory, or the group's collectives
        out["tp_exchange"] = "peer memory, one-shot" if runner.peer_ar is not None else \
            "process group (%s): %s" % (torch.distributed.get_backend(), getattr(runner, "peer_ar_note", ""))
        if runner.peer_ar is not None:
            pe, pg, perr = time_allreduce_16k(dev, fn=runner.peer_ar.all_reduce)
            out["peer_allreduce_us_16k"] = {"eager": pe, "captured": pg, "capture_error": perr, "world": world,
                                            "timeouts": runner.peer_ar.timeouts()}
    if rank == 0:
        out["roofline"] = time_dominant_kernel(runner, B)
    if rank == 0 and n_gpus == 1 and not args.skip_cpu:
        out["cpu_baseline"] = cpu_baseline(cfg, G, B)
        out["cpu_baseline_opt125m"] = cpu_baseline_opt125m()
    if rank == 0 and n_gpus == 1 and args.quant != "none" and not args.skip_bf16 and not cfg.num_experts:
        # the same harness on the unquantized model: the "x its own bf16 decode" comparator
        runner = None
        torch.cuda.empty_cache()
        r16 = LlamaRunner(cfg, quant="none", kv_cache_dtype=args.kv_dtype, device=dev, max_batch=B, max_seq_len=max_len)
        r16.reset(in_len)
        if graph:
            r16.capture()
        for _ in range(4):
            r16.decode_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n16 = max(8, args.steps // 4)
        for _ in range(n16):
            r16.decode_step()
        torch.cuda.synchronize()
        s16 = (time.perf_counter() - t0) / n16
        out["bf16_tok_s"] = B / s16
        out["speedup_vs_bf16"] = s16 / step_s
    if rank == 0 and n_gpus == 1 and args.quant != "none" and not args.skip_shapes and (B, in_len) == (1, 128):
        # the other fixed (batch, input_len) shapes of the latency benchmark, same model and harness
        runner = r16 = None
        out["other_shapes"] = []
        for b2, in2 in ((1, 2048), (8, 32), (32, 1024)):
            torch.cuda.empty_cache()
            ml2 = (in2 + 4 + 32 + 8 + 255) // 256 * 256
            r2 = LlamaRunner(cfg, quant=args.quant, kv_cache_dtype=args.kv_dtype, group_size=G, device=dev,
                             max_batch=b2, max_seq_len=ml2)
            p2 = torch.randint(0, min(10000, cfg.vocab), (b2, in2), generator=torch.Generator().manual_seed(0))
            ts = []
            for i in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r2.prefill(p2)
                torch.cuda.synchronize()
                if i >= 2:
                    ts.append(time.perf_counter() - t0)
            if graph:
                r2.capture()
            for _ in range(4):
                r2.decode_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(32):
                r2.decode_step()
            torch.cuda.synchronize()
            s2 = (time.perf_counter() - t0) / 32
            bytes2 = decode_bytes_per_step(cfg, b2, in2 + 4 + 16, G, args.quant == "awq", kv_bytes, world)
            out["other_shapes"].append({"batch": b2, "input_len": in2, "tok_s": b2 / s2, "ms_per_step": s2 * 1e3,
                                        "ttft_ms_p50": statistics.median(ts) * 1e3,
                                        "step_hbm_frac": bytes2 / s2 / 1e9 / HBM_PEAK_GBS})
            del r2
    if rank == 0 and n_gpus == 1 and not dist_on and not args.skip_configs and model == "llama3-8b" and args.quant == "gptq" \
            and (B, in_len) == (1, 128):
        # the benchmark's other single-GPU configurations, same harness, short runs: Llama-3-8B AWQ int4 with an
        # fp8 KV cache (hipGraph decode) and Mixtral 8x7B AWQ int4 (routed experts), batch 1
        runner = r16 = r2 = None
        out["other_configs"] = []
        for m2, q2, kv2, in2 in (("llama3-8b", "awq", "fp8", 128), ("llama3-8b", "awq", "fp8", 2048),
                                 ("mixtral-8x7b", "awq", "auto", 128)):
            torch.cuda.empty_cache()
            cfg2 = CONFIGS[m2]
            ml2 = (in2 + 4 + 32 + 8 + 255) // 256 * 256
            r2 = LlamaRunner(cfg2, quant=q2, kv_cache_dtype=kv2, group_size=G, device=dev, max_batch=1, max_seq_len=ml2)
            p2 = torch.randint(0, min(10000, cfg2.vocab), (1, in2), generator=torch.Generator().manual_seed(0))
            ts = []
            for i in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r2.prefill(p2)
                torch.cuda.synchronize()
                if i >= 2:
                    ts.append(time.perf_counter() - t0)
            g2 = graph
            if g2:
                try:
                    r2.capture()
                except Exception as e:  # noqa: BLE001
                    g2 = False
                    r2.graph = None
                    print(f"[bench] {m2}: graph capture failed ({type(e).__name__}: {e}); eager", file=sys.stderr)
            for _ in range(4):
                r2.decode_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(32):
                r2.decode_step()
            torch.cuda.synchronize()
            s2 = (time.perf_counter() - t0) / 32
            bytes2 = decode_bytes_per_step(cfg2, 1, in2 + 4 + 16, G, q2 == "awq", 1 if kv2 == "fp8" else 2, 1)
            out["other_configs"].append({
                "workload": "%s %s-int4 g128 decode, TP=1, batch 1, prompt %d, %s KV, %s" % (
                    m2, q2, in2, "fp8-e4m3" if kv2 == "fp8" else "bf16", "hipGraph" if g2 else "eager"),
                "tok_s": 1 / s2, "ms_per_step": s2 * 1e3, "ttft_ms_p50": statistics.median(ts) * 1e3,
                "step_bytes": bytes2, "step_hbm_frac": bytes2 / s2 / 1e9 / HBM_PEAK_GBS})
            del r2
            r2 = None
    if rank == 0:
        print(json.dumps(out))
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
