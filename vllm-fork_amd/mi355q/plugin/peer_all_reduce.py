"""One-shot all-reduce of small tensors over peer memory, for the tensor-parallel group of this process.

Host side of `mi355q_peer_ar_*` (csrc/allreduce.hip), with the life cycle and the questions the reference's
communicators answer (vllm/distributed/device_communicators/quick_all_reduce.py:44-62 `QuickAllReduce`,
custom_all_reduce.py `CustomAllreduce`, as cuda_communicator.py:275-340 consults them): is it usable for this
group (`disabled`), should this tensor take it (`should_use`), reduce it (`all_reduce`).  One process per GPU;
the IPC handles of the ranks' regions travel through the process group the communicator is built on.

Construction ends with a check on the real group: a series of reductions of random tensors, eagerly and as a
replayed graph, must agree with the group's own all-reduce and be identical on all ranks, with no timeout
counted; every rank learns the common verdict, and a communicator that failed it anywhere is disabled
everywhere (callers then keep the library collective).
"""
import ctypes
import os

import torch
import torch.distributed as dist

from .. import _lib
from .._lib import check
from ..ops import _dt, _stream

DEFAULT_MAX_BYTES = 256 * 1024     # [16 tokens, 8192] in 16 bits; larger sums go to the library collective


class PeerAllReduce:
    def __init__(self, group=None, device=None, max_bytes: int = DEFAULT_MAX_BYTES, self_check: bool = True):
        self.group = group
        self.disabled = True
        self.reason = ""
        self._comm = None
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if device is None:
            device = f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else "cuda:0"
        self.device = torch.device(device)
        self.max_bytes = max_bytes
        if os.environ.get("MI355Q_PEER_AR", "1") == "0":
            self.reason = "switched off by MI355Q_PEER_AR=0"
            return
        # MI355Q_PEER_AR_LOOPBACK=1: a group of one rank exchanges with itself (the host side and the kernel under a
        # process group of the real backend, on a single device)
        least = 1 if os.environ.get("MI355Q_PEER_AR_LOOPBACK") == "1" else 2
        if not least <= self.world <= 8:
            self.reason = f"{self.world} ranks (2..8 served)"
            return
        ok, why = self._setup()
        ok, why = self._agree(ok, why)
        if ok and self_check:
            ok, why = self._agree(*self._self_check())
        if ok and not self._host_group() and self.world > 1:
            ok, why = self._not_slower_than_the_group()
        self.disabled, self.reason = (not ok), why
        if self.disabled:
            self.close()

    # -- construction ------------------------------------------------------------------------------
    def _host_group(self) -> bool:
        return dist.get_backend(self.group) == "gloo"

    def _agree(self, ok: bool, why: str):
        """Every rank learns whether all ranks are fine (the collective runs on every rank whatever its own state)."""
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if self._host_group() else self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag.item()) == 1:
            return True, ""
        return False, why or "another rank could not set the communicator up"

    def _setup(self):
        L = _lib.lib()
        try:
            torch.cuda.set_device(self.device)
            comm = ctypes.c_void_p()
            check(L.mi355q_peer_ar_init(self.rank, self.world, self.max_bytes, ctypes.byref(comm)))
            self._comm = comm
            hb = L.mi355q_peer_ar_handle_bytes()
            buf = (ctypes.c_ubyte * hb)()
            check(L.mi355q_peer_ar_get_handle(self._comm, buf))
            mine = torch.tensor(list(bytes(buf)), dtype=torch.uint8)
            ok, why = True, ""
        except Exception as e:  # noqa: BLE001  (no device, no IPC, ...: whatever it is, the group keeps its collectives)
            hb = L.mi355q_peer_ar_handle_bytes()
            mine = torch.zeros(hb, dtype=torch.uint8)
            ok, why = False, f"set-up failed: {type(e).__name__}: {e}"
        # the exchange runs on every rank, failed or not, so that no rank waits for another for ever
        where = "cpu" if self._host_group() else self.device
        parts = [torch.empty(hb, dtype=torch.uint8, device=where) for _ in range(self.world)]
        dist.all_gather(parts, mine.to(where), group=self.group)
        if not ok:
            return ok, why
        try:
            handles = b"".join(bytes(p.cpu().tolist()) for p in parts)
            check(L.mi355q_peer_ar_open_handles(self._comm, handles))
        except Exception as e:  # noqa: BLE001
            return False, f"opening the peers' regions failed: {type(e).__name__}: {e}"
        return True, ""

    def _self_check(self):
        try:
            gen = torch.Generator(device=self.device).manual_seed(1234 + self.rank)
            host = self._host_group()

            def lib_sum(t):
                if host:
                    c = t.float().cpu()
                    dist.all_reduce(c, group=self.group)
                    return c.to(self.device)
                c = t.float()
                dist.all_reduce(c, group=self.group)
                return c

            def same_everywhere(t):
                lo, hi = t.clone().view(torch.int16).int(), t.clone().view(torch.int16).int()
                if host:
                    lo, hi = lo.cpu(), hi.cpu()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
                dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
                return bool(torch.equal(lo, hi))

            # a rank that has seen a failure keeps going to the end: every rank must issue the same collectives
            fail = ""
            # first a single exchange, and a common verdict on it: where peers' stores never become visible every
            # further call would sit out its time limit
            x = torch.randn(8192, device=self.device, generator=gen).to(torch.bfloat16)
            want = lib_sum(x)
            got = self.all_reduce(x.clone())
            first_ok = torch.allclose(got.float(), want, rtol=2e-2, atol=2e-2 * self.world) and self.timeouts() == 0
            ok, _ = self._agree(bool(first_ok), "")
            if not ok:
                return False, "self-check: the first exchange failed or ran into its time limit on some rank"
            for dtype in (torch.bfloat16, torch.float16):
                for n in (2, 4096, 8192, 12346, self.max_bytes // 2):
                    for _ in range(3):
                        x = torch.randn(n, device=self.device, generator=gen).to(dtype)
                        want = lib_sum(x)
                        got = self.all_reduce(x.clone())
                        if not torch.allclose(got.float(), want, rtol=2e-2, atol=2e-2 * self.world):
                            fail = fail or f"self-check: {n} x {dtype} differs from the group's all-reduce"
                        if not same_everywhere(got):
                            fail = fail or f"self-check: {n} x {dtype}: ranks hold different results"
            # replayed graph: the epoch advances on the device
            x = torch.randn(8192, device=self.device, generator=gen).to(torch.bfloat16)
            y = torch.empty_like(x)
            s = torch.cuda.Stream(device=self.device)
            s.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(s):
                self.all_reduce(x, out=y)
                g = torch.cuda.CUDAGraph()
                # a process group's watchdog thread queries events while this thread captures: only this thread's
                # calls may invalidate the capture
                with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                    self.all_reduce(x, out=y)
                    self.all_reduce(y, out=y)
                for _ in range(5):
                    g.replay()
            torch.cuda.current_stream(self.device).wait_stream(s)
            torch.cuda.synchronize(self.device)
            want = lib_sum(lib_sum(x).to(torch.bfloat16))
            if not torch.allclose(y.float(), want, rtol=3e-2, atol=3e-2 * self.world * self.world):
                fail = fail or "self-check: the replayed graph differs from the group's all-reduce"
            if not same_everywhere(y):
                fail = fail or "self-check: replayed graph: ranks hold different results"
            for rows, n in ((1, 16032), (3, 514)):
                if rows * n * 2 > self.max_bytes:
                    continue
                x = torch.randn((rows, n), device=self.device, generator=gen).to(torch.bfloat16)
                src = x.cpu() if host else x
                parts = [torch.empty_like(src) for _ in range(self.world)]
                dist.all_gather(parts, src, group=self.group)
                if not torch.equal(self.all_gather(x).cpu(), torch.cat(parts, dim=-1).cpu()):
                    fail = fail or f"self-check: all-gather of {rows} x {n} differs from the group's"
            if self.timeouts():
                fail = fail or "self-check: a wait for a peer ran into its time limit"
            return (not fail), fail
        except RuntimeError as e:
            return False, f"self-check failed: {e}"

    def _not_slower_than_the_group(self):
        """The exchange is kept only where it is at least as fast as the group's own all-reduce at the size it is
        for (16 KiB, back-to-back calls; the slowest rank's times decide, so all ranks decide alike)."""
        x = torch.ones(8192, dtype=torch.bfloat16, device=self.device)

        def per_call(fn, reps=100):
            for _ in range(10):
                fn(x)
            torch.cuda.synchronize(self.device)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn(x)
            b.record()
            torch.cuda.synchronize(self.device)
            return a.elapsed_time(b) * 1e3 / reps

        t = torch.tensor([per_call(self.all_reduce), per_call(lambda v: dist.all_reduce(v, group=self.group))],
                         dtype=torch.float32, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        mine, theirs = float(t[0]), float(t[1])
        self.us_16k = {"peer": mine, "group": theirs}
        if self.timeouts() or mine > theirs:
            return False, f"the group's all-reduce is faster at 16 KiB ({theirs:.1f} us against {mine:.1f})"
        return True, ""

    # -- use ---------------------------------------------------------------------------------------
    def should_use(self, x: torch.Tensor) -> bool:
        return (not self.disabled and x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.is_contiguous()
                and x.numel() % 2 == 0 and 0 < x.numel() * 2 <= self.max_bytes)

    def all_reduce(self, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """Sum of x over the group; into `out` (same shape and dtype; default: in place)."""
        if self._comm is None:
            raise RuntimeError("PeerAllReduce: communicator is closed")
        out = x if out is None else out
        if out.shape != x.shape or out.dtype != x.dtype or not out.is_contiguous() or not x.is_contiguous():
            raise RuntimeError("PeerAllReduce: input and output must be contiguous tensors of one shape and dtype")
        check(_lib.lib().mi355q_peer_ar_all_reduce(self._comm, x.data_ptr(), out.data_ptr(), x.numel(), _dt(x), _stream()))
        return out

    def should_gather(self, x: torch.Tensor) -> bool:
        return (not self.disabled and x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.dim() == 2
                and x.is_contiguous() and x.shape[1] % 2 == 0 and 0 < x.numel() * 2 <= self.max_bytes)

    def all_gather(self, x: torch.Tensor) -> torch.Tensor:
        """[rows, n] shards of all ranks -> [rows, world * n], rank order along the last dimension."""
        if self._comm is None:
            raise RuntimeError("PeerAllReduce: communicator is closed")
        if x.dim() != 2 or not x.is_contiguous():
            raise RuntimeError("PeerAllReduce.all_gather: a contiguous [rows, n] tensor is expected")
        out = torch.empty((x.shape[0], self.world * x.shape[1]), dtype=x.dtype, device=x.device)
        check(_lib.lib().mi355q_peer_ar_all_gather(self._comm, x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1],
                                                   _dt(x), _stream()))
        return out

    def timeouts(self) -> int:
        n = ctypes.c_int(0)
        check(_lib.lib().mi355q_peer_ar_timeouts(self._comm, ctypes.byref(n)))
        return n.value

    def close(self) -> None:
        if self._comm is not None:
            _lib.lib().mi355q_peer_ar_destroy(self._comm)
            self._comm = None
        self.disabled = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
