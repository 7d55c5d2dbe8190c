"""Mixed-precision (WNA16) linear-kernel contract, the MI355X kernel that implements it, and the
per-platform kernel list with its chooser.

Mirrors vllm/model_executor/kernels/linear/mixed_precision/MPLinearKernel.py:14-94 (config +
abstract kernel), kernels/linear/__init__.py:724-794 (choose_mp_linear_kernel: first kernel in
the platform's list that is not disabled by VLLM_DISABLED_KERNELS, meets the capability and
can_implement()s the layer; otherwise ValueError listing every reason) and :1091-1137
(register_linear_kernel appends at lowest priority).
"""
import os
from abc import ABC, abstractmethod
from dataclasses import dataclass

import torch

from .. import _lib, ops
from .scalar_type import ScalarType, scalar_types


@dataclass
class MPLinearLayerConfig:
    full_weight_shape: tuple[int, int]       # [in, out] of the unsharded layer
    partition_weight_shape: tuple[int, int]  # [in, out] on this rank
    weight_type: ScalarType
    act_type: torch.dtype
    group_size: int
    zero_points: bool
    has_g_idx: bool
    out_type: torch.dtype | None = None


class MPLinearKernel(ABC):
    @classmethod
    @abstractmethod
    def get_min_capability(cls) -> int: ...

    @classmethod
    @abstractmethod
    def can_implement(cls, c: MPLinearLayerConfig) -> tuple[bool, str | None]: ...

    def __init__(self, c: MPLinearLayerConfig, w_q_param_name: str, w_s_param_name: str,
                 w_zp_param_name: str | None = None, w_gidx_param_name: str | None = None) -> None:
        ok, why = self.can_implement(c)
        if not ok:
            raise ValueError(f"{type(self).__name__} cannot implement this layer: {why}")
        if c.zero_points and w_zp_param_name is None:
            raise ValueError("zero_points=True needs the name of the zero-point parameter")
        if c.has_g_idx and w_gidx_param_name is None:
            raise ValueError("has_g_idx=True needs the name of the g_idx parameter")
        self.config = c
        self.w_q_name, self.w_s_name = w_q_param_name, w_s_param_name
        self.w_zp_name, self.w_gidx_name = w_zp_param_name, w_gidx_param_name

    @abstractmethod
    def process_weights_after_loading(self, layer: torch.nn.Module) -> None: ...

    @abstractmethod
    def apply_weights(self, layer: torch.nn.Module, x: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor: ...

    def _get_weight_params(self, layer):
        return (getattr(layer, self.w_q_name), getattr(layer, self.w_s_name),
                getattr(layer, self.w_zp_name or "", None), getattr(layer, self.w_gidx_name or "", None))


def _plain(t: torch.Tensor) -> torch.nn.Parameter:
    return torch.nn.Parameter(t.data if isinstance(t, torch.nn.Parameter) else t, requires_grad=False)


def _set_param(layer: torch.nn.Module, name: str, t: torch.Tensor | None) -> None:
    """Swap in a plain nn.Parameter (the reference's replace_parameter: loader subclasses are
    dropped once loading is over so that graph capture sees ordinary tensors)."""
    if name in layer._parameters:
        del layer._parameters[name]
    elif hasattr(layer, name):
        delattr(layer, name)
    layer.register_parameter(name, None if t is None else _plain(t))


class MI355XW4A16LinearKernel(MPLinearKernel):
    """4-bit weights x 16-bit activations on gfx950 through libmi355q.

    After process_weights_after_loading the layer holds
      <w_q>  : int32 [K*N/8]   kernel tile layout (ops.w4a16_repack)
      <w_s>  : act  [N/tile, K/G, tile]  group scales, checkpoint values, strip-major
      w_zc   : act  [N/tile, K/G, tile]  (magic bias + zero point) per group, asymmetric formats only
    and `mi355q_weight`, the ops.PackedW4 view over them that apply_weights passes to the GEMM.

    Accepted checkpoint layouts, told apart by the loader attributes of the parameters:
      weight  int32 [K/8, N] or [N, K/8], packed along K  (GPTQ, compressed-tensors)
              int32 [K, N/8], packed along N, sequential nibbles
      scales  [K/G, N] or [N, K/G]
      zeros   int32 [K/G, N/8] or [N/8, K/G], sequential nibbles, value z (uint4) --
              GPTQ's historical z-1 storage and AWQ's nibble order are selected with
              `zp_mode` / `src_layout` by the linear method that owns the checkpoint format.
    """

    SUPPORTED_QUANT_TYPES = (scalar_types.uint4b8, scalar_types.uint4)

    def __init__(self, c, w_q_param_name, w_s_param_name, w_zp_param_name=None, w_gidx_param_name=None,
                 src_layout: int | None = None, zp_mode: int | None = None):
        super().__init__(c, w_q_param_name, w_s_param_name, w_zp_param_name, w_gidx_param_name)
        self.src_layout, self.zp_mode = src_layout, zp_mode

    @classmethod
    def get_min_capability(cls) -> int:
        return 95  # gfx950

    @classmethod
    def can_implement(cls, c: MPLinearLayerConfig) -> tuple[bool, str | None]:
        if c.weight_type not in cls.SUPPORTED_QUANT_TYPES:
            return False, f"Quant type {c.weight_type} not supported; supported: {list(cls.SUPPORTED_QUANT_TYPES)}"
        if c.act_type not in (torch.float16, torch.bfloat16):
            return False, "Only float16/bfloat16 activations are supported"
        if c.has_g_idx and c.partition_weight_shape[0] != c.full_weight_shape[0] and c.zero_points:
            # rows of one scale group are then spread over the ranks; the regrouped layout built for that
            # case (process_weights_after_loading) pads groups with rows that multiply to zero, which
            # needs the constant zero point of a symmetric format
            return False, ("Activation reordering (g_idx) with the input dimension sharded is only supported "
                           "for symmetric weights")
        if c.weight_type == scalar_types.uint4 and not c.zero_points:
            return False, "uint4 weights need zero points (symmetric checkpoints use uint4b8)"
        K, N = c.partition_weight_shape
        G = K if c.group_size == -1 else c.group_size
        if G <= 0 or K % G:
            return False, f"Input features {K} not divisible by group size {G}"
        if K % 32 or G % 32:
            return False, f"Input features ({K}) and group size ({G}) must be multiples of 32"
        if N % 64:
            return False, f"Output features ({N}) must be a multiple of 64"
        return True, None

    # -- load time ---------------------------------------------------------
    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        c = self.config
        K, N = c.partition_weight_shape
        G = K if c.group_size == -1 else c.group_size
        w_q, w_s, w_zp, _ = self._get_weight_params(layer)

        q = w_q.data
        packed_dim = getattr(w_q, "packed_dim", None)
        in_dim = getattr(w_q, "input_dim", 0)
        layout = self.src_layout
        if tuple(q.shape) == (K // 8, N) and (packed_dim is None or packed_dim == in_dim):
            layout = _lib.W_GPTQ if layout is None else layout
        elif tuple(q.shape) == (N, K // 8):
            q, layout = q.t().contiguous(), _lib.W_GPTQ
        elif tuple(q.shape) == (K, N // 8):
            layout = _lib.W_NSEQ if layout is None else layout
        else:
            raise ValueError(f"{self.w_q_name}: shape {tuple(q.shape)} is none of [K/8,N], [N,K/8], [K,N/8] "
                             f"for K={K}, N={N}")
        if c.has_g_idx and K != c.full_weight_shape[0]:
            # named through the class: the fork-side kernel class borrows this function without inheriting
            return MI355XW4A16LinearKernel._process_sharded_act_order(self, layer, q, layout)
        s = w_s.data
        if tuple(s.shape) == (N, K // G) and (K // G != N or getattr(w_s, "output_dim", 1) == 0):
            s = s.t()
        if tuple(s.shape) != (K // G, N):
            raise ValueError(f"{self.w_s_name}: shape {tuple(w_s.shape)} != ({K // G}, {N})")
        zp_mode, z = _lib.ZP_CONST, None
        if c.zero_points:
            if w_zp is None:
                raise ValueError("layer has no zero-point parameter but the config asks for zero points")
            z = w_zp.data
            if tuple(z.shape) == (N // 8, K // G) and (K // G != N // 8 or getattr(w_zp, "output_dim", 1) == 0):
                z = z.t()
            if tuple(z.shape) != (K // G, N // 8):
                raise ValueError(f"{self.w_zp_name}: shape {tuple(w_zp.shape)} != ({K // G}, {N // 8})")
            zp_mode = self.zp_mode if self.zp_mode is not None else _lib.ZP_NSEQ
        # Activation order (GPTQ desc_act): row k belongs to group g_idx[k].  Sorting the rows by
        # group makes groups contiguous again (what gptq_shuffle + the b_q_perm gather do in the
        # reference, q_gemm.cu:224-228,1857-1868); the same permutation is applied to the input
        # features at run time.
        layer.mi355q_perm = None
        if c.has_g_idx:
            w_g = getattr(layer, self.w_gidx_name) if self.w_gidx_name else None
            perm = _act_order_perm(None if w_g is None else w_g.data, K, G)
            if perm is not None:
                if layout != _lib.W_GPTQ:
                    raise ValueError("activation reordering is defined for weights packed along K (GPTQ layout)")
                q = _permute_packed_rows(q.contiguous(), perm)
                layer.mi355q_perm = perm
        w = ops.w4a16_repack(q.contiguous(), s.contiguous().to(c.act_type), None if z is None else z.contiguous(),
                             G, layout, zp_mode)
        _set_param(layer, self.w_q_name, w.packed)
        _set_param(layer, self.w_s_name, w.scales)
        if self.w_zp_name is not None:
            _set_param(layer, self.w_zp_name, None)   # folded into w_zc (or unused: symmetric)
        if self.w_gidx_name is not None:
            _set_param(layer, self.w_gidx_name, None)
        _set_param(layer, "w_zc", w.zc)
        layer.mi355q_weight = ops.PackedW4(getattr(layer, self.w_q_name).data, getattr(layer, self.w_s_name).data,
                                           None if w.zc is None else layer.w_zc.data, K, N, G, w.tile)

    def _process_sharded_act_order(self, layer: torch.nn.Module, q: torch.Tensor, layout: int) -> None:
        """desc_act checkpoint on a row-parallel layer: this rank holds K of the K_full input features, row k
        belongs to group g_idx[k] of the K_full / G groups, whose scales every rank holds in full
        (auto_gptq.py:367-378).  A group has any number of its rows here, so the rows are sorted by group and
        every group is padded to whole sub-groups of 32 rows with rows of the constant zero point (they
        multiply to zero whatever input they meet); the kernel then runs with group size 32, each sub-group
        carrying its group's scales, and the input features are gathered into the same order at run time
        (padding rows read feature 0)."""
        c = self.config
        (K, N), K_full = c.partition_weight_shape, c.full_weight_shape[0]
        G, SUB = c.group_size, 32
        if layout != _lib.W_GPTQ:
            raise ValueError("activation reordering is defined for weights packed along K (GPTQ layout)")
        _, w_s, _, w_g = self._get_weight_params(layer)
        scales = w_s.data
        if tuple(scales.shape) != (K_full // G, N):
            raise ValueError(f"{self.w_s_name}: shape {tuple(scales.shape)} != ({K_full // G}, {N}) "
                             "(scales of a reordered checkpoint are held in full on every rank)")
        if w_g is None or w_g.numel() != K:
            raise ValueError(f"g_idx must have {K} entries")
        dev = q.device
        g = w_g.data.to(device=dev, dtype=torch.int64)
        if int(g.min()) < 0 or int(g.max()) >= K_full // G:
            raise ValueError("g_idx names a group outside the checkpoint's scales")
        order = torch.argsort(g, stable=True)
        gs = g[order]
        counts = torch.bincount(gs, minlength=K_full // G)
        padded = (counts + SUB - 1) // SUB * SUB
        starts, first = torch.cumsum(padded, 0) - padded, torch.cumsum(counts, 0) - counts
        K2 = (int(padded.sum()) + 127) // 128 * 128
        dest = starts[gs] + (torch.arange(K, device=dev) - first[gs])
        shifts = torch.arange(0, 32, 4, device=dev, dtype=torch.int32)
        rows = ((q[:, None, :] >> shifts[None, :, None]) & 0xF).reshape(K, N)
        new_rows = torch.full((K2, N), 8, dtype=torch.int32, device=dev)
        new_rows[dest] = rows[order]
        perm = torch.zeros(K2, dtype=torch.int64, device=dev)
        perm[dest] = order
        group_of_sub = torch.zeros(K2 // SUB, dtype=torch.int64, device=dev)
        group_of_sub[: int(padded.sum()) // SUB] = torch.repeat_interleave(torch.arange(K_full // G, device=dev), padded // SUB)
        new_rows = new_rows.reshape(K2 // 8, 8, N)
        packed = torch.zeros((K2 // 8, N), dtype=torch.int32, device=dev)
        for i in range(8):
            packed |= new_rows[:, i, :] << (4 * i)
        w = ops.w4a16_repack(packed, scales[group_of_sub].contiguous().to(c.act_type), None, SUB, _lib.W_GPTQ, _lib.ZP_CONST)
        _set_param(layer, self.w_q_name, w.packed)
        _set_param(layer, self.w_s_name, w.scales)
        if self.w_zp_name is not None:
            _set_param(layer, self.w_zp_name, None)
        _set_param(layer, self.w_gidx_name, None)
        _set_param(layer, "w_zc", None)
        layer.mi355q_perm = perm
        layer.mi355q_weight = ops.PackedW4(getattr(layer, self.w_q_name).data, getattr(layer, self.w_s_name).data, None,
                                           K2, N, SUB, w.tile)

    # -- run time ----------------------------------------------------------
    def apply_weights(self, layer, x, bias=None):
        perm = getattr(layer, "mi355q_perm", None)
        if perm is not None:
            x = x.index_select(-1, perm)
        return ops.w4a16_gemm(x, layer.mi355q_weight, bias=bias)

    def apply_weights_silu_mul(self, layer, x, bias=None):
        """x = [.., 2K] gate_up output; SiluAndMul runs in the GEMM prologue."""
        if getattr(layer, "mi355q_perm", None) is not None:   # reordered inputs: activation first, then gather
            act = torch.empty(x.shape[:-1] + (x.shape[-1] // 2,), dtype=x.dtype, device=x.device)
            ops.silu_and_mul(act, x)
            return self.apply_weights(layer, act, bias)
        return ops.w4a16_gemm(x, layer.mi355q_weight, bias=bias, fuse_silu_mul=True)

    def apply_weights_add_rms_norm(self, layer, x, residual, norm_weight, epsilon, bias=None):
        """fused_add_rms_norm(x, residual) then the GEMM; returns (output, new residual)."""
        if getattr(layer, "mi355q_perm", None) is not None:
            if residual is None:
                residual = x
                normed = torch.empty_like(x)
                ops.rms_norm(normed, x, norm_weight, epsilon)
            else:
                normed, residual = x.clone(), residual.clone()
                ops.fused_add_rms_norm(normed, residual, norm_weight, epsilon)
            return self.apply_weights(layer, normed, bias), residual
        return ops.w4a16_gemm_add_rms_norm(x, residual, norm_weight, epsilon, layer.mi355q_weight, bias=bias)


class MI355XW8A16LinearKernel(MPLinearKernel):
    """8-bit symmetric integer weights (GPTQ 8 bits, uint8b128) x 16-bit activations on gfx950.

    After process_weights_after_loading the layer holds <w_q> as uint8 [N, K] (values biased by 128)
    and <w_s> as fp32 [N, K/G].  Checkpoint layouts: weight int32 [K/4, N] packed along K with scales
    [K/G, N] (GPTQ), or weight int32 [N, K/4] with scales [N, K/G] (compressed-tensors).  Activation
    reordering is not served (8-bit checkpoints with desc_act are refused)."""

    @classmethod
    def get_min_capability(cls) -> int:
        return 95

    @classmethod
    def can_implement(cls, c: MPLinearLayerConfig) -> tuple[bool, str | None]:
        if c.weight_type != scalar_types.uint8b128:
            return False, f"Quant type {c.weight_type} not supported; supported: [{scalar_types.uint8b128}]"
        if c.act_type not in (torch.float16, torch.bfloat16):
            return False, "Only float16/bfloat16 activations are supported"
        if c.zero_points:
            return False, "8-bit weights with zero points are not supported"
        if c.has_g_idx:
            return False, "Activation reordering (g_idx) is not supported for 8-bit weights"
        K, N = c.partition_weight_shape
        G = K if c.group_size == -1 else c.group_size
        if G <= 0 or K % G or G % 16 or K % 16:
            return False, f"Input features {K} / group size {G} must be multiples of 16 and divide"
        return True, None

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        c = self.config
        K, N = c.partition_weight_shape
        G = K if c.group_size == -1 else c.group_size
        w_q, w_s, _, _ = self._get_weight_params(layer)
        k_major = tuple(w_q.shape) == (K // 4, N) and (K // 4 != N or getattr(w_q, "output_dim", 1) != 0)
        if k_major:                                   # GPTQ: int32 [K/4, N], scales [K/G, N]
            w, s = ops.gptq8_repack(w_q.data, w_s.data, G)
        elif tuple(w_q.shape) == (N, K // 4):         # compressed-tensors: int32 [N, K/4], scales [N, K/G]
            if tuple(w_s.shape) != (N, K // G):
                raise ValueError(f"{self.w_s_name}: shape {tuple(w_s.shape)} != ({N}, {K // G})")
            # four consecutive k per word, least significant byte first: the bytes already are [N, K]
            w = w_q.data.contiguous().view(torch.uint8).reshape(N, K)
            s = w_s.data.to(torch.float32).contiguous()
        else:
            raise ValueError(f"{self.w_q_name}: shape {tuple(w_q.shape)} is neither ({K // 4}, {N}) nor ({N}, {K // 4})")
        _set_param(layer, self.w_q_name, w)
        _set_param(layer, self.w_s_name, s)
        if self.w_zp_name is not None:
            _set_param(layer, self.w_zp_name, None)
        if self.w_gidx_name is not None:
            _set_param(layer, self.w_gidx_name, None)
        layer.mi355q_group = G

    def apply_weights(self, layer, x, bias=None):
        out = ops.wi8a16_gemm(x, getattr(layer, self.w_q_name).data, getattr(layer, self.w_s_name).data,
                              layer.mi355q_group)
        if bias is not None:
            out.add_(bias)     # on the rounded product, as the reference's output.add_(bias)
        return out


def _act_order_perm(g_idx: torch.Tensor | None, K: int, G: int) -> torch.Tensor | None:
    """Row order that sorts a GPTQ g_idx by group (stable), or None when the rows already are in
    group order (checkpoints written without activation reordering carry a trivial or empty g_idx).
    A g_idx that was never filled in (all zeros with more than one group) is refused like any other
    assignment that does not give every group exactly group_size rows."""
    if g_idx is None or g_idx.numel() == 0:
        return None
    if g_idx.numel() != K:
        raise ValueError(f"g_idx has {g_idx.numel()} entries for {K} input features")
    g = g_idx.to(torch.int64)
    want = torch.arange(K, device=g.device, dtype=torch.int64) // G
    if torch.equal(g, want):
        return None
    perm = torch.argsort(g, stable=True)
    if not torch.equal(g[perm], want):
        raise ValueError("g_idx does not assign exactly group_size rows to every group")
    return perm


def _permute_packed_rows(qweight: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
    """qweight int32 [K/8, N] (8 consecutive k per word, nibble i at bit 4 i) with its K rows
    taken in the order `perm`."""
    K8, N = qweight.shape
    shifts = torch.arange(0, 32, 4, device=qweight.device, dtype=torch.int32)
    rows = ((qweight[:, None, :] >> shifts[None, :, None]) & 0xF).reshape(K8 * 8, N)
    rows = rows.index_select(0, perm.to(qweight.device)).reshape(K8, 8, N)
    out = torch.zeros((K8, N), dtype=torch.int32, device=qweight.device)
    for i in range(8):
        out |= rows[:, i, :] << (4 * i)
    return out


# platform name -> ordered kernel classes.  "rocm" is the only platform this package serves.
_POSSIBLE_KERNELS: dict[str, list[type[MPLinearKernel]]] = {"rocm": [MI355XW4A16LinearKernel,
                                                                  MI355XW8A16LinearKernel]}


def register_linear_kernel(kernel_class: type, platform: str = "rocm", kernel_type: str = "mp") -> None:
    if kernel_type != "mp":
        raise ValueError(f"Unrecognized kernel type: {kernel_type}")
    _POSSIBLE_KERNELS.setdefault(platform, []).append(kernel_class)


def _device_capability() -> int | None:
    if not torch.cuda.is_available():
        return None
    major, minor = torch.cuda.get_device_capability()
    return major * 10 + minor


def choose_mp_linear_kernel(config: MPLinearLayerConfig, compute_capability: int | None = None,
                            platform: str = "rocm") -> type[MPLinearKernel]:
    if compute_capability is None:
        compute_capability = _device_capability()
    disabled = {k.strip() for k in os.environ.get("VLLM_DISABLED_KERNELS", "").split(",") if k.strip()}
    reasons = []
    for kernel in _POSSIBLE_KERNELS.get(platform, []):
        if kernel.__name__ in disabled:
            reasons.append(f" {kernel.__name__} disabled by environment variable")
            continue
        if compute_capability is not None and kernel.get_min_capability() > compute_capability:
            reasons.append(f"{kernel.__name__} requires capability {kernel.get_min_capability()}, current compute "
                           f" capability is {compute_capability}")
            continue
        ok, why = kernel.can_implement(config)
        if ok:
            return kernel
        reasons.append(f" {kernel.__name__} cannot implement due to: {why}")
    raise ValueError("Failed to find a kernel that can implement the WNA16 linear layer. Reasons: \n" + "\n".join(reasons))
