"""Registration of this package's kernels inside a running vLLM (only imported when `vllm` is
importable).  Uses the fork's public registration points and nothing else:

  * register_linear_kernel(cls, PlatformEnum.ROCM, "mp") appends at the lowest priority
    (kernels/linear/__init__.py:1091-1137), so the classes are then moved to the front of
    _POSSIBLE_KERNELS[PlatformEnum.ROCM] to win choose_mp_linear_kernel (:724-794);
  * register_linear_kernel(cls, PlatformEnum.ROCM, "fp8") likewise appends to _POSSIBLE_FP8_KERNELS
    (:352-360), from where init_fp8_linear_kernel (:615-685) picks the first class that can implement
    the layer: the fp8 x fp8 kernel class is moved to the front as well, so the fork's own
    Fp8LinearMethod (fp8.py:354-361) runs its per-tensor / per-token checkpoints on it;
  * register_backend(AttentionBackendEnum.CUSTOM, "<path>") (v1/attention/backends/registry.py
    :242-294), selected with --attention-backend CUSTOM; MI355Q_OVERRIDE_ROCM_ATTN=1 also
    replaces the fork's ROCM_ATTN entry, so that the platform's default choice
    (platforms/rocm.py:470-484) lands on this backend;
  * register_quantization_config("auto_awq" / "awq" / "awq_marlin") (quantization/__init__.py
    :59-106): the fork's ROCm AWQ method calls ops.awq_gemm / awq_dequantize directly
    (auto_awq.py:934-958) instead of asking for an MPLinearKernel, so the linear method is
    replaced by a subclass that repacks once and runs the MI355X GEMM.

Every class handed to vLLM is a real subclass of the fork's own base class, defined here with
explicit delegation to the implementation in this package.
"""
import os


def make_mp_kernel_classes():
    """(W4A16 class, W8A16 class) deriving from the fork's MPLinearKernel."""
    import torch
    from vllm.model_executor.kernels.linear.mixed_precision.MPLinearKernel import MPLinearKernel as VllmMPLinearKernel

    from . import mp_kernel

    def _shadow(impl, c):
        # vLLM's ScalarType objects are another class than this package's: match them by name
        mine = {str(t): t for t in impl.SUPPORTED_QUANT_TYPES}
        wt = mine.get(str(c.weight_type))
        if wt is None:
            return None, f"Quant type {c.weight_type} not supported; supported: {list(mine)}"
        return mp_kernel.MPLinearLayerConfig(tuple(c.full_weight_shape), tuple(c.partition_weight_shape), wt,
                                             c.act_type, c.group_size, c.zero_points, c.has_g_idx, c.out_type), None

    def _derive(impl, name):
        class Kernel(VllmMPLinearKernel):
            __doc__ = impl.__doc__
            SUPPORTED_QUANT_TYPES = impl.SUPPORTED_QUANT_TYPES

            @classmethod
            def get_min_capability(cls) -> int:
                return impl.get_min_capability()

            @classmethod
            def can_implement(cls, c):
                shadow, why = _shadow(impl, c)
                if shadow is None:
                    return False, why
                return impl.can_implement(shadow)

            def __init__(self, c, w_q_param_name, w_s_param_name, w_zp_param_name=None, w_gidx_param_name=None,
                         src_layout=None, zp_mode=None):
                ok, why = self.can_implement(c)
                if not ok:
                    raise ValueError(f"{name} cannot implement this layer: {why}")
                VllmMPLinearKernel.__init__(self, c, w_q_param_name, w_s_param_name, w_zp_param_name,
                                            w_gidx_param_name)
                self.src_layout, self.zp_mode = src_layout, zp_mode

            def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
                impl.process_weights_after_loading(self, layer)

            def apply_weights(self, layer, x, bias=None):
                return impl.apply_weights(self, layer, x, bias)

        for extra in ("apply_weights_silu_mul", "apply_weights_add_rms_norm"):
            if hasattr(impl, extra):
                setattr(Kernel, extra, getattr(impl, extra))
        Kernel.__name__ = Kernel.__qualname__ = name
        return Kernel

    if not hasattr(mp_kernel.MI355XW8A16LinearKernel, "SUPPORTED_QUANT_TYPES"):
        mp_kernel.MI355XW8A16LinearKernel.SUPPORTED_QUANT_TYPES = (mp_kernel.scalar_types.uint8b128,)
    return (_derive(mp_kernel.MI355XW4A16LinearKernel, "MI355XW4A16LinearKernel"),
            _derive(mp_kernel.MI355XW8A16LinearKernel, "MI355XW8A16LinearKernel"))


def make_fp8_kernel_class():
    """FP8ScaledMMLinearKernel subclass of the fork: activation quantisation and the scaled product both
    run in libmi355q (ScaledMMLinearKernel.py:108-187 drives it: quantise unless the caller hands over an
    already quantised activation, then apply_scaled_mm)."""
    import torch
    from vllm.model_executor.kernels.linear.scaled_mm.ScaledMMLinearKernel import (FP8ScaledMMLinearKernel,
                                                                                   ScaledMMLinearKernel)

    from .. import ops

    class _Quant:
        """Stands where the base class keeps its QuantFP8 module: same call, (x, scale, scale_ub) ->
        (fp8 tensor, scale)."""

        def __init__(self, static: bool, per_token: bool):
            self.static, self.per_token = static, per_token

        def __call__(self, x, scale=None, scale_ub=None):
            if self.static:
                return ops.scaled_fp8_quant(x, scale=scale)
            return ops.scaled_fp8_quant(x, scale_ub=scale_ub, use_per_token_if_dynamic=self.per_token)

    class MI355XFP8ScaledMMLinearKernel(FP8ScaledMMLinearKernel):
        @classmethod
        def is_supported(cls, compute_capability=None):
            return True, None

        @classmethod
        def can_implement(cls, c):
            a, w = c.activation_quant_key.scale, c.weight_quant_key.scale
            if not (a.group_shape.is_per_tensor() or (a.group_shape.is_per_token() and not a.static)):
                return False, f"activation scales must be per tensor or dynamic per token, got {a.group_shape}"
            if not (w.group_shape.is_per_tensor() or w.group_shape.is_per_channel()):
                return False, f"weight scales must be per tensor or per output channel, got {w.group_shape}"
            if c.out_dtype not in (None, torch.bfloat16, torch.float16) or c.input_dtype not in (torch.bfloat16, torch.float16):
                return False, "activations and outputs must be float16 or bfloat16"
            K = c.weight_shape[1] if len(c.weight_shape) == 2 else 0
            if K % 16:
                return False, f"the input size per rank ({K}) must be a multiple of 16"
            return True, None

        def __init__(self, c, layer_param_names):
            # what FP8ScaledMMLinearKernel.__init__ (ScaledMMLinearKernel.py:111-121) sets up, with this
            # package's quantiser where it builds a QuantFP8 module
            a = c.activation_quant_key.scale
            self.quant_fp8 = _Quant(a.static, a.group_shape.is_per_token())
            self.fp8_dtype = ops.FP8_DTYPE
            ScaledMMLinearKernel.__init__(self, c, layer_param_names)

        def apply_scaled_mm(self, *, A, B, out_dtype, As, Bs, bias, output_shape):
            # B is the [K, N] transpose view the fork's linear method leaves in layer.weight (fp8.py:400):
            # its transpose is the checkpoint's [N, K] tensor with contiguous rows
            w = B.t()
            if w.stride(-1) != 1:
                w = w.contiguous()
            return ops.fp8_scaled_mm(A, w, out_dtype, As, Bs, bias).view(*output_shape)

    return MI355XFP8ScaledMMLinearKernel


def make_awq_config_class():
    """AutoAWQConfig subclass of the fork whose plain linear method runs the MI355X kernel."""
    import torch
    from vllm.model_executor.layers.quantization import auto_awq as ref

    from .. import _lib, ops

    class AutoAWQLinearMethod(ref.AutoAWQLinearMethod):   # the name is what WEIGHT_LOADER_V2_SUPPORTED lists
        """Checkpoint tensors are created and loaded by the fork's BaseAWQLinearMethod
        (auto_awq.py:846-919); after loading they are repacked once into the kernel layout."""

        def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
            qc = self.quant_config
            K = layer.qweight.shape[0]
            G = K if qc.group_size == -1 else qc.group_size
            w = ops.w4a16_repack(layer.qweight.data.contiguous(), layer.scales.data.contiguous(),
                                 layer.qzeros.data.contiguous(), G, _lib.W_AWQ, _lib.ZP_AWQ)
            for name, t in (("qweight", w.packed), ("scales", w.scales), ("qzeros", w.zc)):
                delattr(layer, name)
                layer.register_parameter(name, torch.nn.Parameter(t, requires_grad=False))
            layer.mi355q_weight = ops.PackedW4(layer.qweight.data, layer.scales.data, layer.qzeros.data, w.K, w.N, G,
                                               w.tile)

        def apply(self, layer, x, bias=None):
            return ops.w4a16_gemm(x, layer.mi355q_weight, bias=bias)

    class AutoAWQConfig(ref.AutoAWQConfig):
        def get_quant_method(self, layer, prefix):
            m = super().get_quant_method(layer, prefix)
            if type(m) is ref.AutoAWQLinearMethod:
                ok = self.weight_bits == 4 and (self.group_size == -1 or self.group_size % 32 == 0)
                return AutoAWQLinearMethod(self) if ok else m
            return m

    return AutoAWQConfig


def hook_into_vllm() -> None:
    from vllm.model_executor.kernels.linear import _POSSIBLE_FP8_KERNELS, _POSSIBLE_KERNELS, register_linear_kernel
    from vllm.model_executor.layers.quantization import register_quantization_config
    from vllm.platforms.interface import PlatformEnum
    from vllm.v1.attention.backends.registry import AttentionBackendEnum, register_backend

    lst = _POSSIBLE_KERNELS.setdefault(PlatformEnum.ROCM, [])
    have = {k.__name__ for k in lst}
    for pos, kernel in enumerate(make_mp_kernel_classes()):
        if kernel.__name__ in have:
            continue
        register_linear_kernel(kernel, PlatformEnum.ROCM, "mp")
        lst.insert(pos, lst.pop(lst.index(kernel)))

    fp8_lst = _POSSIBLE_FP8_KERNELS.setdefault(PlatformEnum.ROCM, [])
    if "MI355XFP8ScaledMMLinearKernel" not in {k.__name__ for k in fp8_lst}:
        fp8_kernel = make_fp8_kernel_class()
        register_linear_kernel(fp8_kernel, PlatformEnum.ROCM, "fp8")
        fp8_lst.insert(0, fp8_lst.pop(fp8_lst.index(fp8_kernel)))

    path = "mi355q.plugin.vllm_attention."
    register_backend(AttentionBackendEnum.CUSTOM, path + "MI355XAttentionBackend")
    if os.environ.get("MI355Q_OVERRIDE_ROCM_ATTN", "0") == "1":
        register_backend(AttentionBackendEnum.ROCM_ATTN, path + "MI355XRocmAttnBackend")

    awq = make_awq_config_class()
    for name in ("auto_awq", "awq", "awq_marlin"):
        register_quantization_config(name)(awq)
