"""Registration of this package's kernels inside a running vLLM (only imported when `vllm` is
importable).  Uses the reference's public registration points and nothing else:

  * register_linear_kernel(cls, PlatformEnum.ROCM, "mp") appends at the lowest priority
    (kernels/linear/__init__.py:1091-1137), so the class is moved to the front of
    _POSSIBLE_KERNELS[PlatformEnum.ROCM] to win choose_mp_linear_kernel (:724-794);
  * register_backend(AttentionBackendEnum.CUSTOM, "<path>") (v1/attention/backends/registry.py
    :242-294), selected with --attention-backend CUSTOM.

The vLLM-side kernel class derives from vLLM's own MPLinearKernel so isinstance checks in the
fork hold; its behaviour is the implementation in mp_kernel.MI355XW4A16LinearKernel.
"""


def hook_into_vllm() -> None:
    from vllm.model_executor.kernels.linear import _POSSIBLE_KERNELS, register_linear_kernel
    from vllm.model_executor.kernels.linear.mixed_precision.MPLinearKernel import MPLinearKernel as VllmMPLinearKernel
    from vllm.platforms.interface import PlatformEnum

    from . import mp_kernel

    impl = mp_kernel.MI355XW4A16LinearKernel
    body = {k: v for k, v in vars(impl).items() if not k.startswith("__") or k == "__init__"}
    body["__doc__"] = impl.__doc__

    def can_implement(cls, c):
        # vLLM's ScalarType objects compare by value with their own kind only: go through the name
        mine = {str(t): t for t in impl.SUPPORTED_QUANT_TYPES}
        wt = mine.get(str(c.weight_type))
        if wt is None:
            return False, f"Quant type {c.weight_type} not supported; supported: {list(mine)}"
        shadow = mp_kernel.MPLinearLayerConfig(c.full_weight_shape, c.partition_weight_shape, wt, c.act_type,
                                               c.group_size, c.zero_points, c.has_g_idx, c.out_type)
        return impl.can_implement.__func__(cls, shadow)

    body["can_implement"] = classmethod(can_implement)
    kernel = type("MI355XW4A16LinearKernel", (VllmMPLinearKernel,), body)
    if not any(k.__name__ == kernel.__name__ for k in _POSSIBLE_KERNELS.get(PlatformEnum.ROCM, [])):
        register_linear_kernel(kernel, PlatformEnum.ROCM, "mp")
        lst = _POSSIBLE_KERNELS[PlatformEnum.ROCM]
        lst.insert(0, lst.pop(lst.index(kernel)))

    try:
        from vllm.v1.attention.backends.registry import AttentionBackendEnum, register_backend
        register_backend(AttentionBackendEnum.CUSTOM, "mi355q.plugin.attention.Mi355AttentionBackend")
    except Exception:  # attention registry layout differs between vLLM versions; linear kernels still work
        pass
