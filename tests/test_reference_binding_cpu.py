"""The plug-in's vLLM-side classes bound into the real fork, in the authoring container.

Imports the reference from /root/reference (absent third-party modules stubbed, as the golden
generator does) and checks what `mi355q.plugin.register()` promises inside a vLLM process:
the mixed-precision kernels win `choose_mp_linear_kernel` and are real `MPLinearKernel`
subclasses that can be instantiated and run their load-time repack, the attention backend
resolves through `AttentionBackendEnum.CUSTOM` and passes the fork's `validate_configuration`,
and the AWQ config override resolves through `get_quantization_config`.

Skipped wherever the reference tree is absent (the GPU box): nothing of it travels.
"""
import importlib
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import pytest
import torch

REF = os.environ.get("MI355Q_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "vllm")), reason="reference tree not present")


def _stubbed(fn):
    """Run fn(), giving every absent third-party module it tries to import a stand-in."""
    while True:
        try:
            return fn()
        except ModuleNotFoundError as e:
            if e.name is None or e.name.startswith(("vllm", "tests", "mi355q")):
                raise
            m = MagicMock(name=e.name)
            m.__path__ = []
            m.__spec__ = None
            sys.modules[e.name] = m


def _imp(name):
    return _stubbed(lambda: importlib.import_module(name))


@pytest.fixture(scope="module")
def vllm_ref():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    ns = types.SimpleNamespace()
    ns.linear = _imp("vllm.model_executor.kernels.linear")
    ns.mpk = _imp("vllm.model_executor.kernels.linear.mixed_precision.MPLinearKernel")
    ns.platform = _imp("vllm.platforms.interface")
    ns.backend = _imp("vllm.v1.attention.backend")
    ns.registry = _imp("vllm.v1.attention.backends.registry")
    ns.quant = _imp("vllm.model_executor.layers.quantization")
    ns.scalar = _imp("vllm.scalar_type")
    # what the hooks import themselves: stub the absent third-party modules they pull in first
    _imp("vllm.model_executor.layers.quantization.auto_awq")
    _imp("vllm.model_executor.layers.linear")
    _imp("mi355q.plugin.vllm_attention")
    from mi355q.plugin import vllm_hooks
    vllm_hooks.hook_into_vllm()
    vllm_hooks.hook_into_vllm()   # every vLLM process calls the plug-in; twice must equal once
    ns.hooks = vllm_hooks
    return ns


class _RocmPlatform:
    """What choose_mp_linear_kernel reads of current_platform (kernels/linear/__init__.py:745-752)."""

    def __init__(self, enum):
        self._enum = enum

    def get_device_capability(self):
        return (9, 5)


def _choose(vllm_ref, monkeypatch, cfg):
    monkeypatch.setattr(vllm_ref.linear, "current_platform", _RocmPlatform(vllm_ref.platform.PlatformEnum.ROCM))
    monkeypatch.setattr(vllm_ref.linear, "_get_linear_backend", lambda: "auto")
    return vllm_ref.linear.choose_mp_linear_kernel(cfg)


def test_registration_is_idempotent_and_first(vllm_ref):
    lst = vllm_ref.linear._POSSIBLE_KERNELS[vllm_ref.platform.PlatformEnum.ROCM]
    names = [k.__name__ for k in lst]
    assert names[:2] == ["MI355XW4A16LinearKernel", "MI355XW8A16LinearKernel"]
    assert names.count("MI355XW4A16LinearKernel") == 1 and names.count("MI355XW8A16LinearKernel") == 1
    assert "TritonW4A16LinearKernel" in names   # the fork's own kernels stay behind ours
    for k in lst[:2]:
        assert issubclass(k, vllm_ref.mpk.MPLinearKernel)


@pytest.mark.parametrize("wtype,zp,expect", [("uint4b8", False, "MI355XW4A16LinearKernel"),
                                             ("uint4", True, "MI355XW4A16LinearKernel"),
                                             ("uint8b128", False, "MI355XW8A16LinearKernel")])
def test_choose_mp_linear_kernel_picks_the_mi355x_class(vllm_ref, monkeypatch, wtype, zp, expect):
    st = vllm_ref.scalar.scalar_types
    cfg = vllm_ref.mpk.MPLinearLayerConfig(full_weight_shape=(4096, 6144), partition_weight_shape=(4096, 6144),
                                           weight_type=getattr(st, wtype), act_type=torch.bfloat16, group_size=128,
                                           zero_points=zp, has_g_idx=False)
    assert _choose(vllm_ref, monkeypatch, cfg).__name__ == expect


def test_refusals_fall_through_to_the_forks_kernels(vllm_ref, monkeypatch):
    st = vllm_ref.scalar.scalar_types
    cfg = vllm_ref.mpk.MPLinearLayerConfig((4096, 6144), (4096, 6144), st.uint4b8, torch.float32, 128, False, False)
    k = vllm_ref.linear._POSSIBLE_KERNELS[vllm_ref.platform.PlatformEnum.ROCM][0]
    ok, why = k.can_implement(cfg)
    assert not ok and "float16/bfloat16" in why
    with pytest.raises(ValueError, match="Failed to find a kernel"):
        monkeypatch.setenv("VLLM_DISABLED_KERNELS", "")
        bad = vllm_ref.mpk.MPLinearLayerConfig((4096, 6144), (4096, 6144), st.uint4b8, torch.float32, 100, False, False)
        _choose(vllm_ref, monkeypatch, bad)


def test_kernel_instance_runs_the_load_time_repack(vllm_ref, monkeypatch):
    """GPTQ checkpoint tensors on a plain module -> the chosen kernel class -> kernel-layout parameters;
    the packed nibbles unpack to the checkpoint's values (host arithmetic of the same C entry point)."""
    import oracle as orc
    from mi355q import _lib, ops
    st = vllm_ref.scalar.scalar_types
    K, N, G = 256, 128, 128
    cfg = vllm_ref.mpk.MPLinearLayerConfig((K, N), (K, N), st.uint4b8, torch.bfloat16, G, False, False)
    kernel_cls = _choose(vllm_ref, monkeypatch, cfg)
    kernel = kernel_cls(cfg, w_q_param_name="qweight", w_s_param_name="scales", w_zp_param_name="qzeros",
                        w_gidx_param_name="g_idx")
    assert isinstance(kernel, vllm_ref.mpk.MPLinearKernel)
    rng = np.random.default_rng(0)
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    layer = torch.nn.Module()
    layer.register_parameter("qweight", torch.nn.Parameter(torch.from_numpy(orc.gptq_pack(q)), requires_grad=False))
    scales = torch.from_numpy(rng.random((K // G, N), dtype=np.float32)).to(torch.bfloat16)
    layer.register_parameter("scales", torch.nn.Parameter(scales, requires_grad=False))
    layer.register_parameter("qzeros", torch.nn.Parameter(torch.zeros((K // G, N // 8), dtype=torch.int32), requires_grad=False))
    layer.register_parameter("g_idx", torch.nn.Parameter(torch.arange(K, dtype=torch.int32) // G, requires_grad=False))
    kernel.process_weights_after_loading(layer)
    w = layer.mi355q_weight
    assert (w.K, w.N, w.G) == (K, N, G) and layer.qzeros is None and layer.g_idx is None
    back = np.empty((K, N), dtype=np.int32)
    _lib.check(_lib.lib().mi355q_w4a16_unpack_host(w.packed.numpy().ctypes.data, back.ctypes.data, K, N, w.tile))
    assert np.array_equal(back, q)
    assert torch.equal(ops.scales_row_major(w), scales)
    with pytest.raises(RuntimeError, match="device tensors"):   # no CPU compute path exists behind apply_weights
        kernel.apply_weights(layer, torch.zeros(1, K, dtype=torch.bfloat16))


def _validate(cls, vllm_ref, **over):
    kw = dict(head_size=128, dtype=torch.bfloat16, kv_cache_dtype="fp8", block_size=16, use_mla=False, has_sink=False,
              use_sparse=False, use_mm_prefix=False, use_per_head_quant_scales=False, device_capability=None,
              attn_type=vllm_ref.backend.AttentionType.DECODER)
    kw.update(over)
    return cls.validate_configuration(**kw)


def test_attention_backend_resolves_and_validates(vllm_ref):
    enum = vllm_ref.registry.AttentionBackendEnum
    cls = enum.CUSTOM.get_class()
    assert issubclass(cls, vllm_ref.backend.AttentionBackend) and cls.get_name() == "CUSTOM"
    assert enum[cls.get_name()] is enum.CUSTOM          # attention.py:423 looks the name up in the enum
    assert _validate(cls, vllm_ref) == []
    assert _validate(cls, vllm_ref, kv_cache_dtype="auto", block_size=32, head_size=64, dtype=torch.float16) == []
    assert _validate(cls, vllm_ref, head_size=80) == ["head_size not supported"]
    assert _validate(cls, vllm_ref, kv_cache_dtype="fp8_e5m2") == ["kv_cache_dtype not supported"]
    assert "sliding window not supported" in _validate(cls, vllm_ref, has_sliding_window=True)
    assert "attention sinks not supported" in _validate(cls, vllm_ref, has_sink=True)
    assert "MLA not supported" in _validate(cls, vllm_ref, use_mla=True)
    assert cls.get_kv_cache_shape(10, 16, 8, 128) == (2, 10, 16, 8, 128)
    assert cls.get_kv_cache_block_dim(16, 8, 128) == 1
    assert issubclass(cls.get_impl_cls(), vllm_ref.backend.AttentionImpl)
    assert issubclass(cls.get_builder_cls(), vllm_ref.backend.AttentionMetadataBuilder)
    assert cls.get_builder_cls().get_cudagraph_support(None, None) == vllm_ref.backend.AttentionCGSupport.ALWAYS


def test_attention_impl_and_builder_take_the_forks_arguments(vllm_ref):
    cls = vllm_ref.registry.AttentionBackendEnum.CUSTOM.get_class()
    # positional arguments exactly as attention.py:409-422 passes them
    impl = cls.get_impl_cls()(32, 128, 0.0883, 8, None, None, "fp8", None, vllm_ref.backend.AttentionType.DECODER, None)
    assert (impl.num_heads, impl.num_kv_heads, impl.head_size, impl.kv_cache_dtype) == (32, 8, 128, "fp8")
    assert impl.fused_rope_kvcache_supported() and impl.dcp_world_size == 1
    with pytest.raises(NotImplementedError):
        cls.get_impl_cls()(32, 128, 0.0883, 8, None, 4096, "auto", None, vllm_ref.backend.AttentionType.DECODER, None)
    spec = types.SimpleNamespace(block_size=16)
    builder = cls.get_builder_cls()(spec, ["model.layers.0.self_attn.attn"], types.SimpleNamespace(), torch.device("cpu"))
    assert builder.block_size == 16 and builder.layer_names == ["model.layers.0.self_attn.attn"]
    common = types.SimpleNamespace(num_actual_tokens=3, max_query_len=1, max_seq_len=40,
                                   query_start_loc=torch.arange(4, dtype=torch.int32),
                                   seq_lens=torch.tensor([40, 7, 9], dtype=torch.int32),
                                   block_table_tensor=torch.zeros((3, 4), dtype=torch.int32),
                                   slot_mapping=torch.zeros(3, dtype=torch.int64), causal=True)
    md = builder.build(common_prefix_len=0, common_attn_metadata=common, fast_build=False)
    assert (md.num_actual_tokens, md.max_seq_len, md.use_cascade) == (3, 40, False)
    assert md.block_table is common.block_table_tensor
    md2 = builder.build_for_cudagraph_capture(common)
    assert md2.seq_lens.tolist() == [1, 1, 1]
    with pytest.raises(NotImplementedError):
        builder.build(common_prefix_len=8, common_attn_metadata=common)


def test_rocm_attn_override_is_opt_in(vllm_ref, monkeypatch):
    enum = vllm_ref.registry.AttentionBackendEnum
    assert not enum.ROCM_ATTN.is_overridden()
    monkeypatch.setenv("MI355Q_OVERRIDE_ROCM_ATTN", "1")
    vllm_ref.hooks.hook_into_vllm()
    try:
        cls = enum.ROCM_ATTN.get_class()
        assert cls.get_name() == "ROCM_ATTN" and issubclass(cls, enum.CUSTOM.get_class())
    finally:
        enum.ROCM_ATTN.clear_override()


def test_awq_config_override_resolves_and_repacks(vllm_ref, monkeypatch):
    import oracle as orc
    from mi355q import _lib
    param_mod = _imp("vllm.model_executor.parameter")   # its loaders ask for the TP rank: single rank here
    monkeypatch.setattr(param_mod, "get_tensor_model_parallel_rank", lambda: 0)
    monkeypatch.setattr(param_mod, "get_tensor_model_parallel_world_size", lambda: 1)
    ref_awq = _imp("vllm.model_executor.layers.quantization.auto_awq")
    linear_mod = _imp("vllm.model_executor.layers.linear")
    for name in ("auto_awq", "awq", "awq_marlin"):
        cls = _stubbed(lambda: vllm_ref.quant.get_quantization_config(name))
        assert issubclass(cls, ref_awq.AutoAWQConfig) and cls is not ref_awq.AutoAWQConfig
    cfg = cls.from_config({"quant_method": "awq", "bits": 4, "group_size": 128, "zero_point": True, "version": "gemm"})
    layer_probe = MagicMock(spec=linear_mod.LinearBase)
    method = _stubbed(lambda: cfg.get_quant_method(layer_probe, "model.layers.0.self_attn.qkv_proj"))
    assert isinstance(method, ref_awq.AutoAWQLinearMethod) and type(method) is not ref_awq.AutoAWQLinearMethod
    assert type(method).__name__ in linear_mod.WEIGHT_LOADER_V2_SUPPORTED
    # checkpoint tensors created by the fork's own create_weights, then our load-time repack (host arithmetic)
    K, N, G = 256, 128, 128
    rng = np.random.default_rng(1)
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    z = rng.integers(0, 16, size=(K // G, N), dtype=np.int32)
    layer = torch.nn.Module()
    method.create_weights(layer, K, [N], K, N, torch.float16, weight_loader=lambda p, w: p.data.copy_(w))
    assert tuple(layer.qweight.shape) == (K, N // 8) and tuple(layer.qzeros.shape) == (K // G, N // 8)
    layer.qweight.data.copy_(torch.from_numpy(orc.pack_cols(q, awq_order=True)))
    layer.qzeros.data.copy_(torch.from_numpy(orc.pack_cols(z, awq_order=True)))
    layer.scales.data.copy_(torch.from_numpy(rng.random((K // G, N), dtype=np.float32)).to(torch.float16))
    method.process_weights_after_loading(layer)
    w = layer.mi355q_weight
    back = np.empty((K, N), dtype=np.int32)
    _lib.check(_lib.lib().mi355q_w4a16_unpack_host(w.packed.numpy().ctypes.data, back.ctypes.data, K, N, w.tile))
    assert np.array_equal(back, q)
    from mi355q import ops
    zc = ops.scales_row_major(w, zc=True).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(zc, (0x6400 + z).astype(np.uint16))      # fp16(1024 + z)


def test_fp8_scaled_mm_kernel_is_chosen_by_the_forks_selector(vllm_ref, monkeypatch):
    """init_fp8_linear_kernel (kernels/linear/__init__.py:615-685) on a ROCm platform returns the MI355X class
    for per-tensor and dynamic per-token activations, leaves per-group (block) activations to the fork's
    block kernels, and the instance quantises through this package's op."""
    qu = _imp("vllm.model_executor.layers.quantization.utils.quant_utils")
    smm = _imp("vllm.model_executor.kernels.linear.scaled_mm.ScaledMMLinearKernel")
    lst = vllm_ref.linear._POSSIBLE_FP8_KERNELS[vllm_ref.platform.PlatformEnum.ROCM]
    names = [k.__name__ for k in lst]
    assert names[0] == "MI355XFP8ScaledMMLinearKernel" and names.count("MI355XFP8ScaledMMLinearKernel") == 1
    assert "ROCmFP8ScaledMMLinearKernel" in names and issubclass(lst[0], smm.FP8ScaledMMLinearKernel)
    monkeypatch.setattr(vllm_ref.linear, "current_platform", _RocmPlatform(vllm_ref.platform.PlatformEnum.ROCM))
    monkeypatch.setattr(vllm_ref.linear, "_get_linear_backend", lambda: "auto", raising=False)
    for act in (qu.kFp8StaticTensorSym, qu.kFp8DynamicTensorSym, qu.kFp8DynamicTokenSym):
        k = vllm_ref.linear.init_fp8_linear_kernel(activation_quant_key=act, weight_quant_key=qu.kFp8StaticTensorSym,
                                                   input_dtype=torch.bfloat16, out_dtype=torch.bfloat16,
                                                   weight_shape=(6144, 4096))
        assert type(k).__name__ == "MI355XFP8ScaledMMLinearKernel"
        assert k.quant_fp8.static == act.scale.static and k.quant_fp8.per_token == act.scale.group_shape.is_per_token()
    cfg = smm.FP8ScaledMMLinearLayerConfig(weight_quant_key=qu.kFp8StaticTensorSym, activation_quant_key=qu.kFp8DynamicTokenSym,
                                           weight_shape=(6144, 4100), input_dtype=torch.bfloat16, out_dtype=torch.bfloat16)
    ok, why = lst[0].can_implement(cfg)
    assert not ok and "multiple of 16" in why
    with pytest.raises(RuntimeError, match="device tensors"):   # no CPU compute path behind apply_weights
        layer = torch.nn.Module()
        layer.weight = torch.zeros(64, 32).to(torch.float8_e4m3fn).t()
        layer.weight_scale, layer.input_scale = torch.ones(1), None
        k.apply_weights(layer, torch.zeros(2, 64, dtype=torch.bfloat16))
