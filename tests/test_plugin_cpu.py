"""Host logic of the plug-in surface (no GPU): registries, checkpoint-config parsing, weight
creation shapes, tensor-parallel shard loading and the load-time repack.  Patterned on the
reference's tests/quantization/test_register_quantization_config.py, test_auto_gptq.py and
tests/kernels/quantization/test_mixed_precision... selection tests."""
import numpy as np
import pytest
import torch

import oracle as orc
from mi355q import _lib
from mi355q.plugin import (AutoAWQConfig, AutoGPTQConfig, MI355XW4A16LinearKernel, MPLinearLayerConfig,
                           choose_mp_linear_kernel, get_quantization_config, register_quantization_config,
                           resolve_quantization)
from mi355q.plugin import parallel
from mi355q.plugin.gptq import get_dynamic_override
from mi355q.plugin.linear import (ColumnParallelLinear, MergedColumnParallelLinear, QKVParallelLinear,
                                  RowParallelLinear, UnquantizedLinearMethod)
from mi355q.plugin.registry import QuantizationConfig
from mi355q.plugin.scalar_type import scalar_types

GPTQ_CFG = {"quant_method": "gptq", "bits": 4, "group_size": 128, "desc_act": False, "sym": True}
AWQ_CFG = {"quant_method": "awq", "w_bit": 4, "q_group_size": 128, "zero_point": True, "version": "gemm"}


@pytest.fixture
def tp():
    """Pretend to be rank r of n without a process group (the loaders only read rank and size)."""
    def _set(rank, size):
        parallel._STATE.update(rank=rank, size=size, group=None)
    yield _set
    parallel._STATE.update(rank=0, size=1, group=None)


# ---------------------------------------------------------------------------
# registries and config parsing
# ---------------------------------------------------------------------------
def test_scalar_types():
    t = scalar_types.uint4b8
    assert (t.size_bits, t.bias, t.min(), t.max(), str(t)) == (4, 8, -8, 7, "uint4b8")
    assert scalar_types.uint4.max() == 15 and not scalar_types.uint4.has_bias()
    assert scalar_types.uint8b128.size_bits == 8 and scalar_types.uint4b8.id != scalar_types.uint4.id


def test_registry_lookup_and_override():
    assert get_quantization_config("auto_gptq") is AutoGPTQConfig
    assert get_quantization_config("auto_awq") is AutoAWQConfig
    with pytest.raises(ValueError, match="Invalid quantization method"):
        get_quantization_config("no_such_method")
    with pytest.raises(ValueError, match="subclass"):
        register_quantization_config("bad")(dict)

    @register_quantization_config("custom_quant")
    class Custom(AutoGPTQConfig):
        pass

    assert get_quantization_config("custom_quant") is Custom
    # re-registering a name swaps the class in
    register_quantization_config("custom_quant")(AutoGPTQConfig)
    assert get_quantization_config("custom_quant") is AutoGPTQConfig


def test_resolve_quantization_from_checkpoint_config():
    assert resolve_quantization(GPTQ_CFG) == "auto_gptq"
    assert resolve_quantization(GPTQ_CFG, "gptq_marlin") == "auto_gptq"
    assert resolve_quantization(AWQ_CFG) == "auto_awq"
    with pytest.raises(ValueError, match="Unknown quantization"):
        resolve_quantization({"quant_method": "nope"})


def test_gptq_config_parsing():
    c = AutoGPTQConfig.from_config(dict(GPTQ_CFG, lm_head=True))
    assert (c.weight_bits, c.group_size, c.desc_act, c.is_sym, c.pack_factor) == (4, 128, False, True, 8)
    assert c.quant_type == scalar_types.uint4b8 and c.lm_head_quantized
    assert AutoGPTQConfig.from_config(dict(GPTQ_CFG, bits=8)).quant_type == scalar_types.uint8b128
    # one group per channel makes act-order meaningless
    assert AutoGPTQConfig.from_config(dict(GPTQ_CFG, group_size=-1, desc_act=True)).desc_act is False
    with pytest.raises(ValueError, match="Unsupported quantization config"):
        AutoGPTQConfig.from_config(dict(GPTQ_CFG, sym=False))
    with pytest.raises(ValueError, match="Cannot find any of"):
        AutoGPTQConfig.from_config({"bits": 4})
    assert AutoGPTQConfig.get_config_filenames() == ["quantize_config.json"]


def test_gptq_dynamic_rules_and_module_list():
    dyn = {r"+:.*\.layers\.1\..*": {"group_size": 64}, r"-:.*\.layers\.2\..*": {}}
    c = AutoGPTQConfig.from_config(dict(GPTQ_CFG, dynamic=dyn))
    assert get_dynamic_override(c, "model.layers.1.mlp.down_proj", "group_size", 128) == 64
    assert get_dynamic_override(c, "model.layers.2.mlp.down_proj") is False
    assert get_dynamic_override(c, "model.layers.0.mlp.down_proj", "group_size", 128) == 128
    l1 = RowParallelLinear(512, 256, bias=False, params_dtype=torch.bfloat16, quant_config=c,
                           prefix="model.layers.1.mlp.down_proj")
    assert l1.quant_method.quant_config.group_size == 64 and l1.scales.shape == (8, 256)
    assert c.group_size == 128   # the override lives on a per-layer copy
    l2 = RowParallelLinear(512, 256, bias=False, params_dtype=torch.bfloat16, quant_config=c,
                           prefix="model.layers.2.mlp.down_proj")
    assert isinstance(l2.quant_method, UnquantizedLinearMethod) and l2.weight.shape == (256, 512)
    # explicit module list: anything not named stays 16-bit; fused modules are judged by their shards
    c2 = AutoGPTQConfig.from_config(dict(GPTQ_CFG, modules_in_block_to_quantize=[["self_attn.q_proj", "self_attn.k_proj",
                                                                                  "self_attn.v_proj"]]))
    c2.packed_modules_mapping = {"qkv_proj": ["q_proj", "k_proj", "v_proj"]}
    qkv = QKVParallelLinear(256, 64, 4, 2, bias=False, params_dtype=torch.bfloat16, quant_config=c2,
                            prefix="model.layers.0.self_attn.qkv_proj")
    assert hasattr(qkv, "qweight")
    mlp = RowParallelLinear(512, 256, bias=False, params_dtype=torch.bfloat16, quant_config=c2,
                            prefix="model.layers.0.mlp.down_proj")
    assert isinstance(mlp.quant_method, UnquantizedLinearMethod)
    c3 = AutoGPTQConfig.from_config(GPTQ_CFG)
    c3.maybe_update_config("m", safetensors_metadata={"model.layers.0.mlp.down_proj.qweight": {"dtype": "I32"},
                                                      "model.layers.0.mlp.down_proj.scales": {"dtype": "F16"},
                                                      "lm_head.weight": {"dtype": "BF16"}})
    assert c3.modules_in_block_to_quantize == ["model.layers.0.mlp.down_proj"]


def test_awq_config_parsing_and_skip_list():
    c = AutoAWQConfig.from_config(dict(AWQ_CFG, modules_to_not_convert=["mlp.gate"]))
    assert (c.weight_bits, c.group_size, c.zero_point, c.pack_factor) == (4, 128, True, 8)
    assert c.quant_type == scalar_types.uint4 and c.full_config["quant_method"] == "awq"
    with pytest.raises(ValueError, match="Unsupported num_bits"):
        AutoAWQConfig.from_config(dict(AWQ_CFG, w_bit=8))
    skipped = ColumnParallelLinear(256, 128, bias=False, params_dtype=torch.float16, quant_config=c,
                                   prefix="model.layers.0.mlp.gate")
    assert isinstance(skipped.quant_method, UnquantizedLinearMethod)
    lin = ColumnParallelLinear(256, 128, bias=False, params_dtype=torch.float16, quant_config=c,
                               prefix="model.layers.0.mlp.up_proj")
    assert lin.qweight.shape == (256, 16) and lin.qzeros.shape == (2, 16) and lin.scales.shape == (2, 128)


def test_cache_scale_name_mapping():
    m = QuantizationConfig.map_cache_scale_name
    assert m("model.layers.0.self_attn.kv_scale") == "model.layers.0.self_attn.attn.k_scale"
    assert m("model.layers.0.self_attn.k_proj.k_scale") == "model.layers.0.self_attn.attn.k_scale"
    assert m("model.layers.0.self_attn.v_scale") == "model.layers.0.self_attn.attn.v_scale"
    assert m("model.layers.0.self_attn.attn.v_scale") == "model.layers.0.self_attn.attn.v_scale"
    assert m("model.layers.0.mlp.down_proj.weight") == "model.layers.0.mlp.down_proj.weight"


# ---------------------------------------------------------------------------
# kernel selection
# ---------------------------------------------------------------------------
def _kcfg(**kw):
    base = dict(full_weight_shape=(4096, 6144), partition_weight_shape=(4096, 6144), weight_type=scalar_types.uint4b8,
                act_type=torch.bfloat16, group_size=128, zero_points=False, has_g_idx=False)
    base.update(kw)
    return MPLinearLayerConfig(**base)


def test_kernel_selection_and_reasons(monkeypatch):
    assert choose_mp_linear_kernel(_kcfg()) is MI355XW4A16LinearKernel
    assert choose_mp_linear_kernel(_kcfg(weight_type=scalar_types.uint4, zero_points=True)) is MI355XW4A16LinearKernel
    # 8-bit symmetric GPTQ weights have a kernel of their own (no reordering, no zero points)
    from mi355q.plugin.mp_kernel import MI355XW8A16LinearKernel
    assert choose_mp_linear_kernel(_kcfg(weight_type=scalar_types.uint8b128)) is MI355XW8A16LinearKernel
    ok, reason = MI355XW4A16LinearKernel.can_implement(_kcfg(weight_type=scalar_types.uint8b128))
    assert not ok and "not supported" in reason
    with pytest.raises(ValueError, match="Failed to find a kernel"):
        choose_mp_linear_kernel(_kcfg(weight_type=scalar_types.uint8b128, has_g_idx=True))
    # activation reordering is served with the whole of K on the rank and, for symmetric weights, with K sharded
    assert choose_mp_linear_kernel(_kcfg(has_g_idx=True)) is MI355XW4A16LinearKernel
    assert choose_mp_linear_kernel(_kcfg(has_g_idx=True, partition_weight_shape=(2048, 6144))) is MI355XW4A16LinearKernel
    for bad, why in ((dict(has_g_idx=True, partition_weight_shape=(2048, 6144), weight_type=scalar_types.uint4,
                           zero_points=True), "only supported for symmetric"),
                     (dict(act_type=torch.float32), "float16/bfloat16"),
                     (dict(partition_weight_shape=(4096, 100)), "multiple of 64"),
                     (dict(group_size=100), "not divisible"), (dict(weight_type=scalar_types.uint4), "zero points")):
        ok, reason = MI355XW4A16LinearKernel.can_implement(_kcfg(**bad))
        assert not ok and why in reason
        with pytest.raises(ValueError, match="Failed to find a kernel"):
            choose_mp_linear_kernel(_kcfg(**bad))
    with pytest.raises(ValueError, match="requires capability 95"):
        choose_mp_linear_kernel(_kcfg(), compute_capability=94)
    monkeypatch.setenv("VLLM_DISABLED_KERNELS", "MI355XW4A16LinearKernel")
    with pytest.raises(ValueError, match="disabled by environment variable"):
        choose_mp_linear_kernel(_kcfg())


def test_gptq_create_weights_shapes_llama8b():
    c = AutoGPTQConfig.from_config(GPTQ_CFG)
    qkv = QKVParallelLinear(4096, 128, 32, 8, bias=False, params_dtype=torch.bfloat16, quant_config=c, prefix="qkv_proj")
    assert qkv.qweight.shape == (512, 6144) and qkv.qweight.dtype == torch.int32
    assert qkv.scales.shape == (32, 6144) and qkv.qzeros.shape == (32, 768) and qkv.g_idx.shape == (4096,)
    gu = MergedColumnParallelLinear(4096, [14336, 14336], bias=False, params_dtype=torch.bfloat16, quant_config=c,
                                    prefix="gate_up_proj")
    assert gu.qweight.shape == (512, 28672)
    down = RowParallelLinear(14336, 4096, bias=False, params_dtype=torch.bfloat16, quant_config=c, prefix="down_proj")
    assert down.qweight.shape == (1792, 4096) and down.scales.shape == (112, 4096)


# ---------------------------------------------------------------------------
# tensor-parallel shard loading + load-time repack
# ---------------------------------------------------------------------------
def _gptq_tensors(rng, K, N, G):
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    s = orc.f32_to_h16((0.05 * rng.random((K // G, N)) + 0.01).astype(np.float32))
    z = rng.integers(0, 16, size=(K // G, N), dtype=np.int32)
    return q, s, z


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _bf(bits):
    return _t(bits.view(np.int16)).view(torch.bfloat16)


def _unpacked(layer):
    w = layer.mi355q_weight
    back = np.empty((w.K, w.N), dtype=np.int32)
    _lib.check(_lib.lib().mi355q_w4a16_unpack_host(w.packed.data_ptr(), back.ctypes.data, w.K, w.N, w.tile))
    return back


@pytest.mark.parametrize("tp_size", [1, 2, 4])
def test_gptq_tp_sharding_of_every_layer_kind(tp, tp_size):
    """Each rank must end up with exactly the rows/columns of the full matrices that the
    reference's loaders give it (parameter.py:148-230), including the replicated KV heads when
    there are fewer KV heads than ranks, and the packed tensors must shard on packed units."""
    rng = np.random.default_rng(5)
    hidden, head, H, KVH, inter, G = 256, 64, 8, 2, 512, 64
    c = AutoGPTQConfig.from_config(dict(GPTQ_CFG, group_size=G))
    full = {n: _gptq_tensors(rng, K, N, G) for n, (K, N) in
            {"q": (hidden, H * head), "k": (hidden, KVH * head), "v": (hidden, KVH * head),
             "gate": (hidden, inter), "up": (hidden, inter), "down": (inter, hidden), "o": (H * head, hidden)}.items()}

    def load(layer, name, shard_id=None):
        q, s, z = full[name]
        args = () if shard_id is None else (shard_id,)
        layer.weight_loader(layer.qweight, _t(orc.gptq_pack(q)), *args)
        layer.weight_loader(layer.scales, _bf(s), *args)
        layer.weight_loader(layer.qzeros, _t(orc.pack_cols(z)), *args)

    for rank in range(tp_size):
        tp(rank, tp_size)
        qkv = QKVParallelLinear(hidden, head, H, KVH, bias=False, params_dtype=torch.bfloat16, quant_config=c, prefix="qkv")
        for sid in "qkv":
            load(qkv, sid, sid)
        gu = MergedColumnParallelLinear(hidden, [inter, inter], bias=False, params_dtype=torch.bfloat16, quant_config=c,
                                        prefix="gate_up")
        load(gu, "gate", 0)
        load(gu, "up", 1)
        down = RowParallelLinear(inter, hidden, bias=False, params_dtype=torch.bfloat16, quant_config=c, prefix="down")
        load(down, "down")
        o = RowParallelLinear(H * head, hidden, bias=False, params_dtype=torch.bfloat16, quant_config=c, prefix="o")
        load(o, "o")

        hq = H // tp_size * head
        kv_rank = rank // max(tp_size // KVH, 1)
        hkv = max(KVH // tp_size, 1) * head
        exp_qkv = np.concatenate([full["q"][0][:, rank * hq:(rank + 1) * hq],
                                  full["k"][0][:, kv_rank * hkv:(kv_rank + 1) * hkv],
                                  full["v"][0][:, kv_rank * hkv:(kv_rank + 1) * hkv]], axis=1)
        exp_qkv_z = np.concatenate([full["q"][2][:, rank * hq:(rank + 1) * hq],
                                    full["k"][2][:, kv_rank * hkv:(kv_rank + 1) * hkv],
                                    full["v"][2][:, kv_rank * hkv:(kv_rank + 1) * hkv]], axis=1)
        assert np.array_equal(orc.unpack_rows(qkv.qweight.data.numpy()), exp_qkv)
        assert np.array_equal(orc.unpack_cols(qkv.qzeros.data.numpy()), exp_qkv_z)
        ni = inter // tp_size
        exp_gu = np.concatenate([full["gate"][0][:, rank * ni:(rank + 1) * ni], full["up"][0][:, rank * ni:(rank + 1) * ni]], axis=1)
        exp_gu_s = np.concatenate([full["gate"][1][:, rank * ni:(rank + 1) * ni], full["up"][1][:, rank * ni:(rank + 1) * ni]], axis=1)
        assert np.array_equal(orc.unpack_rows(gu.qweight.data.numpy()), exp_gu)
        assert np.array_equal(gu.scales.data.view(torch.int16).numpy().view(np.uint16), exp_gu_s)
        assert np.array_equal(orc.unpack_rows(down.qweight.data.numpy()), full["down"][0][rank * ni:(rank + 1) * ni])
        gpr = ni // G
        assert np.array_equal(down.scales.data.view(torch.int16).numpy().view(np.uint16),
                              full["down"][1][rank * gpr:(rank + 1) * gpr])
        # load-time repack (host arithmetic here) keeps exactly those values
        for layer, exp in ((qkv, exp_qkv), (gu, exp_gu), (down, full["down"][0][rank * ni:(rank + 1) * ni])):
            layer.process_weights_after_loading()
            assert np.array_equal(_unpacked(layer), exp)
            assert layer.qweight.dtype == torch.int32 and layer.qweight.dim() == 1
            assert type(layer.qweight) is torch.nn.Parameter and layer.qzeros is None and layer.w_zc is None


def test_fused_on_disk_tensors_are_split(tp):
    """A checkpoint that already stores qkv / gate_up fused (no shard id) is cut into its
    logical matrices before sharding (linear.py:1129-1176)."""
    rng = np.random.default_rng(6)
    hidden, head, H, KVH, G = 128, 64, 4, 2, 64
    c = AutoGPTQConfig.from_config(dict(GPTQ_CFG, group_size=G))
    N = (H + 2 * KVH) * head
    q, s, z = _gptq_tensors(rng, hidden, N, G)
    for rank in range(2):
        tp(rank, 2)
        qkv = QKVParallelLinear(hidden, head, H, KVH, bias=False, params_dtype=torch.bfloat16, quant_config=c, prefix="qkv")
        qkv.weight_loader(qkv.qweight, _t(orc.gptq_pack(q)))
        qkv.weight_loader(qkv.qzeros, _t(orc.pack_cols(z)))
        hq, hkv = H // 2 * head, KVH // 2 * head
        cols = np.r_[rank * hq:(rank + 1) * hq, H * head + rank * hkv:H * head + (rank + 1) * hkv,
                     (H + KVH) * head + rank * hkv:(H + KVH) * head + (rank + 1) * hkv]
        assert np.array_equal(orc.unpack_rows(qkv.qweight.data.numpy()), q[:, cols])
        assert np.array_equal(orc.unpack_cols(qkv.qzeros.data.numpy()), z[:, cols])


def test_awq_tp_sharding_and_repack(tp):
    rng = np.random.default_rng(8)
    K, N, G = 256, 256, 64
    c = AutoAWQConfig.from_config(dict(AWQ_CFG, q_group_size=G))
    q, s, z = _gptq_tensors(rng, K, N, G)
    qw, qz = orc.pack_cols(q, awq_order=True), orc.pack_cols(z, awq_order=True)
    for rank in range(2):
        tp(rank, 2)
        col = ColumnParallelLinear(K, N, bias=False, params_dtype=torch.float16, quant_config=c, prefix="up_proj")
        row = RowParallelLinear(K, N, bias=False, params_dtype=torch.float16, quant_config=c, prefix="down_proj")
        for layer in (col, row):
            layer.weight_loader(layer.qweight, _t(qw))
            layer.weight_loader(layer.qzeros, _t(qz))
            layer.weight_loader(layer.scales, _t(s.view(np.int16)).view(torch.float16))
            layer.process_weights_after_loading()
        h = N // 2
        assert np.array_equal(_unpacked(col), q[:, rank * h:(rank + 1) * h])
        assert np.array_equal(_unpacked(row), q[rank * (K // 2):(rank + 1) * (K // 2)])
        # zc = 1024 + z in fp16 (the magic-number bias of the fp16 kernel plus the zero point)
        from mi355q import ops
        zc_rm = ops.scales_row_major(col.mi355q_weight, zc=True).contiguous()
        zc = orc.h16_to_f32(zc_rm.view(torch.int16).numpy().view(np.uint16), orc.FP16)
        assert np.array_equal(zc, 1024.0 + z[:, rank * h:(rank + 1) * h])


def test_row_parallel_rejects_misaligned_groups(tp):
    tp(0, 4)
    c = AutoAWQConfig.from_config(AWQ_CFG)
    with pytest.raises(ValueError, match="too large tensor parallel size"):
        RowParallelLinear(256, 128, bias=False, params_dtype=torch.float16, quant_config=c, prefix="down_proj")


def test_loader_shape_mismatch_is_an_error():
    c = AutoGPTQConfig.from_config(GPTQ_CFG)
    lin = ColumnParallelLinear(256, 128, bias=False, params_dtype=torch.bfloat16, quant_config=c, prefix="x")
    with pytest.raises(ValueError, match="does not fit"):
        lin.weight_loader(lin.qweight, torch.zeros(16, 128, dtype=torch.int32))


# ---------------------------------------------------------------------------
# fp8 weight-only
# ---------------------------------------------------------------------------
def test_fp8_config_parsing():
    from mi355q.plugin import Fp8Config
    from mi355q.plugin.fp8 import Fp8LinearMethod, Fp8PerTensorOnlineLinearMethod
    assert get_quantization_config("fp8") is Fp8Config
    c = Fp8Config.from_config({"quant_method": "fp8", "activation_scheme": "static", "ignored_layers": ["lm_head"]})
    assert c.is_checkpoint_fp8_serialized and c.activation_scheme == "static" and c.ignored_layers == ["lm_head"]
    c = Fp8Config.from_config({"quant_method": "fp8", "activation_scheme": "dynamic", "weight_block_size": [128, 128],
                               "modules_to_not_convert": ["mlp.gate"]})
    assert c.weight_block_size == [128, 128] and c.ignored_layers == ["mlp.gate"]
    with pytest.raises(ValueError, match="Unsupported activation scheme"):
        Fp8Config(True, "per_token")
    with pytest.raises(ValueError, match="dynamic activation scheme"):
        Fp8Config(True, "static", weight_block_size=[128, 128])
    with pytest.raises(ValueError, match="fp8-serialized"):
        Fp8Config(False, "dynamic", weight_block_size=[128, 128])
    lin = ColumnParallelLinear(256, 128, bias=False, params_dtype=torch.bfloat16, quant_config=c, prefix="mlp.up_proj")
    assert isinstance(lin.quant_method, Fp8LinearMethod) and lin.weight.dtype == torch.float8_e4m3fn
    assert lin.weight.shape == (128, 256) and lin.weight_scale_inv.shape == (1, 2)
    skipped = ColumnParallelLinear(256, 128, bias=False, params_dtype=torch.bfloat16, quant_config=c, prefix="mlp.gate")
    assert isinstance(skipped.quant_method, UnquantizedLinearMethod)
    online = ColumnParallelLinear(256, 128, bias=False, params_dtype=torch.bfloat16,
                                  quant_config=Fp8Config(False, "dynamic"), prefix="mlp.up_proj")
    assert isinstance(online.quant_method, Fp8PerTensorOnlineLinearMethod) and online.weight.dtype == torch.bfloat16
    assert Fp8Config.map_cache_scale_name("model.layers.0.self_attn.k_proj.output_scale") == \
        "model.layers.0.self_attn.attn.k_scale"


def test_fp8_per_tensor_scales_expand_per_channel_and_shard(tp):
    from mi355q.plugin import Fp8Config
    c = Fp8Config.from_config({"quant_method": "fp8", "activation_scheme": "static"})
    c.linear_mode = "w8a16"
    rng = np.random.default_rng(3)
    hidden, head, H, KVH = 128, 64, 4, 2
    full = {n: torch.from_numpy(rng.integers(0, 120, size=(w, hidden), dtype=np.uint8)).view(torch.float8_e4m3fn)
            for n, w in (("q", H * head), ("k", KVH * head), ("v", KVH * head))}
    scales = {"q": 0.5, "k": 0.25, "v": 2.0}
    for rank in range(2):
        tp(rank, 2)
        qkv = QKVParallelLinear(hidden, head, H, KVH, bias=False, params_dtype=torch.bfloat16, quant_config=c, prefix="qkv")
        for sid in "qkv":
            qkv.weight_loader(qkv.weight, full[sid], sid)
            qkv.weight_loader(qkv.weight_scale, torch.tensor(scales[sid]), sid)
            qkv.weight_loader(qkv.input_scale, torch.tensor(1.0), sid)
        qkv.process_weights_after_loading()
        hq, hkv = H // 2 * head, KVH // 2 * head
        exp = torch.cat([full["q"].view(torch.uint8)[rank * hq:(rank + 1) * hq],
                         full["k"].view(torch.uint8)[rank * hkv:(rank + 1) * hkv],
                         full["v"].view(torch.uint8)[rank * hkv:(rank + 1) * hkv]])
        assert torch.equal(qkv.weight.data.view(torch.uint8), exp)
        assert qkv.weight_scale.tolist() == [0.5] * hq + [0.25] * hkv + [2.0] * hkv
        assert qkv.input_scale is None and qkv.mi355q_scale_block is None


def test_fp8_w8a8_requantises_fused_shards_to_the_largest_scale(tp):
    """process_fp8_weight_tensor_strategy (fp8_utils.py:1321-1346 -> w8a8_utils.py:76-107): one weight scale for
    the fused layer, shards with a smaller scale dequantised through fp16 and requantised; the static input
    scale becomes the largest of the shards' (fp8.py:397-399); a fused checkpoint that loads a single scale
    is left alone."""
    from mi355q.plugin import Fp8Config
    tp(0, 1)
    c = Fp8Config.from_config({"quant_method": "fp8", "activation_scheme": "static"})
    assert c.linear_mode == "w8a8"
    rng = np.random.default_rng(5)
    hidden, inter = 64, 48
    gu = MergedColumnParallelLinear(hidden, [inter, inter], bias=False, params_dtype=torch.bfloat16, quant_config=c, prefix="gate_up")
    w = [torch.from_numpy(rng.integers(0, 120, size=(inter, hidden), dtype=np.uint8)).view(torch.float8_e4m3fn) for _ in range(2)]
    for i, (ws, xs) in enumerate([(0.5, 0.125), (0.25, 0.75)]):
        gu.weight_loader(gu.weight, w[i], i)
        gu.weight_loader(gu.weight_scale, torch.tensor(ws), i)
        gu.weight_loader(gu.input_scale, torch.tensor(xs), i)
    gu.process_weights_after_loading()
    assert gu.weight_scale.tolist() == [0.5] and gu.input_scale.tolist() == [0.75]
    assert torch.equal(gu.weight.data[:inter].view(torch.uint8), w[0].view(torch.uint8))      # 0.5 / 0.5: unchanged
    want = ((w[1].to(torch.float16) * 0.25).float() * (1 / 0.5)).to(torch.float8_e4m3fn)
    assert torch.equal(gu.weight.data[inter:].view(torch.uint8), want.view(torch.uint8))
    # one scale in the checkpoint for the whole fused weight: the other slot keeps its sentinel
    gu2 = MergedColumnParallelLinear(hidden, [inter, inter], bias=False, params_dtype=torch.bfloat16,
                                     quant_config=Fp8Config(True, "dynamic"), prefix="gate_up")
    both = torch.cat(w)
    gu2.weight.data.copy_(both)
    gu2.weight_scale.data[0] = 0.125
    gu2.process_weights_after_loading()
    assert gu2.weight_scale.tolist() == [0.125] and torch.equal(gu2.weight.data.view(torch.uint8), both.view(torch.uint8))
    assert gu2.input_scale is None
    with pytest.raises(ValueError, match="Unsupported fp8 linear mode"):
        Fp8Config(True, "dynamic", linear_mode="w4a8")


def test_fp8_block_scales_shard_on_tiles(tp):
    from mi355q.plugin import Fp8Config
    c = Fp8Config.from_config({"quant_method": "fp8", "activation_scheme": "dynamic", "weight_block_size": [64, 32]})
    rng = np.random.default_rng(4)
    hidden, inter = 128, 256
    sg = torch.from_numpy(rng.random((inter // 64, hidden // 32)).astype(np.float32))
    su = torch.from_numpy(rng.random((inter // 64, hidden // 32)).astype(np.float32))
    sd = torch.from_numpy(rng.random((hidden // 64, inter // 32)).astype(np.float32))
    for rank in range(2):
        tp(rank, 2)
        gu = MergedColumnParallelLinear(hidden, [inter, inter], bias=False, params_dtype=torch.bfloat16, quant_config=c,
                                        prefix="gate_up")
        gu.weight_loader(gu.weight_scale_inv, sg, 0)
        gu.weight_loader(gu.weight_scale_inv, su, 1)
        assert gu.weight_scale_inv.shape == (4, 4)
        assert torch.equal(gu.weight_scale_inv.data, torch.cat([sg[rank * 2:(rank + 1) * 2], su[rank * 2:(rank + 1) * 2]]))
        down = RowParallelLinear(inter, hidden, bias=False, params_dtype=torch.bfloat16, quant_config=c, prefix="down")
        down.weight_loader(down.weight_scale_inv, sd)
        assert torch.equal(down.weight_scale_inv.data, sd[:, rank * 4:(rank + 1) * 4])
    tp(0, 4)
    with pytest.raises(ValueError, match="not divisible by weight quantization block_k"):
        RowParallelLinear(inter, hidden, bias=False, params_dtype=torch.bfloat16,
                          quant_config=Fp8Config(True, "dynamic", weight_block_size=[64, 128]), prefix="down")


# ---------------------------------------------------------------------------
# mixture of experts
# ---------------------------------------------------------------------------
def test_moe_config_and_loader_conversions(tp):
    import moe_ref
    from mi355q.plugin import FusedMoE, MoeWNA16Config
    from mi355q.plugin.moe import convert_awq_tensor, convert_gptq_int4_qzeros
    c = MoeWNA16Config.from_config(dict(AWQ_CFG, bits=4, group_size=64, quant_method="awq"))
    assert c.has_zp and c.bit8_pack_factor == 2
    assert MoeWNA16Config.override_quantization_method(dict(GPTQ_CFG), "moe_wna16") == "moe_wna16"
    assert MoeWNA16Config.override_quantization_method(dict(GPTQ_CFG), None) is None
    assert not MoeWNA16Config.is_moe_wna16_compatible(dict(GPTQ_CFG, desc_act=True))
    with pytest.raises(ValueError, match="only support gptq and awq"):
        MoeWNA16Config.from_config({"quant_method": "fp8", "bits": 4, "group_size": 64})
    # AWQ tensors -> byte layout: value-exact against the oracle's unpackers
    rng = np.random.default_rng(11)
    K, N, G = 64, 32, 32
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    z = rng.integers(0, 16, size=(K // G, N), dtype=np.int32)
    qw = convert_awq_tensor(_t(orc.pack_cols(q, awq_order=True)), "qweight").numpy()
    qz = convert_awq_tensor(_t(orc.pack_cols(z, awq_order=True)), "qzeros").numpy()
    assert qw.shape == (N, K // 2) and np.array_equal(moe_ref.unpack_moe_weight(qw), q)
    assert qz.shape == (N // 2, K // G) and np.array_equal(moe_ref.unpack_moe_zeros(qz), z)
    zg = rng.integers(0, 15, size=(K // G, N), dtype=np.int32)   # stored z - 1 < 15
    got = convert_gptq_int4_qzeros(_t(orc.pack_cols(zg))).numpy()
    assert np.array_equal((got & 0xF), zg[:, 0::2] + 1) and np.array_equal(got >> 4, zg[:, 1::2] + 1)

    # layer shapes, group-size halving, TP sharding of the expert tensors, host repack
    E, H, I = 4, 128, 192          # I/2 = 96 per rank: group 64 does not divide it -> 32
    full = MoeWNA16Config.from_config(dict(GPTQ_CFG, group_size=64))
    w1 = rng.integers(0, 16, size=(E, H, I), dtype=np.int32)     # checkpoint orientation [K, N]
    w2 = rng.integers(0, 16, size=(E, I, H), dtype=np.int32)
    s1 = rng.random((E, H // 64, I)).astype(np.float32)
    for rank in range(2):
        tp(rank, 2)
        moe = FusedMoE(E, 2, H, I, params_dtype=torch.bfloat16, quant_config=full, prefix="mlp.experts")
        assert moe.group_size == 32 and moe.group_size_div_factor == 2
        assert moe.w13_qweight.shape == (E, I, H // 2) and moe.w2_qweight.shape == (E, H, I // 4)
        assert moe.w13_scales.shape == (E, I, H // 32) and not hasattr(moe, "w13_qzeros")
        for e in range(E):
            for sid in ("w1", "w3"):
                moe.weight_loader(moe.w13_qweight, _t(orc.gptq_pack(w1[e])), f"experts.{e}.{sid}.qweight", sid, e)
                moe.weight_loader(moe.w13_scales, torch.from_numpy(s1[e]).to(torch.bfloat16),
                                  f"experts.{e}.{sid}.scales", sid, e)
            moe.weight_loader(moe.w2_qweight, _t(orc.gptq_pack(w2[e])), f"experts.{e}.w2.qweight", "w2", e)
            assert moe.weight_loader(moe.w2_qweight, torch.zeros(1), f"experts.{e}.w2.g_idx", "w2", e, True) is False
        half = I // 2
        for e in range(E):
            got13 = moe_ref.unpack_moe_weight(moe.w13_qweight.data[e].numpy())           # [K=H, N=2*half]
            exp = w1[e][:, rank * half:(rank + 1) * half]
            assert np.array_equal(got13, np.concatenate([exp, exp], axis=1))
            got2 = moe_ref.unpack_moe_weight(moe.w2_qweight.data[e].numpy())             # [K=half, N=H]
            assert np.array_equal(got2, w2[e][rank * half:(rank + 1) * half])
            sc = moe.w13_scales.data[e].float().numpy()                                  # [2*half, H/32]
            exp_s = torch.from_numpy(s1[e]).to(torch.bfloat16).float().numpy()[:, rank * half:(rank + 1) * half].T
            assert np.array_equal(sc[:half], np.repeat(exp_s, 2, axis=1))
        moe.process_weights_after_loading()
        assert moe.w13_packed.E == E and moe.w13_packed.K == H and moe.w13_packed.N == 2 * half
        w = moe.w2_packed.expert(1)
        back = np.empty((w.K, w.N), dtype=np.int32)
        _lib.check(_lib.lib().mi355q_w4a16_unpack_host(w.packed.data_ptr(), back.ctypes.data, w.K, w.N, w.tile))
        assert np.array_equal(back, w2[1][rank * half:(rank + 1) * half])


def test_act_order_row_permutation_helpers():
    """GPTQ desc_act: rows sorted by group (stable), packed rows moved nibble by nibble."""
    from mi355q.plugin.mp_kernel import _act_order_perm, _permute_packed_rows
    rng = np.random.default_rng(4)
    K, N, G = 256, 16, 64
    assert _act_order_perm(None, K, G) is None
    assert _act_order_perm(torch.empty(0, dtype=torch.int32), K, G) is None
    with pytest.raises(ValueError, match="exactly group_size rows"):                  # a g_idx that was never loaded
        _act_order_perm(torch.zeros(K, dtype=torch.int32), K, G)
    assert _act_order_perm(torch.zeros(K, dtype=torch.int32), K, K) is None          # one group: all zeros is right
    assert _act_order_perm(torch.arange(K, dtype=torch.int32) // G, K, G) is None     # already in group order
    g_idx = rng.permutation(np.arange(K) // G).astype(np.int32)
    perm = _act_order_perm(torch.from_numpy(g_idx), K, G)
    assert np.array_equal(g_idx[perm.numpy()], np.arange(K) // G)
    assert np.array_equal(perm.numpy(), np.argsort(g_idx, kind="stable"))
    with pytest.raises(ValueError, match="exactly group_size rows"):
        bad = g_idx.copy()
        bad[bad == 1] = 0
        _act_order_perm(torch.from_numpy(bad), K, G)
    with pytest.raises(ValueError, match="entries for"):
        _act_order_perm(torch.zeros(K - 8, dtype=torch.int32), K, G)
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    got = _permute_packed_rows(torch.from_numpy(orc.gptq_pack(q)), perm)
    assert np.array_equal(got.numpy(), orc.gptq_pack(q[perm.numpy()]))


# ---------------------------------------------------------------------------
# quantized output head (`lm_head: true` checkpoints)
# ---------------------------------------------------------------------------
def test_quantized_lm_head_gets_a_gptq_method_and_vocab_shards(tp):
    """gptq_utils.py:140-156: a ParallelLMHead takes the quantized linear method when the config says
    lm_head, the unquantized embedding method when it is excluded by a dynamic rule, and nothing
    special otherwise; its packed tensors are sharded along the vocabulary."""
    from mi355q.plugin.gptq import AutoGPTQLinearMethod
    from mi355q.plugin.linear import ParallelLMHead, UnquantizedEmbeddingMethod
    V, H, G = 1024, 256, 128
    rng = np.random.default_rng(7)
    q, s, _ = _gptq_tensors(rng, H, V, G)
    qw = _t(orc.gptq_pack(q))
    for size in (1, 2):
        for rank in range(size):
            tp(rank, size)
            c = AutoGPTQConfig.from_config(dict(GPTQ_CFG, lm_head=True))
            head = ParallelLMHead(V, H, params_dtype=torch.bfloat16, quant_config=c, prefix="lm_head")
            assert isinstance(head.quant_method, AutoGPTQLinearMethod)
            n = V // size
            assert head.qweight.shape == (H // 8, n) and head.scales.shape == (H // G, n)
            head.weight_loader(head.qweight, qw)
            head.weight_loader(head.scales, _bf(s))
            assert torch.equal(head.qweight.data, qw[:, rank * n:(rank + 1) * n])
            assert torch.equal(head.scales.data, _bf(s)[:, rank * n:(rank + 1) * n])
    tp(0, 1)
    plain = ParallelLMHead(V, H, params_dtype=torch.bfloat16, quant_config=AutoGPTQConfig.from_config(GPTQ_CFG),
                           prefix="lm_head")
    assert isinstance(plain.quant_method, UnquantizedEmbeddingMethod) and plain.weight.shape == (V, H)
    excluded = AutoGPTQConfig.from_config(dict(GPTQ_CFG, lm_head=True, dynamic={"-:lm_head": {}}))
    head = ParallelLMHead(V, H, params_dtype=torch.bfloat16, quant_config=excluded, prefix="lm_head")
    assert isinstance(head.quant_method, UnquantizedEmbeddingMethod)
    awq = AutoAWQConfig.from_config(dict(AWQ_CFG, lm_head=True))
    assert awq.get_quant_method(ParallelLMHead(V, H, params_dtype=torch.bfloat16), "lm_head") is not None
    assert AutoAWQConfig.from_config(AWQ_CFG).get_quant_method(ParallelLMHead(V, H, params_dtype=torch.bfloat16), "lm_head") is None
    # vocabulary padded to 64 * tp rows; the last rank's padding rows are zero
    tp(1, 2)
    odd = ParallelLMHead(1000, H, params_dtype=torch.bfloat16)
    assert odd.num_embeddings_padded == 1024 and odd.weight.shape == (512, H)
    w = torch.randn(1000, H).to(torch.bfloat16)
    odd.weight_loader(odd.weight, w)
    assert torch.equal(odd.weight.data[:488], w[512:]) and not odd.weight.data[488:].any()


def test_act_order_with_the_input_dimension_sharded_regroups_rows():
    """A rank of a row-parallel desc_act layer: rows sorted by group, groups padded to sub-groups of 32 with
    rows of the zero point, scales repeated per sub-group, inputs gathered in the same order; the packed
    weight unpacks to exactly those rows (host arithmetic of the repack entry point)."""
    from mi355q import _lib, ops
    from mi355q.plugin.mp_kernel import MI355XW4A16LinearKernel, MPLinearLayerConfig
    rng = np.random.default_rng(11)
    K_full, K, N, G = 1024, 256, 128, 128
    cfg = MPLinearLayerConfig(full_weight_shape=(K_full, N), partition_weight_shape=(K, N), weight_type=scalar_types.uint4b8,
                              act_type=torch.bfloat16, group_size=G, zero_points=False, has_g_idx=True)
    assert MI355XW4A16LinearKernel.can_implement(cfg) == (True, None)
    g_full = rng.permutation(np.arange(K_full) // G).astype(np.int32)
    g = g_full[256:512]                                                  # rank 1 of 4
    q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
    scales = torch.from_numpy(rng.random((K_full // G, N), dtype=np.float32)).to(torch.bfloat16)
    layer = torch.nn.Module()
    for name, t in (("qweight", torch.from_numpy(orc.gptq_pack(q))), ("scales", scales),
                    ("qzeros", torch.zeros((K_full // G, N // 8), dtype=torch.int32)), ("g_idx", torch.from_numpy(g))):
        layer.register_parameter(name, torch.nn.Parameter(t, requires_grad=False))
    kernel = MI355XW4A16LinearKernel(cfg, "qweight", "scales", "qzeros", "g_idx")
    kernel.process_weights_after_loading(layer)
    w, perm = layer.mi355q_weight, layer.mi355q_perm.numpy()
    counts = np.bincount(g, minlength=K_full // G)
    K2 = (int(((counts + 31) // 32 * 32).sum()) + 127) // 128 * 128
    assert (w.K, w.N, w.G) == (K2, N, 32) and perm.shape == (K2,) and layer.g_idx is None and w.zc is None
    back = np.empty((K2, N), dtype=np.int32)
    _lib.check(_lib.lib().mi355q_w4a16_unpack_host(w.packed.numpy().ctypes.data, back.ctypes.data, K2, N, w.tile))
    s_rows = ops.scales_row_major(w).float().numpy()                     # [K2 / 32, N]
    # the matrix the kernel multiplies, scattered back to the shard's own row order, is the definition
    W2 = (back - 8).astype(np.float64) * np.repeat(s_rows, 32, axis=0)
    W = np.zeros((K, N))
    np.add.at(W, perm, W2)
    want = (q - 8).astype(np.float64) * scales.float().numpy()[g]
    assert np.array_equal(W, want)
    pad_rows = np.ones(K2, dtype=bool)
    order = np.argsort(g, kind="stable")
    first, starts = np.cumsum(counts) - counts, np.cumsum((counts + 31) // 32 * 32) - (counts + 31) // 32 * 32
    pad_rows[starts[g[order]] + (np.arange(K) - first[g[order]])] = False
    assert np.all(back[pad_rows] == 8)                                   # padding multiplies to zero
    with pytest.raises(ValueError, match="held in full on every rank"):
        bad = torch.nn.Module()
        for name, t in (("qweight", torch.from_numpy(orc.gptq_pack(q))), ("scales", scales[:2]),
                        ("qzeros", torch.zeros((2, N // 8), dtype=torch.int32)), ("g_idx", torch.from_numpy(g))):
            bad.register_parameter(name, torch.nn.Parameter(t, requires_grad=False))
        MI355XW4A16LinearKernel(cfg, "qweight", "scales", "qzeros", "g_idx").process_weights_after_loading(bad)
