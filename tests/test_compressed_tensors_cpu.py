"""compressed-tensors config handling on the CPU: group parsing, target / ignore matching, scheme
resolution.  Cases follow the reference's own checks in
vllm/model_executor/layers/quantization/compressed_tensors/{compressed_tensors.py, utils.py} and the
layer / scheme pairs listed in tests/quantization/test_compressed_tensors.py (W4A16 / W8A16 pack-quantized
group and channel, fp8 W8A16, fp8 W8A8 static and dynamic) -- the checkpoints those tests download are
not available offline, so configs of the same shape are written out here."""
import numpy as np
import pytest
import torch

from mi355q.plugin.compressed_tensors import (CompressedTensorsConfig, CompressedTensorsKVCacheMethod,
                                               CompressedTensorsLinearMethod, CompressedTensorsW8A16Fp8,
                                               CompressedTensorsWNA16, QuantizationArgs, find_matched_target,
                                               is_activation_quantization_format, should_ignore_layer)
from mi355q.plugin.linear import (MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear,
                                  UnquantizedLinearMethod)
from mi355q.plugin.registry import get_quantization_config, resolve_quantization
from mi355q.plugin.scalar_type import scalar_types

FUSED = {"qkv_proj": ["q_proj", "k_proj", "v_proj"], "gate_up_proj": ["gate_proj", "up_proj"]}


def w4a16_config(**weights):
    w = {"num_bits": 4, "type": "int", "symmetric": True, "strategy": "group", "group_size": 128, "dynamic": False}
    w.update(weights)
    return {"quant_method": "compressed-tensors", "format": "pack-quantized", "ignore": ["lm_head"],
            "config_groups": {"group_0": {"targets": ["Linear"], "weights": w, "input_activations": None}}}


def make(cfg_dict):
    cfg = CompressedTensorsConfig.from_config(cfg_dict)
    cfg.packed_modules_mapping = FUSED
    return cfg


def test_registry_name_and_resolution():
    assert get_quantization_config("compressed-tensors") is CompressedTensorsConfig
    assert resolve_quantization({"quant_method": "compressed-tensors"}) == "compressed-tensors"
    assert CompressedTensorsConfig.get_name() == "compressed-tensors"
    assert CompressedTensorsConfig.get_config_filenames() == []
    assert torch.bfloat16 in CompressedTensorsConfig.get_supported_act_dtypes()


def test_quantization_args_defaults_follow_the_library():
    assert QuantizationArgs.model_validate({"num_bits": 4, "group_size": 128}).strategy == "group"
    assert QuantizationArgs.model_validate({"num_bits": 4, "group_size": -1}).strategy == "channel"
    assert QuantizationArgs.model_validate({"num_bits": 8}).strategy == "tensor"
    assert QuantizationArgs.model_validate({"num_bits": 4, "group_size": 64, "actorder": True}).actorder == "group"
    assert QuantizationArgs.model_validate({"num_bits": 4, "group_size": 64, "actorder": "static"}).actorder == "weight"
    assert QuantizationArgs.model_validate({"num_bits": 4, "group_size": 64, "actorder": False}).actorder is None
    assert QuantizationArgs.model_validate({"num_bits": 8, "type": "float", "strategy": "block",
                                            "block_structure": "128x128"}).block_structure == [128, 128]
    with pytest.raises(ValueError):
        QuantizationArgs.model_validate({"num_bits": 4, "strategy": "group"})
    with pytest.raises(ValueError):
        QuantizationArgs.model_validate({"num_bits": 4, "group_size": 64, "actorder": "sideways"})
    # unknown keys (observer settings ...) are kept aside, not an error
    assert QuantizationArgs.model_validate({"num_bits": 8, "observer": "minmax"}).extra == {"observer": "minmax"}


def test_activation_quantization_formats():
    for f in ("naive-quantized", "int-quantized", "float-quantized", "nvfp4-pack-quantized"):
        assert is_activation_quantization_format(f)
    for f in ("pack-quantized", "dense", None):
        assert not is_activation_quantization_format(f)


def test_ignore_matching_exact_regex_and_fused():
    assert should_ignore_layer("lm_head", ["lm_head"])
    assert not should_ignore_layer("model.layers.0.mlp.down_proj", ["lm_head"])
    assert should_ignore_layer("model.layers.3.mlp.down_proj", ["re:.*layers\\.3\\..*"])
    assert not should_ignore_layer(None, ["lm_head"])
    ign = ["model.layers.0.self_attn.q_proj", "model.layers.0.self_attn.k_proj", "model.layers.0.self_attn.v_proj"]
    assert should_ignore_layer("model.layers.0.self_attn.qkv_proj", ign, FUSED)
    assert not should_ignore_layer("model.layers.1.self_attn.qkv_proj", ign, FUSED)
    with pytest.raises(ValueError, match="same scheme"):
        should_ignore_layer("model.layers.0.self_attn.qkv_proj", ign[:2], FUSED)


def test_target_matching_order_name_then_fused_then_class():
    lin = RowParallelLinear(64, 64, bias=False, params_dtype=torch.bfloat16, prefix="x", return_bias=False)
    assert find_matched_target("model.layers.0.mlp.down_proj", lin, ["Linear"]) == "Linear"
    assert find_matched_target("model.layers.0.mlp.down_proj", lin, ["Linear", "re:.*down_proj"]) == "re:.*down_proj"
    assert find_matched_target("model.layers.0.mlp.down_proj", lin, ["Embedding"]) is None
    targets = ["re:.*\\.q_proj", "re:.*\\.k_proj", "re:.*\\.v_proj"]
    assert find_matched_target("model.layers.0.self_attn.qkv_proj", lin, targets, FUSED) == "re:.*\\.q_proj"
    assert find_matched_target("model.layers.0.self_attn.qkv_proj", lin, targets[:2], FUSED) is None


@pytest.mark.parametrize("weights,qtype,G,zp", [
    ({}, scalar_types.uint4b8, 128, False),
    ({"strategy": "channel", "group_size": None}, scalar_types.uint4b8, -1, False),
    ({"num_bits": 8}, scalar_types.uint8b128, 128, False),
    ({"symmetric": False}, scalar_types.uint4, 128, True),
])
def test_wna16_scheme_resolution_and_parameters(weights, qtype, G, zp):
    cfg = make(w4a16_config(**weights))
    lin = MergedColumnParallelLinear(512, [256, 256], bias=False, params_dtype=torch.bfloat16, quant_config=cfg,
                                     prefix="model.layers.0.mlp.gate_up_proj", return_bias=False)
    assert isinstance(lin.quant_method, CompressedTensorsLinearMethod)
    s = lin.scheme
    assert isinstance(s, CompressedTensorsWNA16)
    assert (s.quant_type, s.group_size, not s.symmetric, s.has_g_idx) == (qtype, G, zp, False)
    bits = s.num_bits
    assert lin.weight_packed.shape == (512, 512 * bits // 32) and lin.weight_packed.dtype == torch.int32
    assert (lin.weight_packed.input_dim, lin.weight_packed.output_dim, lin.weight_packed.packed_dim) == (1, 0, 1)
    assert lin.weight_scale.shape == (512, 1 if G == -1 else 512 // G)
    assert lin.weight_shape.shape == (2,)
    assert hasattr(lin, "weight_zero_point") == zp
    if zp:
        assert lin.weight_zero_point.shape == (512 * bits // 32, 512 // G) and lin.weight_zero_point.packed_dim == 0
    assert not hasattr(lin, "weight_g_idx")
    assert type(s.kernel).__name__ == ("MI355XW8A16LinearKernel" if bits == 8 else "MI355XW4A16LinearKernel")


def test_ignored_and_unmatched_layers_stay_unquantized():
    cfg = make(w4a16_config())
    head = RowParallelLinear(512, 256, bias=False, params_dtype=torch.bfloat16, quant_config=cfg, prefix="lm_head",
                             return_bias=False)
    assert isinstance(head.quant_method, UnquantizedLinearMethod)
    d = w4a16_config()
    d["config_groups"]["group_0"]["targets"] = ["re:.*down_proj"]
    cfg = make(d)
    qkv = QKVParallelLinear(512, 64, 8, 2, bias=False, params_dtype=torch.bfloat16, quant_config=cfg,
                            prefix="model.layers.0.self_attn.qkv_proj", return_bias=False)
    assert isinstance(qkv.quant_method, UnquantizedLinearMethod)
    down = RowParallelLinear(512, 256, bias=False, params_dtype=torch.bfloat16, quant_config=cfg,
                             prefix="model.layers.0.mlp.down_proj", return_bias=False)
    assert isinstance(down.quant_method, CompressedTensorsLinearMethod)


def test_actorder_group_adds_g_idx_and_weight_ordering_does_not():
    cfg = make(w4a16_config(actorder="group"))
    lin = RowParallelLinear(512, 256, bias=False, params_dtype=torch.bfloat16, quant_config=cfg,
                            prefix="model.layers.0.mlp.down_proj", return_bias=False)
    assert lin.scheme.has_g_idx and lin.weight_g_idx.shape == (512,)
    cfg = make(w4a16_config(actorder="weight"))
    lin = RowParallelLinear(512, 256, bias=False, params_dtype=torch.bfloat16, quant_config=cfg,
                            prefix="model.layers.0.mlp.down_proj", return_bias=False)
    assert not lin.scheme.has_g_idx and not hasattr(lin, "weight_g_idx")


def fp8_config(strategy="channel", input_activations=None, block=None):
    w = {"num_bits": 8, "type": "float", "symmetric": True, "strategy": strategy, "dynamic": False}
    if block:
        w["block_structure"] = block
    return {"quant_method": "compressed-tensors", "format": "float-quantized", "ignore": ["lm_head"],
            "config_groups": {"group_0": {"targets": ["Linear"], "weights": w,
                                          "input_activations": input_activations}}}


@pytest.mark.parametrize("strategy,act,static", [
    ("channel", None, False),
    ("tensor", {"num_bits": 8, "type": "float", "symmetric": True, "strategy": "tensor", "dynamic": False}, True),
    ("channel", {"num_bits": 8, "type": "float", "symmetric": True, "strategy": "token", "dynamic": True}, False),
    ("block", None, False),
])
def test_fp8_checkpoints_resolve_to_the_weight_only_scheme(strategy, act, static, monkeypatch):
    monkeypatch.setenv("MI355Q_FP8_LINEAR", "w8a16")
    cfg = make(fp8_config(strategy, act, [128, 128] if strategy == "block" else None))
    lin = MergedColumnParallelLinear(512, [256, 256], bias=False, params_dtype=torch.bfloat16, quant_config=cfg,
                                     prefix="model.layers.0.mlp.gate_up_proj", return_bias=False)
    s = lin.scheme
    assert type(s) is CompressedTensorsW8A16Fp8 and s.is_static_input_scheme == static
    assert lin.weight.dtype == torch.float8_e4m3fn and lin.weight.shape == (512, 512)
    want = {"channel": (512, 1), "tensor": (2,), "block": (4, 4)}[strategy]
    assert lin.weight_scale.shape == want
    assert hasattr(lin, "input_scale") == static
    assert cfg.has_blocked_weights() == (strategy == "block")


@pytest.mark.parametrize("block,dynamic,w8a8", [([128, 128], True, True), ([64, 64], True, False), ([128, 128], False, False)])
def test_block_quantised_fp8_w8a8_checkpoints(block, dynamic, w8a8, monkeypatch):
    """schemes/compressed_tensors_w8a8_fp8.py:69-79: 128 x 128 weight blocks with dynamic activation scales take the
    activation-quantising scheme (one scale per token and 128-wide group); other block shapes, and static
    activation scales (which the reference rejects, :72), are served weight-only."""
    from mi355q.plugin.compressed_tensors import CompressedTensorsW8A8Fp8
    monkeypatch.delenv("MI355Q_FP8_LINEAR", raising=False)
    act = {"num_bits": 8, "type": "float", "symmetric": True, "strategy": "group" if dynamic else "tensor",
           "dynamic": dynamic}
    if dynamic:
        act["group_size"] = block[0]
    cfg = make(fp8_config("block", act, block))
    lin = MergedColumnParallelLinear(512, [256, 256], bias=False, params_dtype=torch.bfloat16, quant_config=cfg,
                                     prefix="model.layers.0.mlp.gate_up_proj", return_bias=False)
    assert type(lin.scheme) is (CompressedTensorsW8A8Fp8 if w8a8 else CompressedTensorsW8A16Fp8)
    assert lin.weight_scale.shape == (512 // block[0], 512 // block[1])
    lin.process_weights_after_loading()
    assert lin.weight_scale.dtype == torch.float32 and lin.mi355q_scale_block == tuple(block)
    assert getattr(lin, "input_scale", None) is None
    with pytest.raises(ValueError, match="128 x 128"):
        CompressedTensorsW8A8Fp8(QuantizationArgs.model_validate(
            {"num_bits": 8, "type": "float", "strategy": "block", "block_structure": [64, 128]}), False)


@pytest.mark.parametrize("strategy,dynamic", [("tensor", False), ("tensor", True), ("channel", False), ("channel", True)])
def test_fp8_w8a8_checkpoints_resolve_to_the_activation_quantising_scheme(strategy, dynamic, monkeypatch):
    """compressed_tensors.py:831-846: an fp8 W8A8 checkpoint with tensor / channel weight scales takes the
    scheme that quantises activations; fused per-tensor shards end on one scale (the largest), their fp8
    values rewritten for it (fp8_utils.py:1321-1346)."""
    from mi355q.plugin.compressed_tensors import CompressedTensorsW8A8Fp8
    monkeypatch.delenv("MI355Q_FP8_LINEAR", raising=False)
    act = {"num_bits": 8, "type": "float", "symmetric": True, "strategy": "token" if dynamic else "tensor",
           "dynamic": dynamic}
    cfg = make(fp8_config(strategy, act, None))
    lin = MergedColumnParallelLinear(64, [32, 48], bias=False, params_dtype=torch.bfloat16, quant_config=cfg,
                                     prefix="model.layers.0.mlp.gate_up_proj", return_bias=False)
    assert type(lin.scheme) is CompressedTensorsW8A8Fp8 and lin.scheme.is_static_input_scheme == (not dynamic)
    g = torch.Generator().manual_seed(3)
    ws, scales = [], [0.02, 0.05]
    for i, width in enumerate([32, 48]):
        w8 = (torch.randn((width, 64), generator=g) * 20).to(torch.float8_e4m3fn)
        lin.weight_loader(lin.weight, w8, i)
        if strategy == "tensor":
            lin.weight_loader(lin.weight_scale, torch.tensor(scales[i]), i)
            ws.append(w8.float() * scales[i])
        else:
            sc = 0.01 + 0.02 * torch.rand((width, 1), generator=g)
            lin.weight_loader(lin.weight_scale, sc, i)
            ws.append(w8.float() * sc)
        if not dynamic:
            lin.weight_loader(lin.input_scale, torch.tensor(0.1 * (i + 1)), i)
    lin.process_weights_after_loading()
    assert lin.weight_scale.shape == ((1,) if strategy == "tensor" else (80,))
    if strategy == "tensor":
        assert float(lin.weight_scale) == pytest.approx(0.05)
    # the stored values times the stored scale still are the checkpoint's weights (to fp8 rounding at the new scale)
    deq = lin.weight.float() * lin.weight_scale.reshape(-1, 1)
    torch.testing.assert_close(deq, torch.cat(ws), rtol=0.07, atol=0.05 * 2 ** -9)
    if dynamic:
        assert lin.input_scale is None
    else:
        assert lin.input_scale.shape == (1,) and float(lin.input_scale) == pytest.approx(0.2)


def test_schemes_outside_the_weight_only_path_are_named_and_refused():
    lin = RowParallelLinear(512, 256, bias=False, params_dtype=torch.bfloat16, prefix="x", return_bias=False)
    int8_act = {"num_bits": 8, "type": "int", "symmetric": True, "strategy": "token", "dynamic": True}
    d = w4a16_config(num_bits=8, strategy="channel", group_size=None)
    d["format"] = "int-quantized"
    d["config_groups"]["group_0"]["input_activations"] = int8_act
    with pytest.raises(NotImplementedError, match="W8A8"):
        make(d).get_quant_method(lin, "model.layers.0.mlp.down_proj")
    nv = {"quant_method": "compressed-tensors", "format": "nvfp4-pack-quantized",
          "config_groups": {"g": {"targets": ["Linear"], "weights": {
              "num_bits": 4, "type": "float", "symmetric": True, "strategy": "tensor_group", "group_size": 16}}}}
    with pytest.raises(NotImplementedError, match="fp4"):
        make(nv).get_quant_method(lin, "model.layers.0.mlp.down_proj")
    # pack-quantized needs group or channel scales; widths the format defines but no kernel serves
    with pytest.raises(NotImplementedError, match="No compressed-tensors compatible scheme"):
        make(w4a16_config(strategy="tensor", group_size=None)).get_quant_method(lin, "model.layers.0.mlp.down_proj")
    with pytest.raises(ValueError, match="num_bits = 3"):
        make(w4a16_config(num_bits=3)).get_quant_method(lin, "model.layers.0.mlp.down_proj")
    with pytest.raises(ValueError, match="Asymmetric"):
        make(w4a16_config(num_bits=8, symmetric=False)).get_quant_method(lin, "model.layers.0.mlp.down_proj")
    with pytest.raises(DeprecationWarning):
        CompressedTensorsConfig.from_config({**w4a16_config(), "sparsity_config": {"targets": ["Linear"]}})
    with pytest.raises(NotImplementedError, match="transform_config"):
        CompressedTensorsConfig.from_config({**w4a16_config(), "transform_config": {"config_groups": {"a": {}}}})


def test_attention_only_groups_are_dropped_and_kv_cache_scheme_is_validated():
    d = w4a16_config()
    d["config_groups"]["group_1"] = {"targets": ["LlamaAttention"], "weights": None,
                                     "input_activations": {"num_bits": 8, "type": "float"}}
    cfg = make(d)
    assert list(cfg.target_scheme_map) == ["Linear"]
    ok = {"num_bits": 8, "type": "float", "strategy": "tensor", "symmetric": True, "dynamic": False}
    CompressedTensorsKVCacheMethod.validate_kv_cache_scheme(None)
    CompressedTensorsKVCacheMethod.validate_kv_cache_scheme(ok)
    for bad in ({**ok, "num_bits": 4}, {**ok, "type": "int"}, {**ok, "strategy": "attn_head"},
                {**ok, "symmetric": False}):
        with pytest.raises(NotImplementedError):
            CompressedTensorsKVCacheMethod.validate_kv_cache_scheme(bad)


def test_mapper_renames_layer_paths_only():
    class Mapper:
        def _map_name(self, name):
            return None if name.startswith("drop.") else "language_model." + name

    d = w4a16_config()
    d["config_groups"]["group_0"]["targets"] = ["Linear", "re:.*proj", "model.layers.0.mlp.down_proj", "drop.me.x"]
    d["ignore"] = ["lm_head", "model.embed.x"]
    cfg = make(d)
    cfg.apply_vllm_mapper(Mapper())
    assert list(cfg.target_scheme_map) == ["Linear", "re:.*proj", "language_model.model.layers.0.mlp.down_proj"]
    assert cfg.ignore == ["lm_head", "language_model.model.embed.x"]


# -- routed experts ---------------------------------------------------------------------------------------
def _ct_expert_tensors(rng, K, N, G):
    """One expert projection as a pack-quantized checkpoint holds it: weight_packed [N, K/8] int32 (eight
    consecutive input channels per word, value + 8), weight_scale [N, K/G], weight_shape [N, K]."""
    q = rng.integers(0, 16, size=(N, K), dtype=np.int64)
    words = np.zeros((N, K // 8), dtype=np.int64)
    for j in range(8):
        words |= q[:, j::8] << (4 * j)
    packed = torch.from_numpy(words.astype(np.uint32).view(np.int32))
    scale = torch.from_numpy((0.02 * rng.random((N, K // G)) + 0.005).astype(np.float32)).to(torch.bfloat16)
    return q, packed, scale, torch.tensor([N, K], dtype=torch.int64)


@pytest.mark.parametrize("tp", [(0, 1), (1, 2)])
def test_routed_experts_resolve_to_the_wna16_expert_method_and_load_transposed(tp, monkeypatch):
    """compressed_tensors_moe.py:28-76 (scheme of expert 0's projections, "Linear" targets cover the fused
    experts) and compressed_tensors_moe_wna16.py:136-330 (transposed parameter shapes, sharding along the
    transposed dims)."""
    from mi355q.plugin import FusedMoE, CompressedTensorsWNA16MoEMethod
    from mi355q.plugin import moe as moe_mod
    rank, size = tp
    monkeypatch.setattr(moe_mod, "get_tensor_model_parallel_rank", lambda: rank)
    monkeypatch.setattr(moe_mod, "get_tensor_model_parallel_world_size", lambda: size)
    E, H, I, G = 2, 256, 128 * size, 64
    cfg = make(w4a16_config(group_size=G))
    moe = FusedMoE(E, 2, H, I, params_dtype=torch.bfloat16, quant_config=cfg, prefix="model.layers.0.mlp.experts")
    assert isinstance(moe.quant_method, CompressedTensorsWNA16MoEMethod)
    Ip = I // size
    assert moe.w13_weight_packed.shape == (E, H // 8, 2 * Ip) and moe.w13_weight_packed.dtype == torch.int32
    assert moe.w2_weight_packed.shape == (E, Ip // 8, H)
    assert moe.w13_weight_scale.shape == (E, H // G, 2 * Ip) and moe.w2_weight_scale.shape == (E, Ip // G, H)
    assert moe.w13_weight_shape.shape == (E, 2)
    rng = np.random.default_rng(5)
    for sid, (K, N) in (("w1", (H, I)), ("w3", (H, I)), ("w2", (I, H))):
        q, packed, scale, shape = _ct_expert_tensors(rng, K, N, G)
        tgt = "w2" if sid == "w2" else "w13"
        name = f"experts.1.{sid}"
        assert moe.weight_loader(getattr(moe, tgt + "_weight_packed"), packed, name + ".weight_packed", sid, 1, True)
        assert moe.weight_loader(getattr(moe, tgt + "_weight_scale"), scale, name + ".weight_scale", sid, 1, True)
        assert moe.weight_loader(getattr(moe, tgt + "_weight_shape"), shape, name + ".weight_shape", sid, 1, True)
        assert not moe.weight_loader(getattr(moe, tgt + "_weight_packed"), torch.zeros(K, dtype=torch.int32),
                                     name + ".weight_g_idx", sid, 1, True)
        got_w = getattr(moe, tgt + "_weight_packed").data[1]
        got_s = getattr(moe, tgt + "_weight_scale").data[1]
        if sid == "w2":      # input channels sharded: rows of the transposed tensors
            assert torch.equal(got_w, packed.t()[rank * Ip // 8:(rank + 1) * Ip // 8])
            assert torch.equal(got_s, scale.t()[rank * Ip // G:(rank + 1) * Ip // G])
        else:                # output channels sharded, gate before up
            c0 = 0 if sid == "w1" else Ip
            assert torch.equal(got_w[:, c0:c0 + Ip], packed.t()[:, rank * Ip:(rank + 1) * Ip])
            assert torch.equal(got_s[:, c0:c0 + Ip], scale.t()[:, rank * Ip:(rank + 1) * Ip])
        assert getattr(moe, tgt + "_weight_shape").data[1].tolist() == [N, K]


def test_routed_expert_schemes_outside_the_kernel_are_refused():
    """The checks of the branch without Marlin (compressed_tensors_moe_wna16.py:118-127) and of the format
    (compressed_tensors_moe.py:76-89)."""
    from mi355q.plugin import FusedMoE

    def build(d):
        return FusedMoE(2, 2, 256, 128, params_dtype=torch.bfloat16, quant_config=make(d), prefix="model.layers.0.mlp.experts")

    with pytest.raises(ValueError, match="channelwise"):
        build(w4a16_config(strategy="channel", group_size=None))
    with pytest.raises(ValueError, match="actorder"):
        build(w4a16_config(actorder="group"))
    asym = build(w4a16_config(symmetric=False))     # zero points: eight output channels per word
    assert asym.w13_weight_zero_point.shape == (2, 2, 2 * 128 // 8) and asym.w2_weight_zero_point.shape == (2, 1, 256 // 8)
    with pytest.raises(ValueError, match="4-bit"):
        build(w4a16_config(num_bits=8, strategy="group"))
    d = w4a16_config()
    d["ignore"] = ["re:.*experts.*"]
    with pytest.raises(NotImplementedError, match="not quantized"):
        build(d)
    build(w4a16_config(actorder="weight"))      # reordered when quantized: nothing to do at run time
