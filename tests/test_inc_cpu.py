"""AutoRound / INC checkpoint config (`quant_method: "auto-round"`): the unit cases of the reference's
tests/quantization/test_auto_round.py:142-470 against this package's mirror of
vllm/model_executor/layers/quantization/inc/."""
import pytest

from mi355q.plugin import resolve_quantization
from mi355q.plugin.awq import AutoAWQLinearMethod
from mi355q.plugin.gptq import AutoGPTQLinearMethod
from mi355q.plugin.inc import (INCConfig, INCLayerConfig, INCLinearMethod, INCMxfp4Scheme, INCMxfp8Scheme,
                               INCWNA16LinearScheme, INCWna16Scheme, resolve_scheme)
from mi355q.plugin.linear import LinearBase, UnquantizedLinearMethod
from mi355q.plugin.moe import MoeWNA16Method


class DummyLayer:
    pass


class DummyFusedMoE:
    pass


class ParallelLMHead:   # the parser recognises the head by its class name
    pass


def make_config(**overrides) -> INCConfig:
    kwargs = dict(weight_bits=4, group_size=128, sym=True, packing_format="auto_round:auto_gptq",
                  block_name_to_quantize=None, extra_config=None, data_type="int", backend="auto")
    kwargs.update(overrides)
    return INCConfig(**kwargs)


def test_exact_match():
    c = make_config(extra_config={"layers.0.self_attn.q_proj": {"bits": 8, "group_size": 64, "sym": False}})
    lc = c.config_parser.resolve(DummyLayer(), "layers.0.self_attn.q_proj")
    assert (lc.bits, lc.group_size, lc.sym, lc.quantized) == (8, 64, False, True)


def test_model_prefix_early_exit():
    c = make_config(extra_config={"model.layers.1.mlp.gate_proj": {"bits": 16}})
    assert isinstance(c.get_quant_method(DummyLayer(), "layers.1.mlp.gate_proj"), UnquantizedLinearMethod)


def test_regex_match_and_invalid_regex():
    c = make_config(extra_config={r"layers\.\d+\.self_attn\.(q|k|v)_proj": {"bits": 8, "group_size": 64, "sym": False}})
    lc = c.config_parser.resolve(DummyLayer(), "layers.3.self_attn.q_proj")
    assert (lc.bits, lc.group_size, lc.sym) == (8, 64, False)
    c = make_config(extra_config={"[invalid": {"bits": 8, "group_size": 64, "sym": False}})
    lc = c.config_parser.resolve(DummyLayer(), "layers.0.self_attn.q_proj")
    assert (lc.bits, lc.group_size, lc.sym) == (4, 128, True)


def test_block_name_to_quantize_and_lm_head():
    c = make_config(block_name_to_quantize=["layers.1"])
    lc = c.config_parser.resolve(DummyLayer(), "layers.0.self_attn.q_proj")
    assert (lc.bits, lc.group_size, lc.sym, lc.quantized) == (16, -1, True, False)
    assert make_config(block_name_to_quantize="a,b").block_name_to_quantize == ["a", "b"]
    lc = make_config().config_parser.resolve(ParallelLMHead(), "lm_head")
    assert lc.quantized is False and lc.bits == 16
    c = make_config(extra_config={"lm_head": {"bits": 4, "group_size": 128, "sym": True}})
    lc = c.config_parser.resolve(ParallelLMHead(), "model.language_model.lm_head")
    assert lc.quantized is True and (lc.bits, lc.group_size, lc.sym) == (4, 128, True)


def test_fused_layers_need_consistent_configs():
    c = make_config(extra_config={"layers.0.block_sparse_moe.experts.0.w1": {"bits": 4, "group_size": 128, "sym": True},
                                  "layers.0.block_sparse_moe.experts.0.w2": {"bits": 8, "group_size": 128, "sym": True}})
    with pytest.raises(ValueError, match="requires consistent quant config"):
        c.config_parser.resolve(DummyFusedMoE(), "layers.0.block_sparse_moe")
    c = make_config(extra_config={"layers.0.self_attn.q_proj": {"bits": 4, "group_size": 128, "sym": True},
                                  "layers.0.self_attn.k_proj": {"bits": 8, "group_size": 128, "sym": True},
                                  "layers.0.self_attn.v_proj": {"bits": 4, "group_size": 128, "sym": True}})
    c.packed_modules_mapping = {"qkv_proj": ["q_proj", "k_proj", "v_proj"]}
    with pytest.raises(ValueError, match="requires consistent quant config"):
        c.config_parser.resolve(DummyLayer(), "layers.0.self_attn.qkv_proj")
    # consistent shards resolve to the shared value; an unrelated name containing the key does not trigger
    c.extra_config["layers.0.self_attn.k_proj"]["bits"] = 4
    assert c.config_parser.resolve(DummyLayer(), "layers.0.self_attn.qkv_proj").bits == 4
    assert c.config_parser.resolve(DummyLayer(), "layers.5.linear_attn.in_proj_qkvz").bits == 4


def test_validation_of_the_global_values():
    with pytest.raises(ValueError, match="weight_bits=5 is not one of"):
        make_config(weight_bits=5)
    with pytest.raises(ValueError, match="data_type='nf4' is not one of"):
        make_config(data_type="nf4")
    with pytest.raises(ValueError, match="packing_format='gguf' is not one of"):
        make_config(packing_format="gguf")
    with pytest.raises(ValueError, match="backend='exllama' is not one of"):
        make_config(backend="exllama")
    with pytest.raises(ValueError, match="only goes with a microscaling"):
        make_config(packing_format="auto_round:llm_compressor")
    c = make_config(weight_bits=8, group_size=32, packing_format="auto_round:llm_compressor", data_type="mx_fp")
    assert c.is_mxfp8 and c.weight_bits == 8
    with pytest.raises(AssertionError, match="needs group_size=32"):
        make_config(weight_bits=8, group_size=128, packing_format="auto_round:llm_compressor", data_type="mx_fp")
    with pytest.raises(AssertionError, match="needs act_dynamic=True"):
        INCConfig.from_config({"bits": 8, "group_size": 32, "sym": True, "packing_format": "auto_round:llm_compressor",
                               "data_type": "mx_fp", "act_bits": 8, "act_data_type": "mx_fp", "act_group_size": 32,
                               "act_sym": True, "act_dynamic": False})


def test_scheme_resolution():
    lc = INCLayerConfig(bits=4, group_size=32, sym=True, packing_format="", backend="", data_type="mx_fp", quantized=True)
    assert lc.is_mxfp4 and not lc.is_mxfp8
    assert isinstance(resolve_scheme(lc), INCMxfp4Scheme)
    lc8 = INCLayerConfig(bits=8, group_size=32, sym=True, packing_format="", backend="", data_type="mx_fp", quantized=True)
    assert isinstance(resolve_scheme(lc8), INCMxfp8Scheme)
    with pytest.raises(NotImplementedError, match="microscaling"):
        resolve_scheme(lc).get_linear_method(None, DummyLayer(), "layers.0.mlp.down_proj", lc)
    lcw = INCLayerConfig(bits=4, group_size=128, sym=True, packing_format="auto_round:auto_gptq", backend="auto",
                         data_type="int", quantized=True)
    assert isinstance(resolve_scheme(lcw), INCWna16Scheme)
    c = INCConfig.from_config({"quant_method": "auto-round", "bits": 4, "group_size": 32, "sym": True,
                               "packing_format": "auto_round:llm_compressor", "data_type": "mx_fp4e2m1"})
    assert c.config_parser.resolve(DummyLayer(), "model.layers.0.mlp.down_proj").is_mxfp4


def test_layers_route_to_the_gptq_and_awq_methods():
    lin = object.__new__(LinearBase)
    c = INCConfig.from_config({"quant_method": "auto-round", "bits": 4, "group_size": 128, "sym": True,
                               "extra_config": {"model.layers.0.mlp.gate": {"bits": 16, "data_type": "float"}}})
    m = c.get_quant_method(lin, "model.layers.0.mlp.down_proj")
    assert isinstance(m, INCLinearMethod) and isinstance(m.scheme, INCWNA16LinearScheme)
    inner = m.scheme.inner_method
    assert isinstance(inner, AutoGPTQLinearMethod)
    assert (inner.quant_config.weight_bits, inner.quant_config.group_size, inner.quant_config.is_sym,
            inner.quant_config.desc_act) == (4, 128, True, False)
    assert isinstance(c.get_quant_method(lin, "model.layers.0.mlp.gate"), UnquantizedLinearMethod)
    assert c.get_quant_method(DummyLayer(), "model.layers.0.input_layernorm") is None

    a = INCConfig.from_config({"quant_method": "auto-round", "bits": 4, "group_size": 64, "sym": False,
                               "packing_format": "auto_round:auto_awq"})
    inner = a.get_quant_method(lin, "model.layers.2.self_attn.o_proj").scheme.inner_method
    assert isinstance(inner, AutoAWQLinearMethod)
    assert (inner.quant_config.group_size, inner.quant_config.zero_point) == (64, True)

    inner8 = make_config(weight_bits=8).get_quant_method(lin, "model.layers.0.mlp.down_proj").scheme.inner_method
    assert isinstance(inner8, AutoGPTQLinearMethod) and inner8.quant_config.weight_bits == 8
    with pytest.raises(NotImplementedError, match="Only 4-bit and 8-bit symmetric"):
        make_config(weight_bits=2).get_quant_method(lin, "model.layers.0.mlp.down_proj")
    with pytest.raises(NotImplementedError, match="Only 4-bit and 8-bit symmetric"):
        make_config(sym=False).get_quant_method(lin, "model.layers.0.mlp.down_proj")

    moe = c.get_quant_method(DummyFusedMoE(), "model.layers.0.mlp.experts")
    assert isinstance(moe, MoeWNA16Method) and moe.quant_config.linear_quant_method == "gptq" and not moe.quant_config.has_zp
    moe = a.get_quant_method(DummyFusedMoE(), "model.layers.0.mlp.experts")
    assert moe.quant_config.linear_quant_method == "awq" and moe.quant_config.has_zp


def test_auto_round_checkpoints_resolve_to_inc():
    assert INCConfig.override_quantization_method({"quant_method": "auto-round"}, user_quant=None) == "inc"
    assert INCConfig.override_quantization_method({"quant_method": "gptq"}, user_quant=None) is None
    assert INCConfig.get_config_filenames() == ["quantization_config.json"]
    assert resolve_quantization({"quant_method": "auto-round", "bits": 4, "group_size": 128, "sym": True}) == "inc"
    c = make_config()
    assert c.get_layer_config(DummyLayer(), "layers.0.mlp.up_proj") == (4, 128, True)
    assert "INCConfig(weight_bits=4" in repr(c)
