"""Randomly drawn shapes through the decode and prompt GEMMs (whatever kernel and form the planner picks for each) and
through both prompt attention kernels, against the oracle: the cases of tools/fuzz_decode.py, tools/fuzz_mm.py and
tools/fuzz_prefill_attention.py with fixed seeds (the tools take other seeds and counts on the command line)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [5, 6])
def test_decode_gemm_random_shapes(seed):
    """1 to 16 rows, GPTQ / AWQ, group sizes 32 to K, every strip width, bias, and the fused add + RMSNorm / SiLU
    forms where the layer takes them; a strip width the library chose itself must never be refused."""
    assert _tool("fuzz_decode").run(60, seed, verbose=False) == 0


@pytest.mark.parametrize("seed", [11, 12])
def test_prompt_gemm_random_shapes(seed):
    """Tolerance of the W4A16 tests (rtol = atol = 1e-2, tests/kernels/quantization/test_triton_w4a16.py:162)."""
    assert _tool("fuzz_mm").run(20, seed, verbose=False) == 0


@pytest.mark.parametrize("seed", [13, 14])
def test_decode_batch_gemm_random_shapes(seed):
    """13 to 32 rows through the many-row entry on wide layers: the streaming rows kernel in its forms (fragments
    gathered per lane or whole rows through LDS, 32- and 112-wide strips) and the 32-row tiles with K-slices."""
    assert _tool("fuzz_mm").run(16, seed, verbose=False, decode_batches=True) == 0


@pytest.mark.parametrize("seed", [21, 22])
def test_prompt_attention_random_configurations(seed):
    """Bars of test_paged_prefill_attention_vs_oracle for each kernel (partitioned 3e-3, running softmax 4e-3, fp8 cache 1e-2)."""
    from mi355q import _lib
    try:
        assert _tool("fuzz_prefill_attention").run(12, seed, verbose=False) == 0
    finally:
        _lib.set_option("MI355Q_PREFILL_FLASH_MIN", None)
        _lib.set_option("MI355Q_PREFILL_FLASH_WGS", None)
        _lib.set_option("MI355Q_PREFILL_FLASH_CTL", None)
