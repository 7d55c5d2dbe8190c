"""mi355q: MI355X-native (gfx950) quantized-decode path behind the plugin surface of
the vLLM fork (quantization configs, linear kernels, attention backend, custom ops)."""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
