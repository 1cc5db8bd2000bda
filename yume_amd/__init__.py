"""yume_amd — MI355X (gfx950) transform-and-visibility path of the Yume engine.

Product code only: HIP kernels + C ABI (csrc/, libyume_hip.so), the ctypes binding (_lib), numpy-facing
wrappers (api), the synthetic-scene harness (synth) and the engine-facing system mirror (systems).
Nothing here imports the CPU oracle (oracle/ is test infrastructure).
"""
from . import _lib  # noqa: F401
from .api import Context, Scene, T68  # noqa: F401

__all__ = ["Context", "Scene", "T68"]
