"""MI355X (gfx950) native two-tower retrieval training hot path.

Drop-in for the path the reference declares but never implemented
(``/root/reference/src/models/__init__.py:1``, ``src/training/__init__.py:1``;
schema ``configs/data_config.yaml:54-71``; entry point ``pyproject.toml:67``).
All compute runs in hand-written HIP kernels behind the C ABI of
``include/twotower_hip.h``; see DESIGN.md.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
__version__ = "0.1.0"
