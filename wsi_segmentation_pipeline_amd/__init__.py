"""MI355X-native (gfx950) per-patch CNN inference path for whole-slide images.

Product code: HIP kernels + C ABI in ``csrc/`` (built to ``lib/libwsi_hip.so``), the ctypes
binding ``native``, the launch/weight-prepack host side ``engine`` and the slide/tile host logic
``slide``.  The reference-named drop-in modules (``resnets_shift``, ``models.models``,
``utils.eval`` ...) at the repository root are thin mirrors of the reference API over this package.
"""
from . import native  # noqa: F401

__all__ = ['native']
