"""`from fast_slic_amd.hip import SlicHip` -- the same pattern as fast_slic/avx2.py:1-14."""
from ._binding import SlicModel, is_supported_arch, ARCH_NAME
from .base_slic import BaseSlic, SlicHip

if not is_supported_arch(ARCH_NAME):   # pragma: no cover
    raise ImportError("fast_slic_amd is not configured with hip/gfx950 support")

Slic = SlicHip
