"""MI355X-native SLIC superpixels behind the fast_slic surface (fast_slic/__init__.py:1-4)."""
from .base_slic import BaseSlic, SlicHip, LSCHip, SlicRealDistHip, SlicRealDistL2Hip, SlicRealDistNoQHip
from ._binding import (SlicModel, Engine, get_supported_archs, is_supported_arch, enforce_connectivity,
                       make_params, device_count, CLUSTER_DTYPE, ARCH_NAME)

Slic = SlicHip
LSC = LSCHip
SlicRealDist = SlicRealDistHip
SlicRealDistL2 = SlicRealDistL2Hip
SlicRealDistNoQ = SlicRealDistNoQHip
supported_archs = tuple(get_supported_archs())
