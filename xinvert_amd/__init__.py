"""xinvert_amd -- MI355X-native SOR inversion engine behind the xinvert call boundary.

Layout (only what the hot path needs):
  csrc/          hand-written HIP kernels for gfx950 + the C-ABI (include/xinv.h)
  _lib.py        ctypes binding of libxinv_hip.so (no CPU fallback)
  core.py        inv_standard2D / inv_general2D / inv_standard3D / inv_general3D ...   (reference xinvert/core.py)
  apps.py        invert_Poisson / invert_Stommel / invert_GillMatsuno / invert_omega, cal_flow
                 (reference xinvert/apps.py)
  dist.py        batch-axis sharding across ranks (one process per GPU) + flags gather
  field.py       minimal labelled array standing in for xarray.DataArray
"""
from .field import Field                                           # noqa: F401
from .core import (inv_standard2D, inv_standard2D_test, inv_general2D, inv_general2D_bih,    # noqa: F401
                   inv_standard3D, inv_general3D)
from .apps import (invert_Poisson, invert_Stommel, invert_StommelMunk, invert_GillMatsuno,  # noqa: F401
                   invert_Fofonoff, invert_BrethertonHaidvogel, invert_omega, invert_3DOcean,
                   invert_RefState, invert_PV2D, invert_Eliassen, invert_GillMatsuno_test,
                   invert_Stommel_test, invert_StommelArons, invert_geostrophic,
                   animate_iteration, cal_flow, default_iParams, default_mParams)

__version__ = '0.1.0'
