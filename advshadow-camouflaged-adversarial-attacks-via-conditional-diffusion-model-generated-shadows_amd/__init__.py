"""MI355X-native AdvShadow hot path: DDIM reverse sampling through the UNet eps-predictor,
shadow composite, victim forward and the ASR/PSNR/SSIM reduction, as hand-written gfx950 HIP
kernels behind the reference's own Python entry points.

The directory name (fixed by the build contract) is not a valid Python identifier; import the
package through the ``advshadow_amd`` alias module at the repository root.
"""
from . import _lib  # noqa: F401
from ._lib import AdvsError, LIB_PATH  # noqa: F401

__all__ = ["AdvsError", "LIB_PATH"]
