from .ddim import DDIMDiffusion  # noqa: F401
from .ddpm import DDPMDiffusion  # noqa: F401
from .plms import PLMSDiffusion  # noqa: F401
