from .ddim import DDIMDiffusion  # noqa: F401
