"""Initialisers ``generate()`` relies on (utils/initializer.py:29-63,81-96,158-175,216-256)."""
import logging

import torch

from ..model.networks.cspdarkunet import CSPDarkUnet
from ..model.networks.unet import UNet
from ..model.samples.ddim import DDIMDiffusion
from ..model.samples.ddpm import DDPMDiffusion
from ..model.samples.plms import PLMSDiffusion
from .checkpoint import read_ckpt

logger = logging.getLogger(__name__)


def device_initializer(device_id=0, is_train=False):
    """utils/initializer.py:29-63.  The HIP path has no CPU mode, so a missing GPU is an error."""
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: the MI355X path cannot fall back to the CPU")
    return torch.device("cuda", device_id)


def network_initializer(network, device):
    """utils/initializer.py:81-96 (unknown names fall back to unet, as in the reference)."""
    if network == "cspdarkunet":
        return CSPDarkUnet
    if network != "unet":
        logger.warning("[%s]: Setting network error, we has been automatically set to unet.", device)
    return UNet


def sample_initializer(sample, image_size, device):
    """utils/initializer.py:158-175 (unknown names fall back to ddpm, as in the reference)."""
    if sample == "ddim":
        return DDIMDiffusion(img_size=image_size, device=device)
    if sample == "plms":
        return PLMSDiffusion(img_size=image_size, device=device)
    if sample != "ddpm":
        logger.warning("[%s]: Setting sample error, we has been automatically set to ddpm.", device)
    return DDPMDiffusion(img_size=image_size, device=device)


def generate_initializer(ckpt_path, args, device):
    """utils/initializer.py:216-256: checkpoint metadata overrides the arguments when present."""
    state = read_ckpt(ckpt_path, "cpu")
    state = state if isinstance(state, dict) else {}

    def pick(name):
        v = state.get(name)
        return v if v is not None else getattr(args, name, None)

    return pick("conditional"), pick("network"), pick("image_size"), pick("num_classes"), pick("act")
