"""Drop-in for ``tools/generate.py::generate(args)`` (tools/generate.py:26-88) on MI355X.

Reads the same ``args`` attributes (weight_path, conditional, network, image_size, num_classes, act,
generate_name, sample, num_images, use_ema, image_format, result_path, class_name, cfg_scale) and
writes the same files under ``os.path.join(args.result_path, str(time.time()))``.  Optional extra
attributes: ``compute_dtype`` ("fp32" | "bf16"), ``sample_steps`` (DDIM steps, reference default
500), ``x_T`` (injected start noise), ``seed``.
"""
import argparse
import logging
import os
import time

import torch

from ..utils.checkpoint import load_ckpt
from ..utils.initializer import device_initializer, generate_initializer, network_initializer, sample_initializer
from ..utils.utils import check_and_create_dir, save_images, save_one_image_in_images

logger = logging.getLogger(__name__)


def generate(args):
    logger.info("Start generation.")
    device = device_initializer()
    conditional, network, image_size, num_classes, act = generate_initializer(args.weight_path, args, device)
    result_path = os.path.join(args.result_path, str(time.time()))
    check_and_create_dir(result_path)
    Network = network_initializer(network, device)
    diffusion = sample_initializer(args.sample, image_size, device)
    steps = getattr(args, "sample_steps", None)
    if steps and hasattr(diffusion, "sample_steps"):
        diffusion = type(diffusion)(sample_steps=steps, img_size=image_size, device=device)
    expand = args.image_size if image_size != args.image_size else None
    dtype = getattr(args, "compute_dtype", "fp32")
    if getattr(args, "seed", None) is not None:
        torch.manual_seed(args.seed)
    x_T = getattr(args, "x_T", None)
    num_images = args.num_images
    if conditional:
        model = Network(num_classes=num_classes, device=device, image_size=image_size, act=act, compute_dtype=dtype).to(device)
        load_ckpt(args.weight_path, model, device, is_train=False, is_use_ema=args.use_ema)
        if args.class_name == -1:
            y = torch.arange(num_classes).long().to(device)
            num_images = num_classes
        else:
            y = torch.Tensor([args.class_name] * num_images).long().to(device)
        x = diffusion.sample(model=model, n=num_images, labels=y, cfg_scale=args.cfg_scale, x_T=x_T)
    else:
        model = Network(device=device, image_size=image_size, act=act, compute_dtype=dtype).to(device)
        load_ckpt(args.weight_path, model, device, is_train=False)
        x = diffusion.sample(model=model, n=num_images, x_T=x_T)
    save_images(x, os.path.join(result_path, f"{args.generate_name}.{args.image_format}"))
    save_one_image_in_images(x, result_path, args.generate_name, image_size=expand, image_format=args.image_format)
    logger.info("Finish generation.")
    return result_path


def build_parser():
    """Same flags as tools/generate.py:92-155 (``type=bool`` kept: any non-empty string is True)."""
    p = argparse.ArgumentParser()
    p.add_argument("--conditional", type=bool, default=True)
    p.add_argument("--generate_name", type=str, default="df")
    p.add_argument("--image_size", type=int, default=64)
    p.add_argument("--image_format", type=str, default="jpg")
    p.add_argument("--num_images", type=int, default=1)
    p.add_argument("--use_ema", type=bool, default=True)
    p.add_argument("--weight_path", type=str, required=True)
    p.add_argument("--result_path", type=str, default="results/vis")
    p.add_argument("--sample", type=str, default="ddpm")
    p.add_argument("--network", type=str, default="unet")
    p.add_argument("--act", type=str, default="silu")
    p.add_argument("--num_classes", type=int, default=10)
    p.add_argument("--class_name", type=int, default=0)
    p.add_argument("--cfg_scale", type=int, default=3)
    p.add_argument("--compute_dtype", type=str, default="fp32")
    p.add_argument("--sample_steps", type=int, default=None)
    return p


if __name__ == "__main__":
    generate(build_parser().parse_args())
