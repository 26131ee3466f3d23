"""Attack-success-rate evaluation (ASR_fast.py:67-126, test.py:123-148) on the GPU.

``preprocess_image`` and ``compute_asr`` keep the reference signatures; files are decoded on the host
(PIL), everything after that -- Pillow-exact resize to 224, ToTensor, victim forward, argmax --
runs batched on the device.  ``evaluate_batch`` is the folder-free form the generation pipeline uses
(uint8 sampler output in, per-image predictions out).
"""
import json
import os

import numpy as np
import torch
from PIL import Image

from .imageops import preprocess_batch, resize_u8, to_tensor
from .metrics import argmax_rows

_EXTS = ("png", "jpg", "jpeg", "bmp", "gif")


def load_label_maps(config_path):
    """ASR_fast.py:67-75,99: (label_to_int, int_to_label) from an ``id2label`` JSON (config*.json)."""
    with open(config_path, "r") as f:
        id2label = json.load(f)["id2label"]
    label_to_int = {label: int(i) for i, label in id2label.items()}
    return label_to_int, {v: k for k, v in label_to_int.items()}


def preprocess_image(image_path, device="cuda"):
    """ASR_fast.py:90-97: open -> RGB -> Resize((224,224)) -> ToTensor -> [1,3,224,224] (no normalisation)."""
    arr = np.asarray(Image.open(image_path).convert("RGB"), dtype=np.uint8)
    t = torch.from_numpy(arr.copy()).to(device)[None]
    return to_tensor(resize_u8(t, 224, 224))


def _logits(model, x):
    out = model(x)
    return out.logits if hasattr(out, "logits") else out        # HF models (ASR_fast.py:114, commented)


def evaluate_batch(images_u8_nchw, model, size=224, mean=None, std=None, jpeg_quality=None):
    """uint8 [n,3,S,S] (GPU) -> int32 [n] predicted class indices.  ``jpeg_quality=75`` reproduces the ``.jpg``
    files the reference's generate() writes and ASR_fast.py reads back (bit-exact pixels, no file)."""
    x = preprocess_batch(images_u8_nchw, size, mean, std, jpeg_quality)
    return argmax_rows(_logits(model, x))


def compute_asr(folder_path, model, int_to_label, batch_size=64, device="cuda"):
    """ASR_fast.py:101-126: fraction of images whose predicted label differs from the label encoded
    in the file name (``name.rsplit('_', 1)[0]``)."""
    names = [f for f in os.listdir(folder_path) if f.lower().endswith(_EXTS)]
    total = len(names)
    successful = 0
    for i in range(0, total, batch_size):
        chunk = names[i:i + batch_size]
        xs = [preprocess_image(os.path.join(folder_path, f), device) for f in chunk]
        pred = argmax_rows(_logits(model, torch.cat(xs, 0))).cpu().tolist()
        for f, p in zip(chunk, pred):
            if int_to_label[p] != f.rsplit("_", 1)[0]:
                successful += 1
    print(total)
    print(successful)
    return successful / total
