"""Checkpoint reading for generation (utils/checkpoint.py:21-117 of the reference).

Both on-disk formats load: the IDDM dictionary ``{"model", "ema_model", "num_classes",
"conditional", "image_size", "network", "act", ...}`` (utils/checkpoint.py:143-147) and a bare
``state_dict`` (diff_model.py:574).  Files are opened with ``weights_only=True``: nothing in a
checkpoint is executed.
"""
import logging
from collections import OrderedDict

import torch

logger = logging.getLogger(__name__)


def read_ckpt(ckpt_path, device="cpu"):
    return torch.load(ckpt_path, map_location=device, weights_only=True)


def load_model_ckpt(model, model_ckpt, is_train=False, is_pretrain=False, is_distributed=False):
    """Strip a DistributedDataParallel ``module.`` prefix and keep only tensors whose shape matches
    the model (utils/checkpoint.py:85-117: mismatched entries are silently dropped)."""
    if is_train:
        raise NotImplementedError("training-time checkpoint handling is outside the generation path")
    own = model.state_dict()
    renamed = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in model_ckpt.items()}
    kept = {k: v for k, v in renamed.items() if k in own and tuple(own[k].shape) == tuple(v.shape)}
    dropped = sorted(set(renamed) - set(kept))
    if dropped:
        logger.warning("checkpoint entries ignored (unknown name or shape mismatch): %s", dropped[:8])
    own.update(kept)
    model.load_state_dict(OrderedDict(own))


def load_ckpt(ckpt_path, model, device, optimizer=None, is_train=False, is_pretrain=False, is_distributed=False,
              is_use_ema=False):
    """utils/checkpoint.py:21-68 for ``is_train=False``: pick ``ema_model`` or ``model``."""
    state = read_ckpt(ckpt_path, device)
    if not (isinstance(state, dict) and ("model" in state or "ema_model" in state)):
        load_model_ckpt(model, state)                         # bare state_dict
        return
    assert state.get("model") is not None or state.get("ema_model") is not None, \
        "Error!! Checkpoint model and ema_model are not None. Please check checkpoint's structure."
    if state.get("model") is None or (is_use_ema and state.get("ema_model") is not None):
        weights = state["ema_model"]
    else:
        weights = state["model"]
    load_model_ckpt(model, weights, is_train=is_train, is_pretrain=is_pretrain, is_distributed=is_distributed)
    logger.info("[%s]: Successfully load model checkpoint.", device)
