"""Checkpoint helpers behind the reference's names (yolov6/utils/checkpoint.py:11-60).

A YOLOv6 checkpoint is a pickled dict of module OBJECTS (core/engine.py:192-200 writes
`{'model': deepcopy(model).half(), 'ema': ..., 'optimizer': ..., 'epoch': ...}`), so the class paths
`yolov6.models.yolo.Model`, `yolov6.layers.common.RepVGGBlock`, ... are part of the ABI.  After
`yolov6_amd.install_as_yolov6()` those paths resolve to this package's classes and a checkpoint written by the
reference un-pickles straight onto the HIP path."""
import os
import shutil

import torch

from .torch_utils import fuse_model

_STRIPPED_KEYS = ("optimizer", "ema", "updates")


def _read(path, map_location=None):
    # module pickles cannot satisfy torch >= 2.6's weights_only default
    return torch.load(path, map_location=map_location, weights_only=False)


def _module_of(ckpt):
    return ckpt["ema"] if ckpt.get("ema") else ckpt["model"]


def load_state_dict(weights, model, map_location=None):
    """Copy every tensor of the checkpoint's `model` whose name and shape match into `model` (the rest keeps its
    initialisation) - how the reference fine-tunes from pretrained weights."""
    source = _read(weights, map_location)["model"].float().state_dict()
    own = model.state_dict()
    model.load_state_dict({k: v for k, v in source.items() if k in own and own[k].shape == v.shape}, strict=False)
    return model


def load_checkpoint(weights, map_location=None, inplace=True, fuse=True):
    """The checkpoint's EMA model (else its plain model) as an fp32 eval-mode module, BatchNorm folded when `fuse`."""
    model = _module_of(_read(weights, map_location)).float()
    if fuse:
        model = fuse_model(model)
    return model.eval()


def save_checkpoint(ckpt, is_best, save_dir, model_name=""):
    """Write `<save_dir>/<model_name>.pt`; the best one is duplicated as best_ckpt.pt."""
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, f"{model_name}.pt")
    torch.save(ckpt, path)
    if is_best:
        shutil.copyfile(path, os.path.join(save_dir, "best_ckpt.pt"))


def strip_optimizer(ckpt_dir, epoch):
    """Shrink best_ckpt.pt / last_ckpt.pt for release: EMA weights become the model, optimizer state is dropped,
    parameters are frozen fp16."""
    for tag in ("best", "last"):
        path = os.path.join(ckpt_dir, f"{tag}_ckpt.pt")
        if not os.path.exists(path):
            continue
        ckpt = _read(path, torch.device("cpu"))
        ckpt["model"] = _module_of(ckpt)
        ckpt.update({k: None for k in _STRIPPED_KEYS}, epoch=epoch)
        ckpt["model"].half().requires_grad_(False)
        torch.save(ckpt, path)
