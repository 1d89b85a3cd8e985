"""f4 — checkpoint files in the reference's layout.

The reference writes `{"epoch", "state_dict", "optimizer", "best_psnr", "best_rot"}` with torch.save (utils/train_utils.py:167-169,
called from the trainers) — the state_dict keys carry DistributedDataParallel's `module.` prefix — and reads them back in two ways:
whole-model resume (utils/exp_utils.py:152-182, strict) and stage hand-over, where only the `encoder_3d.` / `rotate.` / `render.`
sub-trees of a GT-pose checkpoint are loaded into a model that has additional pose networks (utils/exp_utils.py:185-216, strict per
sub-module). Same function names and arguments here, so the trainers can switch imports; loading goes through `load_state_dict`,
which also drops the packed-weight caches of the HIP inference path (convops.PackedModule).
"""
import os
import warnings

import torch


def save_checkpoint(state, checkpoint="checkpoint", filename="checkpoint.pth.tar"):
    """utils/train_utils.py:167-169."""
    os.makedirs(checkpoint, exist_ok=True)
    path = os.path.join(checkpoint, filename)
    torch.save(state, path)
    return path


def _read(path, device):
    if not os.path.isfile(path):
        raise ValueError("=> no checkpoint found at '{}'".format(path))
    ckpt = torch.load(path, map_location=device if device is not None else torch.device("cpu"), weights_only=False)
    sd = ckpt["state_dict"]
    if sd and next(iter(sd)).startswith("module."):                 # written from a DistributedDataParallel wrapper
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    return ckpt, sd


def resume_training(model, optimizer, output_dir, cpt_name="cpt_last.pth.tar", strict=True, device=None):
    """utils/exp_utils.py:152-182 -> (model, optimizer, start_epoch, best_psnr, best_rot)."""
    ckpt, sd = _read(os.path.join(output_dir, cpt_name), device)
    missing = set(model.state_dict().keys()) - set(sd.keys())
    if missing:
        warnings.warn("checkpoint lacks %d of the model's tensors, e.g. %s" % (len(missing), sorted(missing)[:4]))
    model.load_state_dict(sd, strict=strict)
    if optimizer is not None and "optimizer" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer"])
    return model, optimizer, ckpt.get("epoch", 0), ckpt.get("best_psnr", 0.0), ckpt.get("best_rot", float("inf"))


def load_encoder_pretrained(model, resume_root, cpt_name="cpt_last.pth.tar", strict=True, device=None):
    """utils/exp_utils.py:185-216: the reconstruction sub-trees (`encoder_3d`, `rotate`, `render`) of a checkpoint into `model`,
    each with strict key matching; the pose networks of `model` keep their weights."""
    _, sd = _read(os.path.join(resume_root, cpt_name), device)
    for name in ("rotate", "encoder_3d", "render"):
        prefix = name + "."
        sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        target = getattr(model, name)
        if len(sub) != len(target.state_dict()):
            warnings.warn("%s: checkpoint holds %d tensors, the module %d" % (name, len(sub), len(target.state_dict())))
        target.load_state_dict(sub, strict=strict)
    return model
