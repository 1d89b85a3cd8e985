"""TEST INFRASTRUCTURE — import the reference's own hot-path modules in THIS container.

Puts `oracle/shims` (third-party stand-ins) and `/root/reference` on sys.path and returns the
reference modules. Only `oracle/make_golden.py` and ad-hoc validation use this; it cannot run
on the GPU box (no /root/reference there) and nothing in forge_amd/, bench.py or the gpu tests
imports it.
"""
import importlib
import os
import sys

REFERENCE_ROOT = os.environ.get("FORGE_REFERENCE_ROOT", "/root/reference")
SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models"))


def import_reference():
    """Returns a dict of reference modules (models.rotate, models.fusion, ...)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    import numpy as np
    if not hasattr(np, "float"):
        np.float = float            # models/model_utils.py:45 uses the removed alias
    for p in (REFERENCE_ROOT, SHIMS):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REFERENCE_ROOT)
    sys.path.insert(0, SHIMS)
    names = ["models.fusion", "models.rotate", "models.encoder", "models.volume_render",
             "models.model_single_pose_estimator", "models.pose_estimator_3d", "utils.geo_utils"]
    mods = {n: importlib.import_module(n) for n in names}
    try:  # models.model pulls pose_estimator_2d which downloads weights in its constructor
        pe2d = importlib.import_module("models.pose_estimator_2d")
        import torchvision.models as tvm
        pe2d.model_zoo.load_url = lambda url, **kw: tvm.ResNet(tvm.Bottleneck, [3, 4, 6, 3]).state_dict()
        mods["models.model"] = importlib.import_module("models.model")
    except Exception as e:  # pragma: no cover - joint model is a "next" row
        mods["models.model"] = None
        mods["models.model.error"] = repr(e)
    return mods


def kubric_config(img_size=256, volume_size=1.0, n_pts_per_ray=64, min_depth=0.5, max_depth=2.0,
                  use_gt_pose=True, parameter="all", dataset_name="kubric"):
    """config/kubric/gt_pose.yaml:10-60 as an attribute-style object."""
    from easydict import EasyDict
    return EasyDict({
        "dataset": {"name": dataset_name, "img_size": img_size, "num_frame": 5},
        "network": {"padding_mode": "zeros", "rot_representation": "quat",
                    "scale_rotate": 0.01, "scale_translate": 0.01, "backbone": "resnet"},
        "render": {"n_pts_per_ray": n_pts_per_ray, "volume_size": volume_size, "min_depth": min_depth,
                   "max_depth": max_depth, "camera_z": 1.5, "k_size": 5},
        "train": {"use_gt_pose": use_gt_pose, "canonicalize": True, "parameter": parameter},
    })
