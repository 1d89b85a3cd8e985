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


def import_reference_eval():
    """kubric_eval.py of the reference (do_refinement, kubric_eval.py:412-530) importable in this container: its plotting / metric / dataset imports
    (skimage, cv2, imageio, lpips; dataset.kubric / dataset.gso pull torchvision.transforms.functional) are not installed and not needed by the
    refinement loop - they are replaced by inert stand-in modules for the duration of the import. Returns the module."""
    import importlib.abc
    import importlib.machinery
    import types
    import_reference()

    class _Stub(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            s_ = _Stub(self.__name__ + "." + k)
            setattr(self, k, s_)
            return s_

        def __call__(self, *a, **kw):
            return _Stub("call")

    class _AutoStub(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, name, path, target=None):
            if name.split(".")[0] in ("skimage", "lpips", "cv2", "imageio", "matplotlib"):
                return importlib.machinery.ModuleSpec(name, self, is_package=True)
            return None

        def create_module(self, spec):
            mod = _Stub(spec.name)
            mod.__path__ = []
            return mod

        def exec_module(self, module):
            pass
    finder = _AutoStub()
    sys.meta_path.append(finder)
    for name, cls in (("dataset.kubric", "Kubric"), ("dataset.gso", "GSO")):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            setattr(mod, cls, type(cls, (), {}))
            sys.modules[name] = mod
    try:
        return importlib.import_module("kubric_eval")
    finally:
        sys.meta_path.remove(finder)
