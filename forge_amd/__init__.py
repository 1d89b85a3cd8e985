"""forge_amd — MI355X (gfx950) native implementation of the FORGE reconstruction hot path
(UT-Austin-RPL/FORGE: encoder lift -> voxel pose warp -> ConvGRU fusion -> volume render),
behind the reference's own nn.Module surface. See DESIGN.md / INTEGRATION.md.

    from forge_amd.model import FORGE                                   # models.model.FORGE
    from forge_amd.model_single_pose_estimator import FORGE_poseEstimator3D
"""
__version__ = "0.2.1"


def invalidate_packed(module):
    """Drop the packed-weight / folded-BatchNorm caches of every fused HIP inference path below `module`. Needed only after
    editing parameters through `.data` (which bypasses the version counters the caches key on); optimizer steps, load_state_dict,
    .to() and train()/eval() are tracked automatically."""
    from .convops import invalidate_packed as _inv
    _inv(module)


def install_reference_aliases():
    """Make `from models.model import FORGE` (the import lines of kubric_train_*.py, demo.py,
    kubric_eval.py) resolve to this package: registers forge_amd's modules under the reference's
    `models.*` names in sys.modules. Call once before importing the reference's entry scripts."""
    import importlib
    import sys
    import types
    pkg = types.ModuleType("models")
    pkg.__path__ = []
    sys.modules["models"] = pkg
    for name in ("model", "model_single_pose_estimator", "encoder", "fusion", "rotate", "volume_render",
                 "pose_estimator_3d", "pose_estimator_2d"):
        mod = importlib.import_module("forge_amd." + name)
        sys.modules["models." + name] = mod
        setattr(pkg, name, mod)
