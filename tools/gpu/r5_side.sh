cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | grep -v Warn | tail -12
import json, sys, torch
sys.path.insert(0, ".")
import bench
from forge_amd.model import FORGE
dev = torch.device("cuda:0")
for flag in (True, False, True, False):
    FORGE.pose2d_side_stream = flag
    for e in bench.joint_configs(dev, steps=6)[:1]:
        print("side stream", flag, e.get("name"), "eager %.2f" % e.get("ms_per_step", -1), "graph", (e.get("hipgraph_replay") or {}).get("ms_per_step"), (e.get("hipgraph_replay") or {}).get("error"))
PY
timeout 900 python -m pytest tests -m gpu -q -x -k "joint_training_step or joint_mode or joint_finetune or forge_joint_forward" 2>&1 | tail -3
