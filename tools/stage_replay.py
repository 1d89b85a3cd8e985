#!/usr/bin/env python
"""Per-stage time of the b-scene inference step as the hipGraph replays it: every stage (ResNet trunk, conv1, rotate, fusion, heads,
ray-march + conv_rgb) is captured into its own hipGraph on static inputs and its replay is timed (no host launch gaps, unlike the HIP-event
split of an eager pass in bench.py's stages_ms). STAGE_SCENES=b."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from forge_amd import synthetic as syn  # noqa: E402
from forge_amd.flopmeter import stage_replay_ms  # noqa: E402
from forge_amd.model import FORGE  # noqa: E402


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    b = int(os.environ.get("STAGE_SCENES", "1"))
    model = FORGE(syn.kubric_config())
    model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
    model = model.to(dev).eval()
    sample = {k: v.to(dev) for k, v in syn.make_sample(b, 5, 256, 1.5, seed=1000).items()}
    res = stage_replay_ms(model, sample, dev)
    print("scenes %d  " % b + "  ".join("%s %.3f" % kv for kv in res.items()) + "  sum %.3f ms" % sum(res.values()))
