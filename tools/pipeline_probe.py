#!/usr/bin/env python
"""Throughput of the b = 1 inference step with TWO steps in flight: two hipGraphs of the same model (separate static buffers) replayed
alternately on two HIP streams, so that the under-filling ResNet-trunk launches of scene n+1 can share the chip with the ConvGRU launches of
scene n. Compares against back-to-back replays on one stream. PIPE_DEPTH=2|3, PIPE_STEPS=40."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from forge_amd import synthetic as syn  # noqa: E402
from forge_amd.graph import GraphedForward  # noqa: E402
from forge_amd.model import FORGE  # noqa: E402

dev = torch.device("cuda:0")
depth = int(os.environ.get("PIPE_DEPTH", "2"))
scenes = int(os.environ.get("PIPE_SCENES", "1"))
steps = int(os.environ.get("PIPE_STEPS", "40"))
cfg = syn.kubric_config()
model = FORGE(cfg)
model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
model = model.to(dev).eval()
ds = syn.SyntheticDataset(1.5)
samples = [{k: v.to(dev) for k, v in syn.make_sample(scenes, 5, 256, 1.5, seed=1000 + i).items()} for i in range(depth)]
from forge_amd import convops as co  # noqa: E402
wt = os.environ.get("PIPE_WINO_TILE")                      # experiment: pin the tile of the Winograd point-GEMM launches (A..D)
if wt:
    _orig_tile = co.wino_gemm_tile
    co.wino_gemm_tile = lambda R, Cout, Cin: wt if (os.environ.get("PIPE_WINO_TILE_N", "") in ("", str(Cout))) else _orig_tile(R, Cout, Cin)
rule = os.environ.get("PIPE_WINO_RULE")                    # experiment: a Python expression of (R, Cout, Cin) naming the tile, e.g. "'I' if Cout >= 256 else 'A'"
if rule:
    co.wino_gemm_tile = eval("lambda R, Cout, Cin: " + rule)
tmap = dict(kv.split(":") for kv in os.environ.get("PIPE_TILE_MAP", "").split(",") if kv)   # experiment: rename planned tiles, e.g. "D:H,B:I"
if tmap:
    _orig_plan = co.conv_plan

    def _mapped_plan(*a, **k):
        t, ks = _orig_plan(*a, **k)
        return tmap.get(t, t), ks
    co.conv_plan = _mapped_plan
ft = os.environ.get("PIPE_FORCE_TILE")                     # experiment: pin the tile of EVERY conv_igemm launch (and no split-K)
if ft:
    co.STATE.plan_override = (ft, 1)
graphs = [GraphedForward(model, s, ds, dev) for s in samples]
co.STATE.plan_override = None
streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
ref = [g(s)[0].clone() for g, s in zip(graphs, samples)]
torch.cuda.synchronize()


def run_seq(n):
    for i in range(n):
        graphs[i % depth](samples[i % depth])


def run_pipe(n):
    for i in range(n):
        k = i % depth
        with torch.cuda.stream(streams[k]):
            graphs[k](samples[k])


for name, fn in (("one stream, back to back", run_seq), ("%d streams, %d steps in flight" % (depth, depth), run_pipe), ("one stream, back to back", run_seq)):
    fn(2 * depth)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("scenes %d  %-32s %.3f ms/step  %.1f views/s" % (scenes, name, dt / steps * 1e3, scenes * 5 * steps / dt))
run_pipe(depth)
torch.cuda.synchronize()
print("pipelined outputs equal the sequential ones:", all(torch.equal(g.static_out[0], r) for g, r in zip(graphs, ref)))
