"""Spatial partitioning experiment: the pipelined b = 1 step with each HIP stream restricted to a subset of the CUs (hipExtStreamCreateWithCUMask),
so that every step in flight owns a slice of the chip instead of time-sharing all of it. CUMASK_MODE = none | halves | quarters | interleave2 | interleave4."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from forge_amd import synthetic as syn
from forge_amd.graph import GraphedForward
from forge_amd.model import FORGE

dev = torch.device("cuda:0")
mode = os.environ.get("CUMASK_MODE", "halves")
steps = int(os.environ.get("PIPE_STEPS", "40"))
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
NCU = torch.cuda.get_device_properties(dev).multi_processor_count


def masked_stream(bits):
    words = (NCU + 31) // 32
    arr = (ctypes.c_uint32 * words)()
    for b in bits:
        arr[b // 32] |= (1 << (b % 32))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


if mode == "none":
    depth, masks = int(os.environ.get("PIPE_DEPTH", "4")), None
elif mode == "halves":
    depth, masks = 2, [range(0, NCU // 2), range(NCU // 2, NCU)]
elif mode == "quarters":
    depth, masks = 4, [range(i * NCU // 4, (i + 1) * NCU // 4) for i in range(4)]
elif mode == "interleave2":
    depth, masks = 2, [range(0, NCU, 2), range(1, NCU, 2)]
elif mode == "interleave4":
    depth, masks = 4, [range(i, NCU, 4) for i in range(4)]
elif mode == "xcd2":          # bits b with (b % 8) in one half of the XCDs, if consecutive CU ids alternate XCDs
    depth, masks = 2, [[b for b in range(NCU) if (b % 8) < 4], [b for b in range(NCU) if (b % 8) >= 4]]
elif mode == "halves4":       # 4 steps in flight, two per half
    depth, masks = 4, [range(0, NCU // 2), range(NCU // 2, NCU), range(0, NCU // 2), range(NCU // 2, NCU)]
streams = [torch.cuda.Stream(device=dev) for _ in range(depth)] if masks is None else [masked_stream(list(m)) for m in masks]
cfg = syn.kubric_config()
model = FORGE(cfg)
model.load_state_dict(syn.seeded_state_dict(model.state_dict(), 0))
model = model.to(dev).eval()
ds = syn.SyntheticDataset(1.5)
samples = [{k: v.to(dev) for k, v in syn.make_sample(1, 5, 256, 1.5, seed=1000 + i).items()} for i in range(depth)]
graphs = [GraphedForward(model, s, ds, dev) for s in samples]
ref = [g(s)[0].clone() for g, s in zip(graphs, samples)]
torch.cuda.synchronize()


def run(n):
    for i in range(n):
        k = i % depth
        with torch.cuda.stream(streams[k]):
            graphs[k](samples[k])


run(2 * depth)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ok = all(torch.equal(g.static_out[0], r) for g, r in zip(graphs, ref))
print("CUs %d mode %-12s depth %d: %.3f ms/step  %.1f views/s  outputs equal: %s" % (NCU, mode, depth, dt / steps * 1e3, 5 * steps / dt, ok))
