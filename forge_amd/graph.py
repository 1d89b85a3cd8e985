"""hipGraph capture of the fixed-shape inference step.

The hot path at b=1 is ~90 kernel launches of 20-1000 us each; launched eagerly from Python the host becomes
the bottleneck for the short ones (ResNet layers, heads, conv_rgb). The MI355X-native answer is a HIP graph:
capture `model(sample, dataset, device)` once (eval mode, no autograd, static shapes and buffers) and replay
it per step. Everything in the forward is capture-safe: no host synchronisation, no host->device copies, all
launches go to torch's current stream (the capture stream), outputs live in the graph's private pool.
"""
import torch


class GraphedForward:
    """Callable replaying a captured `model(sample, dataset, device)`.

    g = GraphedForward(model, sample, dataset, device); outs = g(new_sample)
    `new_sample` tensors are copied into the static input buffers (device-to-device when already resident);
    the returned tensors are the graph's static outputs (overwritten by the next replay).
    """

    def __init__(self, model, sample, dataset, device, warmup=3):
        if model.training:
            raise RuntimeError("GraphedForward captures the inference step: call model.eval() first")
        self.model, self.dataset, self.device = model, dataset, device
        self.static_in = {k: (v.to(device).clone() if torch.is_tensor(v) else v) for k, v in sample.items()}
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(warmup):                       # weight packing, allocator warm-up
                model(self.static_in, dataset, device)
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: only calls made by THIS thread are checked against the capture; the RCCL watchdog thread of a live
        # process group may keep polling its events while we capture
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.static_out = model(self.static_in, dataset, device)

    def __call__(self, sample=None):
        if sample is not None:
            for k, v in sample.items():
                if torch.is_tensor(v) and v is not self.static_in[k]:
                    self.static_in[k].copy_(v, non_blocking=True)
                    if v.is_cuda:                         # the copy reads v on the CURRENT stream (PipelinedForward: a private one): the caching
                        v.record_stream(torch.cuda.current_stream(self.device))   # allocator must not hand v's block out again before it ran
        self.graph.replay()
        return self.static_out


class PipelinedForward:
    """`depth` steps in flight: `depth` hipGraphs of the same model (shared weights, separate static input / output buffers and private
    pools) replayed round-robin on `depth` HIP streams.

    Why: at one scene per step the ResNet trunk is ~60 launches of 320-1280 workgroups with 2-64 K-steps each - a third of the step's
    time at 0.4 of the matrix-core peak, bound by per-launch ramp / prologue / epilogue phases in which most CUs idle (every workgroup of
    such a launch is co-resident, so the phases do not overlap inside a launch) - while the ConvGRU launches of the same step are
    8192-workgroup, MFMA-bound kernels. With the next scene's step in flight on a second hardware queue the dispatcher fills the trunk's
    idle slots with the previous scene's GEMM workgroups: measured 8.19 -> 7.35 ms per step (2 in flight) -> 7.02 ms (3), bit-identical
    outputs (tools/pipeline_probe.py). This is a throughput device: the latency of one step stays that of a single replay.

        p = PipelinedForward(model, sample, dataset, device, depth=3)
        for s in samples: out = p(s)        # returns the static outputs of the slot used; valid ONLY after p.wait(), overwritten `depth` calls later
        p.wait()

    The sample's device tensors are read by a copy on the slot's private stream (ordered after the caller's stream; `record_stream` keeps their
    memory from being reused before the copy ran): the caller may drop or let the allocator recycle a sample right after p(sample), but must
    not overwrite it IN PLACE before p.wait() (or an event on the slot's stream) - the copy may still be pending.
    """

    def __init__(self, model, sample, dataset, device, depth=3, warmup=3):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.device, self.depth = device, depth
        self.slots = [GraphedForward(model, sample, dataset, device, warmup=warmup if i == 0 else 1) for i in range(depth)]
        self.streams = [torch.cuda.Stream(device=device) for _ in range(depth)]
        self.calls = 0

    def __call__(self, sample=None):
        k = self.calls % self.depth
        self.calls += 1
        st = self.streams[k]
        st.wait_stream(torch.cuda.current_stream(self.device))         # inputs written on the caller's stream are visible to the replay
        with torch.cuda.stream(st):
            return self.slots[k](sample)

    def wait(self):
        """Make the caller's stream wait for every step in flight (then the returned static outputs may be read on it)."""
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            cur.wait_stream(st)


class GraphedCall:
    """hipGraph capture of an arbitrary fixed-shape, capture-safe inference callable `fn()` whose inputs are tensors the caller keeps
    alive and overwrites in place between replays (e.g. FORGE.reconstruct on resident feature volumes). g = GraphedCall(fn, device);
    out = g() replays and returns the static outputs."""

    def __init__(self, fn, device, warmup=3):
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.static_out = fn()

    def __call__(self):
        self.graph.replay()
        return self.static_out


class GraphedStep:
    """hipGraph capture of a whole fixed-shape optimisation step: `step_fn()` must run forward, loss, backward and the optimizer step on
    STATIC tensors (inputs that are overwritten in place between replays, parameters, a `capturable=True` optimizer) and return the
    loss tensor. The eager training step of the b = 1 scene configuration is host-bound (~600 kernel launches from Python per step);
    the replay is not.

        opt = torch.optim.Adam(params, lr=..., capturable=True)
        def step_fn():
            loss = loss_of(model(static_sample, dataset, device))
            loss.backward(); torch.nn.utils.clip_grad_norm_(params, 10.0); opt.step()
            return loss.detach()
        g = GraphedStep(step_fn, opt)          # runs `warmup` eager steps (kernel loading, weight packing, optimizer state), then captures
        for batch in loader:
            for k, v in batch.items(): static_sample[k].copy_(v, non_blocking=True)
            loss = g()                         # one replay = one optimisation step

    Everything inside must be capture-safe: no host synchronisation (`.item()`, `torch.inverse`, `torch.tensor(..., device=cuda)`), no
    host->device copies. BatchNorm running statistics and Adam's step counters are device tensors updated in place by the replay."""

    def __init__(self, step_fn, optimizer, warmup=3):
        self.optimizer = optimizer
        for _ in range(warmup):
            optimizer.zero_grad(set_to_none=True)
            step_fn()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)              # gradients are (re)allocated inside the graph's private pool
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss = step_fn()

    def __call__(self):
        self.graph.replay()
        return self.loss
