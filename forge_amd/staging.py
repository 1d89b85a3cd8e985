"""f4 — host->device staging of a Kubric-schema sample dict (dataset/kubric.py:390-402).

The reference moves its inputs with one `.to(device)` per key per forward (models/model.py:50,91-95,117-119: seven pageable
copies, each a synchronous hipMemcpy). Here every hot-path tensor of the sample is packed into ONE pinned host buffer and moved
with ONE asynchronous copy on the current stream; the model then reads views of that single device buffer. Samples whose
tensors already live on the device (bench.py, hipGraph replay, a prefetching loader) pass through untouched.
"""
import torch

# float tensors FORGE.forward / FORGE_poseEstimator3D.forward read (everything else in the dict is left where it is)
HOT_KEYS = ("images", "fg_probabilities", "K_cv2", "cam_extrinsics_cv2_canonicalized", "cam_poses_cv2_canonicalized",
            "cam_extrinsics_cv2", "cam_poses_cv2", "cam_poses_rel_cv2")

_PINNED = {}        # device index -> (pinned host buffer, event recorded after the last copy out of it)


def stage_sample(sample, device, keys=HOT_KEYS):
    """Returns a dict with the same keys as `sample` whose hot-path tensors are float32 tensors on `device`.
    Host tensors travel through one pinned buffer and one async copy (a fresh device buffer per call: tensors of an earlier
    forward that autograd still holds are never overwritten)."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    todo = [k for k in keys if k in sample and torch.is_tensor(sample[k]) and
            (sample[k].device != device or sample[k].dtype != torch.float32)]
    if not todo:
        return sample
    out = dict(sample)
    host = [k for k in todo if sample[k].device.type == "cpu"]
    for k in todo:                                   # device-resident tensors of another dtype / device: a plain conversion
        if k not in host:
            out[k] = sample[k].to(device=device, dtype=torch.float32)
    if not host:
        return out
    if device.type != "cuda":
        for k in host:
            out[k] = sample[k].to(device=device, dtype=torch.float32)
        return out
    sizes = [sample[k].numel() for k in host]
    offs, total = [], 0
    for n in sizes:                                  # 64-float (256-byte) aligned segments
        offs.append(total)
        total += (n + 63) // 64 * 64
    idx = device.index if device.index is not None else torch.cuda.current_device()
    buf, ev = _PINNED.get(idx, (None, None))
    if ev is not None:
        ev.synchronize()                             # the previous copy out of the pinned buffer has finished
    if buf is None or buf.numel() < total:
        buf = torch.empty(total, dtype=torch.float32).pin_memory()
    for k, o, n in zip(host, offs, sizes):
        buf[o:o + n].copy_(sample[k].reshape(-1))    # host-side pack (converts other float dtypes to fp32)
    with torch.cuda.device(idx):
        dev_buf = torch.empty(total, dtype=torch.float32, device=device)
        dev_buf.copy_(buf[:total], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
    _PINNED[idx] = (buf, ev)
    for k, o, n in zip(host, offs, sizes):
        out[k] = dev_buf[o:o + n].view(sample[k].shape)
    return out
