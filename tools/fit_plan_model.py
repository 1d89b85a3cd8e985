#!/usr/bin/env python
"""Fit the (eff, ov) constants of the tiles of csrc/conv_igemm.hip: plan_conv on the sweeps tools/conv_plan_sweep.py wrote
(gpurun_out/plan_sweep*.json). Python replica of plan_conv; a coordinate search over the constants of the tiles named in FIT_TILES with the
others fixed; prints the loss of the chosen plans against the per-shape best (sum of microseconds) before and after."""
import itertools, json, math, os, sys

TILES = {  # id: [bm, bn, occ, eff, ov] - the constants of csrc/conv_igemm.hip: plan_conv as they stand (tests/test_abi_and_surface.py keeps the two in step)
    "A": [128, 128, 2, 1.00, 1.0], "B": [64, 128, 3, 1.06, 0.5], "C": [128, 64, 3, 0.93, 0.5], "D": [64, 64, 5, 0.90, 1.0], "E": [128, 32, 4, 0.90, 0.5],
}
RENAME = dict(kv.split(":") for kv in os.environ.get("FIT_RENAME", "").split(",") if kv)   # e.g. G:A,H:D,I:B,J:C,K:E for sweeps that carried experimental letters
SPLITS = [1, 2, 3, 4, 6, 8]
WS = 128 << 20


def g(o):
    return 0.8 if o <= 1 else 0.85 if o == 2 else 0.96 if o == 3 else 1.0


def model_us(t, M, Cout, Cin, ntaps, k):
    bm, bn, occ, eff, ov = t
    nsteps = ntaps * (Cin // 32)
    K = float(ntaps * Cin)
    ntn = -(-Cout // bn)
    nb = -(-M // bm) * ntn
    wgs, slots = nb * k, 256 * occ
    full, rem = divmod(wgs, slots)
    tile_us = bm * bn * (-(-nsteps // k)) / (8000.0 * eff)
    us = full * occ * tile_us
    rounds = full
    if rem > 0:
        r = -(-rem // 256)
        us += r * tile_us * g(occ) / g(r)
        rounds += 1
    us = max(us, (M * K * 4.0 * ntn + Cout * K * 4.0 + M * Cout * 4.0) / 4.0e6)
    us += (rounds + 1) * ov * math.sqrt(bm * bn / 16384.0)
    if k > 1:
        us += 3.0 + (k + 1) * M * Cout * 4.0 / 2.0e6
    return us


def choose(tiles, shape, avail):
    M, Cout, Cin, ntaps = shape
    nsteps = ntaps * (Cin // 32)
    best, bu = None, 1e300
    for tid, t in tiles.items():
        if tid == "E" and Cout > 32:
            continue
        for k in SPLITS:
            if k > 1 and (nsteps // k < 8 or k * M * Cout * 4 > WS):
                continue
            if "%s%d" % (tid, k) not in avail:
                continue
            us = model_us(t, M, Cout, Cin, ntaps, k)
            if us < bu:
                bu, best = us, "%s%d" % (tid, k)
    return best


def loss(tiles, data):
    tot = best = 0.0
    for d in data:
        c = choose(tiles, d["shape"], d["us"])
        tot += d["us"][c]
        best += min(d["us"].values())
    return tot, best


if __name__ != "__main__":
    raise SystemExit        # imported for the replica only (tests): everything below is the fitting run
files = sys.argv[1:] or ["gpurun_out/plan_sweep_dma.json", "gpurun_out/plan_sweep_dma_s4.json"]
sets = [json.load(open(f)) for f in files]
if RENAME:
    for d in sets:
        for x in d:
            x["us"] = {RENAME[k[0]] + k[1:]: v for k, v in x["us"].items() if k[0] in RENAME}
fit = os.environ.get("FIT_TILES", "ABCDE")
for f, d in zip(files, sets):
    print("%-44s current constants: chosen %.1f us / best %.1f us" % (f, *loss(TILES, d)))
# weight the sets equally (relative loss)
def total(tiles):
    return sum(loss(tiles, d)[0] / loss(tiles, d)[1] for d in sets)
cur = {k: list(v) for k, v in TILES.items()}
for it in range(4):
    for tid in fit:
        bestv, bl = None, 1e300
        for eff, ov in itertools.product([1.0] if tid == "A" else [0.90 + 0.01 * i for i in range(36)], [0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0]):   # only ratios matter: A's eff stays 1
            cur[tid][3], cur[tid][4] = eff, ov
            l = total(cur)
            if l < bl - 1e-9:
                bl, bestv = l, (eff, ov)
        cur[tid][3], cur[tid][4] = bestv
    print("pass", it, {t: cur[t][3:] for t in fit}, "rel loss sum %.4f" % total(cur))
for f, d in zip(files, sets):
    print("%-44s fitted:   chosen %.1f us / best %.1f us" % (f, *loss(cur, d)))
    for x in d:
        c = choose(cur, x["shape"], x["us"])
        b = min(x["us"], key=x["us"].get)
        if x["us"][c] > 1.03 * x["us"][b]:
            print("   ", x["shape"], "chose", c, "%.1f" % x["us"][c], "best", b, "%.1f" % x["us"][b])
