"""Would per-row (8x2 pixel = one 16-lane DPP row) queues pay?  For sampled tiles: per 8x8 block, the number of list
entries with an active pixel (within n_contrib) in the block vs the longest of its four 8x2 row queues / 4x4 / 8x4 groups."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fdgs import synth
from oracle import pyoracle
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
sc = synth.make_scene(synth.CONFIGS[name], seed=0)
o = pyoracle.Oracle(sc); out = o.forward()
W, H = sc["W"], sc["H"]; gx = (W + 15) // 16
rng = np.random.default_rng(0)
tiles = rng.choice(out["ranges"].shape[0], size=150, replace=False)
S = dict(blk=0, r8x2=0, r8x2_sum=0, q4x4=0, q4x4_sum=0, h8x4=0, h8x4_sum=0, fwd_blk=0, fwd_r=0)
for t in tiles:
    r0, r1 = out["ranges"][t]
    if r1 <= r0: continue
    ids = out["point_list"][r0:r1]
    tx, ty = t % gx, t // gx
    px = (tx * 16 + np.arange(16))[None, :].repeat(16, 0).astype(np.float32)
    py = (ty * 16 + np.arange(16))[:, None].repeat(16, 1).astype(np.float32)
    xy = out["means2D"][ids]; co = out["conic_opacity"][ids]
    dx = xy[:, 0, None, None] - px[None]; dy = xy[:, 1, None, None] - py[None]
    power = -0.5 * (co[:, 0, None, None] * dx * dx + co[:, 2, None, None] * dy * dy) - co[:, 1, None, None] * dx * dy
    alpha = np.minimum(0.99, co[:, 3, None, None] * np.exp(power))
    a = (power <= 0) & (alpha >= 1 / 255.) & ((px < W) & (py < H))[None]
    nc = out["n_contrib"][ty*16:ty*16+16, tx*16:tx*16+16]
    ncp = np.zeros((16, 16), np.int64); ncp[:nc.shape[0], :nc.shape[1]] = nc
    n = len(ids)
    for by in range(2):
        for bx in range(2):
            ab = a[:, by*8:by*8+8, bx*8:bx*8+8]                       # [n, 8, 8] cull-level activity (what the cull keeps)
            last = ncp[by*8:by*8+8, bx*8:bx*8+8].max()                # the wave walks entries [0, last)
            ab = ab[:last]
            if len(ab) == 0: continue
            S["blk"] += ab.reshape(len(ab), -1).any(1).sum()
            rows = ab.reshape(len(ab), 4, 2, 8).any(axis=(2, 3))      # [n, 4] 8x2 rows
            S["r8x2"] += rows.sum(0).max(); S["r8x2_sum"] += rows.sum()
            q = ab.reshape(len(ab), 2, 4, 2, 4).any(axis=(2, 4)).reshape(len(ab), 4)
            S["q4x4"] += q.sum(0).max(); S["q4x4_sum"] += q.sum()
            h = ab.reshape(len(ab), 2, 4, 8).any(axis=(2, 3))  # top / bottom 8x4 halves
            S["h8x4"] += h.sum(0).max(); S["h8x4_sum"] += h.sum()
print(name)
print("block-queue iterations (entries alive in the 8x8 block, up to the block's last contributor): %d" % S["blk"])
for k, lab in (("r8x2", "8x2 rows (DPP rows)"), ("q4x4", "4x4 quadrants"), ("h8x4", "8x4 halves")):
    print("%-22s longest-queue iterations %.3f of block iterations;  queue entries total %.2f x block" % (lab, S[k] / S["blk"], S[k + "_sum"] / S["blk"]))
