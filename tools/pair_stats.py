"""How sparse is the blend work?  Fractions of (tile entry x pixel group) with any contributing pixel (oracle-side analysis)."""
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
tiles = rng.choice(out["ranges"].shape[0], size=200, replace=False)
tot = act = act_nc = 0; any_tile = any_strip = any_8x8 = any_4x4 = 0; n_e = 0
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
    a = (power <= 0) & (alpha >= 1 / 255.)
    inside = (px < W) & (py < H)
    a &= inside[None]
    nc = out["n_contrib"][ty*16:ty*16+16, tx*16:tx*16+16]
    ncp = np.zeros((16,16), np.int64); ncp[:nc.shape[0], :nc.shape[1]] = nc
    pos = np.arange(len(ids))[:, None, None]
    a_nc = a & (pos < ncp[None])
    n = len(ids); n_e += n
    tot += n * 256; act += a.sum(); act_nc += a_nc.sum()
    any_tile += a.reshape(n, -1).any(1).sum()
    any_strip += a.reshape(n, 4, 4, 16).any(axis=(2, 3)).sum()
    any_8x8 += a.reshape(n, 2, 8, 2, 8).any(axis=(2, 4)).sum()
    any_4x4 += a.reshape(n, 4, 4, 4, 4).any(axis=(2, 4)).sum()
print(name, "entries/tile %.0f" % (n_e / len(tiles)))
print("active pixel-pairs / all pairs      : %.3f  (within n_contrib: %.3f)" % (act / tot, act_nc / tot))
print("entries with any active px in tile   : %.3f" % (any_tile / n_e))
print("16x4 strips with any active px       : %.3f" % (any_strip / (4 * n_e)))
print("8x8 blocks with any active px        : %.3f" % (any_8x8 / (4 * n_e)))
print("4x4 blocks with any active px        : %.3f" % (any_4x4 / (16 * n_e)))
print("mean n_contrib %.1f  mean T %.3f" % (out["n_contrib"].mean(), out["out_T"].mean()))
