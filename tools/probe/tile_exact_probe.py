"""How many (tile, Gaussian) instances of the tile_cull lists could an EXACT per-tile reach test (block_reaches on the 16x16 tile
rectangle) drop?  C3 on the rig cameras.  (GPU; uses tests/util.py helpers)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from fdgs import synth
import util

dev = torch.device("cuda:0")
for pose in ("rig0", "rig2", "axis"):
    scene = synth.make_scene(synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"], seed=0, pose=pose)
    out, _ = util.run_hip(scene, dev, None, tile_cull=True)
    W, H = scene["W"], scene["H"]
    gx = (W + 15) // 16
    pl = torch.from_numpy(out["point_list"].astype(np.int64)).to(dev)
    tk = torch.from_numpy(out["tile_keys"].astype(np.int64)).to(dev)
    m2 = torch.from_numpy(out["means2D"]).to(dev)[pl]
    co = torch.from_numpy(out["conic_opacity"]).to(dev)[pl]
    A, B, Cc, op = co[:, 0], co[:, 1], co[:, 2], co[:, 3]

    def reaches(rx0, rx1, ry0, ry1):
        dx0, dx1, dy0, dy1 = rx0 - m2[:, 0], rx1 - m2[:, 0], ry0 - m2[:, 1], ry1 - m2[:, 1]
        xin, yin = (dx0 <= 0) & (dx1 >= 0), (dy0 <= 0) & (dy1 >= 0)
        tau = torch.log(255.0 * op) + 0.05
        ex = torch.where(dx0 > 0, dx0, dx1); ey = torch.where(dy0 > 0, dy0, dy1)
        dyv = torch.minimum(torch.maximum(-B * ex / Cc, dy0), dy1)
        qv = 0.5 * (A * ex * ex + Cc * dyv * dyv) + B * ex * dyv
        dxh = torch.minimum(torch.maximum(-B * ey / A, dx0), dx1)
        qh = 0.5 * (A * dxh * dxh + Cc * ey * ey) + B * dxh * ey
        qmin = torch.minimum(torch.where(xin, torch.full_like(qv, 3e38), qv), torch.where(yin, torch.full_like(qh, 3e38), qh))
        ok = (qmin <= tau) | (xin & yin)
        ok |= ~((A > 0) & (Cc > 0) & (A * Cc > 1.00001 * B * B))
        ok &= op >= 0.0039
        return ok
    tx, ty = (tk % gx).float() * 16, (tk // gx).float() * 16
    tile_ok = reaches(tx, tx + 15, ty, ty + 15)
    nblk = torch.zeros_like(tile_ok, dtype=torch.int32)
    for sy in (0, 8):
        for sx in (0, 8):
            nblk += reaches(tx + sx, tx + sx + 7, ty + sy, ty + sy + 7).int()
    n = tile_ok.numel()
    print("%s: instances %d, tile unreachable %.1f %%, zero blocks reached %.1f %%, mean blocks per instance %.2f" % (
        pose, n, 100.0 * float((~tile_ok).float().mean()), 100.0 * float((nblk == 0).float().mean()), float(nblk.float().mean())))
