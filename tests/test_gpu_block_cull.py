"""GPU: adversarial test of the blend kernels' block cull (csrc/blend_common.h, block_reaches).

A list entry is skipped for a whole 8x8 pixel block when a closed-form lower bound of the conic quadratic over the block
says no pixel can reach alpha >= 1/255.  One false reject silently drops a contribution, so the bound is checked against
brute force -- the blend kernels' own per-pixel test on every pixel centre of the block (fdgs_debug_block_reaches) and an
independent float64 evaluation -- on millions of tuples aimed at its weak spots: condition numbers up to 1e6, opacities
around the 1/255 threshold, means on block edges and corners and far outside, huge offsets (fp32 cancellation in the
quadratic), image-border blocks narrower than 8 pixels."""
import numpy as np
import pytest
import torch

import util  # noqa: F401
from oracle import pyoracle

pytestmark = pytest.mark.gpu


def _tuples(n, rng):
    bx = rng.integers(0, 200, n).astype(np.float32) * 8
    by = rng.integers(0, 150, n).astype(np.float32) * 8
    w = np.where(rng.random(n) < 0.15, rng.integers(1, 8, n), 8).astype(np.float32)   # clamped edge blocks
    h = np.where(rng.random(n) < 0.15, rng.integers(1, 8, n), 8).astype(np.float32)
    rx0, rx1, ry0, ry1 = bx, bx + w - 1, by, by + h - 1
    # mean: inside, on edges / corners, near (within a few sigma) or far
    kind = rng.integers(0, 6, n)
    ex = np.choose(rng.integers(0, 2, n), [rx0, rx1])
    ey = np.choose(rng.integers(0, 2, n), [ry0, ry1])
    far = 10.0 ** rng.uniform(0, 3.5, n) * rng.choice([-1, 1], n)
    mx = np.select([kind == 0, kind == 1, kind == 2, kind == 3], [rx0 + rng.random(n) * (w - 1), ex, ex + rng.normal(0, 2, n), ex + rng.normal(0, 30, n)], ex + far)
    far = 10.0 ** rng.uniform(0, 3.5, n) * rng.choice([-1, 1], n)
    my = np.select([kind == 0, kind == 1, kind == 2, kind == 4], [ry0 + rng.random(n) * (h - 1), ey, ey + rng.normal(0, 2, n), ey + rng.normal(0, 30, n)], ey + far)
    # covariance: sigma_major in [0.3, 3000] px, condition number (variance ratio) up to 1e6, random orientation; + 0.3 low pass
    s1 = 10.0 ** rng.uniform(-0.5, 3.5, n)
    s2 = s1 / np.sqrt(10.0 ** rng.uniform(0, 6, n))
    th = rng.uniform(0, np.pi, n)
    c, s = np.cos(th), np.sin(th)
    a = (c * s1) ** 2 + (s * s2) ** 2 + 0.3
    b = c * s * (s1 ** 2 - s2 ** 2)
    d = (s * s1) ** 2 + (c * s2) ** 2 + 0.3
    det = a * d - b * b
    A, B, Cc = d / det, -b / det, a / det
    # the cancellation case: a long thin splat whose major axis points at the block from D pixels away, offset sideways
    # by a few minor-axis sigmas -- the quadratic is small while its terms are ~ (D / sigma_minor)^2
    aim = rng.random(n) < 0.25
    D = 10.0 ** rng.uniform(0.5, 3.5, n) * rng.choice([-1, 1], n)
    side = rng.normal(0, 2.0, n) * np.sqrt(s2 ** 2 + 0.3)
    cxb, cyb = rx0 + rng.random(n) * (w - 1), ry0 + rng.random(n) * (h - 1)
    mx = np.where(aim, cxb + D * c - side * s, mx)
    my = np.where(aim, cyb + D * s + side * c, my)
    op = np.select([rng.random(n) < 0.3, rng.random(n) < 0.5], [rng.uniform(0.0039, 0.0041, n), 10.0 ** rng.uniform(-2.4, 0, n)], rng.random(n))
    return np.stack([mx, my, A, B, Cc, op, rx0, rx1, ry0, ry1], axis=1).astype(np.float32)


def _brute64(t):
    """float64: max alpha over the pixel centres of the block (reference semantics forward.cu:585-590)."""
    t = t.astype(np.float64)
    best = np.zeros(len(t))
    for j in range(8):
        for i in range(8):
            px, py = t[:, 6] + i, t[:, 8] + j
            ok = (px <= t[:, 7]) & (py <= t[:, 9])
            dx, dy = t[:, 0] - px, t[:, 1] - py
            power = -0.5 * (t[:, 2] * dx * dx + t[:, 4] * dy * dy) - t[:, 3] * dx * dy
            alpha = np.where(power > 0, 0.0, np.minimum(0.99, t[:, 5] * np.exp(np.minimum(power, 0.0))))
            best = np.maximum(best, np.where(ok, alpha, 0.0))
    return best


def test_block_reaches_never_rejects_a_contributing_entry(gpu_device):
    from fdgs import _capi
    rng = np.random.default_rng(7)
    total = rejected = accepted_needlessly = needed = needed_oracle = disagree = 0
    for _ in range(4):
        t = _tuples(600_000, rng)
        d_t = torch.from_numpy(t).to(gpu_device)
        d_o = torch.empty((len(t), 2), dtype=torch.uint8, device=gpu_device)
        with torch.cuda.device(gpu_device):
            rc = _capi.lib.fdgs_debug_block_reaches(len(t), d_t.data_ptr(), d_o.data_ptr(), _capi.current_stream_handle(gpu_device))
        assert rc == 0, _capi.last_error()
        o = d_o.cpu().numpy()
        reach, brute = o[:, 0].astype(bool), o[:, 1].astype(bool)
        bad = brute & ~reach
        assert not bad.any(), "false rejects vs the kernels' own per-pixel test: %d, first tuple %r" % (bad.sum(), t[bad][0])
        # third column: the ORACLE's per-pixel test (its own fp32 arithmetic: accurate expf, no contraction) -- an entry that both
        # the bound and the HIP per-pixel test reject but the oracle accepts would change a pixel against the reference
        orc = pyoracle.block_any_pixel_passes(t)
        bad_o = orc & ~reach
        assert not bad_o.any(), "false rejects vs the oracle's per-pixel test: %d, first tuple %r" % (bad_o.sum(), t[bad_o][0])
        needed_oracle += int(orc.sum()); disagree += int((orc != brute).sum())
        a64 = _brute64(t)
        bad64 = (a64 >= (1.0 / 255.0) * (1.0 + 1e-4)) & ~reach   # clearly above the threshold in exact arithmetic
        assert not bad64.any(), "false rejects vs float64: %d, first tuple %r (alpha %g)" % (bad64.sum(), t[bad64][0], a64[bad64][0])
        total += len(t); needed += int(brute.sum()); accepted_needlessly += int((reach & ~brute).sum()); rejected += int((~reach).sum())
    print("block_reaches: %d tuples, %d need the entry, %d rejected, %d accepted although no pixel passes (%.2f %% of the accepted)" % (
        total, needed, rejected, accepted_needlessly, 100.0 * accepted_needlessly / max(1, total - rejected)))
    print("block_reaches: the oracle's per-pixel test needs %d entries; it differs from the HIP per-pixel test on %d tuples (alpha "
          "within rounding of 1/255), none of them rejected by the bound" % (needed_oracle, disagree))
    assert total >= 2_000_000 and needed > 100_000 and rejected > 100_000 and needed_oracle > 100_000
