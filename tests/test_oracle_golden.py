"""CPU: the oracle restatement (oracle/fdgs_oracle.c) against the golden fixtures produced by the verbatim
CPU build of the reference kernels.  Integer / index outputs and depth bits must be identical; floats agree to
rounding (both are -ffp-contract=off builds that follow the same operation order); gradients to atomics order."""
import numpy as np
import pytest

import golden_util
from util import run_oracle

INT_KEYS = ("radii", "tiles_touched", "point_offsets", "clamped", "keys_sorted", "point_list", "ranges", "n_contrib")
FLOAT_KEYS = ("out_color", "out_flow", "out_depth", "out_T", "out_means3D", "means2D", "depths", "cov3D", "rgb",
              "conic_opacity")


@pytest.mark.parametrize("name", golden_util.NAMES)
def test_port_oracle_matches_golden(name):
    scene, up, fw, bw = golden_util.load(name)
    out, grads = run_oracle(scene, up, kind="port")
    assert out["R"] == fw["R"]
    for k in INT_KEYS:
        np.testing.assert_array_equal(out[k], fw[k], err_msg="%s %s" % (name, k))
    vis = fw["radii"] > 0
    for k in FLOAT_KEYS:
        a, b = out[k], fw[k]
        if a.shape[0] == vis.shape[0] and a.ndim <= 2 and k not in ("out_depth", "out_T"):
            a, b = a[vis], b[vis]
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32), err_msg="%s %s bits" % (name, k))
    for k, b in bw.items():
        a = grads[k].reshape(b.shape)
        scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
        err = float(np.abs(a - b).max()) if b.size else 0.0
        assert err <= 2e-5 * scale, "%s %s: %g (scale %g)" % (name, k, err, scale)


def test_golden_fixtures_present():
    assert len(golden_util.NAMES) >= 4
