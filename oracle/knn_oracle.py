"""TEST INFRASTRUCTURE -- CPU restatement (numpy, float32) of the reference's ``distCUDA2``
(simple-knn/simple_knn.cu:143-220, spatial.cu:15-27): for every point the mean of the squared distances to its three
nearest OTHER points (index != own index; duplicates count with distance 0; FLT_MAX stands in for missing neighbours).

The reference finds them exactly (Morton order + 1024-point boxes, pruned with a valid upper bound, :158-190), so the
result does not depend on the search order; what has to match is the arithmetic: ``d.x*d.x + d.y*d.y + d.z*d.z`` in
float32 (:134-135) and ``(best[0] + best[1] + best[2]) / 3.0f`` with best[] ascending (:191).

Parity pin: tests/test_oracle_knn.py checks this restatement bit for bit against the reference's OWN device code
(simple_knn.cu:28-191 -- coord2Morton, boxMinMax under the fiber block emulator, boxMeanDist -- compiled verbatim into
oracle/_ref/liboracle_ref.so by oracle/refbuild/build_ref.py; only the host function :193-220, which needs cub / thrust, is
restated with std:: in oracle/refbuild/ref_knn.cpp) and against an independent exact nearest-neighbour search (scipy
cKDTree in float64).  Only tests may import this module.
"""
import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)


def dist2_knn3(points: np.ndarray, chunk: int = 1024) -> np.ndarray:
    p = np.ascontiguousarray(points, dtype=np.float32)
    P = p.shape[0]
    out = np.empty((P,), np.float32)
    for s in range(0, P, chunk):
        q = p[s:s + chunk]
        dx = p[None, :, 0] - q[:, None, 0]          # point - ref, simple_knn.cu:134
        dy = p[None, :, 1] - q[:, None, 1]
        dz = p[None, :, 2] - q[:, None, 2]
        d = (dx * dx + dy * dy) + dz * dz             # float32, left to right
        d[np.arange(q.shape[0]), np.arange(s, s + q.shape[0])] = np.inf   # i == idx is skipped (:170, :185)
        if P - 1 >= 3:
            best = np.sort(np.partition(d, 2, axis=1)[:, :3], axis=1)
        else:
            best = np.sort(d, axis=1)
            best = np.concatenate([best[:, :max(P - 1, 0)], np.full((q.shape[0], 3 - max(P - 1, 0)), FLT_MAX, np.float32)], 1)
        best = best.astype(np.float32)
        with np.errstate(over="ignore"):
            out[s:s + chunk] = ((best[:, 0] + best[:, 1]) + best[:, 2]) / np.float32(3.0)
    return out
