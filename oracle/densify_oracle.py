"""TEST INFRASTRUCTURE -- CPU restatement (numpy) of the reference's densification / pruning
(scene/gaussian_model.py:391-610, utils/general_utils.py:79-133), on the layout this repository trains with
(one ``_features [P,M,3]`` tensor instead of f_dc / f_rest).  Only tests may import this module.

Parity pin: tests/test_oracle_densify.py compares it with fixtures produced by running the REFERENCE's own
``GaussianModel.densify_and_prune`` on the CPU (tests/golden/make_golden_densify.py), with the normal samples the
reference drew injected here.

state  = {"params": {name: array}, "exp_avg": {name: array}, "exp_avg_sq": {name: array},
          "xyz_gradient_accum" [P,1], "t_gradient_accum" [P,1] (4D), "denom" [P,1], "max_radii2D" [P]}
names  = _xyz _features _opacity _scaling _rotation (_t _scaling_t (_rotation_r))
"""
import numpy as np

F32 = np.float32


def build_rotation(r):
    """utils/general_utils.py:79-100"""
    norm = np.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])
    q = r / norm[:, None]
    R = np.zeros((q.shape[0], 3, 3), F32)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def build_rotation_4d(l, r):
    """utils/general_utils.py:113-133 (A = (M_l @ M_r).flip(1, 2))"""
    q_l = l / np.linalg.norm(l, axis=-1, keepdims=True).astype(F32)
    q_r = r / np.linalg.norm(r, axis=-1, keepdims=True).astype(F32)
    a, b, c, d = q_l[:, 0], q_l[:, 1], q_l[:, 2], q_l[:, 3]
    p, q, rr, s = q_r[:, 0], q_r[:, 1], q_r[:, 2], q_r[:, 3]
    M_l = np.stack([a, -b, -c, -d, b, a, -d, c, c, d, a, -b, d, -c, b, a]).reshape(4, 4, -1).transpose(2, 0, 1)
    M_r = np.stack([p, q, rr, s, -q, p, -s, rr, -rr, s, p, -q, -s, -rr, q, p]).reshape(4, 4, -1).transpose(2, 0, 1)
    A = (M_l @ M_r).astype(F32)
    return A[:, ::-1, ::-1].copy()


def _cat(state, new_params):
    """cat_tensors_to_optimizer + densification_postfix (gaussian_model.py:434-485): params appended, optimizer state
    extended with zeros, statistics reset to zero for ALL points."""
    for k, v in new_params.items():
        state["params"][k] = np.concatenate([state["params"][k], v.astype(F32)], 0)
        state["exp_avg"][k] = np.concatenate([state["exp_avg"][k], np.zeros_like(v, F32)], 0)
        state["exp_avg_sq"][k] = np.concatenate([state["exp_avg_sq"][k], np.zeros_like(v, F32)], 0)
    P = state["params"]["_xyz"].shape[0]
    if "_t" in state["params"]:
        state["t_gradient_accum"] = np.zeros((P, 1), F32)
    state["xyz_gradient_accum"] = np.zeros((P, 1), F32)
    state["denom"] = np.zeros((P, 1), F32)
    state["max_radii2D"] = np.zeros((P,), F32)


def _prune(state, mask):
    """prune_points / _prune_optimizer (gaussian_model.py:391-432)"""
    keep = ~mask
    for k in state["params"]:
        state["params"][k] = state["params"][k][keep]
        state["exp_avg"][k] = state["exp_avg"][k][keep]
        state["exp_avg_sq"][k] = state["exp_avg_sq"][k][keep]
    for k in ("xyz_gradient_accum", "denom", "max_radii2D", "t_gradient_accum"):
        if k in state:
            state[k] = state[k][keep]


def densify_and_prune(state, max_grad, min_opacity, extent, max_screen_size, max_grad_t=None, prune_only=False,
                      percent_dense=0.01, N=2, rot_4d=True, gaussian_dim=4, samples=None, samples_t=None):
    """gaussian_model.py:584-610.  ``samples`` ([k*N, 4] under rot_4d, else [k*N, 3]) / ``samples_t`` ([k*N, 1]): the
    draws of torch.normal(mean=0, std=stds) in densify_and_split (:503, :510, :515), injected."""
    state = {k: ({n: a.copy() for n, a in v.items()} if isinstance(v, dict) else v.copy()) for k, v in state.items()}
    pr = state["params"]
    if not prune_only:
        with np.errstate(divide="ignore", invalid="ignore"):
            grads = (state["xyz_gradient_accum"] / state["denom"]).astype(F32)
        grads[np.isnan(grads)] = 0.0
        # ---- densify_and_clone (:545-569)
        scal = np.exp(pr["_scaling"])
        sel = (np.linalg.norm(grads, axis=-1) >= max_grad) & (scal.max(1) <= percent_dense * extent)
        _cat(state, {k: v[sel] for k, v in pr.items()})
        pr = state["params"]
        # ---- densify_and_split (:487-543)
        n_init = pr["_xyz"].shape[0]
        padded = np.zeros((n_init,), F32)
        padded[:grads.shape[0]] = grads[:, 0]
        scal = np.exp(pr["_scaling"])
        sel = (padded >= max_grad) & (scal.max(1) > percent_dense * extent)
        rep = lambda a: np.concatenate([a[sel]] * N, 0)  # noqa: E731   tensor[mask].repeat(N, 1)
        new = {"_scaling": np.log(rep(scal) / F32(0.8 * N)), "_rotation": rep(pr["_rotation"]), "_features": rep(pr["_features"]),
               "_opacity": rep(pr["_opacity"])}
        if not rot_4d:
            rots = np.concatenate([build_rotation(pr["_rotation"][sel])] * N, 0)
            new["_xyz"] = np.einsum("nij,nj->ni", rots, samples.astype(F32)).astype(F32) + rep(pr["_xyz"])
            if gaussian_dim == 4:
                new["_t"] = samples_t.astype(F32) + rep(pr["_t"])
                new["_scaling_t"] = np.log(rep(np.exp(pr["_scaling_t"])) / F32(0.8 * N))
        else:
            rots = np.concatenate([build_rotation_4d(pr["_rotation"][sel], pr["_rotation_r"][sel])] * N, 0)
            xyzt = np.concatenate([pr["_xyz"], pr["_t"]], 1)
            new_xyzt = np.einsum("nij,nj->ni", rots, samples.astype(F32)).astype(F32) + rep(xyzt)
            new["_xyz"], new["_t"] = new_xyzt[:, 0:3], new_xyzt[:, 3:4]
            new["_scaling_t"] = np.log(rep(np.exp(pr["_scaling_t"])) / F32(0.8 * N))
            new["_rotation_r"] = rep(pr["_rotation_r"])
        n_new = N * int(sel.sum())
        _cat(state, new)
        _prune(state, np.concatenate([sel, np.zeros((n_new,), bool)]))
        pr = state["params"]
    opacity = 1.0 / (1.0 + np.exp(-pr["_opacity"]))
    prune_mask = (opacity < min_opacity)[:, 0]
    if max_screen_size:
        big_vs = state["max_radii2D"] > max_screen_size
        big_ws = np.exp(pr["_scaling"]).max(1) > 0.1 * extent
        prune_mask = prune_mask | big_vs | big_ws
    _prune(state, prune_mask)
    return state
