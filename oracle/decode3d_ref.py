"""CPU restatement (numpy fp64 + SciPy L-BFGS-B) of the reference 3D box decode.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The arithmetic of this step lives in a third-party dependency that is not vendored in
/root/reference: ``scipy.optimize.minimize(method='L-BFGS-B')`` (reference pins no version;
this image has SciPy 1.15.3, a C port of L-BFGS-B 3.0).  The restatement therefore calls the same
SciPy routine with the same options, and restates only the reference's own objective / gradient /
driver loop (utils/model_utils.py:155-177, 206-234, 264-312).

Pinned by: tests/golden/decode3d_cases.npz (outputs of the reference function itself).
"""
import warnings
import numpy as np
from scipy.optimize import minimize

# utils/model_utils.py:275-282 : corners in order i(x), j(y), k(z) in {1,-1}, times 0.5
COR = 0.5 * np.array([[i, j, k] for i in (1, -1) for j in (1, -1) for k in (1, -1)], dtype=np.float64).T  # (3,8)

# utils/model_utils.py:290-291
OPTIONS = {'disp': None, 'maxcor': 10, 'ftol': 2.220446049250313e-09, 'gtol': 1e-05, 'eps': 1e-08,
           'maxfun': 15000, 'maxiter': 15000, 'iprint': -1, 'maxls': 20, 'finite_diff_rel_step': None}


def aim_fun(K, UV):
    """utils/model_utils.py:155-177 (aimFun): sum of squared reprojection errors, cost=1e-4."""
    cor_t, uv_t = COR.T, np.asarray(UV, np.float64).reshape(8, 2)
    cost = 1e-4

    def fun(x):
        obj = 0
        for cor, uv in zip(cor_t, uv_t):
            xc = cor[0] * x[2] * x[1] + cor[2] * x[4] * x[0] + x[5]
            yc = cor[1] * x[3] + x[6]
            zc = -cor[0] * x[2] * x[0] + cor[2] * x[4] * x[1] + x[7]
            obj += (xc * K[0, 0] / (zc + cost) + K[0, 2] - uv[0]) ** 2
            obj += (yc * K[1, 1] / (zc + cost) + K[1, 2] - uv[1]) ** 2
        return obj
    return fun


def jac_fun(K, UV):
    """utils/model_utils.py:206-234 (jac): analytic gradient, cost=1e-6 (sic, differs from aimFun)."""
    cor_t, uv_t = COR.T, np.asarray(UV, np.float64).reshape(8, 2)
    cost = 1e-6

    def fun(x):
        err = np.zeros((len(x),), dtype=np.float64)
        for cor, uv in zip(cor_t, uv_t):
            xc = cor[0] * x[2] * x[1] + cor[2] * x[4] * x[0] + x[5]
            yc = cor[1] * x[3] + x[6]
            zc = -cor[0] * x[2] * x[0] + cor[2] * x[4] * x[1] + x[7]
            dex = (xc * K[0, 0] / (zc + cost) + K[0, 2] - uv[0]) * 2
            dey = (yc * K[1, 1] / (zc + cost) + K[1, 2] - uv[1]) * 2
            d_x = np.array([cor[2] * x[4], cor[0] * x[2], cor[0] * x[1], 0, cor[2] * x[0], 1, 0, 0])
            d_y = np.array([0, 0, 0, cor[1], 0, 0, 1, 0])
            d_z = np.array([-cor[0] * x[2], cor[2] * x[4], -cor[0] * x[0], 0, cor[2] * x[1], 0, 0, 1])
            gx = K[0, 0] * (d_x * zc - d_z * xc) / (zc ** 2 + cost)
            gy = K[1, 1] * (d_y * zc - d_z * yc) / (zc ** 2 + cost)
            err += dex * gx + dey * gy
        return err
    return fun


def solve_one(cls, UV, K, ref_dim, ref_loc):
    """One object: utils/model_utils.py:292-296.  Returns the scipy OptimizeResult."""
    dim = ref_dim[int(cls)]
    x0 = np.array([0, 1] + [dim[2], dim[0], dim[1]] + list(ref_loc), dtype=np.float64)
    K = np.asarray(K, np.float64).reshape(3, 3)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')   # the reference passes constraints= which L-BFGS-B ignores
        return minimize(aim_fun(K, UV), x0, method='L-BFGS-B', jac=jac_fun(K, UV), options=dict(OPTIONS))


def optim_decode_bbox3d(clses, bbox3d_projs, K, ref_dim, ref_loc, return_raw=False):
    """utils/model_utils.py:264-312.  Returns dict(class, Ry, dimension, location, K) of kept
    objects (fun < 0.1); with return_raw also per-object (x, fun, nit, kept)."""
    K = np.asarray(K, np.float64).reshape(3, 3)
    out = {'class': [], 'Ry': [], 'dimension': [], 'location': [], 'K': []}
    raw = {'x': [], 'fun': [], 'nit': [], 'kept': []}
    for cls, UV in zip(clses, bbox3d_projs):
        res = solve_one(cls, np.asarray(UV, np.float64), K, ref_dim, ref_loc)
        kept = bool(res.fun < 0.1)                                           # :298
        raw['x'].append(np.array(res.x)); raw['fun'].append(float(res.fun))
        raw['nit'].append(int(res.nit)); raw['kept'].append(kept)
        if kept:
            x = res.x
            out['Ry'].append(np.arctan2(x[0], x[1]))                          # :300
            out['dimension'].append(np.array([x[3], x[4], x[2]]))             # :302 (h, w, l)
            out['location'].append(np.array([x[5], x[6], x[7]]))              # :303
            out['class'].append(int(cls))
            out['K'].append(K.reshape(9))
    n = len(out['class'])
    res = {'class': out['class'], 'Ry': np.array(out['Ry'], np.float64),
           'dimension': np.array(out['dimension'], np.float64).reshape(n, 3),
           'location': np.array(out['location'], np.float64).reshape(n, 3),
           'K': np.array(out['K'], np.float64).reshape(n, 9)}
    if return_raw:
        return res, {k: np.array(v) for k, v in raw.items()}
    return res


def project_box(dim_hwl, loc, ry, K):
    """Helper for building synthetic key points: project the 8 corners (COR order) of a box with
    dimension (h, w, l), centre ``loc`` and yaw ``ry`` through K.  Same model as aim_fun."""
    K = np.asarray(K, np.float64).reshape(3, 3)
    h, w, l = dim_hwl
    s, c = np.sin(ry), np.cos(ry)
    uv = np.zeros((8, 2))
    for k in range(8):
        cx, cy, cz = COR[:, k]
        xc = cx * l * c + cz * w * s + loc[0]
        yc = cy * h + loc[1]
        zc = -cx * l * s + cz * w * c + loc[2]
        uv[k, 0] = xc * K[0, 0] / zc + K[0, 2]
        uv[k, 1] = yc * K[1, 1] / zc + K[1, 2]
    return uv
