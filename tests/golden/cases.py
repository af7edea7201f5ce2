"""Deterministic INPUTS of the decode-2D golden cases (numpy PCG64 streams), shared by
make_golden.py (which runs the reference on them) and by the tests (which regenerate them instead
of storing megabytes of random logits).  The .npz keeps a small probe of each input to detect
drift of the bit-stream on another machine."""
import numpy as np

DECODE2D_CASES = ['dense', 'sparse_empty_edge', 'plateau', 'thresh05_top30', 'odd']


def decode2d_inputs(name):
    """-> (thresh, topk, [main_kf, offset_fr_main, main_offset, vertex_offset] as fp32 numpy NCHW)."""
    rng = np.random.Generator(np.random.PCG64(7 + DECODE2D_CASES.index(name)))

    def pack(B, H, W, hm):
        return [hm.astype(np.float32),
                (rng.standard_normal((B, 16, H, W)) * 3).astype(np.float32),
                (rng.standard_normal((B, 2, H, W)) * 2).astype(np.float32),
                rng.standard_normal((B, 2, H, W)).astype(np.float32)]

    if name == 'dense':        # >100 peaks per image (cap at topk), full stride-4 map size
        return 0.4, 100, pack(2, 96, 320, rng.standard_normal((2, 3, 96, 320)) * 1.5 - 3)
    if name == 'sparse_empty_edge':   # a few peaks incl. corners; image 1 empty (-> None); threshold edge
        hm = rng.standard_normal((3, 3, 24, 40)) * 0.5 - 6
        hm[0, 0, 5, 7] = 2.0; hm[0, 2, 20, 39] = 1.0; hm[0, 1, 0, 0] = 0.5; hm[0, 1, 23, 0] = 3.0
        hm[2, 0, 11, 11] = -0.30; hm[2, 0, 3, 30] = -0.45   # sigmoid(-0.405)=0.4: one above, one below
        return 0.4, 100, pack(3, 24, 40, hm)
    if name == 'plateau':      # fp32 sigmoid(x>=17) == 1.0; equality-NMS keeps every plateau member
        hm = rng.standard_normal((1, 3, 16, 32)) * 0.3 - 8
        hm[0, 0, 4:6, 4:7] = 20.0
        hm[0, 1, 10, 10] = 1.25; hm[0, 1, 10, 11] = 1.25
        hm[0, 2, 8, 20] = 30.0
        return 0.4, 100, pack(1, 16, 32, hm)
    if name == 'thresh05_top30':   # defaults of models/configs/detault.py:85-86
        return 0.5, 30, pack(2, 48, 160, rng.standard_normal((2, 3, 48, 160)) * 1.5 - 2.5)
    if name == 'odd':          # sizes that are not multiples of the wave / tile widths
        return 0.4, 100, pack(2, 13, 37, rng.standard_normal((2, 3, 13, 37)) * 2 - 2)
    raise KeyError(name)


# ------------------------------------------------------------------------------------------------------
# "Planted" end-to-end decode cases: reference-run logits (an e2e fixture) as background, with the heat map,
# the main-offset and the 16 vertex-offset channels overwritten at a few well separated peaks by the exact
# projection of real cuboids, so that the reference's own Model.inference + optim_decode_bbox3d KEEPS those
# objects (fun < 0.1) next to the natural, non-cuboid detections of the background, which it rejects.
DIM_REF = [[1.52607842, 1.62858147, 3.88396124], [1.76067766, 0.6602296, 0.84220464],
           [1.73712792, 0.59677122, 1.76338868]]          # models/configs/rtm3d_dla34_kitti.yaml:18-27 (h, w, l)
PLANTED_CASES = {
    # name: (background fixture, intrinsics (fx, fy, cx, cy), objects planted per image, seed)
    'planted_small': ('e2e_dla34_small.npz', (180.0, 180.0, 128.0, 64.0), 14, 31),
    'planted_full': ('e2e_dla34_full.npz', None, 40, 32),
}
_COR = 0.5 * np.array([[i, j, k] for i in (1, -1) for j in (1, -1) for k in (1, -1)], dtype=np.float64)   # (8, 3)


def _project(dim_hwl, loc, ry, K):
    """8 corners of a box (utils/model_utils.py:275-281 corner order) through K (9,) -> (8, 2) pixels."""
    h, w, l = dim_hwl
    s, c = np.sin(ry), np.cos(ry)
    xc = _COR[:, 0] * l * c + _COR[:, 2] * w * s + loc[0]
    yc = _COR[:, 1] * h + loc[1]
    zc = -_COR[:, 0] * l * s + _COR[:, 2] * w * c + loc[2]
    return np.stack([xc * K[0] / zc + K[2], yc * K[4] / zc + K[5]], 1)


def plant_cuboids(hm, reg, K, nobj, rng):
    """Overwrite, in place, the logits (hm (B,3,H,W), reg = [offset_fr_main (B,16,H,W), main_offset (B,2,H,W), ...]) at
    ``nobj`` well separated peaks per image with exact cuboid projections.  Returns the planted truth per image."""
    B, C, H, W = hm.shape
    truth = []
    for b in range(B):
        taken, objs = [], []
        while len(objs) < nobj:
            y, x = int(rng.integers(2, H - 2)), int(rng.integers(2, W - 2))
            if any(max(abs(y - yy), abs(x - xx)) < 4 for yy, xx in taken):
                continue
            taken.append((y, x))
            cls = int(rng.integers(0, 3))
            score = 0.5 + 0.45 * (len(objs) + rng.uniform(0, 0.9)) / nobj          # spread over 0.5 .. 0.95, no ties
            ox, oy = rng.uniform(0.05, 0.95, 2)
            z = rng.uniform(8.0, 50.0)
            u, v = 4.0 * (x + ox), 4.0 * (y + oy)
            loc = np.array([(u - K[2]) * z / K[0], (v - K[5]) * z / K[4], z])
            dim = np.array(DIM_REF[cls]) * rng.uniform(0.85, 1.2, 3)
            ry = rng.uniform(-np.pi, np.pi)
            uv = _project(dim, loc, ry, K)
            # heat map: the planted value is a strict 3x3 maximum of its class plane
            lg = np.float32(np.log(score / (1.0 - score)))
            win = hm[b, cls, y - 1:y + 2, x - 1:x + 2]
            np.minimum(win, lg - np.float32(1.0), out=win)
            hm[b, cls, y, x] = lg
            reg[1][b, 0, y, x] = np.float32(np.log(ox / (1 - ox)))
            reg[1][b, 1, y, x] = np.float32(np.log(oy / (1 - oy)))
            ctr = np.array([x + ox, y + oy])
            reg[0][b, :, y, x] = (uv / 4.0 - ctr[None, :]).reshape(16).astype(np.float32)
            objs.append((cls, y, x, dim, loc, ry))
        truth.append(objs)
    return truth


def planted_inputs(name, background):
    """-> (thresh, topk, K (9,) fp64, [4 logit maps fp32 NCHW], truth list per image of (cls, y, x, dim, loc, ry)).
    ``background``: the loaded e2e fixture named in PLANTED_CASES (np.load result)."""
    fixture, intr, nobj, seed = PLANTED_CASES[name]
    rng = np.random.Generator(np.random.PCG64(seed))
    hm = np.array(background['logits_main_kf'], np.float32)
    B, C, H, W = hm.shape
    if 'logits_offset_fr_main' in background:
        reg = [np.array(background['logits_' + n], np.float32) for n in ('offset_fr_main', 'main_offset', 'vertex_offset')]
    else:   # the full-size fixture stores the regression heads sub-sampled: seeded backgrounds instead
        reg = [(rng.standard_normal((B, c, H, W)) * s).astype(np.float32) for c, s in ((16, 3.0), (2, 2.0), (2, 1.0))]
    if intr is None:
        K = np.array(background['K'], np.float64)
    else:
        K = np.array([intr[0], 0, intr[2], 0, intr[1], intr[3], 0, 0, 1], np.float64)
    truth = plant_cuboids(hm, reg, K, nobj, rng)
    return 0.4, 100, K, [hm] + reg, truth
