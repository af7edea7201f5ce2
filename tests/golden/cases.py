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
