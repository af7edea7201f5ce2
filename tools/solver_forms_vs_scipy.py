"""CPU: the two forms of the product's L-BFGS-B (host build of rtm3d_amd/csrc/lbfgsb.h: published subspace step / direct two-loop
form) against the reference's SciPy results (through the oracle) on N synthetic cuboid projections with Gaussian vertex noise:
how often does a correct reimplementation end further than 1e-4 from SciPy, and is the product's direct form worse than the
published one?   python tools/solver_forms_vs_scipy.py [N] [vertex noise sigma in px] >> profiles/r03_solver_forms_vs_scipy.txt"""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import decode3d_ref
from tests.golden.cases import DIM_REF, _project
from rtm3d_amd import weights
K9 = weights.synth_intrinsics().reshape(9)
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', '_build', 'libhost_lbfgsb.so'))
rng = np.random.default_rng(11)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
cls = rng.integers(0, 3, N)
uv = np.zeros((N, 16), np.float32)
for i in range(N):
    z = rng.uniform(6.0, 60.0)
    u, v = rng.uniform(50, 1230), rng.uniform(120, 370)
    loc = np.array([(u - K9[2]) * z / K9[0], (v - K9[5]) * z / K9[4], z])
    dim = np.array(DIM_REF[cls[i]]) * rng.uniform(0.85, 1.2, 3)
    ry = rng.uniform(-np.pi, np.pi)
    p = _project(dim, loc, ry, K9).reshape(16)
    uv[i] = (p + rng.normal(0, noise, 16)).astype(np.float32)      # a little regression noise, like a network's
_, raw = decode3d_ref.optim_decode_bbox3d(cls, uv.reshape(N, 8, 2), K9.reshape(3, 3), DIM_REF, [0, -0.5, 20], return_raw=True)
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
Kn = np.ascontiguousarray(np.tile(K9, (N, 1))); dim = np.ascontiguousarray(DIM_REF, np.float64); loc = np.array([0, -0.5, 20.0])
c64 = np.ascontiguousarray(cls, np.int64)
def bp(xs):
    return np.concatenate([np.arctan2(xs[:, 0:1], xs[:, 1:2]), xs[:, 3:5], xs[:, 2:3], xs[:, 5:8]], 1)
def ad(a, b):
    d = np.abs(a - b); d[:, 0] = np.minimum(d[:, 0], 2 * np.pi - d[:, 0]); return d
print('objects', N, 'kept by the reference', int(raw['kept'].sum()), 'vertex noise sigma', noise)
for form in ('lb_solve_batch', 'lb_solve_batch_direct'):
    x = np.zeros((N, 8)); f = np.zeros(N); nit = np.zeros(N, np.int32); st = np.zeros(N, np.int32)
    getattr(lib, form)(N, P(c64), P(uv), P(Kn), P(dim), P(loc), P(x), P(f), P(nit), P(st))
    both = raw['kept'] & (f < 0.1)
    d = ad(bp(x[both]), bp(raw['x'][both])).max(1)
    q = np.percentile(d, [50, 90, 99, 99.9])
    print('%-22s keep mismatches %d  nit differs %d  box L-inf p50 %.1e p90 %.1e p99 %.1e p99.9 %.1e max %.1e  >1e-4: %d of %d' %
          (form, int((raw['kept'] != (f < 0.1)).sum()), int((nit != raw['nit'])[both].sum()), q[0], q[1], q[2], q[3], d.max(), int((d > 1e-4).sum()), len(d)))
