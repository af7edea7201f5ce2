import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rtm3d_amd
from rtm3d_amd import _lib
lib = _lib.load()
g3 = np.load('tests/golden/decode3d_cases.npz')
dev = torch.device('cuda', 0)
def run(N_rep, scalar=False):
    cl = np.tile(g3['clses'], N_rep); uv = np.tile(g3['uv'], (N_rep, 1, 1)).reshape(-1, 16); N = len(cl)
    d_cls = torch.as_tensor(cl, device=dev); d_uv = torch.as_tensor(uv, device=dev); d_K = torch.as_tensor(np.tile(g3['K'], (N, 1)), device=dev)
    d_dim = torch.as_tensor(g3['dim_ref'], device=dev); d_loc = torch.as_tensor(g3['ref_loc'], device=dev)
    x = torch.zeros(N, 8, dtype=torch.float64, device=dev); f = torch.zeros(N, dtype=torch.float64, device=dev)
    nit = torch.zeros(N, dtype=torch.int32, device=dev); st = torch.zeros(N, dtype=torch.int32, device=dev)
    fn = lib.rtm3d_decode3d_scalar if scalar else lib.rtm3d_decode3d
    s = torch.cuda.current_stream().cuda_stream
    def call(): _lib.check(fn(ctypes.c_void_p(s), N, d_cls.data_ptr(), d_uv.data_ptr(), d_K.data_ptr(), d_dim.data_ptr(), 3, d_loc.data_ptr(), x.data_ptr(), f.data_ptr(), nit.data_ptr(), st.data_ptr()))
    call(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); call(); e1.record(); torch.cuda.synchronize()
    print('N=%d scalar=%s kernel %.3f ms; nit mean %.1f max %d' % (N, scalar, e0.elapsed_time(e1), nit.float().mean().item(), nit.max().item()))
for rep in (1, 4, 16, 50):
    run(rep)
run(1, True); run(50, True)
