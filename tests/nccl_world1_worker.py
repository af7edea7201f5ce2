"""Child process of tests/test_gpu_parity.py::test_pipeline_under_nccl_world1: the two-stream pipeline with its
collective forced through RCCL (backend "nccl", world size 1) on the side stream, against the serial path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', sys.argv[1])
    os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    dist.init_process_group(backend='nccl', device_id=dev)
    import rtm3d_amd
    from rtm3d_amd import weights, distributed as rdist
    from rtm3d_amd.pipeline import Detect3DPipeline
    from tests.util import pack_records_reference
    bb = 'RESNET-18'
    cfg = rtm3d_amd.kitti_config(bb)
    m = rtm3d_amd.create_model(cfg).to(dev).eval()
    m.load_state_dict(weights.synth_state_dict(bb, 1, 'trained', heat_bias=-3.5))
    B = 3
    K = torch.as_tensor(np.tile(weights.synth_intrinsics(), (B, 1)), device=dev)
    xs = [weights.synth_images(B, 64, 128, seed=300 + 7 * i).to(dev) for i in range(4)]
    pipe = Detect3DPipeline(m, B, dev, gather='always')
    pipe.time_gather = True
    got = {}
    for i, x in enumerate(xs):
        k = pipe.submit(x, K)
        if k >= 1:
            got[k - 1] = pipe.results(k - 1).clone()
    got[len(xs) - 1] = pipe.results(len(xs) - 1).clone()
    us = pipe.gather_us(len(xs) - 1)
    pipe.drain()
    for i, x in enumerate(xs):
        det, boxes, _ = m.detect3d(x, K)
        ref = pack_records_reference(det.n, det.cls, det.score, det.mproj, det.verts, det.bbox, 100, boxes)
        torch.cuda.synchronize()
        assert got[i].shape == (B, 100, 32) and torch.equal(got[i], ref), i
    t = torch.tensor([1.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    dist.destroy_process_group()
    print('nccl world-1 pipeline ok, all-gather %.1f us' % us)


if __name__ == '__main__':
    main()
