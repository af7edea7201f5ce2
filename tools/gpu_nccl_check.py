"""Single-rank rehearsal of the collectives bench.py issues at N > 1 (RCCL through torch.distributed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
dist.init_process_group(backend='nccl', device_id=dev)
from rtm3d_amd import distributed as rdist
rec = torch.rand(32, 100, 32, device=dev)
side = torch.cuda.Stream(device=dev, priority=-1)
with torch.cuda.stream(side):
    out = torch.empty((dist.get_world_size() * 32, 100, 32), device=dev)
    dist.all_gather_into_tensor(out, rec)
side.synchronize()
assert torch.equal(out, rec)
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
print('nccl ok', float(t))
dist.destroy_process_group()
