"""Multi-GPU inference: one process per GPU, contiguous batch shards, ONE all-gather of fixed-size
detection records per batch (RCCL over xGMI via torch.distributed backend "nccl").

The reference has no multi-GPU inference (detect.py:18 pins GPU 0); images are independent
(eval-mode BN, per-(b,c) softmax, per-image decode), so the path shards with no data-path
collective except collecting the results.  Record layout (fp32 x 32 per slot):
  [0] cls  [1] score  [2:4] main key-point  [4:20] 8 vertices (x,y)  [20:24] 2D box
  [24:27] dimension (h,w,l)  [27:30] location  [30] Ry  [31] flags: 0 empty, 1 2D only, 2 3D kept
"""
import torch
import torch.distributed as dist

RECORD = 32


def shard_range(total, rank, world):
    """Contiguous split [lo, hi) of `total` images for `rank`; the first total % world ranks get one more.
    Shards may differ by one image: for the fixed-batch pipeline / the all-gather use ``padded_shard``."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_records(n, cls, score, mproj, verts, bbox, topk, boxes=None, out=None):
    """(B,) counts + (B*topk, ...) slot tensors (+ the solver's Boxes3D) -> (B, topk, 32) fp32 records, written by ONE
    HIP launch on the current stream (rtm3d_pack_records; no ATen kernels between the 2D decode and the collective).
    Device tensors only: there is no CPU path."""
    import ctypes
    from . import _lib
    from .model_utils import FUN_ACCEPT
    if not n.is_cuda:
        raise RuntimeError('rtm3d_amd.distributed.pack_records needs CUDA (ROCm) tensors; there is no CPU path')
    B = n.shape[0]
    dev = n.device
    if cls.shape[0] != B * topk:
        raise ValueError('pack_records: %d slots for %d images x topk %d' % (cls.shape[0], B, topk))
    lib = _lib.load()
    with torch.cuda.device(dev):
        if out is None:
            out = torch.empty(B, topk, RECORD, dtype=torch.float32, device=dev)
        x, fun, st = (boxes.x.data_ptr(), boxes.fun.data_ptr(), boxes.status.data_ptr()) if boxes is not None else (0, 0, 0)
        _lib.check(lib.rtm3d_pack_records(ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream), B, int(topk), n.data_ptr(),
                                          cls.data_ptr(), score.data_ptr(), mproj.data_ptr(), verts.data_ptr(), bbox.data_ptr(),
                                          x, fun, st, float(FUN_ACCEPT), out.data_ptr()), 'pack_records')
    return out


def padded_shard(total, rank, world):
    """Equal-size shards for the fixed-shape pipeline and the all-gather: every rank processes ``per = ceil(total/world)``
    image slots [lo, lo+per); the slots at or beyond ``total`` are padding (feed any image, e.g. a repeat of the last
    one) and are dropped by ``trim_gathered``.  Returns (lo, hi_valid, per)."""
    per = -(-total // world)
    lo = min(rank * per, total)
    return lo, min(lo + per, total), per


def trim_gathered(rec, total):
    """Drop the padding slots of ``padded_shard`` from a gathered (world*per, topk, 32) record tensor."""
    return rec[:total]


def gathered_buffer(rec_like, group=None):
    """A (world*b, topk, 32) buffer for ``all_gather_records(out=...)``: allocate it once per pipeline slot so that the
    per-step path neither allocates on the side stream nor hands a side-stream block to another stream's consumers."""
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    return torch.zeros((world * rec_like.shape[0],) + tuple(rec_like.shape[1:]), dtype=rec_like.dtype, device=rec_like.device)


def all_gather_records(rec, group=None, check_shapes=False, always=False, out=None):
    """(b, topk, 32) per rank -> (world*b, topk, 32) ordered by global image index.  One collective.
    ``out``: preallocated result (``gathered_buffer``); without it a fresh tensor is allocated on the current stream."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not always):
        return rec
    world = dist.get_world_size(group)
    rec = rec.contiguous()
    if check_shapes:
        # a one-off guard for new callers (not for the per-step path): uneven shards would make the collective write
        # out of step; all ranks must hand in the same (b, topk, 32)
        shp = torch.tensor(list(rec.shape), dtype=torch.int64, device=rec.device)
        lo, hi = shp.clone(), shp.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
        if not torch.equal(lo, hi):
            raise ValueError('all_gather_records: shard shapes differ across ranks (%s..%s); pad with padded_shard()'
                             % (lo.tolist(), hi.tolist()))
    shape = (world * rec.shape[0],) + tuple(rec.shape[1:])
    if out is None:
        out = torch.empty(shape, dtype=rec.dtype, device=rec.device)
    elif tuple(out.shape) != shape or out.dtype != rec.dtype or out.device != rec.device or not out.is_contiguous():
        raise ValueError('all_gather_records: out must be a contiguous %s %s tensor on %s, got %s %s on %s'
                         % (shape, rec.dtype, rec.device, tuple(out.shape), out.dtype, out.device))
    if rec.is_cuda and dist.get_backend(group) == 'gloo':
        # REHEARSAL transport only (bench.py --rehearse-one-gpu: several ranks sharing one GPU, where RCCL refuses duplicate
        # devices): gloo has no CUDA all-gather, so the records are staged through host memory, synchronously.  The product
        # transport is backend "nccl" (RCCL over xGMI); nothing else takes this branch.
        host = torch.empty(shape, dtype=rec.dtype)
        dist.all_gather_into_tensor(host, rec.cpu(), group=group)
        out.copy_(host)
        return out
    dist.all_gather_into_tensor(out, rec, group=group)
    return out


def unpack_records(rec):
    """(B, topk, 32) -> per-image lists like Model.inference (+ 3D fields), on the records' device."""
    B = rec.shape[0]
    out = []
    for b in range(B):
        r = rec[b][rec[b, :, 31] > 0]
        if r.shape[0] == 0:
            out.append(None)
            continue
        out.append({'cls': r[:, 0].to(torch.int64), 'score': r[:, 1], 'm_proj': r[:, 2:4], 'verts': r[:, 4:20].view(-1, 8, 2),
                    'bbox2d': r[:, 20:24], 'dimension': r[:, 24:27], 'location': r[:, 27:30], 'Ry': r[:, 30],
                    'kept3d': r[:, 31] > 1})
    return out
