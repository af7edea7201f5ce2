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
    """Contiguous split [lo, hi) of `total` images for `rank`; the first total % world ranks get one more."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_records(n, cls, score, mproj, verts, bbox, topk, boxes=None):
    """(B,) counts + (B*topk, ...) slot tensors -> (B, topk, 32) fp32 records (device-side torch ops)."""
    B = n.shape[0]
    rec = torch.zeros(B, topk, RECORD, dtype=torch.float32, device=n.device)
    valid = torch.arange(topk, device=n.device)[None, :] < n[:, None].to(torch.int64)
    rec[..., 0] = cls.view(B, topk).to(torch.float32)
    rec[..., 1] = score.view(B, topk)
    rec[..., 2:4] = mproj.view(B, topk, 2)
    rec[..., 4:20] = verts.view(B, topk, 16)
    rec[..., 20:24] = bbox.view(B, topk, 4)
    flag = valid.to(torch.float32)
    if boxes is not None:
        kept = boxes.kept.view(B, topk) & valid
        rec[..., 24:27] = boxes.dimension.view(B, topk, 3).to(torch.float32)
        rec[..., 27:30] = boxes.location.view(B, topk, 3).to(torch.float32)
        rec[..., 30] = boxes.Ry.view(B, topk).to(torch.float32)
        flag = flag + kept.to(torch.float32)
    rec[..., 31] = flag
    return rec * (flag > 0).to(torch.float32)[..., None]


def all_gather_records(rec, group=None):
    """(b, topk, 32) per rank -> (world*b, topk, 32) ordered by global image index.  One collective."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return rec
    world = dist.get_world_size(group)
    out = torch.empty((world * rec.shape[0],) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, rec.contiguous(), group=group)
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
