"""Stream-level pipelining of the hot path on one GPU.

The 3D decode is latency-bound fp64 work on a few hundred wavefronts; the network is MFMA/HBM-bound
work on thousands of workgroups.  Running them on two HIP streams lets the decode of batch i fill
idle issue slots while batch i+1's backbone is already running:

    main stream : forward(i) -> decode2d(i) -> [event A_i] ............. forward(i+1) -> ...
    side stream :                               wait A_i -> decode3d(i) -> pack -> all-gather(i) -> [event B_i]

Outputs are multi-buffered (slot i % depth); the main stream waits for B_{i-depth} before decode2d
overwrites slot i % depth.  No host synchronisation inside; `results(i)` hands out the records of a
finished step after waiting on B_i.

Small batches (<= SMALL_BATCH images): the solve of a handful of objects is ONE wavefront per object walking a serial
fp64 iteration (1.46 ms for 15 objects at bs=1 when this was measured, ~1.0 ms now; profiles/r02_small_batch.txt) - as long as the whole network
(1.27 ms).  A single side stream would serialise the decodes of consecutive steps and bound the step at that latency, so
small batches use depth 3 and one side stream per slot: decode3d(i) and decode3d(i+1) overlap each other and the step
is bound by the network again.
"""
import os

import numpy as np
import torch

from .model import Detections
from .model_utils import Boxes3D, decode3d_slots
from . import distributed as rdist


SMALL_BATCH = 2      # batches whose 3D decode (a serial fp64 iteration per object, ~1 ms) is about as long as the network


class Detect3DPipeline(object):
    def __init__(self, model, batch, device, dim_ref=None, ref_loc=(0.0, -0.5, 20.0), gather=True, depth=None, decode3d=True,
                 side_cus=0, side_streams=None, sparse_heads=False, solver_form=None):
        self.model, self.B, self.dev = model, batch, torch.device(device)
        # sparse_heads: this call surface hands out detection records, never the dense logits, so the regression branches are
        # evaluated at the detected peaks only (Model.decode2d_sparse); the records agree with the dense path's to fp16 round-off
        self.sparse_heads = bool(sparse_heads)
        # search direction of the 3D decode: 'direct' | 'published' (model_utils.solver_form_id); checked here, not at the first submit
        from .model_utils import solver_form_id, DEFAULT_SOLVER_FORM
        solver_form_id(solver_form)
        self.solver_form = DEFAULT_SOLVER_FORM if solver_form is None else solver_form
        self.heads = 'peaks' if self.sparse_heads else 'dense' 
        self.topk = int(model.config.DETECTOR.TOPK_CANDIDATES)
        dim_ref = dim_ref if dim_ref is not None else model.config.DETECTOR.dim_ref
        if len(dim_ref) < getattr(model, '_num_classes', 0):
            raise IndexError('dim_ref has %d rows for %d classes' % (len(dim_ref), model._num_classes))
        # device-resident constants: a pageable host->device copy inside submit() would block the host
        self.dim_ref = torch.as_tensor(np.asarray(dim_ref, np.float64), device=self.dev)
        self.ref_loc = torch.as_tensor(np.asarray(ref_loc, np.float64), device=self.dev)
        self.gather = gather
        self.decode3d = decode3d          # False: diagnostic only (measures what the 3D decode costs the pipeline)
        small = batch <= SMALL_BATCH
        depth = depth if depth else (3 if small else 2)
        self.depth = depth
        n_side = side_streams if side_streams else (depth if small else 1)
        if not 1 <= n_side <= depth:
            raise ValueError('side_streams must be in 1..depth (%d), got %d' % (depth, n_side))
        with torch.cuda.device(self.dev):
            self.sides = [self._make_side_stream(side_cus, n_side) for _ in range(n_side)]
            self.det = [Detections(batch, self.topk, self.dev) for _ in range(depth)]
            self.boxes = [Boxes3D(batch * self.topk, self.dev) for _ in range(depth)]
            self.ev_a = [torch.cuda.Event() for _ in range(depth)]
            self.ev_b = [torch.cuda.Event() for _ in range(depth)]
            self.rec_local = [torch.zeros(batch, self.topk, rdist.RECORD, dtype=torch.float32, device=self.dev) for _ in range(depth)]
            # per-slot copies of everything the side stream reads or writes, allocated ONCE on the construction stream: the
            # gathered records (no allocation on the side stream per step, no side-stream block handed to main-stream
            # consumers) and the intrinsics (the caller may drop or overwrite its K tensor right after submit())
            gathers = bool(gather) and torch.distributed.is_available() and torch.distributed.is_initialized() and \
                (gather == 'always' or torch.distributed.get_world_size() > 1)
            self.rec_all = [rdist.gathered_buffer(self.rec_local[0]) for _ in range(depth)] if gathers else None
            self.K_slot = [torch.zeros(batch, 9, dtype=torch.float64, device=self.dev) for _ in range(depth)]
            # optional event pair around the collective (bench diagnostics; timing events are not free, so off by default)
            self.time_gather = False
            self.ev_g0 = [torch.cuda.Event(enable_timing=True) for _ in range(depth)]
            self.ev_g1 = [torch.cuda.Event(enable_timing=True) for _ in range(depth)]
        self.rec = [None] * depth
        self.count = 0

    def _make_side_stream(self, side_cus, n_side=1):
        """Side stream confined to `side_cus` CUs: the decode's long-lived waves stay off the CUs that the
        256x256-tile convolutions of the next batch need whole (they cannot co-reside: VGPR/LDS)."""
        if side_cus and side_cus > 0:
            import ctypes
            from . import _lib
            lib = _lib.load()
            h = ctypes.c_void_p()
            _lib.check(lib.rtm3d_stream_create_cumask(self.dev.index, int(side_cus), ctypes.byref(h)), 'stream_create_cumask')
            self._side_handle = h
            return torch.cuda.ExternalStream(h.value, device=self.dev)
        # one side stream: high priority, so the decode's few waves are not queued behind the next batch's thousands of
        # workgroups.  Several side streams (small batches): default priority - three high-priority streams next to a graph
        # replay measured 4.3 ms per bs=1 step against 1.36 ms at priority 0 (profiles/r02_small_batch.txt)
        prio = os.environ.get('RTM3D_SIDE_PRIO')
        prio = int(prio) if prio is not None else (-1 if n_side == 1 else 0)
        return torch.cuda.Stream(device=self.dev, priority=prio)

    def submit_uint8(self, images, K_per_image, size, resize_to=None):
        """The same step fed by camera images: ``images`` = list of B uint8 (h, w, 3) CUDA tensors of any
        sizes; Resize (longest side -> resize_to) + letterbox into the (H, W) canvas + normalise run as two launches that
        write the network's own fp16 NHWC4 input tensor (rtm3d_amd.preprocess.preprocess_batch), then the plan is replayed
        on it.  K_per_image must already carry the Resize / padding bookkeeping (preprocess.resize_K / adjust_K)."""
        from . import preprocess
        if len(images) != self.B:
            raise ValueError('Detect3DPipeline was built for batches of %d images, got %d' % (self.B, len(images)))
        cfg = self.model.config
        H, W = int(size[0]), int(size[1])

        def feed():
            preprocess.preprocess_batch(images, (H, W), cfg.DATASET.MEAN, cfg.DATASET.STD, resize_to=resize_to, model=self.model, heads=self.heads)
            return self.model.forward_logits(None, preloaded=(self.B, H, W), out='reuse', heads=self.heads)
        return self._submit(feed, K_per_image)

    def submit(self, x, K_per_image):
        """Enqueue one batch; returns its step index.  Asynchronous."""
        if x.dim() != 4 or x.shape[0] != self.B:
            # the slots are preallocated for `batch` images: a larger shard would write out of bounds on the device, a
            # smaller one would leave the previous step's detections in the unused rows (pad the last shard instead:
            # rtm3d_amd.distributed.padded_shard)
            raise ValueError('Detect3DPipeline was built for batches of %d images, got input of shape %s'
                             % (self.B, tuple(x.shape)))
        return self._submit(lambda: self.model.forward_logits(x, out='reuse', heads=self.heads), K_per_image)

    def _submit(self, run_network, K_per_image):
        if not isinstance(K_per_image, torch.Tensor) or K_per_image.numel() != self.B * 9 or not K_per_image.is_cuda:
            raise ValueError('K_per_image must be a CUDA tensor with %d x 9 intrinsics' % self.B)
        i = self.count
        s = i % self.depth
        main = torch.cuda.current_stream(self.dev)
        if i >= self.depth:
            main.wait_event(self.ev_b[s])                 # slot s is free again
        logits = run_network()
        if self.sparse_heads:
            self.model.decode2d_sparse(logits, out=self.det[s])
        else:
            self.model.decode2d(logits, out=self.det[s])
        self.K_slot[s].copy_(K_per_image.reshape(self.B, 9), non_blocking=True)     # main stream, ordered before A_s
        K_per_image = self.K_slot[s]
        self.ev_a[s].record(main)
        side = self.sides[s % len(self.sides)]
        with torch.cuda.stream(side):
            side.wait_event(self.ev_a[s])
            if self.decode3d:
                decode3d_slots(self.det[s], K_per_image, self.dim_ref, self.ref_loc, out=self.boxes[s], form=self.solver_form)
            d = self.det[s]
            rec = rdist.pack_records(d.n, d.cls, d.score, d.mproj, d.verts, d.bbox, self.topk,
                                     self.boxes[s] if self.decode3d else None, out=self.rec_local[s])
            if self.time_gather:
                self.ev_g0[s].record(side)
            self.rec[s] = rdist.all_gather_records(rec, always=self.gather == 'always',
                                                   out=self.rec_all[s] if self.rec_all is not None else None) if self.gather else rec
            if self.time_gather:
                self.ev_g1[s].record(side)
            self.ev_b[s].record(side)
        self.count += 1
        return i

    def results(self, i, copy=False):
        """(world*B, topk, 32) records of step i (waits for it on the current stream).
        LIFETIME: the tensor is the slot's preallocated buffer, not a fresh one: it is overwritten by step i + depth (with
        world > 1: by that step's all-gather on a side stream).  Read it - on the stream that called results(), which is the one
        ordered behind the step - before `depth` further submits, or pass copy=True for a private clone made on that stream."""
        s = i % self.depth
        torch.cuda.current_stream(self.dev).wait_event(self.ev_b[s])
        return self.rec[s].clone() if copy else self.rec[s]

    def gather_us(self, i):
        """Device time of step i's collective in microseconds (needs time_gather = True before the step; waits for it)."""
        s = i % self.depth
        self.ev_g1[s].synchronize()
        return self.ev_g0[s].elapsed_time(self.ev_g1[s]) * 1e3

    def drain(self):
        for side in self.sides:
            side.synchronize()
        torch.cuda.current_stream(self.dev).synchronize()
