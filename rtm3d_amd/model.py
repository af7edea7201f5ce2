"""Model facade with the reference's call surface (models/model.py:9-75) on top of
librtm3d_hip.so.

    model = create_model(cfg); model.to('cuda'); model.eval(); model.load_state_dict(sd)
    (clses, m_scores, m_projs, v_projs_regress, bboxes_2d), pred_logits = model(imgs)

``imgs`` is a float32 CUDA tensor (B, 3, H, W), H and W multiples of 32.  Everything numerical runs
in the HIP library; this file only owns the fp32 state dict, caches one recorded plan per input
shape and turns the fixed-size device outputs of the decode kernel into the reference's
per-image lists (``None`` for images without detections, models/model.py:33-44).
There is no CPU path: a CPU input tensor or a missing library raises.
"""
import ctypes
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from . import plan as plan_mod
from .weight_cache import WeightCache
from .weights import parse_backbone, state_dict_spec, synth_state_dict, head_table


GRAPH_MAX_BATCH = 8
MAX_PLANS = 3


class _Namespace(object):
    """Place-holder for the sub-module attributes the reference exposes (model.backbone, ...)."""
    def __init__(self, owner, prefix):
        self._owner, self._prefix = owner, prefix

    def state_dict(self):
        n = len(self._prefix) + 1
        return OrderedDict((k[n:], v) for k, v in self._owner._sd.items() if k.startswith(self._prefix + '.'))


class Detections(object):
    """Fixed-size device-resident results of one batch (no host synchronisation needed to make them)."""
    __slots__ = ('n', 'cls', 'score', 'mproj', 'verts', 'bbox', 'topk')

    def __init__(self, B, topk, device):
        self.topk = topk
        self.n = torch.zeros(B, dtype=torch.int32, device=device)
        self.cls = torch.zeros(B * topk, dtype=torch.int64, device=device)
        self.score = torch.zeros(B * topk, dtype=torch.float32, device=device)
        self.mproj = torch.zeros(B * topk, 2, dtype=torch.float32, device=device)
        self.verts = torch.zeros(B * topk, 8, 2, dtype=torch.float32, device=device)
        self.bbox = torch.zeros(B * topk, 4, dtype=torch.float32, device=device)


class Model(object):
    def __init__(self, config, backbone=None):
        self.config = config
        self._backbone_name = backbone if isinstance(backbone, str) else config.MODEL.BACKBONE
        parse_backbone(self._backbone_name)
        if int(config.MODEL.OUT_CHANNELS) != 256:
            raise NotImplementedError('the HIP kernels are built for OUT_CHANNELS=256 (both shipped configs), got %r' % (config.MODEL.OUT_CHANNELS,))
        # MODEL.HEADER_NUM_CONV (models/nets/header.py:12-13): one dilation-6 conv + (n - 1) dilation-1 convs per branch; any n >= 1
        self._num_conv = int(config.MODEL.HEADER_NUM_CONV)
        if self._num_conv < 1:
            raise ValueError('MODEL.HEADER_NUM_CONV must be >= 1, got %r' % (config.MODEL.HEADER_NUM_CONV,))
        # MODEL.KFNs names the backbone outputs the neck fuses (models/nets/keypoint_fpn_fusion.py:11-17).  The plan is built for
        # the four outputs every shipped config names; anything else is refused HERE instead of being silently ignored.  (The
        # reference itself only runs with all four, in order: its backbones always return four maps and KeypointFPNFusion._fpn
        # asserts len(x) == len(KFNs), keypoint_fpn_fusion.py:36.)
        want = ['layer1', 'layer2', 'layer3', 'layer4'] if 'RESNET' in self._backbone_name.upper() else ['level2', 'level3', 'level4', 'level5']
        kfns = config.MODEL.get('KFNs', None) if hasattr(config.MODEL, 'get') else getattr(config.MODEL, 'KFNs', None)
        if kfns is not None and [str(k) for k in kfns] != want:
            raise NotImplementedError('MODEL.KFNs = %r: the HIP plan fuses exactly the four backbone outputs %r of %s '
                                      '(other subsets / orders are not built)' % (list(kfns), want, self._backbone_name))
        # 'rtm3d' (reference main branch) | 'smoke' (head-table variant, SURVEY.md 8 a12: parity unpinned)
        self._head_variant = config.MODEL.get('HEAD_VARIANT', 'rtm3d') if hasattr(config.MODEL, 'get') else getattr(config.MODEL, 'HEAD_VARIANT', 'rtm3d')
        # one heat-map channel per class of cfg.DATASET.OBJs (models/nets/header.py:11)
        objs = getattr(getattr(config, 'DATASET', None), 'OBJs', None)
        self._num_classes = len(objs) if objs else 3
        self._head_channels = [c for _, _, c in head_table(self._head_variant, self._num_classes)]
        self.training = True                     # nn.Module default; detect.py:31 calls eval()
        self.export = False
        self._device = None
        self._plans = OrderedDict()
        self._ws = {}
        self._wcache = None
        self._verify = None                      # (shape key, VerifyPlanF32) of forward_logits_fp32
        self.use_graph = None                    # None: automatic (hipGraph replay for batches <= GRAPH_MAX_BATCH)
        # reference-style initial weights (utils/torch_utils.py:71-83); replaced by load_state_dict
        self._sd = synth_state_dict(self._backbone_name, seed=0, style='init', head_variant=self._head_variant,
                                    num_classes=self._num_classes, header_num_conv=self._num_conv)
        self.backbone = _Namespace(self, 'backbone')
        self.kfpn_fusion = _Namespace(self, 'kfpn_fusion')
        self.detect_header = _Namespace(self, 'detect_header')

    # ------------------------------------------------------------------ nn.Module-like surface
    def state_dict(self):
        return OrderedDict(self._sd)

    def load_state_dict(self, state_dict, strict=True):
        want = OrderedDict((k, shape) for k, shape, _, _ in state_dict_spec(self._backbone_name, self._head_variant, self._num_classes, self._num_conv))
        missing = [k for k in want if k not in state_dict]
        unexpected = [k for k in state_dict if k not in want]
        if strict and (missing or unexpected):
            raise RuntimeError('Error(s) in loading state_dict: missing keys %s, unexpected keys %s' % (missing[:5], unexpected[:5]))
        new = OrderedDict(self._sd)
        for k, shape in want.items():
            if k in state_dict:
                v = torch.as_tensor(state_dict[k]).detach().cpu()
                if tuple(v.shape) != tuple(shape):
                    raise RuntimeError('size mismatch for %s: %s vs %s' % (k, tuple(v.shape), tuple(shape)))
                new[k] = v.to(torch.int64) if k.endswith('num_batches_tracked') else v.to(torch.float32).contiguous()
        self._sd = new
        self._wcache = None              # folded / packed weights belong to the old state dict
        self._verify = None
        self._drop_plans()
        return None

    def _drop_plans(self):
        for p in self._plans.values():
            p.close()
        self._plans = OrderedDict()

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def to(self, device=None, *args, **kwargs):
        if device is not None and not isinstance(device, torch.dtype):
            d = torch.device(device)
            if d.type != 'cuda':
                raise RuntimeError('rtm3d_amd.Model runs on an AMD GPU only (got device %s); there is no CPU path' % d)
            self._device = torch.device('cuda', d.index if d.index is not None else torch.cuda.current_device())
        return self

    def cuda(self, device=None):
        return self.to(torch.device('cuda', device) if isinstance(device, int) else (device or 'cuda'))

    def float(self):
        return self

    def parameters(self):
        return (v for k, v in self._sd.items() if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked')))

    def modules(self):
        return iter([self])

    def __call__(self, x):
        return self.forward(x)

    # ------------------------------------------------------------------ hot path
    def _plan_for(self, B, H, W, device, heads='dense'):
        """heads = 'dense': all branches on the whole map (Model.forward); 'peaks': the heat map alone plus the patch plan that
        evaluates the regression branches at the detected peaks (detect3d(sparse_heads=True), csrc/sparse_heads.hip)."""
        if heads not in ('dense', 'peaks'):
            raise ValueError("heads must be 'dense' or 'peaks'")
        if heads == 'peaks' and self._head_variant not in (None, 'rtm3d'):
            raise NotImplementedError('peaks-only regression heads exist for the rtm3d head table')
        key = (B, H, W, device.index) if heads == 'dense' else (B, H, W, device.index, heads)
        p = self._plans.get(key)
        if p is not None:
            self._plans[key] = self._plans.pop(key)          # most recently used last
        if p is None:
            # a plan owns its activation workspace (about 9 GB at bs=32) and a packed copy of the weights: keep the
            # MAX_PLANS most recently used shapes (a trailing partial batch, or the reference DatasetReader's
            # rectangular mode, otherwise keeps adding contexts until hipMalloc fails) and free the rest
            while len(self._plans) >= MAX_PLANS:
                old = next(iter(self._plans))
                self._plans.pop(old).close()
            if self._wcache is None:
                self._wcache = WeightCache(self._sd)
            ir = plan_mod.build_plan(self._sd, self._backbone_name, B, H, W, self._head_variant, cache=self._wcache,
                                     num_classes=self._num_classes, dense_heads=1 if heads == 'peaks' else None,
                                     header_num_conv=self._num_conv)
            with torch.cuda.device(device):
                p = plan_mod.RealizedPlan(ir, device.index)
                # small batches are bound by launch gaps, not by the kernels: replay those plans as one hipGraph
                use_graph = self.use_graph if self.use_graph is not None else B <= GRAPH_MAX_BATCH
                p.set_graph(use_graph)
                if heads == 'peaks':
                    topk = int(self.config.DETECTOR.TOPK_CANDIDATES)
                    pir = plan_mod.build_peak_plan(self._sd, B * topk, (H // 4, W // 4), self._head_variant, cache=self._wcache,
                                                   num_classes=self._num_classes)
                    p.peak = plan_mod.RealizedPlan(pir, device.index)
                    p.peak.set_graph(False)
                    # the regression logits at the peaks, [slots][16] and [slots][2] fp32 (the patch plan's two outputs)
                    p.peak_out = [torch.zeros(B * topk, c, 1, 1, dtype=torch.float32, device=device) for c in pir.head_channels]
            self._plans[key] = p
            self._wcache.save()          # no-op unless RTM3D_WEIGHT_CACHE_DIR is set
        return p

    def _check_input(self, x):
        if not isinstance(x, torch.Tensor) or x.dim() != 4 or x.shape[1] != 3:
            raise ValueError('expected a (B, 3, H, W) tensor')
        if not x.is_cuda:
            raise RuntimeError('rtm3d_amd.Model.forward needs a CUDA (ROCm) tensor; there is no CPU path')
        if x.dtype != torch.float32:
            x = x.float()
        return x.contiguous()

    def input_tensor(self, B, H, W, device=None, heads='dense'):
        """(device address, border) of the fp16 NHWC4 input tensor of the plan for (B, H, W): the target of
        ``rtm3d_amd.preprocess.preprocess_batch(..., model=self)``.  heads: which plan ('dense' / 'peaks': they own separate
        workspaces) the following ``forward_logits(None, preloaded=..., heads=...)`` will replay."""
        dev = torch.device(device) if device is not None else self._device
        if dev is None or dev.type != 'cuda':
            raise RuntimeError('rtm3d_amd.Model.input_tensor needs a CUDA (ROCm) device; call model.to("cuda") first')
        dev = torch.device('cuda', dev.index if dev.index is not None else torch.cuda.current_device())
        return self._plan_for(B, H, W, dev, heads).input_tensor()

    def forward_logits(self, x, preloaded=None, out=None, heads='dense'):
        """backbone -> neck -> heads: the four fp32 NCHW logit maps (models/model.py:21-23).
        preloaded=(B, H, W): ``x`` is None and the plan's input tensor was filled by preprocess_batch(model=self).
        out: where the logits go instead of four fresh tensors - a tuple of contiguous fp32 CUDA tensors of the logit
        shapes, or ``'reuse'`` for one set of buffers owned by the plan of this shape (every such call returns the SAME
        tensors: consume them, in stream order, before the next call).  A hipGraph replay is keyed by these addresses
        (rtm3d_forward), so a loop that holds on to its fresh outputs would pay a capture per call; with out= it replays one
        graph (the bs=1 detect.py loop).
        heads='peaks': only the heat-map branch is evaluated (a 1-tuple comes back); feed it to ``decode2d_sparse``."""
        if preloaded is not None:
            if x is not None:
                raise ValueError('forward_logits: pass x=None with preloaded=(B, H, W)')
            B, H, W = (int(v) for v in preloaded)
            dev = self._device
            if dev is None:
                raise RuntimeError('forward_logits(preloaded=...): call model.to("cuda") first')
            xptr = 0
        else:
            x = self._check_input(x)
            B, _, H, W = x.shape
            dev = x.device
            xptr = x.data_ptr()
        plan = self._plan_for(B, H, W, dev, heads)
        with torch.cuda.device(dev):
            shapes = [(B, c, H // 4, W // 4) for c in (self._head_channels if heads == 'dense' else self._head_channels[:1])]
            if out is None:
                outs = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in shapes]
            elif isinstance(out, str):
                if out != 'reuse':
                    raise ValueError("forward_logits: out must be a tuple of tensors or 'reuse'")
                outs = getattr(plan, 'logit_buffers', None)
                if outs is None:
                    outs = plan.logit_buffers = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in shapes]
            else:
                outs = list(out)
                if len(outs) != len(shapes):
                    raise ValueError('forward_logits: out needs %d tensors, got %d' % (len(shapes), len(outs)))
                for t, sh in zip(outs, shapes):
                    if not isinstance(t, torch.Tensor) or tuple(t.shape) != sh or t.dtype != torch.float32 or not t.is_cuda \
                            or (dev.index is not None and t.device.index != dev.index) or not t.is_contiguous():
                        raise ValueError('forward_logits: out tensors must be contiguous fp32 %s on %s' % (shapes, dev))
            ptrs = [o.data_ptr() for o in outs] + [0] * (4 - len(outs))
            plan.forward(torch.cuda.current_stream(dev).cuda_stream, xptr, ptrs)
            # which heat map the fused map z inside this plan belongs to (decode2d_sparse refuses any other)
            plan.z_stamp = (getattr(plan, 'z_stamp', (0, 0))[0] + 1, outs[0].data_ptr())
        return tuple(outs)

    def forward_logits_fp32(self, x):
        """VERIFICATION mode (SURVEY.md H2 ii): the same recorded plan executed on fp32 tensors with fp32 weights and fp64
        accumulation by ``rtm3d_amd.verify.VerifyPlanF32`` (simple HIP kernels, about 100x slower than forward_logits).
        Feeding these logits to ``decode2d`` / ``decode3d_slots`` compares the device path with the reference's fp32 CPU path
        without the fp16 storage error; ``forward`` never takes this path.  One executor is kept (the last shape used): its
        activation workspace is twice the fp16 plan's, so verify on a few images, not on the benchmark batch."""
        from .verify import VerifyPlanF32
        x = self._check_input(x)
        B, _, H, W = x.shape
        key = (B, H, W, x.device.index)
        if self._verify is None or self._verify[0] != key:
            self._verify = None                     # free the previous executor's buffers first
            if self._wcache is None:
                self._wcache = WeightCache(self._sd)
            ir = plan_mod.build_plan(self._sd, self._backbone_name, B, H, W, self._head_variant, cache=self._wcache,
                                     num_classes=self._num_classes, header_num_conv=self._num_conv)
            self._verify = (key, VerifyPlanF32(ir, x.device))
        return self._verify[1].forward(x, self._head_channels)

    def check_range(self, x, strict=False):
        """fp16 RANGE REPORT (verification mode; the product path is untouched).  The reference computes in fp32
        (models/model.py:20-27); this build stores every activation and folded weight as fp16 (|x| <= 65504) without a clamp, so
        a checkpoint / input whose activations leave that range yields inf or NaN logits in ``forward``.  ``check_range`` runs the
        recorded plan once in fp32 on ``x`` (``forward_logits_fp32``) and returns ``VerifyPlanF32.range_report()``: one row per
        written tensor slice and per folded weight array with its largest |value|, the head-room to 65504 and an ``overflow``
        flag, largest first.  ``strict=True`` raises ``OverflowError`` naming the first offending tensor in plan order."""
        self.forward_logits_fp32(x)
        B, _, H, W = x.shape
        rows = self._verify[1].range_report(realized=self._plan_for(B, H, W, x.device))
        if strict:
            bad = sorted([r for r in rows if r['overflow']], key=lambda r: r['order'])       # the first one in plan order: the cause
            if bad:
                raise OverflowError('fp16 range exceeded by %s of op %s: max |x| = %.4g > 65504 (%d tensors in all)'
                                    % (bad[0]['tensor'], bad[0]['op'], bad[0]['max_abs'], len(bad)))
        return rows

    def release_verify(self):
        """Free the fp32 verification executor's device buffers."""
        self._verify = None

    def forward(self, x):
        pred_logits = self.forward_logits(x)
        if self.training:
            # the reference returns the raw logits for its loss (models/model.py:24-25)
            return pred_logits
        return self.inference(pred_logits), pred_logits

    def decode2d_sparse(self, heat_logits, out=None, from_forward=None):
        """2D decode with the regression branches evaluated at the detected peaks only (what Model.inference reads of
        them: models/model.py:47-50,124-128): peaks from the dense heat map (NMS + top-k, identical to decode2d's), the z
        samples each peak depends on gathered into patches, the three head convs of branches offset_fr_main / main_offset as a
        patch plan on the MFMA conv kernels, then the sub-pixel / vertex arithmetic of decode2d.  heat_logits: the 1-tuple of
        ``forward_logits(x, heads='peaks')`` (its plan still holds the fused map z of that forward: call this next, on the
        same stream).  Same Detections as decode2d on the dense logits, vertices to fp16 round-off of the network.
        from_forward: when heat_logits is an edited COPY of what forward_logits returned (tests plant peaks), the original
        tuple - the freshness check below is made on it."""
        if self._num_conv != 2:
            # the patch plan's windows (15 -> 5 -> 3 -> 1) are the receptive field of d6 + d1 + the logit conv, plan.build_peak_plan
            raise NotImplementedError('peaks-only regression heads are built for MODEL.HEADER_NUM_CONV = 2 (got %d): use the dense heads' % self._num_conv)
        hm = heat_logits[0] if isinstance(heat_logits, (tuple, list)) else heat_logits
        hm_src = hm if from_forward is None else (from_forward[0] if isinstance(from_forward, (tuple, list)) else from_forward)
        B, _, Hm, Wm = hm.shape
        dev = hm.device
        plan = self._plan_for(B, 4 * Hm, 4 * Wm, dev, 'peaks')
        # the patches are gathered from the fused map z INSIDE the plan: it must still be the z of the forward that produced
        # `hm`.  A plan that was evicted and rebuilt (MAX_PLANS) holds no z at all, and another forward of this shape in between
        # has overwritten it - both would give plausible but wrong vertices, silently
        stamp = getattr(plan, 'z_stamp', None)
        if stamp is None or stamp[1] != hm_src.data_ptr() or tuple(hm_src.shape) != tuple(hm.shape):
            raise RuntimeError('decode2d_sparse: the plan of this shape does not hold the fused map of these heat-map logits (%s); call '
                               "forward_logits(x, heads='peaks') and decode2d_sparse on its result back to back"
                               % ('the plan was rebuilt since' if stamp is None else 'another forward of this shape ran in between'))
        det = self.decode2d((hm,), out=out, peaks_only=True)
        lib = _lib.load()
        topk = det.topk
        zbase, zB, zH, zW, zC, zP = plan.tensor_info(plan.plan.named['z'])
        pbase, pslots, pS, _, pC, pP = plan.peak.tensor_info(plan.peak.plan.named['zp'])
        if pC != 256 or pP != 0:
            raise RuntimeError('decode2d_sparse: the patch plan input must be a borderless 256-channel tensor')
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.rtm3d_gather_peak_patches(ctypes.c_void_p(stream), ctypes.c_void_p(zbase), zH, zW, zC, zP, B, topk,
                                                     det.n.data_ptr(), det.mproj.data_ptr(), ctypes.c_void_p(pbase),
                                                     ctypes.c_void_p(plan.peak.blob_address(plan.peak.yx_blob)),
                                                     pslots, pS, plan.peak.blob_bytes(plan.peak.yx_blob)), 'gather_peak_patches')
            plan.peak.forward(stream, 0, [t.data_ptr() for t in plan.peak_out] + [0, 0])
            _lib.check(lib.rtm3d_decode2d_finish(ctypes.c_void_p(stream), B, topk, det.n.data_ptr(), plan.peak_out[0].data_ptr(),
                                                 plan.peak_out[1].data_ptr(), float(self.config.MODEL.DOWN_SAMPLE),
                                                 det.mproj.data_ptr(), det.verts.data_ptr(), det.bbox.data_ptr()), 'decode2d_finish')
        return det

    def decode2d(self, pred_logits, out=None, peaks_only=None):
        """Fixed-size device results of the 2D decode (no host sync).  peaks_only: classes, scores and integer key points
        alone (default: the 'smoke' head table)."""
        smoke = self._head_variant == 'smoke' if peaks_only is None else bool(peaks_only)
        used = (pred_logits[0],) if smoke else (pred_logits[0], pred_logits[1], pred_logits[2])
        for t in used:
            if not t.is_cuda:
                raise RuntimeError('rtm3d_amd.Model.inference needs CUDA (ROCm) tensors; there is no CPU path')
        used = [t.contiguous().float() for t in used]
        main_kf = used[0]
        offs_ptr, moff_ptr = (0, 0) if smoke else (used[1].data_ptr(), used[2].data_ptr())   # NULL, NULL = peaks only
        B, K, H, W = main_kf.shape
        topk = int(self.config.DETECTOR.TOPK_CANDIDATES)
        dev = main_kf.device
        lib = _lib.load()
        with torch.cuda.device(dev):
            if out is None:
                out = Detections(B, topk, dev)
            elif out.n.shape[0] != B or out.topk != topk or out.cls.shape[0] != B * topk:
                # the kernel writes B*topk slots: a smaller buffer would be overrun, a larger one would keep stale rows
                raise ValueError('decode2d: output slots are for %d images x top-%d, the logits hold %d images and topk is %d'
                                 % (out.n.shape[0], out.topk, B, topk))
            key = ('d2', B, K, H, W, dev.index)
            ws = self._ws.get(key)
            if ws is None:
                ws = torch.empty(int(lib.rtm3d_decode2d_workspace_bytes(B, K, H, W)), dtype=torch.uint8, device=dev)
                self._ws[key] = ws
            _lib.check(lib.rtm3d_decode2d(ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream), main_kf.data_ptr(),
                                          offs_ptr, moff_ptr, B, K, H, W,
                                          float(self.config.DETECTOR.SCORE_THRESH), topk, float(self.config.MODEL.DOWN_SAMPLE),
                                          ws.data_ptr(), out.n.data_ptr(), out.cls.data_ptr(), out.score.data_ptr(),
                                          out.mproj.data_ptr(), out.verts.data_ptr(), out.bbox.data_ptr()), 'decode2d')
        return out

    def detect(self, x):
        """What ``detect.py:56`` takes from the model - ``preds = model(imgs)[0]``: the five per-image lists (classes, scores,
        main key points, 8 vertices, 2D boxes; ``None`` where nothing was kept) - without the dense regression maps that call
        never looks at: the heat map is computed on the whole image, the regression branches at the detected peaks
        (decode2d_sparse).  Same lists as ``model(x)[0]`` up to fp16 round-off of the vertices (<= 1.4e-3 px measured)."""
        if self._head_variant not in (None, 'rtm3d'):
            return self.forward(x)[0]
        return self._lists(self.decode2d_sparse(self.forward_logits(x, heads='peaks')))

    def inference(self, pred_logits):
        """Model.inference (models/model.py:29-75): per-image lists, ``None`` where nothing was kept."""
        det = self.decode2d(pred_logits)
        if self._head_variant == 'smoke':
            n = det.n.cpu().tolist()
            return self._inference_smoke(pred_logits, det, n)
        return self._lists(det)

    def _lists(self, det):
        B, topk = det.n.shape[0], det.topk
        n = det.n.cpu().tolist()                 # the only host synchronisation of the path
        clses, m_scores, m_projs, v_projs_regress, bboxes_2d = ([None] * B for _ in range(5))
        for i in range(B):
            if n[i] == 0:
                continue
            s = slice(i * topk, i * topk + n[i])
            clses[i], m_scores[i], m_projs[i] = det.cls[s], det.score[s], det.mproj[s]
            v_projs_regress[i], bboxes_2d[i] = det.verts[s], det.bbox[s]
        return clses, m_scores, m_projs, v_projs_regress, bboxes_2d

    def _inference_smoke(self, pred_logits, det, n):
        """Head-table variant: per-image (classes, scores, key points x down_sample, 8 regression values)."""
        B, topk = det.n.shape[0], det.topk
        reg = pred_logits[1]
        down = float(self.config.MODEL.DOWN_SAMPLE)
        clses, m_scores, m_projs, regs = ([None] * B for _ in range(4))
        for i in range(B):
            if n[i] == 0:
                continue
            s = slice(i * topk, i * topk + n[i])
            xy = det.mproj[s]
            clses[i], m_scores[i], m_projs[i] = det.cls[s], det.score[s], down * xy
            regs[i] = reg[i][:, xy[:, 1].long(), xy[:, 0].long()].t().contiguous()
        return clses, m_scores, m_projs, regs

    def forward_dict(self, x):
        """Dict view of the eval outputs (offered in addition to the reference tuple)."""
        dets, logits = self.eval().forward(x)
        if self._head_variant == 'smoke':
            c, s, m, r = dets
            return {'clses': c, 'm_scores': s, 'm_projs': m, 'regression': r, 'main_kf_logits': logits[0], 'regression_logits': logits[1]}
        c, s, m, v, b = dets
        return {'clses': c, 'm_scores': s, 'm_projs': m, 'v_projs_regress': v, 'bboxes_2d': b,
                'main_kf_logits': logits[0], 'offset_fr_main_logits': logits[1], 'main_offset_logits': logits[2],
                'vertex_offset_logits': logits[3]}

    # ------------------------------------------------------------------ fused device pipeline
    def detect3d(self, x, K_per_image, dim_ref=None, ref_loc=(0.0, -0.5, 20.0), fp32_verify=False, sparse_heads=False, solver_form=None):
        """forward + 2D decode + 3D decode, all stream-ordered on the device (no host sync).
        K_per_image: (B, 9) float64 CUDA tensor.  Returns (Detections, Boxes3D, logits).
        fp32_verify=True: the network runs in the fp32 verification mode (forward_logits_fp32), the decode kernels are the
        product's own.  sparse_heads=True: the detect.py call surface never reads the dense regression maps, so only the heat
        map is computed densely and the regression branches at the detected peaks (decode2d_sparse); `logits` is then the
        heat map alone.  solver_form: 'direct' | 'published' (model_utils.solver_form_id), None = the default."""
        from .model_utils import decode3d_slots, decode_smoke_slots
        if sparse_heads:
            if fp32_verify:
                raise ValueError('detect3d: sparse_heads and fp32_verify are separate modes')
            logits = self.forward_logits(x, heads='peaks')
            det = self.decode2d_sparse(logits)
        else:
            logits = self.forward_logits_fp32(x) if fp32_verify else self.forward_logits(x)
            det = self.decode2d(logits)
        dim_ref = dim_ref if dim_ref is not None else self.config.DETECTOR.dim_ref
        if len(dim_ref) < self._num_classes:
            # the reference would raise IndexError at dim_ref[cls] (utils/model_utils.py:293); the kernel only clamps
            raise IndexError('dim_ref has %d rows for %d classes' % (len(dim_ref), self._num_classes))
        if self._head_variant == 'smoke':
            boxes = decode_smoke_slots(det, logits[1], K_per_image, dim_ref, float(self.config.MODEL.DOWN_SAMPLE))
        else:
            boxes = decode3d_slots(det, K_per_image, dim_ref, ref_loc, form=solver_form)
        return det, boxes, logits
