"""CPU: the host-side plan (graph wiring without torch.cat, BN folding, deconv phases, weight
packing) reproduces the oracle; and fp16 storage keeps logits within the stated tolerance."""
import os

import numpy as np
import pytest
import torch

from oracle import rtm3d_ref
from rtm3d_amd import plan as plan_mod, weights
from tests.plan_interp import run_plan


@pytest.mark.parametrize('bb', ['DLA-34', 'RESNET-18'])
def test_plan_matches_oracle_fp32(bb):
    sd = weights.synth_state_dict(bb, 3, 'trained')
    x = weights.synth_images(2, 64, 128, seed=5)
    P = plan_mod.build_plan(sd, bb, 2, 64, 128)
    outs, fetch = run_plan(P, x)
    dets, logits, st = rtm3d_ref.model_forward(x, sd, bb, return_stages=True)
    for i in range(4):
        np.testing.assert_allclose(fetch(P.named['feat%d' % i]).numpy(), st['feats'][i].numpy(), atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(fetch(P.named['z']).numpy(), st['z'].numpy(), atol=2e-4, rtol=1e-4)
    for a, b in zip(outs, logits):
        np.testing.assert_allclose(a.numpy(), b.numpy(), atol=5e-4, rtol=1e-4)


@pytest.mark.parametrize('nconv', [1, 3])
def test_plan_header_depths_match_oracle(nconv):
    """MODEL.HEADER_NUM_CONV = 1 / 3 (models/nets/header.py:12-13): the recorded plan - one fused dilation-6 conv, n - 1 grouped
    dilation-1 convs ping-ponging between two tensors, the logit convs reading the last one - is the oracle's function (which the
    reference-run fixtures e2e_dla34_small_nc{1,3}.npz pin); op names as the bench / profiles expect them."""
    bb = 'RESNET-18'
    sd = weights.synth_state_dict(bb, 3, 'trained', header_num_conv=nconv)
    assert ('detect_header.main_kf_header.%d.weight' % (3 * (nconv - 1))) in sd and ('detect_header.main_kf_header.%d.weight' % (3 * nconv)) not in sd
    x = weights.synth_images(1, 64, 128, seed=5)
    P = plan_mod.build_plan(sd, bb, 1, 64, 128, header_num_conv=nconv)
    names = [o['name'] for o in P.ops if o['name'].startswith('heads.')]
    assert names == ['heads.conv_d6'] + ['heads.conv_d1' if k == 1 else 'heads.conv_d1_%d' % k for k in range(1, nconv)] + ['heads.out_convs'], names
    outs, _ = run_plan(P, x)
    _, logits = rtm3d_ref.model_forward(x, sd, bb)
    for a, b in zip(outs, logits):
        np.testing.assert_allclose(a.numpy(), b.numpy(), atol=5e-4, rtol=1e-4)


@pytest.mark.parametrize('bb', ['DLA-34', 'RESNET-18'])
def test_realize_level_rewrites_keep_the_function(bb):
    """Round 4: the rewrites RealizedPlan applies when it records a plan - a DLA block's `project` 1x1 as extra K-steps of the block's
    second conv (rtm3d_conv_desc.tap_dc), the neck's proj+head 1x1 folded into the transposed conv in front of it with the feature read
    from a space-to-depth copy - are exact in real arithmetic.  Checked WITHOUT a GPU: the rewritten op list (the same builders the
    device path uses: _project_folds / _folded_conv / _neck_up_folds / _neck_fold_conv) through the CPU interpreter against the
    original plan: fused map and logits to fp32 round-off; every fold really present."""
    import copy
    sd = weights.synth_state_dict(bb, 3, 'trained')
    x = weights.synth_images(1, 64, 128, seed=5)
    P = plan_mod.build_plan(sd, bb, 1, 64, 128)
    outs0, fetch0 = run_plan(P, x)
    v2_min = plan_mod.V2_MIN_TILES
    plan_mod.V2_MIN_TILES = 1                   # (the neck fold is gated on a launch size the 64 x 128 test image does not reach)
    try:
        R = plan_mod.RealizedPlan.rewrites_only(P)
        tail = R._level_tail_chains()
        nfold = R._neck_up_folds(tail)
        folds = R._project_folds()
        R._plan_s2d_only(nfold, tail)
    finally:
        plan_mod.V2_MIN_TILES = v2_min
    if bb == 'DLA-34':
        assert len(folds) == 3 and len(tail) == 1 and len(nfold) == 3, (folds, tail, nfold)
    else:
        assert len(folds) == 0 and len(nfold) >= 1, (folds, nfold)          # (ResNet's downsample 1x1s are stride 2: not foldable)
    # rewritten plan: wider tensors for the space-to-depth copies, ops replaced
    Q = copy.copy(P)
    Q.tensors = [dict(t) for t in P.tensors]
    for f in nfold:
        Q.tensors[f['hs'].tid]['C'] += 4 * f['Cf']
    skip = set(folds.values()) | {f['up'] for f in nfold}
    neck_by = {f['pj']: f for f in nfold}
    copy_after = {}
    for f in nfold:
        prod = tail[f['tail']][0] if 'tail' in f else f['feat']          # the op that writes the feature (the root of a fused tail, or a conv)
        copy_after[prod] = {'op': 's2d_copy', 'src': P.ops[prod]['out'][0], 'tid': f['hs'].tid, 'coff': P.tensors[f['hs'].tid]['C']}
    # features that exist only as their space-to-depth copy: the ordinary copy is wiped right after the copy is made, the readers
    # are rewritten exactly as the device path rewrites them (a reader of the ordinary copy would then see zeros)
    wipe_after = {}
    for f in nfold:
        if f.get('s2d_only'):
            prod = tail[f['tail']][0] if 'tail' in f else f['feat']
            wipe_after[prod] = {'op': 'zero_slice', 'slice': P.ops[prod]['out'][0]}
    if bb == 'DLA-34':
        assert len(wipe_after) == 2 and len(R._s2d_readers) == 4, (wipe_after.keys(), R._s2d_readers)     # level3 / level4 roots: pool + entry conv each
    ops = []
    for k, op in enumerate(P.ops):
        if k in skip:
            continue
        if k in folds:
            op = R._folded_conv(P.ops[k], P.ops[folds[k]])
        if k in R._s2d_readers:
            kind, hs_, base_, cf_ = R._s2d_readers[k]
            op = {'op': 'maxpool_s2d', 'tid': hs_.tid, 'coff': base_, 'out': op['out']} if kind == 'pool' else R._s2d_input_conv(op, hs_, base_, cf_)
        if k in neck_by:
            op = R._neck_fold_conv(neck_by[k])
            assert len(op['taps'][0]) == 16 + neck_by[k]['Cf'] // 64 and op['cin'] == 64
        ops.append(op)
        if k in copy_after:
            ops.append(copy_after[k])
        if k in wipe_after:
            ops.append(wipe_after[k])
    Q.ops = ops
    assert len(Q.ops) == len(P.ops) - len(folds) + len(wipe_after)        # project ops gone; each neck fold: -2 ops + 1 copy op
    outs1, fetch1 = run_plan(Q, x)
    np.testing.assert_allclose(fetch1(P.named['z']).numpy(), fetch0(P.named['z']).numpy(), atol=2e-5, rtol=1e-5)
    wiped = {(w['slice'].tid, w['slice'].coff) for w in wipe_after.values()}
    for i in range(4):
        sl = P.named['feat%d' % i]
        if (sl.tid, sl.coff) in wiped:
            assert float(fetch1(sl).abs().max()) == 0.0
            continue
        np.testing.assert_allclose(fetch1(sl).numpy(), fetch0(sl).numpy(), atol=1e-5, rtol=1e-5)
    for a, b in zip(outs1, outs0):
        np.testing.assert_allclose(a.numpy(), b.numpy(), atol=5e-5, rtol=1e-5)


def test_plan_fp16_error_budget():
    """fp16 weights/activations with fp32 accumulation: logit error stays below 0.03 * logit scale."""
    bb = 'DLA-34'
    sd = weights.synth_state_dict(bb, 3, 'trained')
    x = weights.synth_images(1, 64, 128, seed=5)
    P = plan_mod.build_plan(sd, bb, 1, 64, 128)
    outs, _ = run_plan(P, x, half=True)
    _, logits = rtm3d_ref.model_forward(x, sd, bb)
    for a, b in zip(outs, logits):
        err = (a - b).abs().max().item()
        assert err < 0.03 * max(1.0, b.abs().max().item()), err


def test_weight_packing_roundtrip():
    rng = np.random.default_rng(0)
    wt = rng.standard_normal((9, 96, 128)).astype(np.float32)
    packed, cout_pad = plan_mod.pack_mfma_weights(wt, 64)
    assert cout_pad == 128
    pk = packed.reshape(2, 9, 2, 64, 8, 8).astype(np.float32)       # nt, tap, q, r, pos, e
    for (nt, t, q, r, c) in [(0, 0, 0, 0, 0), (1, 4, 1, 13, 5), (0, 8, 1, 63, 7), (1, 2, 0, 31, 3)]:
        pos = c ^ (r & 7)
        co = nt * 64 + r
        want = wt[t, co, q * 64 + c * 8: q * 64 + c * 8 + 8] if co < 96 else np.zeros(8)
        np.testing.assert_allclose(pk[nt, t, q, r, pos], want.astype(np.float16).astype(np.float32))


def test_state_dict_spec_counts():
    assert len(weights.state_dict_spec('DLA-34')) == 321
    assert len(weights.state_dict_spec('RESNET-18')) == 207


def test_weight_cache_is_transparent_and_persistent(tmp_path):
    """n2: BN-folded / composed weights through the WeightCache equal the uncached build bit for bit, a second plan of the
    same state dict is served from the cache, the on-disk copy is keyed by the state-dict digest."""
    from rtm3d_amd.weight_cache import WeightCache, state_dict_digest
    sd = weights.synth_state_dict('RESNET-18', 3, 'trained')
    ref = plan_mod.build_plan(sd, 'RESNET-18', 2, 64, 128)
    c = WeightCache(sd, directory=str(tmp_path))
    a = plan_mod.build_plan(sd, 'RESNET-18', 2, 64, 128, cache=c)
    miss = c.misses
    b = plan_mod.build_plan(sd, 'RESNET-18', 1, 96, 160, cache=c)
    assert c.misses == miss and c.hits > 0                # nothing folded twice
    for x, y in zip(ref.ops, a.ops):
        if x['op'] == 'conv':
            np.testing.assert_array_equal(x['w'], y['w']); np.testing.assert_array_equal(x['bias'], y['bias'])
    assert c.save() is not None
    c2 = WeightCache(sd, directory=str(tmp_path))
    d = plan_mod.build_plan(sd, 'RESNET-18', 2, 64, 128, cache=c2)
    assert c2.misses == 0
    for x, y in zip(ref.ops, d.ops):
        if x['op'] == 'conv':
            np.testing.assert_array_equal(x['w'], y['w'])
    sd2 = dict(sd); sd2['backbone.conv1.weight'] = sd['backbone.conv1.weight'] + 1
    assert state_dict_digest(sd2) != state_dict_digest(sd)
    assert WeightCache(sd2, directory=str(tmp_path)).entries == {}


def test_weight_cache_rejects_other_formats_and_corrupt_files(tmp_path, monkeypatch):
    """the on-disk cache is keyed by the state dict AND the pack-format tag (format version, C ABI version, hash of
    the packing code): a file written by another build of the packers is never read back; a truncated / corrupt file is
    ignored and replaced; temporary files are private to the writer and do not stay behind."""
    from rtm3d_amd import weight_cache as wc
    sd = weights.synth_state_dict('RESNET-18', 3, 'trained')
    c = wc.WeightCache(sd, directory=str(tmp_path))
    plan_mod.build_plan(sd, 'RESNET-18', 1, 64, 128, cache=c)
    path = c.save()
    assert path and os.path.exists(path) and [f for f in os.listdir(str(tmp_path)) if 'tmp' in f] == []
    assert wc.WeightCache(sd, directory=str(tmp_path)).entries            # same format: served
    # another pack format: different digest -> different file, nothing loaded
    monkeypatch.setattr(wc, 'PACK_FORMAT_VERSION', wc.PACK_FORMAT_VERSION + 1)
    c_new = wc.WeightCache(sd, directory=str(tmp_path))
    assert c_new.entries == {} and c_new.digest != c.digest and c_new._path() != path
    # even a file that was renamed onto the new digest's path is refused by its own format field
    os.replace(path, c_new._path())
    assert wc.WeightCache(sd, directory=str(tmp_path)).entries == {}
    monkeypatch.undo()
    # corrupt / truncated file under the right name: ignored, recomputed, replaced
    c3 = wc.WeightCache(sd, directory=str(tmp_path))
    with open(c3._path(), 'wb') as f:
        f.write(b'PK\x03\x04 not a zip')
    c4 = wc.WeightCache(sd, directory=str(tmp_path))
    assert c4.entries == {}
    plan_mod.build_plan(sd, 'RESNET-18', 1, 64, 128, cache=c4)
    assert c4.save() and wc.WeightCache(sd, directory=str(tmp_path)).entries


def test_halo_kernel_selection():
    """Which layers the plan sends to the persistent halo kernels: DLA-34's three 64 -> 64 level2 convs (conv64_halo.hip) at any
    batch, its seven 128 -> 128 level3 convs (conv128_halo.hip) only when there is at least one 8 x 32 tile per CU."""
    sd = weights.synth_state_dict('DLA-34', 1, 'trained')
    for B, want128 in ((32, 7), (16, 7), (8, 0), (1, 0)):      # 30 tiles per image: 480 / 240 / 30 at B = 16 / 8 / 1
        P = plan_mod.build_plan(sd, 'DLA-34', B, 384, 1280)
        convs = [op for op in P.ops if op['op'] == 'conv']
        assert sum(1 for op in convs if plan_mod.conv64_eligible(op)) == 3
        c128 = [op for op in convs if plan_mod.conv128_eligible(op, B)]
        assert len(c128) == want128
        assert all(op['name'].startswith('backbone.level3.') and (op['Hm'], op['Wm']) == (48, 160) for op in c128)
    P = plan_mod.build_plan(sd, 'DLA-34', 32, 128, 256)         # 16 x 32 level3 map: 8 x 32 tiles fit, but only 64 of them
    assert not any(plan_mod.conv128_eligible(op, 32) for op in P.ops if op['op'] == 'conv')


def gather_patches_reference(z, peaks):
    """z: (B, C, H, W) fp32 fused map; peaks: [(b, y, x)] -> (n, 15, 15, C) patches in the layout of csrc/sparse_heads.hip: patch
    pixel (5a + i, 5b + j) = z(y + i - 2 + 6(a - 1), x + j - 2 + 6(b - 1)), zero outside the map."""
    B, C, H, W = z.shape
    out = np.zeros((len(peaks), 15, 15, C), np.float32)
    for s, (b, y, x) in enumerate(peaks):
        for r in range(15):
            for c in range(15):
                zy, zx = y + r % 5 - 2 + 6 * (r // 5 - 1), x + c % 5 - 2 + 6 * (c // 5 - 1)
                if 0 <= zy < H and 0 <= zx < W:
                    out[s, r, c] = z[b, :, zy, zx]
    return out


def test_peak_plan_equals_dense_heads_at_the_peaks():
    """Peaks-only regression heads (plan.build_peak_plan, csrc/sparse_heads.hip): the patch plan evaluated on the gathered
    samples of the fused map z gives, for every peak - interior, on every border, in every corner - the value the DENSE
    offset_fr_main / main_offset maps have at that peak (fp32, CPU plan interpreter: the algebra of the patch layout, the
    {0, 5, 10} taps and the zero-padding masks), and the heat-map-only dense plan gives the dense plan's heat map."""
    bb = 'RESNET-18'
    sd = weights.synth_state_dict(bb, 5, 'trained')
    B, H, W = 2, 64, 128
    x = weights.synth_images(B, H, W, seed=3)
    dense = plan_mod.build_plan(sd, bb, B, H, W)
    outs, fetch = run_plan(dense, x)
    z = fetch(dense.named['z']).numpy()
    Hm, Wm = H // 4, W // 4
    hm_only = plan_mod.build_plan(sd, bb, B, H, W, dense_heads=1)
    outs1, _ = run_plan(hm_only, x)
    assert outs1[1] is None and torch.equal(outs1[0], outs[0])
    peaks = [(0, 0, 0), (0, 0, Wm - 1), (1, Hm - 1, 0), (1, Hm - 1, Wm - 1), (0, 1, 1), (1, 0, 7), (0, 9, 0), (1, 8, Wm - 2),
             (0, Hm - 2, 13), (1, 7, 15), (0, 5, 6), (1, 2, 2)]
    slots = len(peaks) + 2                                  # two empty slots at the end
    pp = plan_mod.build_peak_plan(sd, slots, (Hm, Wm))
    patches = np.zeros((slots, 15, 15, 256), np.float32)
    patches[:len(peaks)] = gather_patches_reference(z, peaks)
    patches[len(peaks):] = 7.0                              # stale content of empty slots must not matter to anyone else
    yx = np.array([[y, xx] for _, y, xx in peaks] + [[-1, -1]] * 2)
    pouts, _ = run_plan(pp, None, prefill={pp.named['zp'].tid: patches}, yx=yx)
    assert pouts[0].shape == (slots, 16, 1, 1) and pouts[1].shape == (slots, 2, 1, 1)
    for s, (b, y, xx) in enumerate(peaks):
        np.testing.assert_allclose(pouts[0][s, :, 0, 0].numpy(), outs[1][b, :, y, xx].numpy(), rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(pouts[1][s, :, 0, 0].numpy(), outs[2][b, :, y, xx].numpy(), rtol=2e-4, atol=2e-4)
