"""GPU (-m gpu): every HIP kernel variant on its own, through the C ABI, against a plain PyTorch fp32
reference of the same op (fp16-rounded operands, fp32 math), including ragged shapes: pixel counts
that are not multiples of the 128/256-pixel tiles, channel slices of wider tensors, strides, dilation,
the four transposed-convolution phases, residual + ReLU epilogues."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from rtm3d_amd import plan as plan_mod, _lib     # noqa: E402


def _run(P, feeds, fetch, x_img=None, expect_kernel=None):
    R = plan_mod.RealizedPlan(P, 0)
    if expect_kernel is not None:
        assert expect_kernel in R.kernel_names(), R.kernel_names()
    for s, arr in feeds:
        arr = np.ascontiguousarray(arr, np.float32)
        _lib.check(R.lib.rtm3d_tensor_upload(R.ctx, R.tids[s.tid], s.coff, s.C, arr.ctypes.data_as(ctypes.c_void_p)))
    B = P.B
    xin = torch.zeros(16, device='cuda') if x_img is None else x_img.cuda().contiguous()
    outs = [torch.zeros(B * 16 * 64 * 64, device='cuda') for _ in range(4)]
    R.forward(torch.cuda.current_stream().cuda_stream, xin.data_ptr(), [o.data_ptr() for o in outs])
    torch.cuda.synchronize()
    res = [R.download(s) for s in fetch]
    R.close()
    return res, outs


def h(t):
    return t.half().float()


CONV_CASES = [
    # B, H, W, cin, cout, k, stride, dil, relu, residual, variant, in_extra_ch
    (2, 24, 40, 64, 64, 3, 1, 1, True, True, 0, 0),        # v1 BN=64, residual
    (1, 13, 21, 128, 128, 3, 1, 1, True, False, 0, 64),     # ragged M (273 px), input slice of a wider tensor
    (2, 16, 24, 64, 128, 3, 2, 1, False, False, 0, 0),      # stride 2
    (1, 12, 20, 256, 64, 1, 1, 1, True, False, 0, 0),       # 1x1
    (1, 20, 36, 64, 256, 3, 1, 6, True, False, 0, 0),       # dilation 6 (head conv)
    (4, 96, 160, 64, 128, 3, 1, 1, True, True, 0, 0),       # v1 on a full-chip launch (480 tiles: the one-stage-in-flight kernel)
    (1, 12, 40, 512, 512, 3, 1, 1, True, True, 0, 0),       # deep ring + split-K (32 tiles, 72 K-steps in 8 ranges), residual
    (1, 12, 40, 512, 256, 1, 1, 1, False, False, 0, 64),    # deep ring + split-K of a 1x1 (8 K-steps in 2 ranges)
    (2, 13, 21, 256, 128, 3, 1, 1, True, False, 0, 0),      # deep ring + split-K, ragged M (546 px = 4.27 tiles)
    (3, 24, 40, 64, 256, 3, 1, 1, True, True, 2, 0),        # mfma256, ragged M (2880 = 11.25 tiles), residual
    (1, 16, 20, 128, 512, 3, 1, 6, False, False, 2, 0),     # mfma256 persistent, NT=2, M=320: six XCDs get no tile
    (3, 24, 40, 64, 256, 3, 1, 1, True, False, 2, 0),       # mfma256 persistent, ragged M (11.25 tiles), 9 K-tiles
    (2, 12, 20, 256, 256, 1, 1, 1, False, False, 2, 64),    # mfma256 persistent, 1x1: the minimum of 4 K-tiles
    (4, 96, 160, 256, 256, 1, 1, 1, True, False, 2, 0),     # mfma256 persistent, 240 tiles: every workgroup draws several tickets
    (2, 24, 64, 64, 64, 3, 1, 1, True, True, 5, 64),        # conv64 halo: residual, input slice of a wider tensor, 12 tiles
    (3, 40, 96, 64, 64, 3, 1, 1, False, False, 5, 0),       # conv64 halo: no ReLU, 45 tiles
    (1, 8, 32, 64, 64, 3, 1, 1, True, False, 5, 0),         # conv64 halo: a single tile
    (4, 96, 320, 64, 64, 3, 1, 1, True, True, 5, 0),        # conv64 halo: 480 tiles on 256 workgroups (ticket hand-out)
    (2, 16, 64, 128, 128, 3, 1, 1, True, True, 6, 0),       # conv128 halo: residual, 8 tiles (fewer than workgroups)
    (3, 24, 96, 128, 128, 3, 1, 1, False, False, 6, 64),    # conv128 halo: no ReLU, input slice of a wider tensor, 27 tiles
    (1, 8, 32, 128, 128, 3, 1, 1, True, False, 6, 0),       # conv128 halo: a single tile
    (8, 96, 160, 128, 128, 3, 1, 1, True, True, 6, 0),      # conv128 halo: 480 tiles on 256 workgroups (ticket hand-out)
    (2, 16, 64, 256, 256, 3, 1, 1, True, True, 6, 0),       # conv128 halo: two channel tiles per pixel tile, 4 input chunks, residual
    (2, 16, 64, 64, 128, 3, 2, 1, True, False, 7, 0),       # conv64s2 halo: 8 x 32 output map, 4 tiles
    (3, 24, 128, 64, 128, 3, 2, 1, False, False, 7, 64),    # conv64s2 halo: no ReLU, input slice of a wider tensor, 18 tiles
    (1, 8, 64, 64, 128, 3, 2, 1, True, False, 7, 0),        # conv64s2 halo: a single tile
    (8, 96, 320, 64, 128, 3, 2, 1, True, False, 7, 0),      # conv64s2 halo: 480 tiles on 256 workgroups (ticket hand-out)
    (2, 16, 24, 16, 16, 3, 1, 1, True, False, 3, 0),        # smallc 16->16
    (2, 16, 24, 16, 32, 3, 2, 1, True, False, 3, 0),        # smallc 16->32 s2
    (1, 18, 26, 32, 64, 3, 2, 1, True, False, 3, 0),        # smallc 32->64 s2
    (1, 9, 13, 32, 64, 1, 1, 1, False, False, 3, 0),        # smallc 1x1
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_kernels_vs_torch(case):
    B, H, W, cin, cout, k, stride, dil, relu, use_res, variant, extra = case
    rng = np.random.default_rng(hash(case) % (2 ** 32))
    pad = dil * (k - 1) // 2
    P = plan_mod.Plan(B, H * 4, W * 4)
    xt = P.tensor(H, W, cin + extra, max(pad, 1))
    xs = P.sub(xt, extra, cin)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    yt = P.tensor(Ho, Wo, cout + 8, 1)
    ys = P.sub(yt, 8, cout)
    rs = P.tensor(Ho, Wo, cout, 0) if use_res else None
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    P.conv(xs, ys, w, b, stride=stride, dil=dil, relu=relu, res=rs, name='t')
    P.ops[-1]['variant'] = variant
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    feeds = [(xs, x)]
    if use_res:
        r = rng.standard_normal((B, cout, Ho, Wo)).astype(np.float32)
        feeds.append((rs, r))
    (got,), _ = _run(P, feeds, [ys])
    ref = F.conv2d(h(torch.from_numpy(x)), h(torch.from_numpy(w)), torch.from_numpy(b), stride, pad, dil)
    if use_res:
        ref = ref + h(torch.from_numpy(r))
    if relu:
        ref = ref.relu()
    ref = h(ref).numpy()
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3 * max(1.0, np.abs(ref).max()))


# (the last shape: 320 tiles of a 64-channel layer on 256 workgroups - one chunk per tile, so every K-tile stages the NEXT tile's halo)
@pytest.mark.parametrize('shape', [(1, 8, 32, 64, 1), (2, 16, 64, 256, 4), (3, 24, 96, 128, 2), (80, 16, 64, 64, 1)])
def test_halo_conv256_vs_torch(shape):
    """3x3 dilation-1 convs whose output is covered by 8x32 tiles take the halo-tile kernel
    (conv_mfma256_halo.hip): single and grouped, 1-4 chunks of 64 channels, several tiles per workgroup."""
    B, H, W, cin, G = shape
    rng = np.random.default_rng(B * 100 + cin)
    P = plan_mod.Plan(B, H * 4, W * 4)
    xt = P.tensor(H, W, cin * G, 1)
    yt = P.tensor(H, W, 256 * G, 1)
    ws = [(rng.standard_normal((256, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32) for _ in range(G)]
    bs = [rng.standard_normal(256).astype(np.float32) for _ in range(G)]
    if G == 1:
        P.conv(xt, yt, ws[0], bs[0], relu=True, name='t')
    else:
        P.grouped_conv([P.sub(xt, g * cin, cin) for g in range(G)], [P.sub(yt, g * 256, 256) for g in range(G)], ws, bs, relu=True, name='t')
    P.ops[-1]['variant'] = 2
    x = rng.standard_normal((B, cin * G, H, W)).astype(np.float32)
    (got,), _ = _run(P, [(xt, x)], [yt])
    ref = torch.cat([F.conv2d(h(torch.from_numpy(x[:, g * cin:(g + 1) * cin])), h(torch.from_numpy(ws[g])), torch.from_numpy(bs[g]), 1, 1)
                     for g in range(G)], 1).relu()
    ref = h(ref).numpy()
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3 * max(1.0, np.abs(ref).max()))


# (1, 48, 32, 64, 256): one 64-channel chunk -> every K-tile stages rows of the NEXT tile; (2, 48, 64, 256, 512): four chunks, two channel
# tiles; (40, 48, 32, 128, 256): 240 tiles, a workgroup runs several in a row and the row ring wraps across tiles; (1, 96, 320, 256, 1024):
# one image of the head conv itself; (3, 96, 64, 192, 256): three chunks (the ring base returns to 0 only every eight chunks)
@pytest.mark.parametrize('shape', [(1, 48, 32, 64, 256), (2, 48, 64, 256, 512), (40, 48, 32, 128, 256), (1, 96, 320, 256, 1024), (3, 96, 64, 192, 256)])
def test_lattice_conv256_vs_torch(shape):
    """3x3 DILATION-6 convs (the fused head conv, header.py:12-16) whose map is covered by tiles of 8 lattice rows x 32 columns take
    the row-sub-lattice halo kernel (conv_mfma256_lattice.hip: a ring of 16 halo-row slots); other shapes stay on the generic
    persistent kernel (the (1, 16, 20, ...) dilation-6 case of CONV_CASES)."""
    B, H, W, cin, cout = shape
    rng = np.random.default_rng(B * 1000 + cin + cout)
    P = plan_mod.Plan(B, H * 4, W * 4)
    xt = P.tensor(H, W, cin + 64, 6)
    xs = P.sub(xt, 64, cin)
    yt = P.tensor(H, W, cout, 1)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    P.conv(xs, yt, w, b, dil=6, relu=True, name='t')
    P.ops[-1]['variant'] = 2
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    (got,), _ = _run(P, [(xs, x)], [yt], expect_kernel='conv3x3_mfma256_lattice')
    ref = h(F.conv2d(h(torch.from_numpy(x)), h(torch.from_numpy(w)), torch.from_numpy(b), 1, 6, 6).relu()).numpy()
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3 * max(1.0, np.abs(ref).max()))


def test_lattice_kernel_is_not_taken_for_other_shapes():
    """A dilation-6 conv whose map is NOT covered by tiles of 8 lattice rows x 32 columns (the KITTI letterbox 416 x 1280: H/4 = 104,
    not a multiple of 48) stays on the generic persistent kernel - and is still right."""
    B, H, W, cin, cout = 1, 104, 64, 64, 256
    rng = np.random.default_rng(7)
    P = plan_mod.Plan(B, H * 4, W * 4)
    xt = P.tensor(H, W, cin, 6)
    yt = P.tensor(H, W, cout, 1)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    P.conv(xt, yt, w, b, dil=6, relu=True, name='t')
    P.ops[-1]['variant'] = 2
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    R = plan_mod.RealizedPlan(P, 0)
    names = R.kernel_names()
    R.close()
    assert 'conv3x3_mfma256' in names and 'conv3x3_mfma256_lattice' not in names, names
    (got,), _ = _run(P, [(xt, x)], [yt])
    ref = h(F.conv2d(h(torch.from_numpy(x)), h(torch.from_numpy(w)), torch.from_numpy(b), 1, 6, 6).relu()).numpy()
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3 * max(1.0, np.abs(ref).max()))


def test_conv256_input_beyond_4gb():
    """The persistent 256 x 256 kernel addresses its pixel operand as an SGPR base + a 32-bit byte offset per lane (round 5): a conv
    whose input tensor is 4 GB or larger must not run on it (launch_conv_mfma256 sends it to the one-tile kernel, which carries
    64-bit addresses).  Input = the last 256 channels of an 8 x 96 x 320 x 8704-channel tensor (4.39 GB; the last image lies beyond
    4 GB): wrapped offsets would read other images' pixels - still inside the allocation, so a wrong routing shows as a mismatch."""
    B, H, W, cin, cout, extra = 8, 96, 320, 256, 256, 8448
    rng = np.random.default_rng(11)
    P = plan_mod.Plan(B, H * 4, W * 4)
    xt = P.tensor(H, W, cin + extra, 1)
    xs = P.sub(xt, extra, cin)
    yt = P.tensor(H, W, cout, 1)
    ys = P.sub(yt, 0, cout)
    assert B * (H + 2) * (W + 2) * (cin + extra) * 2 >= 1 << 32
    w = (rng.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    P.conv(xs, ys, w, b, relu=True, name='t')
    P.ops[-1]['variant'] = 2
    # every image different, so that reading image (n - 7.8) instead of image n cannot go unnoticed
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    (got,), _ = _run(P, [(xs, x)], [ys])
    ref = h(F.conv2d(h(torch.from_numpy(x)), h(torch.from_numpy(w)), torch.from_numpy(b)).relu()).numpy()
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3 * max(1.0, np.abs(ref).max()))


def test_split_k_chain_is_deterministic():
    """Two split-K convolutions back to back share the slab and the per-tile arrival counters (the last workgroup of a
    tile re-arms its counter): the chain must match torch, carry the expected kernels, and replay bit-identically."""
    rng = np.random.default_rng(11)
    B, H, W, C = 1, 12, 40, 256
    P = plan_mod.Plan(B, H * 4, W * 4)
    xt, mt, yt = P.tensor(H, W, C, 1), P.tensor(H, W, C, 1), P.tensor(H, W, C, 1)
    w1 = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
    w2 = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
    b1, b2 = rng.standard_normal(C).astype(np.float32), rng.standard_normal(C).astype(np.float32)
    P.conv(xt, mt, w1, b1, relu=True, name='a')
    P.conv(mt, yt, w2, b2, relu=True, res=xt, name='b')
    for op in P.ops:
        op['variant'] = 0
    R = plan_mod.RealizedPlan(P, 0)
    assert R.kernel_names() == ['conv3x3_mfma_deep_splitk'] * 2
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    _lib.check(R.lib.rtm3d_tensor_upload(R.ctx, R.tids[xt.tid], 0, C, x.ctypes.data_as(ctypes.c_void_p)))
    xin = torch.zeros(16, device='cuda')
    o = [torch.zeros(16, device='cuda') for _ in range(4)]
    outs = []
    for it in range(20):
        R.forward(torch.cuda.current_stream().cuda_stream, xin.data_ptr(), [t.data_ptr() for t in o])
        torch.cuda.synchronize()
        outs.append(R.download(yt))
    R.close()
    xh = h(torch.from_numpy(x))
    mid = h(F.conv2d(xh, h(torch.from_numpy(w1)), torch.from_numpy(b1), 1, 1).relu())
    ref = h((F.conv2d(mid, h(torch.from_numpy(w2)), torch.from_numpy(b2), 1, 1) + xh).relu()).numpy()
    np.testing.assert_allclose(outs[0], ref, rtol=4e-3, atol=4e-3 * max(1.0, np.abs(ref).max()))
    for got in outs[1:]:
        np.testing.assert_array_equal(got, outs[0])


def test_persistent_conv_replays_identically():
    """The persistent conv kernel hands out tiles through per-XCD ticket counters that the last draw of a
    launch resets: ten replays of one context (and a second context in between) must all give the
    first launch's bytes."""
    rng = np.random.default_rng(7)
    outs = []
    for rep in range(2):
        P = plan_mod.Plan(2, 96 * 4, 160 * 4)
        xt = P.tensor(96, 160, 256, 1)
        yt = P.tensor(96, 160, 256, 1)
        P.conv(xt, yt, (np.random.default_rng(1).standard_normal((256, 256, 3, 3)) / 48).astype(np.float32), np.zeros(256, np.float32), relu=True, name='t')
        P.ops[-1]['variant'] = 2
        R = plan_mod.RealizedPlan(P, 0)
        x = np.random.default_rng(2).standard_normal((2, 256, 96, 160)).astype(np.float32)
        _lib.check(R.lib.rtm3d_tensor_upload(R.ctx, R.tids[xt.tid], 0, 256, x.ctypes.data_as(ctypes.c_void_p)))
        xin = torch.zeros(16, device='cuda')
        o = [torch.zeros(16, device='cuda') for _ in range(4)]
        for it in range(10):
            R.forward(torch.cuda.current_stream().cuda_stream, xin.data_ptr(), [t.data_ptr() for t in o])
            torch.cuda.synchronize()
            outs.append(R.download(yt))
        R.close()
    assert np.abs(outs[0]).max() > 0
    for got in outs[1:]:
        np.testing.assert_array_equal(got, outs[0])


# (the last two: maps covered by 8 x 32 tiles - the four sub-pixel phases run on the halo-tile kernel, three halo rows per K-tile)
@pytest.mark.parametrize('shape', [(2, 6, 10, 0), (1, 12, 40, 0), (4, 24, 80, 2), (32, 12, 40, 2), (2, 16, 64, 2), (5, 24, 96, 2)])
def test_deconv_phases_vs_torch(shape):
    B, H, W, variant = shape
    rng = np.random.default_rng(B * 1000 + H)
    P = plan_mod.Plan(B, H * 8, W * 8)
    xt = P.tensor(H, W, 256, 1)
    yt = P.tensor(2 * H, 2 * W, 256, 0)
    w = (rng.standard_normal((256, 256, 4, 4)) / 32).astype(np.float32)
    P.deconv(xt, yt, w, name='up')
    P.ops[-1]['variant'] = variant
    x = rng.standard_normal((B, 256, H, W)).astype(np.float32)
    (got,), _ = _run(P, [(xt, x)], [yt])
    ref = h(F.conv_transpose2d(h(torch.from_numpy(x)), h(torch.from_numpy(w)), None, 2, 1)).numpy()
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize('k,stride,pad', [(2, 2, 0), (3, 2, 1)])
def test_maxpool_vs_torch(k, stride, pad):
    rng = np.random.default_rng(k)
    B, H, W, C = 2, 16, 24, 64
    P = plan_mod.Plan(B, H, W)
    xt = P.tensor(H, W, C + 16, 1)
    yt = P.tensor(H // 2, W // 2, C, 0)
    P.maxpool(P.sub(xt, 16, C), yt, k, stride, pad)
    x = np.abs(rng.standard_normal((B, C, H, W))).astype(np.float32)        # post-ReLU inputs
    (got,), _ = _run(P, [(P.sub(xt, 16, C), x)], [yt])
    ref = F.max_pool2d(h(torch.from_numpy(x)), k, stride, pad).numpy()
    np.testing.assert_array_equal(got, ref)


def test_softmax_fuse_with_peaked_inputs():
    """z + sum_i u_i * softmax_HW(u_i) with large peaks (where the softmax term is not negligible)."""
    rng = np.random.default_rng(5)
    B, H, W = 2, 12, 20
    P = plan_mod.Plan(B, H * 4, W * 4)
    z0 = P.tensor(H, W, 256, 0)
    us = [P.tensor(H, W, 256, p) for p in (0, 1, 0)]
    z = P.tensor(H, W, 256, 6)
    P.softmax_fuse(z0, z, us)
    zin = rng.standard_normal((B, 256, H, W)).astype(np.float32)
    uin = [(rng.standard_normal((B, 256, H, W)) * s).astype(np.float32) for s in (1.0, 3.0, 6.0)]
    for u in uin:
        u[:, :, 3, 4] += 9.0
    (got,), _ = _run(P, [(z0, zin)] + list(zip(us, uin)), [z])
    ref = h(torch.from_numpy(zin))
    for u in uin:
        t = h(torch.from_numpy(u))
        ref = ref + t * torch.softmax(t.view(B, 256, -1), -1).view(B, 256, H, W)
    ref = h(ref).numpy()
    np.testing.assert_allclose(got, ref, rtol=3e-3, atol=3e-3 * np.abs(ref).max())


def test_softmax_fuse_with_epilogue_partials():
    """Fusion whose three operands come from transposed convs on the halo-tile kernel: their epilogues emit the
    spatial-softmax partials (softmax_stat_slot) and the fusion skips its reduction pass.  Checked against torch,
    and the producers must really have been marked."""
    B, H, W = 3, 8, 32
    rng = np.random.default_rng(11)
    P = plan_mod.Plan(B, H * 8, W * 8)
    z0 = P.tensor(2 * H, 2 * W, 256, 0)
    xs = [P.tensor(H, W, 256, 1) for _ in range(3)]
    us = [P.tensor(2 * H, 2 * W, 256, 0) for _ in range(3)]
    ws = [(rng.standard_normal((256, 256, 4, 4)) * sc / 32).astype(np.float32) for sc in (1.0, 2.5, 6.0)]
    for x, u, w in zip(xs, us, ws):
        P.deconv(x, u, w, name='up')
        P.ops[-1]['variant'] = 2
    z = P.tensor(2 * H, 2 * W, 256, 6)
    P.softmax_fuse(z0, z, us, name='fuse')
    zin = rng.standard_normal((B, 256, 2 * H, 2 * W)).astype(np.float32)
    xin = [rng.standard_normal((B, 256, H, W)).astype(np.float32) for _ in range(3)]
    R = plan_mod.RealizedPlan(P, 0)
    assert sorted(R._stat_slots.values()) == [0, 1, 2]
    for s_, arr in [(z0, zin)] + list(zip(xs, xin)):
        _lib.check(R.lib.rtm3d_tensor_upload(R.ctx, R.tids[s_.tid], s_.coff, s_.C, np.ascontiguousarray(arr).ctypes.data_as(ctypes.c_void_p)))
    dummy = torch.zeros(16, device='cuda'); outs = [torch.zeros(16, device='cuda') for _ in range(4)]
    for _ in range(2):                                   # replay: the partial buffer is rewritten every forward
        R.forward(torch.cuda.current_stream().cuda_stream, dummy.data_ptr(), [o.data_ptr() for o in outs])
    torch.cuda.synchronize()
    got = R.download(z)
    R.close()
    ref = h(torch.from_numpy(zin))
    for x, w in zip(xin, ws):
        t = h(F.conv_transpose2d(h(torch.from_numpy(x)), h(torch.from_numpy(w)), None, 2, 1))
        ref = ref + t * torch.softmax(t.view(B, 256, -1), -1).view(B, 256, 2 * H, 2 * W)
    ref = h(ref).numpy()
    np.testing.assert_allclose(got, ref, rtol=4e-3, atol=4e-3 * max(1.0, np.abs(ref).max()))


def test_stem_and_headout_vs_torch():
    rng = np.random.default_rng(9)
    B, H, W = 2, 32, 64
    # stem 7x7 s1 3->16 and s2 3->64 straight from the fp32 NCHW image
    for cout, stride in ((16, 1), (64, 2)):
        P = plan_mod.Plan(B, H, W)
        yt = P.tensor(H // stride, W // stride, cout, 1)
        w = (rng.standard_normal((cout, 3, 7, 7)) / 12).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        P.stem_mfma(yt, w, b, stride)
        x = torch.from_numpy(rng.standard_normal((B, 3, H, W)).astype(np.float32))
        (got,), _ = _run(P, [], [yt], x_img=x)
        ref = h(F.conv2d(h(x), h(torch.from_numpy(w)), torch.from_numpy(b), stride, 3).relu()).numpy()
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
    # the four logit convs (halo-tile kernel), sizes that are not multiples of the 8x32 tile; the last case is large enough
    # (batch x tiles x heads >= 4 x CUs) to take the 16-row tiles, with a ragged last tile row and column
    for (B, Hh, Wh) in ((2, 8, 40), (2, 11, 37), (2, 16, 64), (44, 27, 70)):
        P = plan_mod.Plan(B, Hh * 4, Wh * 4)
        ht = P.tensor(Hh, Wh, 1024, 1)
        ws = [(rng.standard_normal((c, 256, 3, 3)) / 48).astype(np.float32) for c in (3, 16, 2, 2)]
        bs = [rng.standard_normal(c).astype(np.float32) for c in (3, 16, 2, 2)]
        P.headout(ht, ws, bs)
        x = rng.standard_normal((B, 1024, Hh, Wh)).astype(np.float32)
        _, outs = _run(P, [(ht, x)], [])
        for g, (w, b) in enumerate(zip(ws, bs)):
            c = w.shape[0]
            got = outs[g][:B * c * Hh * Wh].view(B, c, Hh, Wh).cpu().numpy()
            ref = F.conv2d(h(torch.from_numpy(x[:, g * 256:(g + 1) * 256])), h(torch.from_numpy(w)), torch.from_numpy(b), 1, 1).numpy()
            np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())


def test_resnet34_stage_parity():
    import rtm3d_amd
    from rtm3d_amd import weights
    from oracle import rtm3d_ref
    bb = 'RESNET-34'
    sd = weights.synth_state_dict(bb, 2, 'trained', heat_bias=-3.0)
    cfg = rtm3d_amd.kitti_config(bb)
    m = rtm3d_amd.create_model(cfg).to('cuda:0').eval()
    m.load_state_dict(sd)
    x = weights.synth_images(1, 64, 128, seed=3)
    logits = m.forward_logits(x.cuda())
    _, lref = rtm3d_ref.model_forward(x, sd, bb)
    for a, b in zip(logits, lref):
        assert (a.cpu() - b).abs().max().item() < 0.010 * max(1.0, b.abs().max().item())   # 2 x measured, profiles/r02_logit_error.json


@pytest.mark.parametrize('shape', [(2, 32, 64), (1, 16, 32), (3, 48, 96)])
def test_fused_dla_stem_vs_torch(shape):
    """conv_stem_fused.hip: base_layer 7x7 (3->16) + level0 3x3 (16->16) [+ level1 3x3 stride 2 (16->32)], each + bias +
    ReLU, in one launch, against plain PyTorch fp32 on fp16-rounded operands (the intermediate maps are rounded to fp16 as
    the unfused path stores them); image borders exercise the zero padding of every layer; also equal to the three-launch
    path within fp32 summation order."""
    B, H, W = shape
    rng = np.random.default_rng(B * 7 + H)
    w0 = (rng.standard_normal((16, 3, 7, 7)) / np.sqrt(3 * 49)).astype(np.float32)
    b0 = rng.standard_normal(16).astype(np.float32) * 0.3
    w1 = (rng.standard_normal((16, 16, 3, 3)) / np.sqrt(16 * 9)).astype(np.float32)
    b1 = rng.standard_normal(16).astype(np.float32) * 0.3
    w2 = (rng.standard_normal((32, 16, 3, 3)) / np.sqrt(16 * 9)).astype(np.float32)
    b2 = rng.standard_normal(32).astype(np.float32) * 0.3
    x = rng.standard_normal((B, 3, H, W)).astype(np.float32)

    def build():
        P = plan_mod.Plan(B, H, W)
        t_base = P.tensor(H, W, 16, 1)
        P.stem_mfma(t_base, w0, b0, 1, name='base')
        t_l0 = P.tensor(H, W, 16, 1)
        P.conv(t_base, t_l0, w1, b1, relu=True, name='level0')
        t_l1 = P.tensor(H // 2, W // 2, 32 + 8, 1)
        out = P.sub(t_l1, 8, 32)
        P.conv(t_l0, out, w2, b2, stride=2, relu=True, name='level1')
        return P, t_l0, out

    res = {}
    for fuse, nops in ((True, 2), (2, 3), (False, 4)):
        plan_mod.FUSE_STEM = fuse
        try:
            P, t_l0, out = build()
            R = plan_mod.RealizedPlan(P, 0)
            assert len(R.op_names) == nops, R.op_names         # input4 + (one fused launch | fused pair + level1 | three convs)
            xin = torch.from_numpy(x).cuda()
            outs = [torch.zeros(16, device='cuda') for _ in range(4)]
            R.forward(torch.cuda.current_stream().cuda_stream, xin.data_ptr(), [o.data_ptr() for o in outs])
            torch.cuda.synchronize()
            res[fuse] = (R.download(out), R.download(t_l0) if fuse is not True else None)
            R.close()
        finally:
            plan_mod.FUSE_STEM = True
    mid = h(F.conv2d(h(torch.from_numpy(x)), h(torch.from_numpy(w0)), torch.from_numpy(b0), 1, 3).relu())
    l0 = h(F.conv2d(mid, h(torch.from_numpy(w1)), torch.from_numpy(b1), 1, 1).relu())
    ref = h(F.conv2d(l0, h(torch.from_numpy(w2)), torch.from_numpy(b2), 2, 1).relu()).numpy()
    tol = 4e-3 * max(1.0, np.abs(ref).max())
    for fuse in (True, 2, False):
        np.testing.assert_allclose(res[fuse][0], ref, rtol=4e-3, atol=tol)
    for fuse in (2, False):
        np.testing.assert_allclose(res[fuse][1], l0.numpy(), rtol=3e-3, atol=3e-3 * max(1.0, float(l0.abs().max())))
    np.testing.assert_allclose(res[True][0], res[False][0], rtol=3e-3, atol=tol)


@pytest.mark.parametrize('shape', [(2, 32, 64), (1, 16, 64), (3, 48, 192)])
def test_fused_level_entry_vs_torch(shape):
    """conv32s2_fused.hip: 2x2 max-pool -> 1x1 project (no ReLU) and 3x3 stride-2 conv (ReLU) of one 32-channel map in one
    launch (DLA level2 entry), against plain PyTorch fp32 on fp16-rounded operands and against the three-launch path."""
    B, H, W = shape
    rng = np.random.default_rng(B * 11 + H)
    wp = (rng.standard_normal((64, 32, 1, 1)) / np.sqrt(32)).astype(np.float32); bp = rng.standard_normal(64).astype(np.float32) * 0.3
    wc = (rng.standard_normal((64, 32, 3, 3)) / np.sqrt(32 * 9)).astype(np.float32); bc = rng.standard_normal(64).astype(np.float32) * 0.3
    x = np.abs(rng.standard_normal((B, 32, H, W))).astype(np.float32)          # post-ReLU map (the zero border equals -inf padding of the pool)

    def build():
        P = plan_mod.Plan(B, H * 2, W * 2)
        xt = P.tensor(H, W, 32 + 8, 1)
        xs = P.sub(xt, 8, 32)
        bottom = P.tensor(H // 2, W // 2, 32, 0)
        P.maxpool(xs, bottom, 2, 2, 0, name='level2.downsample')
        resid = P.tensor(H // 2, W // 2, 64, 0)
        P.conv(bottom, resid, wp, bp, name='level2.project')
        mt = P.tensor(H // 2, W // 2, 64 + 64, 1)
        mid = P.sub(mt, 64, 64)
        P.conv(xs, mid, wc, bc, stride=2, relu=True, name='level2.tree1.conv1')
        return P, xs, resid, mid

    res = {}
    for fuse in (True, False):
        plan_mod.FUSE_LEVEL_ENTRY = fuse
        try:
            P, xs, resid, mid = build()
            R = plan_mod.RealizedPlan(P, 0)
            assert len(R.op_names) == (1 if fuse else 3), R.op_names
            R.close()
            (r, m), _ = _run(P, [(xs, x)], [resid, mid])
            res[fuse] = (r, m)
        finally:
            plan_mod.FUSE_LEVEL_ENTRY = True
    xh = h(torch.from_numpy(x))
    ref_r = h(F.conv2d(F.max_pool2d(xh, 2, 2), h(torch.from_numpy(wp)), torch.from_numpy(bp))).numpy()
    ref_m = h(F.conv2d(xh, h(torch.from_numpy(wc)), torch.from_numpy(bc), 2, 1).relu()).numpy()
    for fuse in (True, False):
        np.testing.assert_allclose(res[fuse][0], ref_r, rtol=3e-3, atol=3e-3 * max(1.0, np.abs(ref_r).max()))
        np.testing.assert_allclose(res[fuse][1], ref_m, rtol=3e-3, atol=3e-3 * max(1.0, np.abs(ref_m).max()))


@pytest.mark.parametrize('shape', [(2, 16, 64, True), (1, 8, 32, True), (3, 40, 96, False), (4, 96, 320, True)])
def test_fused_level_tail_vs_torch(shape):
    """conv64_root.hip: 3x3 64->64 conv + residual + ReLU, the tree's 1x1 root over cat[x2, x1] (ReLU) and the next level's 2x2
    max-pool in one launch (DLA level2 tail), against plain PyTorch fp32 on fp16-rounded operands and against the three-launch
    path (x2 is only written by the latter)."""
    B, H, W, with_pool = shape
    rng = np.random.default_rng(B * 13 + H)
    wc = (rng.standard_normal((64, 64, 3, 3)) / np.sqrt(64 * 9)).astype(np.float32); bc = rng.standard_normal(64).astype(np.float32) * 0.3
    wr = (rng.standard_normal((64, 128, 1, 1)) / np.sqrt(128)).astype(np.float32); br = rng.standard_normal(64).astype(np.float32) * 0.3
    t = np.abs(rng.standard_normal((B, 64, H, W))).astype(np.float32)
    x1 = np.abs(rng.standard_normal((B, 64, H, W))).astype(np.float32)

    def build():
        P = plan_mod.Plan(B, H * 4, W * 4)
        tt = P.tensor(H, W, 64 + 8, 1)
        ts = P.sub(tt, 8, 64)
        cat = P.tensor(H, W, 128, 1)
        x2s, x1s = P.sub(cat, 0, 64), P.sub(cat, 64, 64)
        P.conv(ts, x2s, wc, bc, relu=True, res=x1s, name='level2.tree2.conv2')
        ot = P.tensor(H, W, 64 + 16, 1)
        out = P.sub(ot, 16, 64)
        P.conv(cat, out, wr, br, relu=True, name='level2.root')
        pooled = None
        if with_pool:
            pt = P.tensor(H // 2, W // 2, 64 + 24, 1)
            pooled = P.sub(pt, 24, 64)
            P.maxpool(out, pooled, 2, 2, 0, name='level3.downsample')
        return P, ts, x1s, out, pooled

    res = {}
    for fuse in (True, False):
        plan_mod.FUSE_LEVEL_TAIL = fuse
        try:
            P, ts, x1s, out, pooled = build()
            R = plan_mod.RealizedPlan(P, 0)
            assert len(R.op_names) == (1 if fuse else (3 if with_pool else 2)), R.op_names
            if fuse:
                assert R.kernel_names()[0].startswith('conv3x3_c64+root1x1'), R.kernel_names()
            R.close()
            got, _ = _run(P, [(ts, t), (x1s, x1)], [out] + ([pooled] if with_pool else []))
            res[fuse] = got
        finally:
            plan_mod.FUSE_LEVEL_TAIL = True
    th, x1h = h(torch.from_numpy(t)), h(torch.from_numpy(x1))
    x2 = h((F.conv2d(th, h(torch.from_numpy(wc)), torch.from_numpy(bc), 1, 1) + x1h).relu())
    ref_o = h(F.conv2d(torch.cat([x2, x1h], 1), h(torch.from_numpy(wr)), torch.from_numpy(br)).relu())
    tol = 4e-3 * max(1.0, float(ref_o.abs().max()))
    for fuse in (True, False):
        np.testing.assert_allclose(res[fuse][0], ref_o.numpy(), rtol=4e-3, atol=tol)
        if with_pool:
            # the pooled map is the exact 2x2 max of the kernel's OWN fp16 output
            np.testing.assert_array_equal(res[fuse][1], F.max_pool2d(torch.from_numpy(res[fuse][0]), 2, 2).numpy())
    np.testing.assert_allclose(res[True][0], res[False][0], rtol=4e-3, atol=tol)


@pytest.mark.parametrize('case', [(2, 24, 80, 128, 256, 64), (1, 12, 40, 256, 512, 0), (8, 48, 160, 64, 128, 0), (32, 24, 80, 128, 256, 0), (1, 6, 10, 256, 512, 64)])
def test_project_fold_vs_torch(case):
    """RealizedPlan._project_folds: a DLA block's `project` 1x1 (on the pooled input, another channel slice of the tensor the
    block's second conv reads) as extra K-steps of that conv (rtm3d_conv_desc.tap_dc) instead of a launch + residual read:
    against plain PyTorch fp32 on fp16-rounded operands and against the two-launch path, on the 128-pixel kernel (incl. its
    small-launch split-K form) and the persistent 256-pixel one."""
    B, H, W, cb, cout, lead = case
    rng = np.random.default_rng(B * 17 + H)
    w2 = (rng.standard_normal((cout, cout, 3, 3)) / np.sqrt(cout * 9)).astype(np.float32); b2 = rng.standard_normal(cout).astype(np.float32) * 0.3
    wp = (rng.standard_normal((cout, cb, 1, 1)) / np.sqrt(cb)).astype(np.float32); bp = rng.standard_normal(cout).astype(np.float32) * 0.3
    t = np.abs(rng.standard_normal((B, cout, H, W))).astype(np.float32)
    bot = np.abs(rng.standard_normal((B, cb, H, W))).astype(np.float32)

    def build():
        P = plan_mod.Plan(B, H * 8, W * 8)
        big = P.tensor(H, W, lead + cb + cout, 1)
        bs, ms = P.sub(big, lead, cb), P.sub(big, lead + cb, cout)
        resid = P.tensor(H, W, cout, 0)
        P.conv(bs, resid, wp, bp, name='lvl.project')
        ot = P.tensor(H, W, cout, 1)
        P.conv(ms, ot, w2, b2, relu=True, res=resid, name='lvl.tree1.conv2')
        return P, bs, ms, ot

    res = {}
    for fold in (True, False):
        plan_mod.FOLD_PROJECT = fold
        try:
            P, bs, ms, ot = build()
            R = plan_mod.RealizedPlan(P, 0)
            assert len(R.op_names) == (1 if fold else 2), R.op_names
            R.close()
            (o,), _ = _run(P, [(bs, bot), (ms, t)], [ot])
            res[fold] = o
        finally:
            plan_mod.FOLD_PROJECT = True
    th, bh = h(torch.from_numpy(t)), h(torch.from_numpy(bot))
    ref = (F.conv2d(th, h(torch.from_numpy(w2)), torch.from_numpy(b2), 1, 1) + F.conv2d(bh, h(torch.from_numpy(wp)), torch.from_numpy(bp))).relu()
    tol = 4e-3 * max(1.0, float(ref.abs().max()))
    np.testing.assert_allclose(res[True], h(ref).numpy(), rtol=4e-3, atol=tol)
    np.testing.assert_allclose(res[False], h(ref).numpy(), rtol=4e-3, atol=2 * tol)       # (this path also rounds the projected map to fp16)


def test_forward_on_two_streams_is_serialised():
    """one context = one activation workspace and one set of ticket counters.  Two replays issued back to back on
    DIFFERENT streams must not overlap on the device: rtm3d_forward orders a call on a new stream behind the previous replay."""
    import rtm3d_amd
    from rtm3d_amd import weights
    bb = 'RESNET-18'
    m = rtm3d_amd.create_model(rtm3d_amd.kitti_config(bb)).to('cuda:0').eval()
    m.load_state_dict(weights.synth_state_dict(bb, 1, 'trained', heat_bias=-3.0))
    m.use_graph = False
    xa = weights.synth_images(4, 128, 256, seed=1).cuda()
    xb = weights.synth_images(4, 128, 256, seed=2).cuda()
    ra = [t.clone() for t in m.forward_logits(xa)]
    rb = [t.clone() for t in m.forward_logits(xb)]
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(5):
        with torch.cuda.stream(sa):
            ga = m.forward_logits(xa)
        with torch.cuda.stream(sb):
            gb = m.forward_logits(xb)
        torch.cuda.synchronize()
        for a, b in zip(ga, ra):
            assert torch.equal(a, b)
        for a, b in zip(gb, rb):
            assert torch.equal(a, b)


def test_round4_entry_points_refuse_bad_arguments():
    """Error behaviour of what round 4 added to the C ABI: every refusal returns non-zero with a message and records nothing
    (the plan stays usable)."""
    lib = _lib.load()
    ctx = ctypes.c_void_p()
    _lib.check(lib.rtm3d_ctx_create(0, ctypes.byref(ctx)))
    try:
        def tensor(B, H, W, C, pad):
            tid = ctypes.c_int()
            _lib.check(lib.rtm3d_tensor_create(ctx, B, H, W, C, pad, ctypes.byref(tid)))
            return tid.value

        def blob(nbytes):
            arr = np.zeros(nbytes, np.uint8)
            bid = ctypes.c_int()
            _lib.check(lib.rtm3d_blob_create(ctx, arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes, ctypes.byref(bid)))
            return bid.value
        t_in, t_cat, t_out, t_pool, t_s2d = tensor(1, 16, 64, 64, 1), tensor(1, 16, 64, 128, 1), tensor(1, 16, 64, 64, 1), tensor(1, 8, 32, 64, 1), tensor(1, 8, 32, 512, 1)
        wc, bc, wr, br = blob(9 * 64 * 64 * 2), blob(64 * 4), blob(64 * 128 * 2), blob(64 * 4)
        ok = lambda *a: lib.rtm3d_op_conv64_root(ctx, *a)
        base = [t_in, 0, t_cat, 64, 1, wc, bc, wr, br, t_out, 0, 1, t_pool, 0, t_s2d, 256]
        assert ok(*base) == 0
        for idx, bad, what in ((2, t_out, b'overlaps'),          # x1 in the root's output tensor at the same channels
                               (5, br, b'blob size'),            # wrong conv weight blob
                               (12, t_in, b'pooled'),            # pooled output at full resolution
                               (15, 320, b'space-to-depth'),     # 4 x 64 channels do not fit behind offset 320
                               (14, t_out, b'space-to-depth')):  # copy at full resolution
            args = list(base)
            args[idx] = bad
            if idx == 2:
                args[3] = 0
            assert ok(*args) != 0 and what in lib.rtm3d_last_error(), (idx, lib.rtm3d_last_error())
        # conv descriptor: a tap naming channels outside the tensor; a space-to-depth copy on an odd map / with the wrong kernel
        d = _lib.ConvDesc()
        t_a, t_b = tensor(1, 8, 16, 128, 1), tensor(1, 8, 16, 128, 1)
        d.in_tensor, d.out_tensor, d.res_tensor, d.s2d_tensor, d.softmax_stat_slot = t_a, t_b, -1, 0, -1
        d.Hm, d.Wm, d.in_stride, d.out_scale, d.cin, d.cout, d.groups, d.ntaps = 8, 16, 1, 1, 64, 128, 1, 2
        d.kernel, d.bn_tile = 0, 128
        d.w_blob, d.bias_blob = blob(2 * 128 * 64 * 2), blob(128 * 4)
        d.tap_dc[0][1] = 64
        assert lib.rtm3d_op_conv(ctx, ctypes.byref(d)) == 0, lib.rtm3d_last_error()
        d.tap_dc[0][1] = 96
        assert lib.rtm3d_op_conv(ctx, ctypes.byref(d)) != 0 and b'outside' in lib.rtm3d_last_error()
        d.tap_dc[0][1] = 64
        t_half = tensor(1, 4, 8, 512, 0)
        d.s2d_tensor, d.s2d_coff = t_half + 1, 0          # tensor id + 1 (0 = none)
        assert lib.rtm3d_op_conv(ctx, ctypes.byref(d)) == 0, lib.rtm3d_last_error()
        d.s2d_coff = 8
        assert lib.rtm3d_op_conv(ctx, ctypes.byref(d)) != 0 and b'space-to-depth' in lib.rtm3d_last_error()
        d.s2d_coff, d.kernel = 0, 2
        assert lib.rtm3d_op_conv(ctx, ctypes.byref(d)) != 0
        # gather_peak_patches: capacity
        z = torch.zeros(4, device='cuda')
        rc = lib.rtm3d_gather_peak_patches(None, z.data_ptr(), 8, 8, 256, 1, 2, 100, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 150, 15, 1600)
        assert rc != 0 and b'cannot hold' in lib.rtm3d_last_error()
    finally:
        lib.rtm3d_ctx_destroy(ctx)
