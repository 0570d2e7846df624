"""GPU parity of the implicit-GEMM conv2d (fwd / dgrad / wgrad, through the C ABI) against torch CPU fp32
(the reference's conv is torch's: retinanet.py:26-28 etc.).  Tolerance: 1e-3 relative (north_star) — the fp32 MFMA
is an exact fmaf chain, observed error is ~1e-6."""
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda'

# (N, C, H, W, K, R, stride, pad, bias, relu)  — every ResNet-34 / RetinaNet geometry class + ragged edges
CASES = [
    (2, 64, 56, 56, 64, 3, 1, 1, False, False),     # layer1 3x3
    (2, 64, 56, 56, 128, 3, 2, 1, False, False),    # layer2 strided 3x3
    (2, 64, 56, 56, 128, 1, 2, 0, False, False),    # 1x1/2 downsample
    (2, 128, 28, 28, 128, 3, 1, 1, False, False),
    (2, 256, 14, 14, 256, 3, 1, 1, False, False),
    (3, 512, 7, 7, 512, 3, 1, 1, False, False),
    (2, 3, 64, 64, 64, 7, 2, 3, False, False),      # stem 7x7/2, 3 channels (padded to 4 by ops)
    (2, 256, 16, 16, 36, 3, 1, 1, True, False),     # RetinaNet regression output: K=36, bias
    (1, 256, 8, 8, 180, 3, 1, 1, True, False),      # classification output K=180
    (2, 256, 16, 16, 256, 3, 1, 1, True, True),     # head conv + fused ReLU
    (1, 512, 16, 16, 256, 1, 1, 0, True, False),    # FPN lateral 1x1
    (2, 8, 13, 11, 12, 3, 1, 1, True, False),       # ragged: odd sizes, tiny channels
    (1, 4, 5, 5, 4, 3, 2, 1, False, False),         # tiny
    (5, 16, 9, 9, 20, 3, 2, 1, True, True),
    (2, 32, 15, 15, 32, 3, 2, 1, False, False),     # strided dgrad, odd size: one launch per output-parity class
    (2, 16, 24, 20, 16, 7, 2, 3, False, False),     # strided dgrad, 49 taps in four parity classes of ONE launch
    (3, 256, 16, 16, 256, 3, 2, 1, True, False),    # RetinaNet P6: merged classes on the BK=32 kernel
    (2, 256, 4, 4, 256, 3, 1, 1, True, True),       # RetinaNet head tower on P7 at 512x512: 32 output pixels
    (2, 256, 8, 8, 256, 3, 1, 1, True, True),       # ... on P6
    (2, 256, 64, 64, 256, 3, 1, 1, True, True),     # ... on P3 (the bulk of the head FLOPs)
    (2, 256, 4, 4, 36, 3, 1, 1, True, False),       # regression output conv on P7
    (2, 256, 32, 32, 256, 3, 1, 1, True, True),     # head tower on P4 / P5 at 512x512, 2 images: 128- and 32-tile grids
    (2, 256, 16, 16, 256, 3, 1, 1, True, True),
    (16, 256, 16, 16, 256, 3, 1, 1, True, True),    # ... at the benchmark's 16 images
    (16, 256, 8, 8, 256, 3, 1, 1, True, True),
    (16, 256, 4, 4, 256, 3, 1, 1, True, True),
    (2, 256, 16, 16, 180, 3, 1, 1, True, 2),        # classification output conv with the SIGMOID in the epilogue (retinanet.py:286)
    (16, 256, 8, 8, 180, 3, 1, 1, True, 2),
    (1, 256, 64, 64, 180, 3, 1, 1, True, 2),
    (3, 32, 9, 7, 24, 3, 1, 1, True, 2),
    # one tap, C % 4 == 0 but not a multiple of the k block: the KTAIL instantiation (forward: C, dgrad: K) — round 4
    (1024, 204, 1, 1, 1000, 1, 1, 0, True, True),   # the tabular MLP's layers as 1x1 convolutions
    (1024, 1000, 1, 1, 500, 1, 1, 0, True, True),
    (1024, 500, 1, 1, 4, 1, 1, 0, True, False),
    (70, 36, 1, 1, 44, 1, 1, 0, False, False),      # ragged tiles, k tail of 4 in a 32-wide block
    (2, 40, 9, 7, 100, 1, 1, 0, True, True),        # a 1x1 convolution over pixels
    (3, 52, 8, 8, 60, 1, 2, 0, True, False),        # 1x1 / stride 2
]


@pytest.mark.parametrize('case', CASES, ids=[str(c) for c in CASES])
def test_conv2d_fwd_bwd(case):
    from neuralnetworklibrary_amd import ops
    N, C, H, W, K, R, stride, pad, has_bias, relu = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, R, R, generator=g) / (C * R * R) ** 0.5
    b = torch.randn(K, generator=g) if has_bias else None
    xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    bc = b.clone().requires_grad_(True) if has_bias else None
    ref = F.conv2d(xc, wc, bc, stride=stride, padding=pad)
    pre = ref.detach()
    if relu:
        ref = torch.sigmoid(ref) if relu == 2 else F.relu(ref)
    dy = torch.randn(ref.shape, generator=g)
    if relu == 1:
        # the ReLU gate is a step: an output within rounding distance of zero may be gated differently by two correct fp32
        # implementations (a dozen of the 2 M outputs of the largest case) — no gradient flows through those in this test
        dy = dy * (pre.abs() > 1e-4)
    ref.backward(dy)

    xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    bg = b.to(DEV).requires_grad_(True) if has_bias else None
    out = ops.conv2d(xg, wg, bg, stride, pad, relu)
    assert out.shape == ref.shape
    out.backward(dy.to(DEV))
    scale = ref.abs().max().item()
    assert_close(out, ref, rtol=1e-4, atol=1e-5 * scale, msg='y')
    assert_close(xg.grad, xc.grad, rtol=1e-4, atol=1e-5 * xc.grad.abs().max().item(), msg='dx')
    assert_close(wg.grad, wc.grad, rtol=1e-4, atol=1e-5 * wc.grad.abs().max().item(), msg='dw')
    if has_bias:
        assert_close(bg.grad, bc.grad, rtol=1e-4, atol=1e-5 * bc.grad.abs().max().item(), msg='db')


@pytest.mark.parametrize('N,C,h,w,K,bias', [(2, 1024, 16, 16, 256, True), (16, 512, 32, 32, 256, True), (1, 64, 3, 5, 32, False)])
def test_conv_add_upsampled_is_the_fpn_merge(N, C, h, w, K, bias):
    """`P5_upsampled + P4_1(C4)` (reference PyramidFeatures.forward, retinanet.py:131-141) in the lateral 1x1 convolution's
    epilogue: forward and all four gradients (x, weight, bias, the small map) against torch's conv2d + nn.Upsample + add."""
    from neuralnetworklibrary_amd import ops
    g = torch.Generator().manual_seed(N * 7 + C)
    x = torch.randn(N, C, 2 * h, 2 * w, generator=g)
    wt = torch.randn(K, C, 1, 1, generator=g) / C ** 0.5
    b = torch.randn(K, generator=g) if bias else None
    small = torch.randn(N, K, h, w, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (x, wt, small)] + ([b.clone().requires_grad_(True)] if bias else [None])
    ref = F.conv2d(leaves[0], leaves[1], leaves[3]) + torch.nn.Upsample(scale_factor=2, mode='nearest')(leaves[2])
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    dev = [t.to(DEV).requires_grad_(True) for t in (x, wt, small)] + ([b.to(DEV).requires_grad_(True)] if bias else [None])
    out = ops.conv_add_upsampled(dev[0], dev[1], dev[3], dev[2])
    out.backward(dy.to(DEV))
    assert_close(out, ref, rtol=1e-4, atol=1e-5 * ref.abs().max().item(), msg='y')
    for name, a, r in zip(('dx', 'dw', 'dsmall', 'db'), dev, leaves):
        if a is not None:
            assert_close(a.grad, r.grad, rtol=1e-4, atol=1e-5 * r.grad.abs().max().item(), msg=name)


def test_conv2d_linearity_full_size():
    """BASELINE-size property check (layer1 geometry at bs=64): conv(a*x1 + x2) == a*conv(x1) + conv(x2)."""
    from neuralnetworklibrary_amd import ops
    g = torch.Generator(device=DEV).manual_seed(5)
    x1 = torch.randn(64, 64, 56, 56, device=DEV, generator=g)
    x2 = torch.randn(64, 64, 56, 56, device=DEV, generator=g)
    w = torch.randn(64, 64, 3, 3, device=DEV, generator=g) / 24.0
    y = ops.conv2d(2.0 * x1 + x2, w, None, 1, 1)
    y12 = 2.0 * ops.conv2d(x1, w, None, 1, 1) + ops.conv2d(x2, w, None, 1, 1)
    assert_close(y, y12, rtol=1e-4, atol=1e-4, msg='linearity')


# Balanced schedule (IgemmTapsParams::bal): BASELINE-size grids that are not a multiple of the 256 CUs — tail tiles cut into
# k slices (layer3), base split-K 2 + tail (layer4), everything-is-tail (small batch), ragged M, and the bias+ReLU epilogue
# moving into the slab-reduce kernel.  Checked against the plain launch (NNL_IGEMM_BALANCE=0) and torch (GPU fp32 conv is not
# trusted: compare with the CPU fp32 reference on a slice of the batch).
BAL_CASES = [
    (64, 256, 14, 256, 3, 1, 1, False, False),
    (64, 512, 7, 512, 3, 1, 1, False, False),
    (64, 128, 28, 128, 3, 1, 1, False, False),
    (64, 256, 14, 256, 3, 1, 1, True, True),        # bias + ReLU applied by the slab-reduce kernel
    (63, 256, 14, 256, 3, 1, 1, True, False),       # ragged M: the tail tile row is partial
    (62, 512, 7, 512, 3, 1, 1, False, False),       # split-K 2 for every tile, no tail
]


@pytest.mark.parametrize('kernel', ['direct', 'winograd', 'winograd2d'])
@pytest.mark.parametrize('case', BAL_CASES, ids=[str(c) for c in BAL_CASES])
def test_conv2d_balanced_schedule(case, kernel, monkeypatch):
    """kernel: the direct implicit-GEMM kernel (NNL_CONV_WINO=0; its schedule switch is NNL_IGEMM_BALANCE) and the fused Winograd
    F(2,3) kernel that serves these 3x3 / stride 1 cases by default (wino.hip; NNL_WINO_BALANCE), or its 2-D F(2x2,3x3) sibling (wino2.hip,
    forced with NNL_CONV_WINO=3) — the same schedule, the same checks."""
    from neuralnetworklibrary_amd import ops
    from neuralnetworklibrary_amd._lib import lib
    N, C, H, K, R, stride, pad, has_bias, relu = case
    BAL = 'NNL_IGEMM_BALANCE' if kernel == 'direct' else 'NNL_WINO_BALANCE'
    monkeypatch.setenv('NNL_CONV_WINO', {'direct': '0', 'winograd': '2', 'winograd2d': '3'}[kernel]); lib.nnl_reload_env()    # 2 / 3: wherever it applies
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, C, H, H, generator=g)
    w = torch.randn(K, C, R, R, generator=g) / (C * R * R) ** 0.5
    b = torch.randn(K, generator=g) if has_bias else None
    geom = ops._geom(N, H, H, C, K, R, R, stride, pad)
    if os.environ.get(BAL, '1') != '0':          # (the A/B switch turns the schedule off: plain-grid parity only)
        assert lib.nnl_conv2d_fwd_workspace_bytes(geom) > 0, 'case does not exercise the balanced schedule'
    ws_on = lib.nnl_conv2d_fwd_workspace_bytes(geom)

    def run():
        xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
        bg = b.to(DEV).requires_grad_(True) if has_bias else None
        out = ops.conv2d(xg, wg, bg, stride, pad, relu)
        gd = torch.Generator().manual_seed(8)
        dy = torch.randn(out.shape, generator=gd).to(DEV)
        out.backward(dy)
        return out.detach(), xg.grad, wg.grad, dy

    y_bal, dx_bal, dw_bal, dy = run()
    y_again = run()[0]
    assert torch.equal(y_bal, y_again), 'balanced schedule must be bitwise reproducible'
    monkeypatch.setenv(BAL, '0'); lib.nnl_reload_env()
    if kernel == 'direct':
        assert lib.nnl_conv2d_fwd_workspace_bytes(geom) == 0
    else:                                            # the transformed filter stays; the slabs go
        assert 0 < lib.nnl_conv2d_fwd_workspace_bytes(geom) < ws_on
    y_pl, dx_pl, dw_pl, _ = run()
    monkeypatch.delenv(BAL); monkeypatch.delenv('NNL_CONV_WINO'); lib.nnl_reload_env()
    sy, sx = y_pl.abs().max().item(), dx_pl.abs().max().item()
    assert_close(y_bal, y_pl, rtol=1e-5, atol=2e-6 * sy, msg='y balanced vs plain')
    assert_close(dx_bal, dx_pl, rtol=1e-5, atol=2e-6 * sx, msg='dx balanced vs plain')
    assert_close(dw_bal, dw_pl, rtol=1e-5, atol=1e-6 * dw_pl.abs().max().item(), msg='dw')
    # torch CPU fp32 on the last images (they live in the tail tiles) and the first
    for sl in (slice(0, 2), slice(N - 2, N)):
        xc = x[sl].clone().requires_grad_(True)
        ref = F.conv2d(xc, w, b, stride=stride, padding=pad)
        if relu:
            ref = F.relu(ref)
        ref.backward(dy[sl].cpu())
        assert_close(y_bal[sl], ref, rtol=1e-4, atol=1e-5 * sy, msg='y vs torch')
        assert_close(dx_bal[sl], xc.grad, rtol=1e-4, atol=1e-5 * sx, msg='dx vs torch')


@pytest.mark.parametrize('give_first', [False, True], ids=['producer_runs_first', 'consumer_runs_first'])
@pytest.mark.parametrize('stride', [1, 2])
def test_grad_slot_projection_shortcut_hand_over(stride, give_first):
    """A block input feeds conv1 (3x3, the GradSlot consumer) and the projection shortcut (1x1, the producer that parks its input
    gradient in the slot).  Whatever order autograd runs the two backward nodes in — there is no dependency between them — the
    gradient of the input must be the sum of both (the consumer closes the slot; a late producer returns its gradient normally)."""
    from neuralnetworklibrary_amd import ops
    g = torch.Generator().manual_seed(21 + stride)
    N, C, K, H = 2, 32, 48, 12
    x = torch.randn(N, C, H, H, generator=g)
    w1 = torch.randn(K, C, 3, 3, generator=g) * 0.1
    wd = torch.randn(K, C, 1, 1, generator=g) * 0.1
    xr = x.double().requires_grad_(True)
    yr = F.conv2d(xr, w1.double(), None, stride, 1) + F.conv2d(xr, wd.double(), None, stride, 0)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())

    xg = x.to(DEV).requires_grad_(True)
    w1g = w1.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wdg = wd.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    slot = ops.GradSlot()
    if give_first:          # created first => its backward node runs LAST: the consumer has closed the slot by then
        yd = ops.conv2d(xg, wdg, None, stride, 0, give_slot=slot)
        y1 = ops.conv2d(xg, w1g, None, stride, 1, grad_slot=slot)
    else:
        y1 = ops.conv2d(xg, w1g, None, stride, 1, grad_slot=slot)
        yd = ops.conv2d(xg, wdg, None, stride, 0, give_slot=slot)
    (y1 + yd).backward(dy.to(DEV))
    assert slot.tensor is None and slot.closed
    assert_close(xg.grad, xr.grad.float(), 1e-4, 1e-5 * xr.grad.abs().max().item(), 'dx = dgrad(conv1) + dgrad(shortcut)')


DMA_CASES = [c for c in CASES if c[1] % 32 == 0 and c[4] % 4 == 0][:6] + [(16, 64, 56, 56, 64, 3, 1, 1, False, False),
                                                                            (16, 256, 14, 14, 256, 3, 1, 1, True, True)]


@pytest.mark.parametrize('case', DMA_CASES, ids=[str(c) for c in DMA_CASES])
def test_conv2d_lds_dma_staging_variant(case, monkeypatch):
    """The LDS-DMA staging variant of the 64x64 kernels (NNL_IGEMM_DMA, off by default: DESIGN.md switches table) computes the
    same convolution: forward / dgrad through `buffer_load ... lds` into the swizzled unpadded image, three (BK 16) or two
    (BK 32) buffers, on plain and balanced grids."""
    monkeypatch.setenv('NNL_IGEMM_DMA', '3')
    from neuralnetworklibrary_amd._lib import lib
    lib.nnl_reload_env()
    test_conv2d_fwd_bwd(case)


def test_shared_conv_weight_under_gradsync_is_not_aliased():
    """under data parallelism the wgrad kernel writes dW straight into the all-reduce bucket — legal only for
    a weight used ONCE per step.  RetinaNet's heads apply the same convolutions to 5 pyramid levels (reference
    retinanet.py:267-268 / Vision.py:1462-1466): the five per-use gradients must be summed by autograd, not overwrite each
    other.  GradSync at world size 1 (no collective) shows the aliasing by itself."""
    import torch.nn as nn
    from neuralnetworklibrary_amd import dist as nd
    from neuralnetworklibrary_amd.Applications.VisionModels.retinanet import HipConv2d
    torch.manual_seed(0)

    class Head(nn.Module):
        def __init__(self):
            super().__init__()
            self.shared = HipConv2d(16, 16, 3, padding=1)          # used on every level
            self.once = HipConv2d(16, 16, 3, padding=1)            # used once: stays on the in-place path

        def forward(self, xs):
            return sum(self.shared(x).sum() for x in xs) + self.once(xs[0]).square().sum()

    net = Head().to(DEV)
    xs = [torch.randn(2, 16, s, s, device=DEV) for s in (16, 8, 4, 2, 1)]
    net(xs).backward()
    want = {n: p.grad.clone() for n, p in net.named_parameters()}
    for p in net.parameters():
        p.grad = None
    sync = nd.GradSync(net, bucket_mb=25.0)
    for _ in range(2):                                                # second step: buckets re-used
        for p in net.parameters():
            p.grad = None
        sync.begin()
        net(xs).backward()
        sync.finish()
        for n, p in net.named_parameters():
            assert_close(p.grad, want[n], 1e-4, 1e-4, n)
    assert sync.direct_writes >= 2, 'the single-use convolution should still write its gradient in place'


@pytest.mark.parametrize('case', [(16, 64, 16, 16, 64), (2, 128, 32, 32, 128), (12, 256, 14, 14, 256), (2, 64, 64, 64, 36), (48, 128, 8, 6, 320),   # >= 1024 output pairs each
                                  (64, 128, 7, 7, 64), (12, 64, 15, 13, 128)], ids=str)                                                      # odd sizes (2-D domain only)
@pytest.mark.parametrize('domain', ['1d', '2d'])
def test_wgrad_winograd_domain(case, domain, monkeypatch):
    """The weight gradient of a 3x3 / stride 1 / pad 1 convolution computed in a Winograd domain — F(2,3) along the width (igemm_wgrad_kernel
    WINO + wino_wgrad_finish_kernel, forced with NNL_WGRAD_WINO=2, NNL_WGRAD_WINO2D=0) or F(2x2,3x3) over output quads (round 4:
    igemm_wgrad2d_kernel + wino2d_wgrad_finish_kernel, NNL_WGRAD_WINO2D=2; odd heights / widths included) — against the direct wgrad kernel
    (NNL_WGRAD_WINO=0) and torch CPU fp32; every path bitwise reproducible run to run."""
    from neuralnetworklibrary_amd import ops
    from neuralnetworklibrary_amd._lib import lib
    N, C, H, W, K = case
    if domain == '1d' and W % 2:
        pytest.skip('the 1-D domain needs an even width')
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) / (C * 9) ** 0.5
    dy = torch.randn(N, K, H, W, generator=g)

    def run(wino, wino2d):
        monkeypatch.setenv('NNL_WGRAD_WINO', wino); monkeypatch.setenv('NNL_WGRAD_WINO2D', wino2d); lib.nnl_reload_env()
        xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
        ops.conv2d(xg, wg, None, 1, 1, False).backward(dy.to(DEV))
        return wg.grad.detach().clone()

    forced = ('2', '0') if domain == '1d' else ('1', '2')
    dw_w, dw_w2, dw_d = run(*forced), run(*forced), run('0', '0')
    dw_other = run('2', '0') if domain == '2d' and W % 2 == 0 else None
    monkeypatch.delenv('NNL_WGRAD_WINO'); monkeypatch.delenv('NNL_WGRAD_WINO2D'); lib.nnl_reload_env()
    assert torch.equal(dw_w, dw_w2), 'Winograd-domain wgrad must be bitwise reproducible'
    assert not torch.equal(dw_w, dw_d), 'the forced path did not run (same bits as the direct kernel)'
    if dw_other is not None:
        assert not torch.equal(dw_w, dw_other), 'the 2-D path did not run (same bits as the 1-D domain)'
    ref = torch.nn.grad.conv2d_weight(x, (K, C, 3, 3), dy, padding=1)
    sc = ref.abs().max().item()
    assert_close(dw_w, dw_d, rtol=1e-5, atol=2e-6 * sc, msg='dw Winograd vs direct')
    assert_close(dw_w, ref, rtol=1e-4, atol=1e-5 * sc, msg='dw Winograd vs torch')


@pytest.mark.parametrize('mode', ['2', '3'], ids=['winograd', 'winograd2d'])
def test_prepared_winograd_filters_match_per_call_transform(mode, monkeypatch):
    """ops.prepare_forward / prepare_backward transform the Winograd filters of all layers in one launch each (nnl_wino_filter_multi)
    for the layers that took the Winograd kernel at their last call; the convolutions then run nnl_conv2d_fwd_pre / _dgrad_pre on
    them.  Same bits as the per-call transform (NNL_WINO_PREPARE=0), and the window closes with finish_backward()."""
    from neuralnetworklibrary_amd import ops
    from neuralnetworklibrary_amd._lib import lib
    import torch.nn as nn
    from neuralnetworklibrary_amd.Applications.VisionModels.retinanet import HipConv2d
    monkeypatch.setenv('NNL_CONV_WINO', mode); lib.nnl_reload_env()       # the 1-D / 2-D Winograd kernel wherever it applies (small test shapes)
    torch.manual_seed(3)
    net = nn.Sequential(HipConv2d(64, 128, 3, padding=1), nn.ReLU(), HipConv2d(128, 64, 3, padding=1)).to(DEV)
    x = torch.randn(4, 64, 20, 18, device=DEV)
    dy = torch.randn(4, 64, 20, 18, device=DEV)

    def step(prepare):
        monkeypatch.setenv('NNL_WINO_PREPARE', '1' if prepare else '0')
        xg = x.clone().requires_grad_(True)
        for p in net.parameters():
            p.grad = None
        ops.prepare_forward(net)
        n_fwd = len(ops._WINO_U_FWD)
        y = net(xg)
        ops.prepare_backward(net)
        n_bwd = len(ops._WINO_U_BWD)
        try:
            y.backward(dy)
        finally:
            ops.finish_backward()
        assert not ops._WINO_U_FWD and not ops._WINO_U_BWD and not ops._WT_ACTIVE, 'the prepared-filter window stayed open'
        return y.detach(), xg.grad, [p.grad.clone() for p in net.parameters()], n_fwd, n_bwd

    ref = step(False)                       # also teaches _WINO_PREF which layers take the Winograd kernel
    assert ref[3] == 0 and ref[4] == 0
    got = step(True)
    assert got[3] == 2 and got[4] == 2, 'prepared filters were not built for both layers (fwd %d, dgrad %d)' % (got[3], got[4])
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    for a, b in zip(got[2], ref[2]):
        assert torch.equal(a, b)
    monkeypatch.delenv('NNL_CONV_WINO'); lib.nnl_reload_env()


@pytest.mark.parametrize('case', [(2, 64, 12, 10, 64, None), (3, 32, 9, 7, 36, None), (4, 128, 14, 14, 128, (2, 4)), (1, 16, 2, 2, 8, None),
                                  (2, 64, 17, 33, 96, (1, 3))], ids=str)
@pytest.mark.parametrize('chunk', ['0', '32'], ids=['position-major', 'chunk32'])
def test_winograd_2d_debug_entry(case, chunk, monkeypatch):
    """The 2-D F(2x2, 3x3) kernel (csrc/wino2.hip) through its debug entry nnl_debug_conv_wino2_fwd (the dispatcher takes it from ~500
    quad tiles up; profiles/README.md has the measurements): forward with bias / addend / ReLU / BatchNorm partial sums and the flipped dgrad
    filter, odd heights and widths, plain grid and forced k-slicing (in-kernel slab fix-up), against torch CPU fp32."""
    _winograd_2d_debug_entry(case, chunk, '0', monkeypatch)


@pytest.mark.parametrize('case', [(2, 64, 12, 10, 64, None), (8, 512, 7, 7, 512, None), (4, 128, 14, 14, 128, None), (1, 32, 2, 2, 32, None),
                                  (2, 64, 17, 33, 96, None), (8, 256, 14, 14, 256, None)], ids=str)
@pytest.mark.parametrize('pos', ['1', '2', '4'], ids=['cs1', 'cs2', 'cs4'])
def test_winograd_2d_position_split(case, pos, monkeypatch):
    """Round 5, the small-grid mode of the 2-D kernel (wino2_kernel<32, 4, POS>): one position (xi, nu) — or a channel slice of one — per
    workgroup, M slabs, the tile's last arriver applies A^T M A.  Forced with NNL_WINO2_POS = channel slices per position (a count the
    shape does not allow falls back to the planner's pick); the same checks as the debug-entry test, incl. the ResNet-34 14^2 / 7^2 stages
    at 8 images (the shapes it was built for), odd sizes, BatchNorm partials, bitwise repeatability and counters back to zero."""
    _winograd_2d_debug_entry(case, '0', pos, monkeypatch)


def _winograd_2d_debug_entry(case, chunk, pos, monkeypatch):
    from neuralnetworklibrary_amd._lib import lib, ptr, stream, check
    N, C, H, W, K, forced = case
    monkeypatch.setenv('NNL_WINO2_CHUNK', chunk)              # k order: whole C per position (default) / 32-channel chunks outermost
    monkeypatch.setenv('NNL_WINO2_POS', pos)                  # 0: never the position-split instantiation; n: n channel slices per position
    if forced:
        monkeypatch.setenv('NNL_WINO_PLAN_KS', str(forced[0])); monkeypatch.setenv('NNL_WINO_PLAN_S', str(forced[1]))
    lib.nnl_reload_env()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, H, W, C, generator=g)
    w = torch.randn(K, 3, 3, C, generator=g) / (C * 9) ** 0.5
    b = torch.randn(K, generator=g)
    add = torch.randn(N, H, W, K, generator=g)
    piv = torch.randn(K, generator=g) * 0.1
    dy = torch.randn(N, H, W, K, generator=g)
    counters = torch.zeros(4096, dtype=torch.int32, device=DEV)
    wsb = max(lib.nnl_debug_conv_wino2_workspace_bytes(N, H, W, C, K), lib.nnl_debug_conv_wino2_workspace_bytes(N, H, W, K, C))
    ws = torch.empty(wsb // 4 + 4, device=DEV)
    rows = (N * ((H + 1) // 2) * ((W + 1) // 2) + 63) // 64
    part = torch.zeros(rows, K, 2, device=DEV)
    xd, wd, bd, addd, pivd = x.to(DEV), w.to(DEV), b.to(DEV), add.to(DEV), piv.to(DEV)
    y = torch.empty(N, H, W, K, device=DEV)

    def run(xin, filt, bias, addt, out, cc, kk, relu, flip, bn):
        check(lib.nnl_debug_conv_wino2_fwd(ptr(xin), ptr(filt), ptr(bias), ptr(addt), ptr(out), ptr(ws), wsb, ptr(counters), counters.numel(),
                                           ptr(part) if bn else None, ptr(pivd) if bn else None, N, H, W, cc, kk, relu, flip, stream()))

    ref_lin = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), b, padding=1).permute(0, 2, 3, 1)
    sc = ref_lin.abs().max().item()
    run(xd, wd, bd, None, y, C, K, 1, 0, False)
    assert_close(y, torch.relu(ref_lin), rtol=1e-4, atol=1e-5 * sc, msg='2-D Winograd forward + bias + ReLU')
    y1 = y.clone()
    run(xd, wd, bd, None, y, C, K, 1, 0, False)
    assert torch.equal(y, y1), 'bitwise reproducible'
    run(xd, wd, bd, addd, y, C, K, 0, 0, True)
    ref2 = ref_lin + add
    assert_close(y, ref2, rtol=1e-4, atol=1e-5 * sc, msg='2-D Winograd forward + addend')
    d = (ref2 - piv).reshape(-1, K).double()
    s1, s2 = part[:, :, 0].double().sum(0).cpu(), part[:, :, 1].double().sum(0).cpu()
    assert ((s1 - d.sum(0)).abs().max() / d.abs().sum(0).max()).item() < 1e-5
    assert ((s2 - (d * d).sum(0)).abs().max() / (d * d).sum(0).max()).item() < 1e-5
    assert int(counters.abs().sum()) == 0, 'tile counters back to zero'
    if K % 16 == 0:                                           # the dgrad direction: dy has K channels
        wt = w.permute(3, 1, 2, 0).contiguous()               # [C][R][S][K]
        dx = torch.empty(N, H, W, C, device=DEV)
        run(dy.to(DEV), wt.to(DEV), None, None, dx, K, C, 0, 1, False)
        refdx = torch.nn.grad.conv2d_input((N, C, H, W), w.permute(0, 3, 1, 2).contiguous(), dy.permute(0, 3, 1, 2).contiguous(),
                                           padding=1).permute(0, 2, 3, 1)
        assert_close(dx, refdx, rtol=1e-4, atol=1e-5 * refdx.abs().max().item(), msg='2-D Winograd dgrad filter')
    if forced:
        monkeypatch.delenv('NNL_WINO_PLAN_KS'); monkeypatch.delenv('NNL_WINO_PLAN_S')
    monkeypatch.delenv('NNL_WINO2_CHUNK'); monkeypatch.delenv('NNL_WINO2_POS')
    lib.nnl_reload_env()


def test_shared_parameters_fan_out_sums_the_per_call_gradients_in_one_launch():
    """ops.shared_params (round 5): RetinaNet's head convolutions run on five pyramid levels; their weight / bias gradients used to be
    summed by autograd's chain of accumulation kernels (91 ATen adds per step).  Inside the context every call takes an alias of the
    parameter and one node adds the per-call gradients with ONE nnl_sum_tensors launch: same gradients (up to the order of the
    additions), input gradients untouched, nothing left behind in the registry, and outside autograd the parameters are used directly."""
    from neuralnetworklibrary_amd import ops
    from neuralnetworklibrary_amd.Applications.VisionModels.retinanet import HipConv2d
    torch.manual_seed(3)
    conv = HipConv2d(32, 48, 3, padding=1, bias=True).to(DEV)
    g = torch.Generator().manual_seed(9)
    xs = [torch.randn(2, 32, s, s, generator=g).to(DEV).requires_grad_(True) for s in (24, 12, 6, 3, 2)]

    def run(fan):
        conv.weight.grad = conv.bias.grad = None
        for x in xs:
            x.grad = None
        if fan:
            with ops.shared_params([conv], len(xs)):
                outs = [conv(x) for x in xs]
        else:
            outs = [conv(x) for x in xs]
        assert not ops._FAN
        sum((o * o).sum() for o in outs).backward()
        return conv.weight.grad.clone(), conv.bias.grad.clone(), [x.grad.clone() for x in xs]
    w0, b0, x0 = run(False)
    w1, b1, x1 = run(True)
    w2, b2, _ = run(True)
    assert torch.equal(w1, w2) and torch.equal(b1, b2), 'bitwise reproducible'
    assert w1.stride() == conv.weight.stride()                      # the gradient keeps the parameter's KRSC layout
    assert_close(w1, w0, rtol=1e-5, atol=1e-6 * w0.abs().max().item(), msg='weight gradient: one sum launch vs autograd accumulation')
    assert_close(b1, b0, rtol=1e-5, atol=1e-6 * b0.abs().max().item(), msg='bias gradient')
    for a, b in zip(x1, x0):
        assert torch.equal(a, b), 'input gradients do not depend on the fan-out'
    with torch.no_grad(), ops.shared_params([conv], 5):
        assert not ops._FAN                                          # no autograd: nothing to fan out
        assert_close(conv(xs[1]), conv(xs[1]), rtol=0, atol=0, msg='eval')
