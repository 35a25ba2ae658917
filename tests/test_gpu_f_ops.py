"""GPU parity of the dense / gather kernels against fp32 references: torch CPU conv for the
floating-point MFMA kernel (tolerance stated per test), the oracle C op for RoIAlignRotated."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, relu, res_mode
    (2, 20, 24, 64, 256, (3, 3), (1, 1), (1, 1), 1, 1),
    (1, 33, 17, 128, 128, (3, 3), (1, 1), (1, 1), 0, 0),
    (2, 16, 16, 256, 512, (1, 1), (2, 2), (0, 0), 0, 0),
    (1, 40, 36, 4, 64, (7, 7), (2, 2), (3, 3), 1, 0),
    (3, 16, 33, 256, 256, (2, 2), (2, 1), (0, 0), 1, 0),
    (2, 8, 32, 256, 256, (2, 1), (2, 1), (0, 0), 2, 0),
    (2, 32, 32, 4, 16, (3, 3), (1, 1), (1, 1), 1, 0),
    (2, 32, 32, 16, 32, (3, 3), (1, 1), (1, 1), 1, 0),
    (1, 24, 24, 256, 72, (1, 1), (1, 1), (0, 0), 0, 0),
    (1, 16, 24, 512, 256, (1, 1), (1, 1), (0, 0), 0, 2),
    (2, 8, 32, 512, 256, (3, 3), (1, 1), (1, 1), 0, 0),
    (3, 7, 9, 64, 40, (1, 1), (1, 1), (0, 0), 2, 0),      # ragged rows and a ragged channel block, ReLU before the residual add
    (1, 12, 20, 96, 136, (1, 1), (1, 1), (0, 0), 1, 2),   # x2-upsampled residual, 3 channel blocks (last one ragged)
]


WINO_CASES = [
    # N, H, W, Cin, Cout, relu, residual, (ld_out, coff)
    (1, 2, 2, 16, 64, 0, False, None),            # a single tile
    (3, 16, 33, 256, 256, 1, True, None),         # local extractor layer3 shape: odd width, ragged last block
    (2, 15, 21, 64, 64, 2, True, None),           # odd height and width, ReLU before the residual add
    (1, 64, 64, 128, 128, 1, False, None),
    (5, 8, 32, 512, 256, 0, False, (512, 256)),   # fusion output conv writing into a wider buffer
    (2, 7, 5, 32, 192, 1, True, None),            # three channel blocks, tiny image
    (1, 5, 5, 32, 128, 2, True, None),            # wide kernel (32 tiles x 128 channels), a single 32-channel k-tile
    (2, 9, 11, 96, 256, 1, True, None),           # wide kernel, 3 k-tiles, ragged last tile block, odd sizes
    (1, 31, 40, 64, 384, 0, False, (512, 128)),   # wide kernel, three channel blocks into a wider buffer at an offset
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_winograd_matches_direct_and_torch(case):
    """glass_conv3x3_winograd_nhwc vs glass_conv2d_nhwc (same fp32 MFMA, different algebra) and vs torch CPU fp64.
    Tolerance: F(2x2,3x3) in fp32 adds a few ulp of the INPUT transform's scale; 2e-5 of the output range."""
    from glass_amd.ops import native as K
    N, H, W, Cin, Cout, relu, use_res, strided = case
    dev = _dev()
    x = _rand((N, Cin, H, W), 11)
    w = _rand((Cout, Cin, 3, 3), 12, (2.0 / (Cin * 9)) ** 0.5)
    b = _rand((Cout,), 13, 0.1)
    res = _rand((N, Cout, H, W), 14) if use_res else None
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if relu == 2:
        ref = F.relu(ref)
    if res is not None:
        ref = ref + res.double()
    if relu == 1:
        ref = F.relu(ref)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    rd = None if res is None else res.permute(0, 2, 3, 1).contiguous().to(dev)
    kw = dict(padding=1, relu=relu, residual=rd, res_mode=1 if rd is not None else 0)
    if strided is None:
        yw = K.conv2d_nhwc(xd, wd, b.to(dev), winograd=True, **kw)
        yd = K.conv2d_nhwc(xd, wd, b.to(dev), winograd=False, **kw)
    else:
        ld, coff = strided
        bufw = torch.full((N, H, W, ld), 7.0, device=dev)
        bufd = torch.full((N, H, W, ld), 7.0, device=dev)
        K.conv2d_nhwc(xd, wd, b.to(dev), winograd=True, out=bufw, out_coff=coff, **kw)
        K.conv2d_nhwc(xd, wd, b.to(dev), winograd=False, out=bufd, out_coff=coff, **kw)
        torch.cuda.synchronize()
        assert float((bufw[..., :coff] - 7.0).abs().max()) == 0.0      # untouched channels stay untouched
        yw, yd = bufw[..., coff:coff + Cout], bufd[..., coff:coff + Cout]
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    got_w = yw.cpu().permute(0, 3, 1, 2).double()
    got_d = yd.cpu().permute(0, 3, 1, 2).double()
    assert float((got_w - ref).abs().max()) <= 2e-5 * scale
    assert float((got_d - ref).abs().max()) <= 2e-5 * scale
    assert float((got_w - got_d).abs().max()) <= 2e-5 * scale


WINO43_CASES = [
    # N, H, W, Cin, Cout, relu, residual, (ld_out, coff)
    (1, 4, 4, 32, 128, 0, False, None),           # a single tile, a single k-tile
    (3, 16, 33, 256, 256, 1, True, None),         # local extractor layer3 shape: width 33 -> 9 tile columns, 3 ragged
    (2, 15, 21, 64, 128, 2, True, None),          # odd height and width, ReLU before the residual add
    (1, 64, 64, 128, 128, 1, False, None),
    (5, 8, 32, 512, 256, 0, False, (512, 256)),   # fusion output conv writing into a wider buffer
    (2, 7, 5, 32, 384, 1, True, None),            # three channel blocks, tiny image (tiles mostly padding)
    (2, 9, 11, 96, 256, 1, True, None),           # 3 k-tiles, ragged last tile block
    (1, 31, 40, 64, 384, 0, False, (512, 128)),   # three channel blocks into a wider buffer at an offset
    (2, 1, 3, 32, 128, 1, False, None),           # smaller than one tile
    # narrow shape of the kernel (32 tiles x 64 channels, 16-channel k-tiles): the 64-channel layers
    (2, 16, 16, 64, 64, 1, True, None),           # local extractor layer1 / res2 shape
    (3, 13, 9, 32, 64, 2, True, None),            # Cin = 32 (two k-tiles), ragged everything
    (1, 40, 24, 16, 192, 0, False, (256, 64)),    # a single k-tile, three 64-channel blocks into a wider buffer
    (2, 8, 8, 48, 64, 1, False, None),            # Cin = 48: not a multiple of 32 -> narrow
]


@pytest.mark.parametrize("R,H,W,Cin,Cout,relu,use_res,strided", [
    (64, 16, 33, 256, 256, 1, True, None),        # the local extractor's layer3 / layer4 BasicBlock conv2 (+ residual, ReLU after)
    (40, 16, 33, 128, 256, 1, False, None),       # layer3.0 conv1 (128 -> 256)
    (24, 16, 33, 256, 256, 0, False, (512, 256)), # into a wider buffer at a channel offset
    (6, 12, 5, 64, 64, 2, True, None),            # narrow shape, one full tile column + the strip, ReLU before the residual
    (3, 7, 9, 16, 128, 1, True, None),            # Cin = 16 (strip K = 3 x 32)
])
def test_winograd43_ragged_width_split_vs_torch(R, H, W, Cin, Cout, relu, use_res, strided):
    """Maps of width 4 k + 1 (reference local_feature_extraction.py:123: MaxPool2d(2, (2, 1), (0, 1)) -> 16 x 33): the F(4x4,3x3)
    kernel on the full tile columns (glass_conv3x3_winograd43_body_nhwc) + the last pixel column as glass_conv2d_nhwc over the
    last two input columns, against torch CPU fp64 - every column, the last one in particular - with the layer's weights
    prepared ONCE (the load-time 'col1' pack: no launch packs anything)."""
    from glass_amd.ops import native as K
    dev = _dev()
    x = _rand((R, Cin, H, W), 21)
    w = _rand((Cout, Cin, 3, 3), 22, (2.0 / (Cin * 9)) ** 0.5)
    b = _rand((Cout,), 23, 0.1)
    res = _rand((R, Cout, H, W), 24) if use_res else None
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if relu == 2:
        ref = F.relu(ref)
    if res is not None:
        ref = ref + res.double()
    if relu == 1:
        ref = F.relu(ref)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    cw = K.prepare_conv_weights(w.permute(0, 2, 3, 1).contiguous().to(dev), ragged=True)
    assert "col1" in cw.packs and True in cw.packs
    rd = None if res is None else res.permute(0, 2, 3, 1).contiguous().to(dev)
    kw = dict(padding=1, relu=relu, residual=rd, res_mode=1 if rd is not None else 0)
    n0 = K.packs_on_the_fly()
    if strided is None:
        y = K.conv2d_nhwc(xd, cw, b.to(dev), winograd="f43", **kw)
    else:
        ld, coff = strided
        buf = torch.full((R, H, W, ld), 7.0, device=dev)
        K.conv2d_nhwc(xd, cw, b.to(dev), winograd="f43", out=buf, out_coff=coff, **kw)
        torch.cuda.synchronize()
        assert float((buf[..., :coff] - 7.0).abs().max()) == 0.0 and (coff + Cout == ld or float((buf[..., coff + Cout:] - 7.0).abs().max()) == 0.0)
        y = buf[..., coff:coff + Cout]
    assert K.last_conv_path() == "winograd43r" and K.packs_on_the_fly() == n0
    got = y.cpu().permute(0, 3, 1, 2).double()
    scale = float(ref.abs().max())
    e_body = float((got[..., :W - 1] - ref[..., :W - 1]).abs().max()) / scale
    e_last = float((got[..., W - 1] - ref[..., W - 1]).abs().max()) / scale
    print(f"F(4x4,3x3) ragged split [{R},{H},{W},{Cin}]->{Cout}: max err / range = {e_body:.2e} (tile columns), {e_last:.2e} (last column, direct)")
    assert e_body <= 2e-5 and e_last <= 2e-5


def test_roi_align_up2_equals_pooling_the_materialised_upsampled_map():
    """glass_roi_align_rotated_up2: pooling a half-resolution level THROUGH nearest x2 upsampling is bit-identical to pooling the
    materialised upsampled map (same sampling grid, clamps, weights; only the tap address differs)."""
    from glass_amd.ops import native as K
    dev = _dev()
    x = _rand((3, 24, 40, 64), 51).to(dev)
    up = x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2).contiguous()
    g = torch.Generator().manual_seed(52)
    R = 40
    u = torch.rand((R, 5), generator=g)
    boxes = torch.stack([u[:, 0] * 320, u[:, 1] * 192, 8 + u[:, 2] * 200, 4 + u[:, 3] * 90, u[:, 4] * 360 - 180], 1).float().to(dev)
    boxes[0] = torch.tensor([-20.0, -10.0, 100.0, 60.0, 15.0])         # mostly outside
    boxes[1] = torch.tensor([318.0, 190.0, 40.0, 30.0, -30.0])          # across the bottom-right corner (clamps at H-1 / W-1)
    bidx = (torch.arange(R) % 3).to(torch.int32).to(dev)
    for size, sr in (((8, 32), 0), ((7, 7), 2)):
        a = K.roi_align_rotated([x], [0.25], boxes, bidx, size, sr, up2=True)
        b = K.roi_align_rotated([up], [0.25], boxes, bidx, size, sr)
        assert torch.equal(a, b)
    with pytest.raises(Exception):
        K.roi_align_rotated([x.half()], [0.25], boxes, bidx, (8, 32), 0, up2=True)


@pytest.mark.parametrize("R,H,W,Cin,Cout,relu,use_res,forced", [
    (32, 16, 33, 256, 256, 1, True, True),        # one image's 32 RoIs: 272 workgroups on the full grid, 256 on the body grid
    (32, 16, 33, 128, 256, 1, False, True),
    (5, 10, 7, 64, 64, 2, True, True),            # width 4 k + 3: only the F(2x2) split applies; 64-channel kernel
    (3, 6, 9, 32, 128, 0, False, True),
])
def test_winograd22_odd_width_split_vs_torch(R, H, W, Cin, Cout, relu, use_res, forced):
    """Odd-width maps on the F(2x2,3x3) kernels: full tile columns (glass_conv3x3_winograd_body_nhwc) + the last pixel column as
    the strip convolution - the form the small-grid routing picks for the local extractor's 16 x 33 maps when ONE image is in
    flight (reference local_feature_extraction.py:103-132 on 32 RoIs) - against torch CPU fp64, every column."""
    from glass_amd.ops import native as K
    dev = _dev()
    x = _rand((R, Cin, H, W), 31)
    w = _rand((Cout, Cin, 3, 3), 32, (2.0 / (Cin * 9)) ** 0.5)
    b = _rand((Cout,), 33, 0.1)
    res = _rand((R, Cout, H, W), 34) if use_res else None
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if relu == 2:
        ref = F.relu(ref)
    if res is not None:
        ref = ref + res.double()
    if relu == 1:
        ref = F.relu(ref)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    cw = K.prepare_conv_weights(w.permute(0, 2, 3, 1).contiguous().to(dev), ragged=True)
    rd = None if res is None else res.permute(0, 2, 3, 1).contiguous().to(dev)
    n0 = K.packs_on_the_fly()
    y = K.conv2d_nhwc(xd, cw, b.to(dev), padding=1, relu=relu, residual=rd, res_mode=1 if rd is not None else 0,
                      winograd="f22r" if forced else None)
    assert K.last_conv_path() in ("winograd128r", "winogradr"), K.last_conv_path()
    assert forced or K.packs_on_the_fly() == n0
    got = y.cpu().permute(0, 3, 1, 2).double()
    scale = float(ref.abs().max())
    e_body = float((got[..., :W - 1] - ref[..., :W - 1]).abs().max()) / scale
    e_last = float((got[..., W - 1] - ref[..., W - 1]).abs().max()) / scale
    print(f"F(2x2,3x3) odd-width split [{R},{H},{W},{Cin}]->{Cout}: max err / range = {e_body:.2e} (tile columns), {e_last:.2e} (last column)")
    assert e_body <= 5e-6 and e_last <= 5e-6


@pytest.mark.parametrize("M,Kdim,Nout,relu", [(800, 12544, 2048, 1), (100, 12544, 2048, 1), (100, 2048, 11, 0), (37, 4096, 256, 0), (300, 8192, 320, 2)])
def test_linear_splitk_matches_single_slice_and_torch(M, Kdim, Nout, relu):
    """glass_conv2d_nhwc_splitk on linear layers (box head fc1 / fc2 / predictors with one image in flight, reference
    recognizers_hybrid_head.py:320-322: few rows, K = 256 x 7 x 7): k-slices as independent workgroups + an ordered reduction
    with bias / ReLU, against the single-slice kernel and torch CPU fp64; the result is deterministic (two runs bit-identical)."""
    from glass_amd.ops import native as K
    dev = _dev()
    x = _rand((M, Kdim), 41)
    w = _rand((Nout, Kdim), 42, (2.0 / Kdim) ** 0.5)
    b = _rand((Nout,), 43, 0.1)
    ref = x.double() @ w.double().t() + b.double()
    if relu:
        ref = F.relu(ref)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    assert K._splitk_slices(K.default_routing(), M, Kdim, Kdim, Nout) > 1
    # (split=0: since round 6 the 800-row fc layers - 0.75-1.5 rounds of blocks, a long k-loop each - are routed to the bf16-split
    #  1x1 kernel by default, which is checked at the end; this test is about the split-K launch)
    rt = K.default_routing().replace(split=0)
    y = K.linear(xd, wd, bd, relu=relu, routing=rt)
    y2 = K.linear(xd, wd, bd, relu=relu, routing=rt)
    u = K.linear(xd, wd, bd, relu=relu, routing=rt.replace(splitk=False))
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    scale = float(ref.abs().max())
    e, eu = float((y.cpu().double() - ref).abs().max()) / scale, float((u.cpu().double() - ref).abs().max()) / scale
    print(f"split-K linear [{M},{Kdim}]->{Nout}: max err / range = {e:.2e} (single slice {eu:.2e})")
    assert e <= 2e-6 and float((y - u).abs().max()) <= 1e-5 * scale      # (the single slice is the LESS accurate one: one long fp32 chain)
    if M == 800:      # the default route of this shape: exact fp32 products on the bf16 pipes, ONE fp32 chain over K = 12544
        d = K.linear(xd, wd, bd, relu=relu)
        assert K.last_conv_path() == "pointwise_split"
        ed = float((d.cpu().double() - ref).abs().max()) / scale
        print(f"default route (bf16-split 1x1) [{M},{Kdim}]->{Nout}: max err / range = {ed:.2e}")
        assert ed <= 6e-6 and ed <= 1.5 * eu + 1e-6
    # layers the split does not take: a grid that already fills the chip, a short K
    assert K._splitk_slices(K.default_routing(), 8192, 2048, 2048, 512) == 0 and K._splitk_slices(K.default_routing(), 100, 256, 256, 256) == 0


@pytest.mark.parametrize("shape", [(1, 64, 64, 256, 256, 3, 1, 1, True), (1, 32, 32, 512, 512, 3, 1, 1, False), (1, 32, 32, 2048, 512, 1, 1, 0, False),
                                   (2, 16, 16, 256, 72, 3, 1, 1, False), (1, 64, 64, 1024, 512, 1, 2, 0, False), (3, 16, 33, 256, 256, 3, 1, 1, True)])
def test_conv_splitk_matches_the_single_slice_kernel(shape):
    """the same split on convolutions (res4 / res5 3x3 and 1x1 layers, strided 1x1 shortcuts, the RPN head on small levels -
    what one image per call leaves of the trunk): a slice starts in the middle of the (tap, channel) walk; residual + ReLU
    and a channel-offset output go through the reduction kernel."""
    from glass_amd.ops import native as K
    dev = _dev()
    N, H, W, Cin, Cout, k, stride, pad, res = shape
    x = _rand((N, H, W, Cin), 61).to(dev)
    w = _rand((Cout, k, k, Cin), 62, (2.0 / (k * k * Cin)) ** 0.5).to(dev)
    b = _rand((Cout,), 63, 0.1).to(dev)
    Ho, Wo = K.conv_out_size(H, W, k, k, stride, pad)
    r = _rand((N, Ho, Wo, Cout), 64).to(dev) if res else None
    on, off = K.default_routing().replace(winograd=False, pw=False), K.default_routing().replace(winograd=False, pw=False, splitk=False)
    nk = k * k * Cin // 32
    forced = [s_ for s_ in (3, 4, 6, 8, 2) if nk % s_ == 0][0]       # (the cost model decides in production; every count must be right)
    out_a = torch.zeros((N, Ho, Wo, Cout + 8), device=dev)
    out_b = torch.zeros((N, Ho, Wo, Cout + 8), device=dev)
    rule = K._splitk_slices
    K._splitk_slices = lambda *a_: forced
    try:
        a = K.conv2d_nhwc(x, w, b, stride=stride, padding=pad, relu=1, residual=r, res_mode=1 if res else 0, out=out_a, out_coff=4, routing=on)
    finally:
        K._splitk_slices = rule
    u = K.conv2d_nhwc(x, w, b, stride=stride, padding=pad, relu=1, residual=r, res_mode=1 if res else 0, out=out_b, out_coff=4, routing=off)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.permute(0, 3, 1, 2).double().cpu(), b.double().cpu(), stride=stride, padding=pad)
    ref = ref.permute(0, 2, 3, 1)
    if res:
        ref = ref + r.double().cpu()
    ref = F.relu(ref)
    scale = float(ref.abs().max())
    e = float((a[..., 4:4 + Cout].double().cpu() - ref).abs().max()) / scale
    assert e <= 3e-6, e
    assert float((a - u).abs().max()) <= 3e-6 * scale
    assert float(a[..., :4].abs().max()) == 0.0 and float(a[..., 4 + Cout:].abs().max()) == 0.0      # the window's neighbours untouched


@pytest.mark.parametrize("shape", [(1, 64, 64, 256, 256, False, 8), (1, 32, 32, 512, 512, False, 16), (32, 8, 32, 512, 256, False, 4), (1, 128, 128, 256, 256, True, 2),
                                   (3, 16, 33, 256, 256, True, 2), (3, 16, 33, 256, 256, True, (4, True)), (2, 13, 18, 128, 128, True, 4), (1, 64, 64, 256, 256, True, 2)])
def test_winograd43_splitk_matches_the_single_launch_and_torch(shape):
    """glass_conv3x3_winograd43_splitk_nhwc (round 6): k-slices of the F(4x4,3x3) kernel as gridDim.y, raw partial outputs through a
    workspace, ordered reduction with bias / ReLU / residual and a channel-offset output - against the single-launch F(4x4) kernel and
    torch CPU fp64, for every forced slice count incl. ragged maps (H, W not multiples of 4), the 16 x 33 maps in body + strip form,
    and twice (deterministic).  Reference call sites: the 3x3 layers of d2's ResNet / FPN / RPN behind glass_rcnn.py:83,87 and
    fusion_modules.py:156 with ONE image in flight (glass_runner.py:93-96)."""
    from glass_amd.ops import native as K
    dev = _dev()
    N, H, W, Cin, Cout, res, sl = shape
    x = _rand((N, H, W, Cin), 71).to(dev)
    w = K.prepare_conv_weights(_rand((Cout, 3, 3, Cin), 72, (2.0 / (9 * Cin)) ** 0.5).to(dev), "all", ragged=(W % 4 == 1))
    b = _rand((Cout,), 73, 0.1).to(dev)
    r = _rand((N, H, W, Cout), 74).to(dev) if res else None
    kw = dict(padding=1, relu=1, residual=r, res_mode=1 if res else 0)
    out_a = torch.zeros((N, H, W, Cout + 8), device=dev)
    out_b = torch.zeros((N, H, W, Cout + 8), device=dev)
    K._TLS.force_f43k = sl
    try:
        a = K.conv2d_nhwc(x, w, b, out=out_a, out_coff=4, **kw)
        assert K.last_conv_path() == "winograd43k"
        a2 = K.conv2d_nhwc(x, w, b, **kw)
        a3 = K.conv2d_nhwc(x, w, b, **kw)
    finally:
        K._TLS.force_f43k = None
    K._TLS.force_f43k = 0
    try:
        u = K.conv2d_nhwc(x, w, b, winograd="f43", out=out_b, out_coff=4, **kw)
    finally:
        K._TLS.force_f43k = None
    assert torch.equal(a2, a3) and torch.equal(a2, a[..., 4:4 + Cout])
    ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.raw.permute(0, 3, 1, 2).double().cpu(), b.double().cpu(), padding=1).permute(0, 2, 3, 1)
    if res:
        ref = ref + r.double().cpu()
    ref = F.relu(ref)
    scale = float(ref.abs().max())
    e = float((a[..., 4:4 + Cout].double().cpu() - ref).abs().max()) / scale
    print(f"F(4x4) split-K x{sl} [{N},{H},{W},{Cin}]->{Cout}: max err / range = {e:.2e}")
    assert e <= 1e-5, e
    assert float((a - u).abs().max()) <= 1e-5 * scale
    assert float(a[..., :4].abs().max()) == 0.0 and float(a[..., 4 + Cout:].abs().max()) == 0.0      # the window's neighbours untouched


def test_backbone_stem_fused_matches_the_two_launches_and_torch():
    """glass_backbone_stem_fused (conv 7x7 s2 p3 + bias + ReLU + max_pool2d(3, 2, 1) in one kernel, csrc/backbone_stem.hip) against
    the two launches it replaces and, on the smaller shapes, against torch CPU fp64 - shapes that cross the kernel's seams: more
    than one 256-column block (left halo column on the vector ALU), bands shorter than the image, a ragged last band, W / 2 not
    a multiple of 32."""
    from glass_amd.ops import native as K
    dev = _dev()
    w = _rand((64, 3, 7, 7), 31, (2.0 / 147) ** 0.5 / 32)
    b = _rand((64,), 32, 0.1)
    w4 = torch.zeros((64, 7, 7, 4))
    w4[..., :3] = w.permute(0, 2, 3, 1)
    wd, bd = w4.contiguous().to(dev), b.to(dev)
    off = K.default_routing().replace(stem=False)
    for (N, H, W) in ((2, 64, 96), (1, 36, 40), (3, 72, 1040), (1, 264, 1100), (8, 128, 544), (200, 36, 32), (1, 4, 4), (2, 8, 12)):   # (200, 36, 32): 8-row bands, ragged last band (Hp = 9); last two: smaller than one M-block
        x = _rand((N, 3, H, W), 33 + H, 60.0)
        x4 = torch.zeros((N, H, W, 4))
        x4[..., :3] = x.permute(0, 2, 3, 1)
        xd = x4.contiguous().to(dev)
        assert K.backbone_stem_supported(xd, wd)
        y = K.backbone_stem_fused(xd, wd, bd)
        u = K.maxpool2d_nhwc(K.conv2d_nhwc(xd, wd, bd, stride=2, padding=3, relu=1, routing=off), 3, 2, 1)
        torch.cuda.synchronize()
        assert y.shape == u.shape == (N, H // 4, W // 4, 64)
        scale = float(u.abs().max())
        d = float((y - u).abs().max()) / scale
        msg = f"fused stem [{N},{H},{W}]: max |fused - two launches| / range = {d:.2e}"
        assert d <= 2e-6, msg
        if H * W <= 72 * 1040:
            ref = F.max_pool2d(F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=3)), 3, 2, 1)
            e = float((y.cpu().permute(0, 3, 1, 2).double() - ref).abs().max()) / float(ref.abs().max())
            msg += f", vs torch fp64 {e:.2e}"
            assert e <= 2e-6, msg
        print(msg)
    # not a multiple of 4: the two-launch path stays
    assert not K.backbone_stem_supported(torch.zeros((1, 34, 40, 4), device=dev), wd)


@pytest.mark.parametrize("case", WINO43_CASES)
def test_winograd43_matches_direct_and_torch(case):
    """glass_conv3x3_winograd43_nhwc (F(4x4,3x3), points 0, 1, -1, 1/2, -2, inf) vs torch CPU fp64 and vs the direct kernel.
    Tolerance: 2e-5 of the output range (the kernel's documented bound; measured values are printed)."""
    from glass_amd.ops import native as K
    N, H, W, Cin, Cout, relu, use_res, strided = case
    dev = _dev()
    x = _rand((N, Cin, H, W), 11)
    w = _rand((Cout, Cin, 3, 3), 12, (2.0 / (Cin * 9)) ** 0.5)
    b = _rand((Cout,), 13, 0.1)
    res = _rand((N, Cout, H, W), 14) if use_res else None
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if relu == 2:
        ref = F.relu(ref)
    if res is not None:
        ref = ref + res.double()
    if relu == 1:
        ref = F.relu(ref)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    rd = None if res is None else res.permute(0, 2, 3, 1).contiguous().to(dev)
    kw = dict(padding=1, relu=relu, residual=rd, res_mode=1 if rd is not None else 0)
    if strided is None:
        yw = K.conv2d_nhwc(xd, wd, b.to(dev), winograd="f43", **kw)
        # width 4 k + 1: the kernel runs the k full tile columns, the last pixel column is a strip convolution ("winograd43r")
        split = W % 4 == 1 and W >= 5 and (2 * Cin) % 32 == 0
        assert K.last_conv_path() == ("winograd43r" if split else "winograd43")
        yd = K.conv2d_nhwc(xd, wd, b.to(dev), winograd=False, **kw)
        if split:                                    # ... and the unsplit form (a whole ragged tile column) still agrees
            yu = K.conv2d_nhwc(xd, wd, b.to(dev), winograd="f43", routing=K.default_routing().replace(ragged=False), **kw)
            assert K.last_conv_path() == "winograd43"
            assert float((yu - yd).abs().max()) <= 2e-5 * float(ref.abs().max())
            assert torch.equal(yu[:, :, :W - 1], yw[:, :, :W - 1]), "the full tile columns must not depend on how the last column is computed"
    else:
        ld, coff = strided
        bufw = torch.full((N, H, W, ld), 7.0, device=dev)
        bufd = torch.full((N, H, W, ld), 7.0, device=dev)
        K.conv2d_nhwc(xd, wd, b.to(dev), winograd="f43", out=bufw, out_coff=coff, **kw)
        K.conv2d_nhwc(xd, wd, b.to(dev), winograd=False, out=bufd, out_coff=coff, **kw)
        torch.cuda.synchronize()
        assert float((bufw[..., :coff] - 7.0).abs().max()) == 0.0      # untouched channels stay untouched
        assert float((bufw[..., coff + Cout:] - 7.0).abs().max()) == 0.0 if coff + Cout < ld else True
        yw, yd = bufw[..., coff:coff + Cout], bufd[..., coff:coff + Cout]
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    got_w = yw.cpu().permute(0, 3, 1, 2).double()
    got_d = yd.cpu().permute(0, 3, 1, 2).double()
    ew, ed = float((got_w - ref).abs().max()) / scale, float((got_d - ref).abs().max()) / scale
    print(f"F(4x4,3x3) {case[:5]}: max err / range = {ew:.2e} (direct kernel {ed:.2e})")
    assert ew <= 2e-5
    assert float((got_w - got_d).abs().max()) <= 2e-5 * scale


def test_winograd43_on_post_relu_activations_with_a_mean():
    """F(4x4,3x3) error depends on the INPUT's magnitude, not the output's: non-negative (post-ReLU) inputs with a
    large mean are the unfavourable case of the real network.  256 channels, inputs relu(N(1,1)): still <= 2e-5 of range."""
    from glass_amd.ops import native as K
    dev = _dev()
    x = torch.relu(_rand((2, 256, 32, 32), 21) + 1.0)
    w = _rand((256, 256, 3, 3), 22, (2.0 / (256 * 9)) ** 0.5)
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    y = K.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), w.permute(0, 2, 3, 1).contiguous().to(dev), None, padding=1,
                      winograd="f43")
    y2 = K.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), w.permute(0, 2, 3, 1).contiguous().to(dev), None, padding=1,
                       winograd=True)
    scale = float(ref.abs().max())
    e4 = float((y.cpu().permute(0, 3, 1, 2).double() - ref).abs().max()) / scale
    e2 = float((y2.cpu().permute(0, 3, 1, 2).double() - ref).abs().max()) / scale
    print(f"post-ReLU inputs, 256 ch: F(4x4) err/range {e4:.2e}, F(2x2) {e2:.2e}")
    assert e4 <= 2e-5


def test_winograd_rejects_unsupported():
    from glass_amd.ops import native as K
    from glass_amd._lib import GlassLibraryError
    dev = _dev()
    x = torch.zeros((1, 8, 8, 16), device=dev)
    w = torch.zeros((32, 3, 3, 16), device=dev)       # Cout % 64 != 0
    with pytest.raises(GlassLibraryError):
        K.conv2d_nhwc(x, w, None, padding=1, winograd=True)
    K.conv2d_nhwc(x, w, None, padding=1)              # default: silently the direct kernel


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_matches_torch(case):
    from glass_amd.ops import native as K
    N, H, W, Cin, Cout, k, s, p, relu, res_mode = case
    dev = _dev()
    x = _rand((N, Cin, H, W), 1)
    w = _rand((Cout, Cin, k[0], k[1]), 2, (2.0 / (Cin * k[0] * k[1])) ** 0.5)
    b = _rand((Cout,), 3, 0.1)
    if Cin == 4:
        x[:, 3] = 0
    ref = F.conv2d(x, w, b, stride=s, padding=p)
    res = None
    if res_mode == 1:
        res = _rand(tuple(ref.shape), 4)
        ref = ref + res
    elif res_mode == 2:
        res = _rand((N, Cout, ref.shape[2] // 2, ref.shape[3] // 2), 4)
        ref = ref + F.interpolate(res, scale_factor=2.0, mode="nearest")
    if relu == 1:
        ref = F.relu(ref)
    elif relu == 2:
        extra = _rand(tuple(ref.shape), 5)
        ref = F.relu(ref) + extra
        res, res_mode = extra, 1
    y = K.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), w.permute(0, 2, 3, 1).contiguous().to(dev), b.to(dev),
                      stride=s, padding=p, relu=relu,
                      residual=None if res is None else res.permute(0, 2, 3, 1).contiguous().to(dev), res_mode=res_mode)
    torch.cuda.synchronize()
    got = y.cpu().permute(0, 3, 1, 2)
    # fp32 MFMA is an exact-fp32 fma chain; only the summation order differs from MKL-DNN
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=2e-4)


def test_linear_big_k_and_interleaved_output():
    from glass_amd.ops import native as K
    dev = _dev()
    x = _rand((100, 12544), 1)
    w = _rand((2048, 12544), 2, (2.0 / 12544) ** 0.5)
    b = _rand((2048,), 3, 0.1)
    ref = F.relu(F.linear(x, w, b))
    y = K.linear(x.to(dev), w.to(dev), b.to(dev), relu=1)
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=5e-4)
    # channel-interleaved output (used to build the fusion input x[:, order] for free)
    xs = _rand((2, 8, 32, 256), 4)
    ws = _rand((256, 1, 1, 256), 5, 0.1)
    out = torch.zeros((2, 8, 32, 512), device=dev)
    K.conv2d_nhwc(xs.to(dev), ws.to(dev), None, out=out, out_coff=1, out_cstride=2)
    ref2 = F.conv2d(xs.permute(0, 3, 1, 2), ws.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    o = out.cpu()
    np.testing.assert_allclose(o[..., 1::2].numpy(), ref2.numpy(), rtol=1e-4, atol=1e-4)
    assert (o[..., 0::2] == 0).all()


def test_maxpool_matches_torch():
    from glass_amd.ops import native as K
    dev = _dev()
    for (k, s, p, shape) in (((3, 3), (2, 2), (1, 1), (2, 64, 30, 34)), ((2, 2), (2, 2), (0, 0), (3, 32, 16, 16)),
                             ((2, 2), (2, 1), (0, 1), (2, 128, 32, 32))):
        x = _rand(shape, 7)
        ref = F.max_pool2d(x, k, s, p)
        y = K.maxpool2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), k, s, p)
        assert torch.equal(y.cpu().permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("cfgk", ["box", "rec", "img"])
def test_roi_align_rotated_matches_oracle(cfgk):
    from glass_amd.ops import native as K
    from glass_amd.utils.synth import make_boxes
    from oracle import d2ops
    dev = _dev()
    if cfgk == "box":
        feats = [_rand((2, 256, 64 >> i, 80 >> i), 10 + i) for i in range(5)]
        scales = [1.0 / (4 << i) for i in range(5)]
        out_size, sr = (7, 7), 2
    elif cfgk == "rec":
        feats, scales, out_size, sr = [_rand((2, 256, 64, 80), 20)], [0.25], (8, 32), 0
    else:
        feats, scales, out_size, sr = [_rand((2, 4, 256, 320), 30)], [1.0], (128, 128), 2
    boxes = [make_boxes(i, 9, 256, 320) for i in range(2)]
    boxes[0][0] = torch.tensor([5.0, 4.0, 60.0, 30.0, 0.5])      # partly outside
    boxes[0][1] = torch.tensor([160.0, 128.0, 700.0, 400.0, 33.0])  # larger than the image -> top level
    boxes[1][0] = torch.tensor([100.0, 100.0, 3.0, 2.0, -90.0])   # tiny -> lowest level, gh=gw=1 when sr=0
    ref = d2ops.roi_pooler(feats, scales, boxes, out_size, sr)
    bcat = torch.cat(boxes).contiguous()
    bidx = torch.cat([torch.full((len(b),), i, dtype=torch.int32) for i, b in enumerate(boxes)])
    y = K.roi_align_rotated([f.permute(0, 2, 3, 1).contiguous().to(dev) for f in feats], scales, bcat.to(dev), bidx.to(dev),
                            out_size, sr)
    np.testing.assert_allclose(y.cpu().permute(0, 3, 1, 2).numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)


def test_preprocess_and_resize():
    from glass_amd.ops import native as K
    from glass_amd.utils.synth import make_image
    dev = _dev()
    img = make_image(0, 37, 53)
    chw = K.image_u8hwc_to_chw(img.to(dev), (37, 53))
    assert torch.equal(chw.cpu(), img.permute(2, 0, 1).float())
    up = K.image_u8hwc_to_chw(img.to(dev), (44, 64))
    ref = F.interpolate(img.permute(2, 0, 1).float()[None], size=(44, 64), mode="bilinear", align_corners=False)[0]
    np.testing.assert_allclose(up.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-3)
    batch = torch.full((2, 64, 64, 4), 7.0, device=dev)
    mean, std = [103.53, 116.28, 123.675], [1.0, 1.0, 1.0]
    K.preprocess_image(chw, mean, std, batch, 1)
    b = batch.cpu()
    exp = torch.zeros(64, 64, 4)
    exp[:37, :53, :3] = (img.float() - torch.tensor(mean))
    assert torch.equal(b[1], exp) and (b[0] == 7).all()


@pytest.mark.parametrize("mode", ["random", "ties", "constant", "small"])
def test_rpn_topk_decode_matches_stable_sort(mode):
    """chip-wide radix select vs torch's stable descending sort + the oracle's anchor/delta decode; covers
    heavy ties (order = lower flat index first), the exact fallback (constant map) and k > H*W*A."""
    from glass_amd.ops import native as K
    from oracle import d2ops
    dev = _dev()
    N, A, topk = 2, 12, 1000
    shapes = [(40, 52), (20, 26), (10, 13)] if mode != "small" else [(8, 9), (4, 5)]
    g = torch.Generator().manual_seed(3)
    heads, levels, refs = [], [], []
    off = 0
    cells = [d2ops.rotated_cell_anchors(16 << i, (0.2, 0.5, 1.0), (-90, -45, 0, 45)) for i in range(len(shapes))]
    weights = (1.0, 1.0, 1.0, 1.0, 2.0)
    for i, (H, W) in enumerate(shapes):
        head = torch.randn((N, H, W, 6 * A), generator=g)
        if mode == "ties":
            head[..., :A] = torch.round(head[..., :A] * 2) / 2
        elif mode == "constant":
            head[..., :A] = 0.25
        heads.append(head.to(dev).contiguous())
        k = min(topk, H * W * A)
        lg = head[..., :A].reshape(N, -1)
        dl = head[..., A:].reshape(N, -1, 5)
        anchors = d2ops.rotated_grid_anchors(H, W, 4 << i, cells[i])
        srt, idx = lg.sort(dim=1, descending=True, stable=True)
        boxes = torch.stack([d2ops.apply_deltas_rotated(dl[n][idx[n, :k]], anchors[idx[n, :k]], weights) for n in range(N)])
        refs.append((srt[:, :k], boxes, off, k))
        levels.append({"logits": heads[-1], "deltas": heads[-1].view(-1)[A:], "ldl": 6 * A, "ldd": 6 * A, "H": H, "W": W,
                       "stride": 4 << i, "cell_anchors": cells[i].to(dev), "topk": topk, "slot_off": off})
        off += k
    ob = torch.zeros((N, off, 5), device=dev)
    os_ = torch.zeros((N, off), device=dev)
    ol = torch.full((N, off), -1, dtype=torch.int32, device=dev)
    K.rpn_topk_decode(levels, N, A, 0.0, weights, ob, os_, ol)
    torch.cuda.synchronize()
    for i, (srt, boxes, o, k) in enumerate(refs):
        assert torch.equal(os_[:, o:o + k].cpu(), srt), f"level {i}: selected logits / order differ"
        np.testing.assert_allclose(ob[:, o:o + k].cpu().numpy(), boxes.numpy(), rtol=1e-5, atol=1e-4)
        assert (ol[:, o:o + k] == i).all()


def test_winograd_propagates_non_finite_values_like_the_direct_kernel():
    """no ReLU: a NaN / inf in the input must come out as NaN / inf (the reference filters non-finite predictions
    downstream, rotated_fast_rcnn.py:102-107; it never hides them in a conv) - both Winograd kernels and the direct one"""
    from glass_amd.ops import native as K
    dev = _dev()
    for cin, cout in ((32, 128), (16, 64)):
        x = _rand((1, 8, 8, cin), 5).to(dev)
        x[0, 3, 3, 0] = float("nan")
        w = _rand((cout, 3, 3, cin), 6, 0.1).to(dev)
        for relu in (0, 1):
            yw = K.conv2d_nhwc(x, w, None, padding=1, relu=relu, winograd=True)
            yd = K.conv2d_nhwc(x, w, None, padding=1, relu=relu, winograd=False)
            torch.cuda.synchronize()
            nan_w, nan_d = torch.isnan(yw), torch.isnan(yd)
            if relu == 0:
                assert nan_d[0, 2:5, 2:5, :].all(), "direct kernel lost the NaN"
                assert nan_w[0, 2:5, 2:5, :].all(), "Winograd kernel lost the NaN"
            assert not nan_w[0, 6:, 6:, :].any() and not nan_d[0, 6:, 6:, :].any()


def test_winograd_block_channels_rule():
    """which Winograd kernel a layer gets (and therefore which packed-weight layout) is a pure function of (Cout, Cin)"""
    from glass_amd._lib import lib
    L = lib()
    assert L.glass_winograd_block_channels(256, 256) == 128 and L.glass_winograd_block_channels(128, 32) == 128
    assert L.glass_winograd_block_channels(64, 64) == 64 and L.glass_winograd_block_channels(192, 64) == 64
    assert L.glass_winograd_block_channels(256, 48) == 64


def test_winograd_random_shape_sweep():
    """24 seeded random shapes (tiny maps, H or W = 1, odd sizes, ragged tile counts, several channel blocks) :
    Winograd vs the direct kernel, both through the C ABI."""
    from glass_amd.ops import native as K
    dev = _dev()
    rng = np.random.RandomState(123)
    for it in range(24):
        N = int(rng.randint(1, 5))
        H = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 31]))
        W = int(rng.choice([1, 2, 4, 7, 9, 16, 33, 40]))
        Cin = int(rng.choice([16, 32, 48, 64, 80]))
        Cout = int(rng.choice([64, 128, 192]))
        relu = int(rng.randint(0, 3))
        use_res = bool(rng.randint(0, 2))
        g = torch.Generator().manual_seed(1000 + it)
        x = torch.randn((N, H, W, Cin), generator=g).to(dev)
        w = (torch.randn((Cout, 3, 3, Cin), generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(dev)
        b = (torch.randn((Cout,), generator=g) * 0.1).to(dev)
        r = torch.randn((N, H, W, Cout), generator=g).to(dev) if use_res else None
        kw = dict(padding=1, relu=relu, residual=r, res_mode=1 if use_res else 0)
        yw = K.conv2d_nhwc(x, w, b, winograd=True, **kw)
        yd = K.conv2d_nhwc(x, w, b, winograd=False, **kw)
        torch.cuda.synchronize()
        scale = float(yd.abs().max()) + 1e-6
        err = float((yw - yd).abs().max())
        assert err <= 2e-5 * scale, f"case {it}: N={N} H={H} W={W} Cin={Cin} Cout={Cout} relu={relu} res={use_res}: {err / scale:.2e}"


F16_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, relu, res_mode
    (2, 20, 24, 64, 256, (3, 3), (1, 1), (1, 1), 1, 1),
    (1, 40, 36, 4, 64, (7, 7), (2, 2), (3, 3), 1, 0),
    (2, 32, 32, 16, 32, (3, 3), (1, 1), (1, 1), 1, 0),
    (1, 24, 24, 256, 72, (1, 1), (1, 1), (0, 0), 0, 0),
    (1, 16, 24, 512, 256, (1, 1), (1, 1), (0, 0), 0, 2),
    (3, 16, 33, 256, 256, (2, 2), (2, 1), (0, 0), 1, 0),
    (5, 1, 1, 12544, 2048, (1, 1), (1, 1), (0, 0), 1, 0),
]


@pytest.mark.parametrize("case", F16_CASES)
def test_conv_fp16_mode_equals_conv_of_fp16_rounded_operands(case):
    """glass_conv2d_nhwc_f16 (opt-in precision mode): exactly conv(fp16(x), fp16(w)) accumulated in fp32 - compared
    with torch CPU fp64 on the rounded operands, so only the fp32 accumulation order differs (1e-5 of the scale)."""
    from glass_amd.ops import native as K
    N, H, W, Cin, Cout, k, s, p, relu, res_mode = case
    dev = _dev()
    x = _rand((N, Cin, H, W), 21)
    w = _rand((Cout, Cin, k[0], k[1]), 22, (2.0 / (Cin * k[0] * k[1])) ** 0.5)
    b = _rand((Cout,), 23, 0.1)
    if Cin == 4:
        x[:, 3] = 0
    ref = F.conv2d(x.half().double(), w.half().double(), b.double(), stride=s, padding=p)
    res = None
    if res_mode == 1:
        res = _rand(tuple(ref.shape), 24)
        ref = ref + res.double()
    elif res_mode == 2:
        res = _rand((N, Cout, ref.shape[2] // 2, ref.shape[3] // 2), 24)
        ref = ref + F.interpolate(res.double(), scale_factor=2.0, mode="nearest")
    if relu == 1:
        ref = F.relu(ref)
    prev = K.set_conv_precision("fp16")
    try:
        y = K.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), w.permute(0, 2, 3, 1).contiguous().to(dev), b.to(dev),
                          stride=s, padding=p, relu=relu,
                          residual=None if res is None else res.permute(0, 2, 3, 1).contiguous().to(dev), res_mode=res_mode)
        # layers with Cin, Cout multiples of 64 and enough work per input element: input rounded once (glass_cast_f32_to_f16)
        # + the fp16-MFMA kernel; the others: the fp32 template rounding its operands as it stages them.  Same arithmetic.
        packed = Cin % 64 == 0 and Cout % 64 == 0 and k[0] * k[1] * Cout >= 512
        assert K.last_conv_path() == ("packed_fp16" if packed else "direct_fp16")
    finally:
        K.set_conv_precision(prev)
    torch.cuda.synchronize()
    got = y.cpu().permute(0, 3, 1, 2).double()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 1e-5 * scale


def test_cast_f32_to_f16_is_round_to_nearest_even():
    """glass_cast_f32_to_f16 (the operand rounding of the fp16 conv modes applied once): bit-identical to torch's .half() -
    ties to even, subnormals, overflow to inf, signed zeros, NaN stays NaN"""
    import ctypes
    from glass_amd.ops import native as K
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x = torch.cat([torch.randn(4096, generator=g) * 3, torch.randn(2048, generator=g) * 1e-5, torch.randn(1024, generator=g) * 7e4,
                   torch.tensor([0.0, -0.0, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 65504.0, 65520.0, -65520.0, 6e-8, 2.98e-8, 2.99e-8,
                                 float("inf"), -float("inf")])])
    x = torch.cat([x, torch.zeros((-x.numel()) % 4)])
    xd = x.to(dev)
    y = torch.empty(x.shape, dtype=torch.float16, device=dev)
    K.check(K.lib().glass_cast_f32_to_f16(ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(y.data_ptr()), x.numel(),
                                          ctypes.c_void_p(K.stream_handle())), "glass_cast_f32_to_f16")
    torch.cuda.synchronize()
    assert torch.equal(y.cpu().view(torch.int16), x.half().view(torch.int16))
    nan = torch.full((4,), float("nan"), device=dev)
    yn = torch.empty((4,), dtype=torch.float16, device=dev)
    K.check(K.lib().glass_cast_f32_to_f16(ctypes.c_void_p(nan.data_ptr()), ctypes.c_void_p(yn.data_ptr()), 4,
                                          ctypes.c_void_p(K.stream_handle())), "glass_cast_f32_to_f16")
    assert bool(torch.isnan(yn.float()).all())


def test_upload_is_stream_ordered_and_exact():
    """ops.native.upload (pinned staging, non-blocking copy): the values arrive intact even when the staging tensor is
    dropped immediately and many uploads are in flight behind a long-running kernel"""
    from glass_amd.ops import native as K
    dev = _dev()
    big = torch.randn((4096, 4096), device=dev)
    for _ in range(3):
        big = big @ big * 1e-3                      # keep the stream busy while the uploads are enqueued
    ups = [K.upload(list(range(i, i + 37)), torch.int32, dev) for i in range(200)]
    torch.cuda.synchronize()
    for i, u in enumerate(ups):
        assert u.tolist() == list(range(i, i + 37))


def test_conv_output_larger_than_2gib_takes_the_64bit_epilogue():
    """the vector epilogue addresses y with 32-bit buffer offsets; an output span >= 2 GiB must fall back to the
    scalar epilogue with 64-bit addresses (and Winograd must report 'unsupported') - spot-checked against torch CPU"""
    import ctypes
    from glass_amd.ops import native as K
    dev = _dev()
    N, H, W, Cin, Cout = 1, 1024, 1024, 16, 520           # 1024*1024*520*4 B = 2.03 GiB
    g = torch.Generator().manual_seed(7)
    x = torch.randn((N, H, W, Cin), generator=g)
    w = torch.randn((Cout, 1, 1, Cin), generator=g) * 0.2
    b = torch.randn((Cout,), generator=g)
    y = K.conv2d_nhwc(x.to(dev), w.to(dev), b.to(dev), relu=1)
    torch.cuda.synchronize()
    assert y.numel() * 4 >= 0x7fffff00
    idx = torch.randint(0, H * W, (512,), generator=g)
    idx[:4] = torch.tensor([0, 1, H * W - 2, H * W - 1])
    ref = torch.relu(x.view(-1, Cin)[idx] @ w.view(Cout, Cin).t() + b)
    got = y.view(-1, Cout)[idx.to(dev)].cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=2e-4)
    d = K.ConvDesc(1, 1024, 1024, 64, 576, 3, 3, 1, 1, 1, 1, 1024, 1024, 64, 576, 0, 1, 0, 0, 0)   # y span 2.25 GiB
    assert K.lib().glass_winograd_supported(ctypes.byref(d)) == 0


def test_conv_input_larger_than_2gib_takes_the_64bit_loads():
    """the fast load paths use 32-bit buffer offsets; an input span >= 2 GiB must take the generic 64-bit path"""
    from glass_amd.ops import native as K
    dev = _dev()
    H, W, Cin, Cout = 1024, 1024, 520, 16                 # 1024*1024*520*4 B = 2.03 GiB
    g = torch.Generator().manual_seed(8)
    x = torch.randn((1, H, W, Cin), generator=g)
    assert x.numel() * 4 >= 0x7fffff00
    w = torch.randn((Cout, 1, 1, Cin), generator=g) * 0.05
    b = torch.randn((Cout,), generator=g)
    y = K.conv2d_nhwc(x.to(dev), w.to(dev), b.to(dev))
    torch.cuda.synchronize()
    idx = torch.randint(0, H * W, (512,), generator=g)
    idx[:4] = torch.tensor([0, 1, H * W - 2, H * W - 1])
    ref = x.view(-1, Cin)[idx] @ w.view(Cout, Cin).t() + b
    np.testing.assert_allclose(y.view(-1, Cout)[idx.to(dev)].cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-4)


# ------------------------------------------------------------------ fp16 STORAGE (BASELINE configs[4]; MODEL.CONV_PRECISION fp16s)
H16_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, relu, res_mode, x half, y half, residual half
    (2, 20, 24, 64, 256, (3, 3), (1, 1), (1, 1), 1, 1, True, True, True),
    (1, 33, 17, 128, 128, (3, 3), (1, 1), (1, 1), 0, 0, True, True, False),
    (2, 16, 16, 256, 512, (1, 1), (2, 2), (0, 0), 0, 0, True, False, False),      # fp16 in, fp32 out (an exit of the fp16 chain)
    (1, 40, 36, 4, 64, (7, 7), (2, 2), (3, 3), 1, 0, False, True, False),         # stem: fp32 image in, fp16 out (an entry)
    (2, 32, 32, 16, 32, (3, 3), (1, 1), (1, 1), 1, 0, True, True, False),
    (1, 16, 24, 512, 256, (1, 1), (1, 1), (0, 0), 0, 2, True, True, True),        # FPN lateral: x2-upsampled fp16 residual
    (3, 7, 9, 64, 40, (1, 1), (1, 1), (0, 0), 2, 0, True, True, False),           # ragged rows and channel block (vector epilogue off: Cout % 4 == 0 but 40)
    (2, 8, 32, 256, 256, (2, 1), (2, 1), (0, 0), 1, 0, True, True, False),
    (1, 12, 20, 96, 136, (1, 1), (1, 1), (0, 0), 1, 1, True, False, True),        # fp16 residual into an fp32 output
    # the six shapes of conv_h16_kernel: 256 / 128 / 64 pixels x 128 / 64 channels per block (the small cases above and
    # below take the 64-pixel ones)
    (8, 64, 64, 64, 384, (3, 3), (1, 1), (1, 1), 1, 1, True, True, True),         # 256 x 128
    (2, 64, 64, 64, 384, (3, 3), (1, 1), (1, 1), 1, 0, True, True, False),        # 128 x 128
    (8, 64, 64, 128, 192, (1, 1), (1, 1), (0, 0), 0, 1, True, False, True),       # 256 x 64
    (8, 64, 60, 128, 192, (3, 3), (1, 1), (1, 1), 2, 1, True, True, False),       # 128 x 64
    (2, 13, 11, 192, 64, (3, 3), (2, 2), (1, 1), 1, 0, True, False, False),       # ragged last block, stride 2 with padding
    (1, 9, 9, 64, 128, (5, 5), (1, 1), (2, 2), 0, 0, True, True, False),          # 25 taps, most of them padding at the rim
    (4, 1, 1, 1024, 256, (1, 1), (1, 1), (0, 0), 1, 0, True, False, False),       # a linear layer: 4 pixels in a 128-pixel block
]


@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("case", H16_CASES)
def test_conv_fp16_storage_matches_emulation(case, packed):
    """glass_conv2d_nhwc_h16 / glass_conv2d_nhwc_h16_packed (the fp16-MFMA kernel of csrc/conv_h16.hip, taken when the input
    is fp16 and Cin, Cout are multiples of 64): y = fp16?(act(conv(fp16(x), fp16(w)) + bias [+ residual])) with fp32
    accumulation.  fp32 outputs match the emulation to fp32 summation order; fp16 outputs equal the emulation's rounding
    except where the fp32 value sits on a rounding boundary (<= 1 fp16 ulp, rare)."""
    from glass_amd.ops import native as K
    N, H, W, Cin, Cout, k, s, p, relu, res_mode, xh, yh, rh = case
    takes_packed = packed and xh and Cin % 64 == 0 and Cout % 64 == 0
    if packed and not takes_packed:
        pytest.skip("layer is outside the packed kernel's envelope (covered by packed=False)")
    dev = _dev()
    x = _rand((N, Cin, H, W), 31)
    w = _rand((Cout, Cin, k[0], k[1]), 32, (2.0 / (Cin * k[0] * k[1])) ** 0.5)
    b = _rand((Cout,), 33, 0.1)
    if Cin == 4:
        x[:, 3] = 0
    xq = x.half().float() if True else x                      # the kernel rounds fp32 inputs to fp16 as well (operand rounding)
    ref = F.conv2d(xq.double(), w.half().double(), b.double(), stride=s, padding=p)
    res = None
    if res_mode == 1:
        res = _rand(tuple(ref.shape), 34)
    elif res_mode == 2:
        res = _rand((N, Cout, ref.shape[2] // 2, ref.shape[3] // 2), 34)
    if res is not None and rh:
        res = res.half().float()
    if relu == 2:
        ref = F.relu(ref)
    if res_mode == 1:
        ref = ref + res.double()
    elif res_mode == 2:
        ref = ref + F.interpolate(res.double(), scale_factor=2.0, mode="nearest")
    if relu == 1:
        ref = F.relu(ref)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    xd = nhwc(x).to(dev)
    xd = xd.half() if xh else xd
    rd = None
    if res is not None:
        rd = nhwc(res).to(dev)
        rd = rd.half() if rh else rd
    prev, prev_h = K.set_conv_precision("fp16s"), K.set_conv_h16(packed)
    try:
        y = K.conv2d_nhwc(xd, nhwc(w).to(dev), b.to(dev), stride=s, padding=p, relu=relu, residual=rd, res_mode=res_mode,
                          out_dtype=torch.float16 if yh else torch.float32)
        assert K.last_conv_path() == ("packed_fp16" if takes_packed else "direct_fp16")
    finally:
        K.set_conv_precision(prev)
        K.set_conv_h16(prev_h)
    torch.cuda.synchronize()
    assert y.dtype == (torch.float16 if yh else torch.float32)
    got = y.float().cpu().permute(0, 3, 1, 2).double()
    scale = float(ref.abs().max())
    if not yh:
        err = float((got - ref).abs().max()) / scale
        print(f"fp16-storage conv {case[:5]} fp32 out: max err / range {err:.2e}")
        assert err <= 2e-6
    else:
        want = ref.float().half().double()                    # the emulation's own rounding of the exact result
        big = torch.maximum(torch.maximum(got.abs(), want.abs()), torch.tensor(6.2e-5, dtype=torch.float64))
        ulp = torch.ldexp(torch.ones_like(big), torch.frexp(big)[1] - 11)      # fp16 spacing at the larger of the two values
        d = (got - want).abs()
        frac_off = float((d > 0).double().mean())
        print(f"fp16-storage conv {case[:5]} fp16 out: {frac_off:.2e} of the outputs differ from the emulation's rounding "
              f"(max {float((d / ulp).max()):.2f} ulp)")
        # (near zero the fp32 summation-order noise, ~1e-7 of the range, spans several fp16 subnormal steps)
        assert bool((d <= 1.001 * ulp + 2e-6 * scale).all()) and frac_off < 5e-3


def test_maxpool_and_roi_align_on_fp16_tensors():
    from glass_amd.ops import native as K
    from glass_amd.utils.synth import make_boxes
    from oracle import d2ops
    dev = _dev()
    x = _rand((2, 64, 37, 41), 41).half()
    for kk, ss, pp in ((3, 2, 1), (2, 2, 0), ((2, 1), (2, 1), 0)):
        y = K.maxpool2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), kk, ss, pp)
        ref = F.max_pool2d(x.float(), kk, ss, pp)
        assert y.dtype == torch.float16 and torch.equal(y.float().cpu().permute(0, 3, 1, 2), ref)
    feats = [_rand((2, 256, 64 >> i, 80 >> i), 50 + i).half() for i in range(5)]
    scales = [1.0 / (4 << i) for i in range(5)]
    boxes = [make_boxes(i, 9, 256, 320) for i in range(2)]
    ref = d2ops.roi_pooler([f.float() for f in feats], scales, boxes, (7, 7), 2)
    bcat = torch.cat(boxes).contiguous()
    bidx = torch.cat([torch.full((len(b),), i, dtype=torch.int32) for i, b in enumerate(boxes)])
    y = K.roi_align_rotated([f.permute(0, 2, 3, 1).contiguous().to(dev) for f in feats], scales, bcat.to(dev), bidx.to(dev), (7, 7), 2)
    assert y.dtype == torch.float32
    np.testing.assert_allclose(y.cpu().permute(0, 3, 1, 2).numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("R,H,W", [(3, 128, 128), (2, 32, 64), (1, 96, 32)])
def test_fused_local_stem_equals_the_three_separate_kernels(R, H, W):
    """glass_local_stem_fused (conv0_1 + ReLU + conv0_2 + ReLU + maxpool 2x2 in one kernel) vs the separate entries and vs
    torch fp64; borders of the image and of the 32 x 32 tiles included (conv0_2 pads its INPUT with zeros)."""
    from glass_amd.ops import native as K
    dev = _dev()
    x = _rand((R, 3, H, W), 61) * 50.0
    w1 = _rand((16, 3, 3, 3), 62, (2.0 / 27) ** 0.5)
    b1 = _rand((16,), 63, 0.5)
    w2 = _rand((32, 16, 3, 3), 64, (2.0 / 144) ** 0.5)
    b2 = _rand((32,), 65, 0.5)
    ref = F.max_pool2d(F.relu(F.conv2d(F.relu(F.conv2d(x.double(), w1.double(), b1.double(), padding=1)), w2.double(), b2.double(),
                                       padding=1)), 2)
    xd = F.pad(x.permute(0, 2, 3, 1), (0, 1)).contiguous().to(dev)
    w1d = F.pad(w1.permute(0, 2, 3, 1), (0, 1)).contiguous().to(dev)
    w2d = w2.permute(0, 2, 3, 1).contiguous().to(dev)
    assert K.local_stem_supported(xd, w1d, w2d)
    y = K.local_stem_fused(xd, w1d, b1.to(dev), w2d, b2.to(dev))
    t = K.conv2d_nhwc(xd, w1d, b1.to(dev), padding=1, relu=1)
    t = K.conv2d_nhwc(t, w2d, b2.to(dev), padding=1, relu=1)
    y3 = K.maxpool2d_nhwc(t, 2, 2, 0)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    e_ref = float((y.cpu().permute(0, 3, 1, 2).double() - ref).abs().max()) / scale
    e_sep = float((y - y3).abs().max()) / scale
    print(f"fused local stem {R}x{H}x{W}: vs fp64 {e_ref:.2e}, vs separate kernels {e_sep:.2e} (of range)")
    assert tuple(y.shape) == (R, H // 2, W // 2, 32) and e_ref < 2e-6 and e_sep < 2e-6


@pytest.mark.parametrize("R,H,W", [(3, 128, 128), (2, 32, 64)])
def test_fused_local_stem_fp16_storage_equals_the_separate_fp16_kernels(R, H, W):
    """glass_local_stem_fused_h16 ('fp16s' mode) vs glass_conv2d_nhwc_h16 x 2 + glass_maxpool2d_nhwc_h16: the same fp16
    roundings (operands, the stored conv0_1 map, the output), so fp16 outputs agree except where an fp32 sum sits on a rounding
    boundary (<= 1 fp16 ulp, rare) - and vs the fp64 emulation of that arithmetic."""
    from glass_amd.ops import native as K
    dev = _dev()
    x = _rand((R, 3, H, W), 71) * 2.0
    w1 = _rand((16, 3, 3, 3), 72, (2.0 / 27) ** 0.5)
    b1 = _rand((16,), 73, 0.5)
    w2 = _rand((32, 16, 3, 3), 74, (2.0 / 144) ** 0.5)
    b2 = _rand((32,), 75, 0.5)
    q = lambda t: t.half().double()
    t1 = F.relu(F.conv2d(q(x), q(w1), b1.double(), padding=1)).float().half().double()
    ref = F.max_pool2d(F.relu(F.conv2d(t1, q(w2), b2.double(), padding=1)), 2).float().half().double()
    xd = F.pad(x.permute(0, 2, 3, 1), (0, 1)).contiguous().to(dev)
    w1d = F.pad(w1.permute(0, 2, 3, 1), (0, 1)).contiguous().to(dev)
    w2d = w2.permute(0, 2, 3, 1).contiguous().to(dev)
    prev = K.set_conv_precision("fp16s")
    try:
        assert K.local_stem_supported(xd, w1d, w2d)
        y = K.local_stem_fused(xd, w1d, b1.to(dev), w2d, b2.to(dev))
        t = K.conv2d_nhwc(xd, w1d, b1.to(dev), padding=1, relu=1, out_dtype=torch.float16)
        t = K.conv2d_nhwc(t, w2d, b2.to(dev), padding=1, relu=1)
        y3 = K.maxpool2d_nhwc(t, 2, 2, 0)
    finally:
        K.set_conv_precision(prev)
    torch.cuda.synchronize()
    assert y.dtype == torch.float16 and y3.dtype == torch.float16 and tuple(y.shape) == (R, H // 2, W // 2, 32)
    got, sep = y.float().cpu().permute(0, 3, 1, 2).double(), y3.float().cpu().permute(0, 3, 1, 2).double()
    scale = float(ref.abs().max())
    for name, other in (("separate fp16 kernels", sep), ("fp64 emulation", ref)):
        big = torch.maximum(torch.maximum(got.abs(), other.abs()), torch.tensor(6.2e-5, dtype=torch.float64))
        ulp = torch.ldexp(torch.ones_like(big), torch.frexp(big)[1] - 11)
        d = (got - other).abs()
        frac = float((d > 0).double().mean())
        print(f"fused fp16 local stem {R}x{H}x{W} vs {name}: {frac:.2e} of the outputs differ, max |d| {float(d.max()) / scale:.1e} of the range")
        # a conv0_1 value on a rounding boundary moves by one fp16 ulp and shifts the conv0_2 sums that read it by ~1e-4 of
        # the range: still at most one ulp of the (larger) output, on a few outputs
        assert bool((d <= 1.001 * ulp + 1e-3 * scale).all()) and frac < 2e-2


PW_CASES = [
    # N, H, W, Cin, Cout, stride, relu, res_mode, (ld_out, coff)
    (2, 20, 24, 64, 256, 1, 1, 1, None),          # res2 conv3 + residual
    (1, 33, 17, 128, 128, 1, 0, 0, None),         # ragged pixel count
    (2, 16, 16, 256, 512, 2, 0, 0, None),         # stride-2 shortcut
    (1, 16, 24, 512, 256, 1, 0, 2, None),         # FPN lateral with the x2-upsampled residual
    (3, 7, 9, 96, 128, 1, 2, 1, None),            # 3 k-tiles (odd), ReLU before the residual add
    (1, 12, 20, 32, 384, 1, 1, 0, (512, 128)),    # a single k-tile, three channel blocks into a wider buffer at an offset
    (8, 64, 64, 256, 256, 1, 1, 0, None),         # enough pixels for the 256-pixel blocks
    (2, 15, 15, 2048, 512, 2, 1, 0, None),        # 64 k-tiles, odd map with stride 2
]


@pytest.mark.parametrize("case", PW_CASES)
def test_pointwise_matches_direct_and_torch(case):
    """glass_conv1x1_pointwise_nhwc (weight-streaming 1x1 GEMM) vs torch fp64 and vs glass_conv2d_nhwc: same exact-fp32
    arithmetic, different summation order -> 5e-6 of the output range (K up to 2048 terms)."""
    from glass_amd.ops import native as K
    N, H, W, Cin, Cout, stride, relu, res_mode, strided = case
    dev = _dev()
    K.set_pointwise("all")
    prev_split = K.set_pw_split(0)                 # this test is about the fp32-MFMA 1x1 kernel
    x = _rand((N, Cin, H, W), 71)
    w = _rand((Cout, Cin, 1, 1), 72, (2.0 / Cin) ** 0.5)
    b = _rand((Cout,), 73, 0.1)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride)
    res = None
    if res_mode == 1:
        res = _rand(tuple(ref.shape), 74)
    elif res_mode == 2:
        res = _rand((N, Cout, ref.shape[2] // 2, ref.shape[3] // 2), 74)
    if relu == 2:
        ref = F.relu(ref)
    if res_mode == 1:
        ref = ref + res.double()
    elif res_mode == 2:
        ref = ref + F.interpolate(res.double(), scale_factor=2.0, mode="nearest")
    if relu == 1:
        ref = F.relu(ref)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    xd, wd = nhwc(x).to(dev), nhwc(w).to(dev)
    rd = None if res is None else nhwc(res).to(dev)
    kw = dict(stride=stride, relu=relu, residual=rd, res_mode=res_mode)
    Ho, Wo = ref.shape[2], ref.shape[3]
    if strided is None:
        y = K.conv2d_nhwc(xd, wd, b.to(dev), **kw)
        assert K.last_conv_path() == "pointwise"
        K.set_pointwise(False)
        yd = K.conv2d_nhwc(xd, wd, b.to(dev), **kw)
        K.set_pointwise(True)
        assert K.last_conv_path() == "direct"
    else:
        ld, coff = strided
        buf = torch.full((N, Ho, Wo, ld), 7.0, device=dev)
        bufd = torch.full((N, Ho, Wo, ld), 7.0, device=dev)
        K.conv2d_nhwc(xd, wd, b.to(dev), out=buf, out_coff=coff, **kw)
        assert K.last_conv_path() == "pointwise"
        K.set_pointwise(False)
        K.conv2d_nhwc(xd, wd, b.to(dev), out=bufd, out_coff=coff, **kw)
        K.set_pointwise(True)
        torch.cuda.synchronize()
        assert float((buf[..., :coff] - 7.0).abs().max()) == 0.0
        assert coff + Cout == ld or float((buf[..., coff + Cout:] - 7.0).abs().max()) == 0.0
        y, yd = buf[..., coff:coff + Cout], bufd[..., coff:coff + Cout]
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    e = float((y.cpu().permute(0, 3, 1, 2).double() - ref).abs().max()) / scale
    ed = float((y - yd).abs().max()) / scale
    print(f"pointwise {case[:6]}: vs fp64 {e:.2e}, vs implicit-GEMM kernel {ed:.2e} (of range)")
    K.set_pointwise(True)
    K.set_pw_split(prev_split)
    assert e <= 5e-6 and ed <= 5e-6


@pytest.mark.parametrize("products", [9, 6])
@pytest.mark.parametrize("case", PW_CASES + [(8, 32, 32, 512, 2048, 1, 1, 1, None), (1, 64, 64, 1024, 128, 1, 0, 0, (256, 64))])
def test_pointwise_split_matches_direct_and_torch(case, products):
    """glass_conv1x1_pointwise_split_nhwc (every fp32 product as bf16 piece products on the bf16 matrix cores, fp32
    accumulate) vs torch fp64 and vs glass_conv2d_nhwc.  Nine products ARE the exact product, so the kernel is held to the
    fp32-MFMA kernels' own bound (5e-6 of the output range, K up to 2048 terms) and must not be further from fp64 than
    the fp32-MFMA kernels are (the weight-streaming 1x1 kernel, which like this one runs ONE accumulator down the whole K, or
    the implicit-GEMM kernel) by more than rounding noise; the six-product form (opt-in, measurement only) drops < 2^-23
    per product and is held to the same bound."""
    from glass_amd.ops import native as K
    N, H, W, Cin, Cout, stride, relu, res_mode, strided = case
    dev = _dev()
    x = _rand((N, Cin, H, W), 71)
    w = _rand((Cout, Cin, 1, 1), 72, (2.0 / Cin) ** 0.5)
    b = _rand((Cout,), 73, 0.1)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride)
    res = None
    if res_mode == 1:
        res = _rand(tuple(ref.shape), 74)
    elif res_mode == 2:
        res = _rand((N, Cout, ref.shape[2] // 2, ref.shape[3] // 2), 74)
    if relu == 2:
        ref = F.relu(ref)
    if res_mode == 1:
        ref = ref + res.double()
    elif res_mode == 2:
        ref = ref + F.interpolate(res.double(), scale_factor=2.0, mode="nearest")
    if relu == 1:
        ref = F.relu(ref)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    xd, wd = nhwc(x).to(dev), nhwc(w).to(dev)
    rd = None if res is None else nhwc(res).to(dev)
    kw = dict(stride=stride, relu=relu, residual=rd, res_mode=res_mode)
    Ho, Wo = ref.shape[2], ref.shape[3]
    force = f"pws{products}"
    if strided is None:
        y = torch.full((N, Ho, Wo, Cout), float("nan"), device=dev)     # every output must be written
        K.conv2d_nhwc(xd, wd, b.to(dev), out=y, winograd=force, **kw)
        assert K.last_conv_path() == "pointwise_split"
        yd = K.conv2d_nhwc(xd, wd, b.to(dev), winograd=False, **kw)
        assert K.last_conv_path() == "direct"
        yp = K.conv2d_nhwc(xd, wd, b.to(dev), routing=K.default_routing().replace(split=0, pw="all"), **kw)
        assert K.last_conv_path() == "pointwise"
    else:
        ld, coff = strided
        yp = None
        buf = torch.full((N, Ho, Wo, ld), 7.0, device=dev)
        bufd = torch.full((N, Ho, Wo, ld), 7.0, device=dev)
        buf[..., coff:coff + Cout] = float("nan")
        K.conv2d_nhwc(xd, wd, b.to(dev), out=buf, out_coff=coff, winograd=force, **kw)
        assert K.last_conv_path() == "pointwise_split"
        K.conv2d_nhwc(xd, wd, b.to(dev), out=bufd, out_coff=coff, winograd=False, **kw)
        torch.cuda.synchronize()
        assert float((buf[..., :coff] - 7.0).abs().max()) == 0.0
        assert coff + Cout == ld or float((buf[..., coff + Cout:] - 7.0).abs().max()) == 0.0
        y, yd = buf[..., coff:coff + Cout], bufd[..., coff:coff + Cout]
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y).all()), "unwritten or non-finite outputs"
    scale = float(ref.abs().max())
    e = float((y.cpu().permute(0, 3, 1, 2).double() - ref).abs().max()) / scale
    e_direct = float((yd.cpu().permute(0, 3, 1, 2).double() - ref).abs().max()) / scale
    e_pw = e_direct if yp is None else float((yp.cpu().permute(0, 3, 1, 2).double() - ref).abs().max()) / scale
    ed = float((y - yd).abs().max()) / scale
    print(f"pointwise split x{products} {case[:6]}: vs fp64 {e:.2e} (fp32-MFMA kernels: implicit GEMM {e_direct:.2e}, weight-streaming {e_pw:.2e}), "
          f"vs implicit-GEMM kernel {ed:.2e} (of range)")
    assert e <= 5e-6 and ed <= 5e-6 and e <= 1.5 * max(e_direct, e_pw) + 2e-7


DUAL_CASES = [
    # N, H, W, Cin1, Cin2, Cout, stride   (x1 [N,H,W,Cin1] strided, x2 on the output grid)
    (2, 64, 64, 64, 64, 256, 1),          # res2.0: shortcut 64 -> 256 + conv3 64 -> 256
    (2, 64, 64, 256, 128, 512, 2),        # res3.0: shortcut 256 -> 512 stride 2 + conv3 128 -> 512
    (1, 32, 32, 512, 256, 1024, 2),       # res4.0 (64-pixel blocks)
    (3, 16, 16, 1024, 512, 2048, 2),      # res5.0
    (1, 19, 37, 96, 32, 128, 1),          # ragged pixel count (703 rows: a partial last block), one k-tile of the second source
    (1, 21, 35, 32, 160, 128, 2),         # odd map under stride 2 (Ho = 11, Wo = 18), one k-tile of the first source
]


@pytest.mark.parametrize("case", DUAL_CASES)
def test_pointwise_split_dual_source_matches_the_two_launches_and_torch(case):
    """glass_conv1x1_pointwise_split_dual_nhwc: relu([x1 strided | x2] [W1 | W2]^T + b1 + b2) - a bottleneck block's shortcut folded into
    its conv3 - vs torch fp64 of the two convolutions and vs the two single-source launches it replaces (shortcut, then conv3 + residual
    + ReLU).  Same exact products; only the order of the fp32 additions differs (one accumulator down both k-ranges)."""
    from glass_amd.ops import native as K
    N, H, W, C1, C2, Cout, stride = case
    dev = _dev()
    x1 = _rand((N, C1, H, W), 91)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    x2 = _rand((N, C2, Ho, Wo), 92)
    w1 = _rand((Cout, C1, 1, 1), 93, (1.0 / C1) ** 0.5)
    w2 = _rand((Cout, C2, 1, 1), 94, (1.0 / C2) ** 0.5)
    b1, b2 = _rand((Cout,), 95, 0.1), _rand((Cout,), 96, 0.1)
    ref = F.relu(F.conv2d(x1.double(), w1.double(), b1.double(), stride=stride) + F.conv2d(x2.double(), w2.double(), b2.double()))
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    x1d, x2d, w1d, w2d = nhwc(x1).to(dev), nhwc(x2).to(dev), nhwc(w1).to(dev), nhwc(w2).to(dev)
    wd = K.prepare_dual_weights(w1d, w2d)
    assert wd is not None and tuple(wd.shape) == (Cout, 1, 1, C1 + C2)
    y = K.conv1x1_dual_nhwc(x1d, x2d, wd, (b1 + b2).to(dev), stride=stride, relu=1)
    sc = K.conv2d_nhwc(x1d, w1d, b1.to(dev), stride=stride, winograd="pws9")
    y2 = K.conv2d_nhwc(x2d, w2d, b2.to(dev), relu=1, residual=sc, res_mode=1, winograd="pws9")
    torch.cuda.synchronize()
    assert tuple(y.shape) == (N, Ho, Wo, Cout) and bool(torch.isfinite(y).all())
    refn = ref.permute(0, 2, 3, 1)
    scale = float(refn.abs().max())
    e, e2 = float((y.cpu().double() - refn).abs().max()) / scale, float((y2.cpu().double() - refn).abs().max()) / scale
    d = float((y - y2).abs().max()) / scale
    print(f"dual-source split {case}: vs fp64 {e:.2e} (two launches {e2:.2e}), vs the two launches {d:.2e} (of range)")
    assert e <= 5e-6 and d <= 5e-6 and e <= 1.5 * e2 + 2e-7
    # no ReLU, no bias
    y0 = K.conv1x1_dual_nhwc(x1d, x2d, wd, None, stride=stride)
    ref0 = (F.conv2d(x1.double(), w1.double(), stride=stride) + F.conv2d(x2.double(), w2.double())).permute(0, 2, 3, 1)
    assert float((y0.cpu().double() - ref0).abs().max()) / float(ref0.abs().max()) <= 5e-6


def test_pointwise_split_dual_refuses_what_it_cannot_run():
    from glass_amd.ops import native as K
    from glass_amd._lib import GlassLibraryError
    dev = _dev()
    w1, w2 = _rand((128, 1, 1, 64), 1).to(dev), _rand((128, 1, 1, 32), 2).to(dev)
    wd = K.prepare_dual_weights(w1, w2)
    x1, x2 = _rand((1, 16, 16, 64), 3).to(dev), _rand((1, 16, 16, 32), 4).to(dev)
    with pytest.raises(GlassLibraryError):
        K.conv1x1_dual_nhwc(x1, x2[:, :8].contiguous(), wd)                # grids differ
    assert K.prepare_dual_weights(w1, _rand((256, 1, 1, 32), 5).to(dev)) is None         # different Cout
    assert K.prepare_dual_weights(_rand((128, 1, 1, 48), 6).to(dev), w2) is None         # Cin1 not a multiple of 32
    assert not K.dual_supported(x1, x2, wd, 1)      # 256 pixels x 128 channels: 4 workgroups - the routing keeps the two launches
    assert not K.dual_supported(x1, x2, None, 1)


def test_pointwise_split_reads_channel_slices_and_wide_residuals():
    """the input as the first Cin channels of a wider buffer (ldx > Cin) and the residual as the first Cout channels of a wider
    one (ldr > Cout): pixel strides come from the descriptor, not from the channel counts"""
    from glass_amd.ops import native as K
    dev = _dev()
    N, H, W, Cin, Cout = 2, 24, 40, 96, 256
    xw = _rand((N, H, W, Cin + 32), 81).to(dev)
    w = _rand((Cout, 1, 1, Cin), 82, (2.0 / Cin) ** 0.5).to(dev)
    b = _rand((Cout,), 83, 0.1).to(dev)
    rw = _rand((N, H, W, Cout + 64), 84).to(dev)
    y = torch.full((N, H, W, Cout), float("nan"), device=dev)
    K.conv2d_nhwc(xw, w, b, relu=1, residual=rw, res_mode=1, out=y, cin=Cin, winograd="pws9")
    assert K.last_conv_path() == "pointwise_split"
    ref = torch.relu(xw[..., :Cin].double().reshape(-1, Cin) @ w.double().reshape(Cout, Cin).t() + b.double() +
                     rw[..., :Cout].double().reshape(-1, Cout)).reshape(N, H, W, Cout)
    torch.cuda.synchronize()
    e = float((y.double() - ref).abs().max()) / float(ref.abs().max())
    assert bool(torch.isfinite(y).all()) and e <= 2e-6, e


def test_pointwise_split_is_exact_where_fp32_is():
    """Operands whose products and partial sums are all exactly representable (integers < 2^11 times powers of two, K = 256): an
    fp32 fma chain is exact in ANY order, so the nine-product kernel must return the float64 result bit for bit - the three
    low pieces carry real bits here (operands with 20 significant bits), a dropped or mis-scaled piece product would show."""
    from glass_amd.ops import native as K
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    N, H, W, Cin, Cout = 2, 32, 64, 256, 128
    # x: 20-bit integers scaled by 2^-12 (pieces h, m, l all non-zero), w: small integers -> products < 2^24 each?  keep sums exact:
    # |x| < 2^20, |w| <= 4, K = 256 -> |sum| < 2^20 * 4 * 256 = 2^30 with unit 2^-12 spacing needs 42 bits: NOT exact in fp32.
    # So: x = a * 2^-8 with |a| < 2^12 (12 significant bits, two pieces), w = b * 2^-4 with |b| < 2^4 -> products < 2^16 units,
    # sums < 2^24 units of 2^-12: exact in fp32 in any order.
    a = torch.randint(-(1 << 12) + 1, 1 << 12, (N, H, W, Cin), generator=g).float() * 2.0 ** -8
    bq = torch.randint(-15, 16, (Cout, 1, 1, Cin), generator=g).float() * 2.0 ** -4
    ref = (a.reshape(-1, Cin).double() @ bq.reshape(Cout, Cin).double().t()).float().reshape(N, H, W, Cout)
    y9 = K.conv2d_nhwc(a.to(dev), bq.to(dev), None, winograd="pws9")
    yd = K.conv2d_nhwc(a.to(dev), bq.to(dev), None, winograd=False)
    torch.cuda.synchronize()
    assert torch.equal(yd.cpu(), ref), "the implicit-GEMM kernel itself is not exact on this input: the test's premise is wrong"
    assert torch.equal(y9.cpu(), ref)
    # three-piece operands: x with 24 significant bits, w = +-2^k (one piece): every product is exact, partial sums are not
    # in general, so compare the split kernel with float64 at fp32 rounding (one ulp of the largest partial sum per add)
    x24 = (torch.randint(1 << 23, 1 << 24, (N, H, W, Cin), generator=g).float() * 2.0 ** -24)
    wp2 = (2.0 ** torch.randint(-3, 3, (Cout, 1, 1, Cin), generator=g).float()) * (torch.randint(0, 2, (Cout, 1, 1, Cin), generator=g).float() * 2 - 1)
    r64 = x24.reshape(-1, Cin).double() @ wp2.reshape(Cout, Cin).double().t()
    y = K.conv2d_nhwc(x24.to(dev), wp2.to(dev), None, winograd="pws9").cpu().reshape(-1, Cout).double()
    ydir = K.conv2d_nhwc(x24.to(dev), wp2.to(dev), None, winograd=False).cpu().reshape(-1, Cout).double()
    bound = Cin * 2.0 ** -24 * float(x24.abs().max()) * 4.0 * Cin ** 0.5     # sqrt(K) rounding walk on sums up to K * max|x w|
    e9, edir = float((y - r64).abs().max()), float((ydir - r64).abs().max())
    print(f"24-bit operands: split x9 |err| {e9:.3e}, implicit GEMM {edir:.3e} (bound {bound:.3e})")
    assert e9 <= bound and e9 <= 1.5 * edir
