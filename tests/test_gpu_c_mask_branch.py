"""Rotated mask branch: oracle pinned by the reference's own paste functions (CPU), HIP path vs
oracle / golden (GPU)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "mask_paste.npz")


def _golden():
    g = np.load(GOLD, allow_pickle=False)
    H, W = int(g["H"]), int(g["W"])
    R = g["boxes"].shape[0]
    out_bool = np.unpackbits(g["out_bool"])[: R * H * W].reshape(R, H, W).astype(bool)
    return torch.from_numpy(g["masks"]), torch.from_numpy(g["boxes"]), (H, W), out_bool, g["out_u8"]


# ------------------------------------------------------------------------------------------- CPU
def test_oracle_paste_equals_reference_functions():
    """oracle.paste_rotated_masks vs the outputs of the reference's paste_masks_in_image (make_golden --mask):
    same torch ops in the same order on the same machine type -> exact."""
    from oracle import glass_cpu as O
    masks, boxes, hw, out_bool, out_u8 = _golden()
    got = O.paste_rotated_masks(masks, boxes, hw, 0.5)
    assert got.dtype == torch.bool and tuple(got.shape) == out_bool.shape
    assert int((got.numpy() != out_bool).sum()) == 0
    got8 = O.paste_rotated_masks(masks, boxes, hw, -1)
    assert got8.dtype == torch.uint8
    # (the golden pastes RoI by RoI, this call pastes the batch: same kernels, same values)
    assert int(np.abs(got8.numpy().astype(np.int32) - out_u8.astype(np.int32)).max()) == 0


def test_deconv_is_a_1x1_conv_plus_pixel_shuffle():
    """the identity the HIP mask head relies on (rotated_mask_head.py import_weights), in plain torch fp64."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn((3, 8, 5, 7), generator=g, dtype=torch.float64)
    wd = torch.randn((8, 6, 2, 2), generator=g, dtype=torch.float64)          # [Cin, Cout, 2, 2]
    b = torch.randn((6,), generator=g, dtype=torch.float64)
    ref = F.conv_transpose2d(x, wd, b, stride=2)
    w1 = wd.permute(2, 3, 1, 0).reshape(4 * 6, 8, 1, 1)                         # row = (a*2+b)*Cout + co
    y = F.conv2d(x, w1, b.repeat(4))                                            # [N, 4*Cout, H, W]
    N, _, H, W = y.shape
    y = y.view(N, 2, 2, 6, H, W).permute(0, 3, 4, 1, 5, 2).reshape(N, 6, 2 * H, 2 * W)
    assert float((y - ref).abs().max()) < 1e-12


def test_oracle_mask_head_shapes_and_range():
    from glass_amd.utils.synth import make_state_dict
    from oracle import glass_cpu as O
    sd = make_state_dict(1234, parts=("mask",))
    x = torch.randn((3, 256, 14, 14), generator=torch.Generator().manual_seed(2))
    logits = O.mask_head_logits(sd, x)
    assert tuple(logits.shape) == (3, 1, 28, 28)
    p = torch.sigmoid(logits)
    assert 0.05 < float(p.mean()) < 0.95 and float(p.std()) > 0.05          # synthetic weights give a non-trivial mask


# ------------------------------------------------------------------------------------------- GPU
def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_paste_kernel_matches_reference_golden():
    from glass_amd.ops import native as K
    masks, boxes, hw, out_bool, out_u8 = _golden()
    got = K.paste_rotated_masks(masks.to(_dev()), boxes.to(_dev()), hw, 0.5)
    assert got.dtype == torch.bool
    # BIT-EXACT: the kernel executes the reference's torch-CPU fp32 operations in their order, fused exactly where torch
    # fuses (csrc/masks.hip header; the sequence was established by emulation against this golden, which the
    # reference's own paste_masks_in_image produced): 0 differing pixels, 0 differing bytes
    diff = got.cpu().numpy() != out_bool
    print(f"rotated mask paste vs reference golden: {int(diff.sum())} of {diff.size} pixels differ")
    assert diff.sum() == 0, f"{diff.sum()} pixels differ from the reference paste"
    got8 = K.paste_rotated_masks(masks.to(_dev()), boxes.to(_dev()), hw, -1.0)
    assert got8.dtype == torch.uint8
    d8 = np.abs(got8.cpu().numpy().astype(np.int32) - out_u8.astype(np.int32))
    print(f"rotated mask paste (uint8 mode) vs reference golden: max |diff| {int(d8.max())}, {int((d8 > 0).sum())} bytes differ")
    assert int(d8.max()) == 0


@pytest.mark.gpu
def test_paste_kernel_matches_oracle_on_image_sized_canvas():
    from glass_amd.ops import native as K
    from glass_amd.utils.synth import make_boxes
    from oracle import glass_cpu as O
    g = torch.Generator().manual_seed(9)
    R, M, H, W = 12, 28, 333, 517                                # W not a multiple of 4: exercises the ragged store
    masks = torch.sigmoid(F.interpolate(torch.randn((R, 1, 6, 6), generator=g), size=(M, M), mode="bicubic") * 3)[:, 0].contiguous()
    boxes = make_boxes(3, R, H, W)
    ref = O.paste_rotated_masks(masks, boxes, (H, W), 0.5).numpy()
    got = K.paste_rotated_masks(masks.to(_dev()), boxes.to(_dev()), (H, W), 0.5).cpu().numpy()
    assert got.shape == ref.shape
    # the only operation the kernel cannot replay bit for bit is torch's SLEEF cos/sin (<= 1 ulp, not always
    # correctly rounded; the kernel rounds the fp64 value): RoIs whose cos AND sin agree are required to be bit-exact
    a = torch.deg2rad(boxes[:, 4])
    cr_c, cr_s = torch.cos(a.double()).float(), torch.sin(a.double()).float()
    same = ((torch.cos(a) == cr_c) & (torch.sin(a) == cr_s)).numpy()
    per_roi = (got != ref).reshape(R, -1).sum(1)
    print(f"paste vs oracle on {H}x{W}: differing pixels per RoI {per_roi.tolist()}, torch cos/sin correctly rounded: {same.tolist()}")
    assert int(per_roi[same].sum()) == 0
    assert int(per_roi[~same].sum()) <= 8 * int((~same).sum())
    assert K.paste_rotated_masks(masks[:0].to(_dev()), boxes[:0].to(_dev()), (H, W)).shape == (0, H, W)


@pytest.mark.gpu
def test_mask_head_matches_oracle():
    """4 x conv3x3 (Winograd kernel) + deconv-as-1x1 + pixel shuffle + predictor + sigmoid vs torch CPU."""
    import glass_amd  # noqa: F401
    from glass_amd.config import get_glass_cfg
    from glass_amd.modeling.roi_heads.rotated_mask_head import RotatedMaskRCNNConvUpsampleHead
    from glass_amd.structures.core import ShapeSpec
    from glass_amd.utils.synth import make_state_dict
    from oracle import glass_cpu as O
    cfg = get_glass_cfg(os.path.join(ROOT, "configs", "glass_icdar15_mi355x.yaml"))
    sd = make_state_dict(1234, parts=("mask",))
    head = RotatedMaskRCNNConvUpsampleHead(cfg, ShapeSpec(channels=256, height=14, width=14))
    head.import_weights(sd, _dev(), "roi_heads.mask_head.")
    x = torch.randn((5, 256, 14, 14), generator=torch.Generator().manual_seed(4))
    ref = torch.sigmoid(O.mask_head_logits(sd, x))
    got = head.forward_nhwc(x.permute(0, 2, 3, 1).contiguous().to(_dev()))
    assert tuple(got.shape) == (5, 1, 28, 28)
    assert float((got.cpu() - ref).abs().max()) < 1e-4
    assert head.forward_nhwc(torch.zeros((0, 14, 14, 256), device=_dev())).shape == (0, 1, 28, 28)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp16", "fp16s"])
def test_mask_head_in_the_fp16_modes_matches_the_emulating_oracle(prec):
    """with MASK_ON the fp16 modes round the operands of the deconv and of the predictor as well; the oracle's
    emulation restates that: the mask probabilities are closer to it than to the fp32 oracle."""
    import glass_amd  # noqa: F401
    from glass_amd.config import get_glass_cfg
    from glass_amd.modeling.roi_heads.rotated_mask_head import RotatedMaskRCNNConvUpsampleHead
    from glass_amd.ops import native as K
    from glass_amd.structures.core import ShapeSpec
    from glass_amd.utils.synth import make_state_dict
    from oracle import glass_cpu as O
    cfg = get_glass_cfg(os.path.join(ROOT, "configs", "glass_icdar15_mi355x.yaml"))
    sd = make_state_dict(1234, parts=("mask",))
    head = RotatedMaskRCNNConvUpsampleHead(cfg, ShapeSpec(channels=256, height=14, width=14))
    with K.packing_for(prec):
        head.import_weights(sd, _dev(), "roi_heads.mask_head.")
    torch.cuda.synchronize()
    x = torch.randn((5, 256, 14, 14), generator=torch.Generator().manual_seed(4))
    with O.emulate(prec):
        ref = torch.sigmoid(O.mask_head_logits(sd, x))
    ref32 = torch.sigmoid(O.mask_head_logits(sd, x))
    prev = K.set_conv_precision(prec)
    try:
        got = head.forward_nhwc(x.permute(0, 2, 3, 1).contiguous().to(_dev())).cpu()
        assert K.last_conv_path() in ("direct_fp16", "packed_fp16")
    finally:
        K.set_conv_precision(prev)
    d_emu, d_32 = float((got - ref).abs().max()), float((got - ref32).abs().max())
    print(f"[parity] mask head {prec}: max |dp| vs emulating oracle {d_emu:.2e}, vs fp32 oracle {d_32:.2e}")
    # (six chained layers, each re-rounding its fp32 input to fp16: a summation-order difference flips single roundings by one
    #  fp16 ulp; measured 9.0e-4 vs 2.1e-3 against the fp32 oracle)
    assert d_emu < 1.5e-3 and d_emu < d_32


@pytest.mark.gpu
def test_end_to_end_with_mask_inference_matches_oracle():
    """eval-CLI setting MODEL.ROI_MASK_HEAD.MASK_INFERENCE True (reference tools/eval_glass.py:106): pred_masks and
    pred_rboxes appear, raw 28x28 masks match the oracle, pasted masks match after the meta-arch postprocess."""
    import glass_amd
    from glass_amd.config import get_glass_cfg
    from glass_amd.utils.synth import make_boxes, make_image, make_state_dict
    from oracle import glass_cpu as O
    cfg = get_glass_cfg(os.path.join(ROOT, "configs", "glass_icdar15_mi355x.yaml"),
                        ["MODEL.ROI_MASK_HEAD.MASK_INFERENCE", True, "MODEL.MASK_ON", True])
    sd = make_state_dict(1234, parts=("backbone", "rpn", "box", "recog", "mask"))
    model = glass_amd.build_model(cfg)
    model.load_state_dict(sd)
    sizes = [(192, 256), (224, 224)]
    imgs = [make_image(40 + i, h, w).permute(2, 0, 1).float().contiguous() for i, (h, w) in enumerate(sizes)]
    boxes = [make_boxes(50 + i, 6, h, w) for i, (h, w) in enumerate(sizes)]
    ref = O.glass_inference(sd, imgs, cfg, injected_boxes=boxes)
    dev = _dev()
    # raw masks (no postprocess)
    out = model.inference([{"image": im.to(dev)} for im in imgs], do_postprocess=False, override_boxes=[b.to(dev) for b in boxes])
    for r, o in zip(ref, out):
        assert o.has("pred_masks") and o.has("pred_rboxes")
        assert tuple(o.pred_masks.shape) == tuple(r["pred_masks"].shape)
        assert float((o.pred_masks.cpu() - r["pred_masks"]).abs().max()) < 1e-3
    # pasted masks through the meta-arch postprocess at a different output resolution
    inputs = [{"image": im.to(dev), "height": int(h * 1.25), "width": int(w * 1.25)} for im, (h, w) in zip(imgs, sizes)]
    post = model.inference(inputs, override_boxes=[b.to(dev) for b in boxes])
    for n, (r, o) in enumerate(zip(ref, post)):
        inst = o["instances"]
        h, w = sizes[n]
        det = {"pred_boxes": boxes[n].clone(), "scores": torch.ones(len(boxes[n])), "pred_masks": r["pred_masks"]}
        want = O.meta_postprocess(det, (h, w), (int(h * 1.25), int(w * 1.25)), min_box_dim=2)
        got = inst.pred_masks.cpu()
        assert got.dtype == torch.bool and tuple(got.shape) == tuple(want["pred_masks"].shape)
        on = max(int(want["pred_masks"].sum()), 1)
        assert int((got != want["pred_masks"]).sum()) <= max(10, on // 500)
        np.testing.assert_allclose(inst.pred_rboxes.tensor.cpu().numpy(), inst.pred_boxes.tensor.cpu().numpy(), rtol=0, atol=1e-4)
    # the word post-processor keeps the mask fields with the surviving words (reference: `preds[keep]` indexing)
    from glass_amd.postprocess import build_post_processor
    from glass_amd.structures.core import RotatedBoxes
    inst = post[0]["instances"]
    words = build_post_processor(cfg)(inst)
    assert words.has("pred_masks") and words.pred_masks.shape[0] == len(words) and words.pred_masks.dtype == torch.bool
    assert isinstance(words.pred_rboxes, RotatedBoxes) and len(words.pred_rboxes.tensor) == len(words)
