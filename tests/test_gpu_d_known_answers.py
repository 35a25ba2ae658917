"""GPU: the HIP kernels against known answers that come from NEITHER implementation (tests/known_answers.py):
rotation direction of the rotated RoIAlign sampler, the rotated IoU against an independent float64 clipper, the
device min-area-rectangle of the word merge, and the reference's text-decode golden fed through the device
post-processing kernel (rows a12 / (c))."""
import os

import numpy as np
import pytest
import torch

from known_answers import (box_corners, iou_f64, ramp_roi_align_expected, random_box_pairs)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _cfg(opts=()):
    from glass_amd.config import get_glass_cfg
    return get_glass_cfg(os.path.join(ROOT, "configs", "glass_icdar15_mi355x.yaml"), list(opts))


def _ramp_nhwc(H, W, C=4):
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    f = torch.zeros((1, H, W, C))
    f[0, :, :, 0] = 2 * xs + 3 * ys + 1
    f[0, :, :, 1] = -1.5 * xs + 0.25 * ys + 7
    return f


RAMP_BOXES = [
    ((30.0, 24.0, 12.0, 8.0, 90.0), (4, 6), 1.0, 2),
    ((30.0, 24.0, 12.0, 8.0, 30.0), (4, 6), 1.0, 2),
    ((30.0, 24.0, 12.0, 8.0, -45.0), (2, 8), 1.0, 3),
    ((31.5, 22.25, 9.0, 14.0, 135.0), (7, 7), 1.0, 2),
    ((120.0, 96.0, 48.0, 32.0, 30.0), (4, 6), 0.25, 0),
    ((30.0, 24.0, 12.0, 8.0, -90.0), (8, 32), 1.0, 2),
    ((40.0, 30.0, 20.0, 20.0, 17.0), (128, 128), 1.0, 2),      # the image pooler's output size
]


@pytest.mark.parametrize("box,out_hw,scale,sr", RAMP_BOXES)
def test_roi_align_kernel_rotation_direction_on_a_ramp(box, out_hw, scale, sr):
    """roi_align_rotated_kernel: every bin of a linear ramp equals the ramp at the bin centre rotated
    COUNTER-CLOCKWISE (y down) about the box centre - the documented convention, not the oracle's code."""
    from glass_amd.ops import native as K
    dev = _dev()
    f = _ramp_nhwc(64, 80)
    y = K.roi_align_rotated([f.to(dev)], [scale], torch.tensor([box], device=dev), torch.zeros(1, dtype=torch.int32, device=dev),
                            out_hw, sr)
    got = y[0].cpu().numpy()
    want0 = ramp_roi_align_expected(box, out_hw, scale, 2.0, 3.0, 1.0)
    want1 = ramp_roi_align_expected(box, out_hw, scale, -1.5, 0.25, 7.0)
    e0, e1 = np.abs(got[..., 0] - want0).max(), np.abs(got[..., 1] - want1).max()
    print(f"ramp RoIAlign {box} -> max |err| {e0:.2e} / {e1:.2e}")
    assert e0 < 2e-3 and e1 < 2e-3
    wrong = ramp_roi_align_expected(box[:4] + (-box[4],), out_hw, scale, 2.0, 3.0, 1.0)
    assert np.abs(wrong - want0).max() > 1.0


def test_roi_align_kernel_plus_90_explicit_numbers():
    from glass_amd.ops import native as K
    dev = _dev()
    f = _ramp_nhwc(40, 40)
    y = K.roi_align_rotated([f.to(dev)], [1.0], torch.tensor([[10.5, 20.5, 4.0, 4.0, 90.0]], device=dev),
                            torch.zeros(1, dtype=torch.int32, device=dev), (2, 2), 2)[0, :, :, 0].cpu().numpy()
    val = lambda x, yy: 2 * x + 3 * yy + 1
    want = np.array([[val(9, 21), val(9, 19)], [val(11, 21), val(11, 19)]], dtype=np.float32)
    np.testing.assert_allclose(y, want, atol=1e-4)


def test_pairwise_iou_kernel_against_independent_float64_clipping_sweep():
    """glass_pairwise_iou_rotated on 10^4 random / near-degenerate pairs vs the float64 Sutherland-Hodgman clipper
    (and vs the oracle's C restatement of d2's fp32 algorithm, which it must follow value for value)."""
    from glass_amd.ops import native as K
    from oracle import d2ops
    dev = _dev()
    b1, b2, fam = random_box_pairs(10000, 2024)
    got = np.zeros(len(b1), dtype=np.float64)
    for lo in range(0, len(b1), 500):
        m = K.pairwise_iou_rotated(torch.from_numpy(b1[lo:lo + 500]).to(dev), torch.from_numpy(b2[lo:lo + 500]).to(dev))
        got[lo:lo + 500] = m.diagonal().cpu().double().numpy()
    want = np.array([iou_f64(b1[i], b2[i]) for i in range(len(b1))])
    ora = np.array([float(d2ops.lib().d2o_single_box_iou_rotated(d2ops._p(b1[i]), d2ops._p(b2[i]))) for i in range(len(b1))])
    err = np.abs(got - want)
    names = ["generic", "thin", "shared-edge", "identical", "1e-3deg", "nested", "concentric", "far"]
    worst = {names[f]: float(err[fam == f].max()) for f in range(8)}
    print("HIP rotated IoU vs float64 clipping, max |err| per family:", worst)
    for k in ("generic", "thin", "identical", "1e-3deg", "nested", "concentric"):
        assert worst[k] < 2e-5, (k, worst[k])
    assert worst["far"] == 0.0
    e = err[fam == 2]                               # collinear edges: see tests/test_oracle_d2ops.py
    assert np.median(e) < 1e-6 and (e > 1e-5).mean() < 0.02
    # vs the oracle (same fp32 algorithm; sin/cos of the device differ by an ulp)
    eo = np.abs(got - ora)
    print("HIP rotated IoU vs oracle C: max", float(eo[fam != 2].max()), "shared-edge disagreeing fraction",
          float((eo[fam == 2] > 1e-5).mean()))
    assert eo[fam != 2].max() < 2e-5


def test_device_min_area_rect_merges_collinear_boxes_to_their_span():
    """the merge step of postprocess_words_kernel (reference post_processor_rotated_boxes.py:187-216 -> cv2.minAreaRect
    of the pair's 8 corners): two same-height boxes on one axis must become the rectangle spanning both.  Known
    answer = the span rectangle's corner set (representation-independent)."""
    from glass_amd.postprocess.post_processor_rotated_boxes import PostProcessorRotatedBoxes
    from glass_amd.structures.core import Instances, RotatedBoxes
    dev = _dev()
    pp = PostProcessorRotatedBoxes(_cfg())
    g = np.random.default_rng(5)
    for trial in range(24):
        a = float(g.uniform(-80, 80)) if trial else 0.0
        w, h = float(g.uniform(60, 120)), float(g.uniform(18, 30))
        shift = float(g.uniform(0.3, 0.6)) * w                      # overlap 40-70 % of a box: IoA above the merge threshold
        cx, cy = float(g.uniform(300, 500)), float(g.uniform(300, 500))
        t = np.radians(a)
        ux, uy = np.cos(t), -np.sin(t)                               # the box's own x axis in image coordinates (CCW, y down)
        b = torch.tensor([[cx, cy, w, h, a], [cx + shift * ux, cy + shift * uy, w, h, a]], dtype=torch.float32)
        inst = Instances((1000, 1000))
        inst.pred_boxes = RotatedBoxes(b.to(dev))
        inst.scores = torch.tensor([0.9, 0.8], device=dev)
        inst.pred_classes = torch.zeros(2, dtype=torch.int64, device=dev)
        out = pp(inst)
        assert len(out) == 1, (trial, a, len(out))
        got = out.pred_boxes.tensor[0].cpu().numpy().astype(np.float64)
        span = np.array([cx + shift * ux / 2, cy + shift * uy / 2, w + shift, h, a])
        gc = box_corners(got)
        wc = box_corners(span)
        key = lambda c: c[np.lexsort((np.round(c[:, 1], 2), np.round(c[:, 0], 2)))]
        np.testing.assert_allclose(key(gc), key(wc), atol=0.02)
        # polygons are the corners of the merged box
        poly = out.pred_polygons[0].cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(key(poly), key(wc), atol=0.02)


def test_reference_text_decode_golden_through_the_device_kernel(golden_dir):
    """row a12: `TextEncoder.decode_prod_v2` (reference text_encoder.py:81-151) as run by the reference itself
    (tests/golden/text_decode.npz: arg-max indices + probabilities in, texts + scores out) against the text decode
    that runs INSIDE postprocess_words_kernel on the GPU."""
    from glass_amd.ops import native as K
    from glass_amd.postprocess import build_post_processor
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "text_decode.npz"), allow_pickle=False)
    idx, prob = g["idx"], g["prob"]
    R, T = idx.shape
    C = len(g["characters"])
    tp = np.zeros((1, R, T, C), dtype=np.float32)
    for r in range(R):
        for t in range(T):
            tp[0, r, t, :] = min(prob[r, t] * 0.5, (1.0 - prob[r, t]) / (C - 1))
            tp[0, r, t, idx[r, t]] = prob[r, t]
    pp = build_post_processor(_cfg(["POST_PROCESSING.TEXT_THRESHOLD", 0.0]))
    assert pp.text_encoder.character == [str(c) for c in g["characters"]]
    # R well-separated boxes with descending scores (no merges, no NMS)
    boxes = torch.tensor([[[100.0 + 150 * r, 200.0, 80.0, 30.0, 0.0] for r in range(R)]])
    scores = torch.tensor([[0.99 - 0.01 * r for r in range(R)]])
    stop = pp.text_encoder.character.index("[s]")
    out = K.postprocess_words(boxes.to(dev), scores.to(dev), torch.tensor([R], dtype=torch.int32, device=dev),
                              torch.from_numpy(tp).to(dev), None, pp._thresholds(), stop)
    n = int(out["count"][0])
    src = out["src"][0, :n].cpu().tolist()
    chars = out["char"][0, :n].cpu().numpy()
    tlen = out["text_len"][0, :n].cpu().tolist()
    tsc = out["text_score"][0, :n].cpu().numpy()
    want_texts = [str(t) for t in g["texts"]]
    # the reference drops nothing at threshold 0 except what its own score filter removes: every word must be there
    assert sorted(src) == list(range(R))
    for j, r in enumerate(src):
        text = "".join(pp.text_encoder.character[int(c)] for c in chars[j, :tlen[j]])
        assert text == want_texts[r], (r, text, want_texts[r])
        # fp32 product of <= 26 probabilities: the multiplication order may differ from numpy's by a few ulp
        assert abs(float(tsc[j]) - float(g["scores"][r])) <= 1e-5 * float(g["scores"][r]) + 1e-12
