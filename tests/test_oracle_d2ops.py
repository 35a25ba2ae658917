"""CPU: analytic known-answer tests pinning the restated detectron2 v0.6 ops (the reference holds
no fixtures for them).  These pin oracle/d2_ops.c + oracle/d2ops.py, which in turn
are what the HIP kernels are compared against on the GPU."""
import math

import numpy as np
import pytest
import torch

from oracle import d2ops


def test_iou_identities():
    b = torch.tensor([[10.0, 20.0, 8.0, 4.0, 30.0]])
    assert abs(float(d2ops.pairwise_iou_rotated(b, b)) - 1.0) < 1e-5
    # two unit squares offset by 0.5 along x: inter 0.5, union 1.5
    a = torch.tensor([[0.0, 0.0, 1.0, 1.0, 0.0]])
    c = torch.tensor([[0.5, 0.0, 1.0, 1.0, 0.0]])
    assert abs(float(d2ops.pairwise_iou_rotated(a, c)) - 1.0 / 3.0) < 1e-6
    # a 90-degree rotation swaps w/h: (w=4,h=2,90deg) == (w=2,h=4,0deg)
    r1 = torch.tensor([[5.0, 5.0, 4.0, 2.0, 90.0]])
    r2 = torch.tensor([[5.0, 5.0, 2.0, 4.0, 0.0]])
    assert abs(float(d2ops.pairwise_iou_rotated(r1, r2)) - 1.0) < 1e-5
    # square vs the same square rotated by 45 degrees: octagon area = 2*(sqrt(2)-1)*s^2
    s1 = torch.tensor([[0.0, 0.0, 2.0, 2.0, 0.0]])
    s2 = torch.tensor([[0.0, 0.0, 2.0, 2.0, 45.0]])
    inter = 2 * (math.sqrt(2) - 1) * 4
    assert abs(float(d2ops.pairwise_iou_rotated(s1, s2)) - inter / (8 - inter)) < 1e-5
    # disjoint and degenerate
    far = torch.tensor([[100.0, 100.0, 2.0, 2.0, 10.0]])
    assert float(d2ops.pairwise_iou_rotated(s1, far)) == 0.0
    assert float(d2ops.pairwise_iou_rotated(s1, torch.tensor([[0.0, 0.0, 0.0, 2.0, 0.0]]))) == 0.0
    # rotation invariance of IoU
    g = torch.Generator().manual_seed(0)
    for _ in range(20):
        b1 = torch.rand(1, 5, generator=g) * torch.tensor([10, 10, 8, 8, 360.0]) + torch.tensor([0, 0, 1, 1, -180.0])
        b2 = torch.rand(1, 5, generator=g) * torch.tensor([10, 10, 8, 8, 360.0]) + torch.tensor([0, 0, 1, 1, -180.0])
        i1 = float(d2ops.pairwise_iou_rotated(b1, b2))
        i2 = float(d2ops.pairwise_iou_rotated(b2, b1))
        assert abs(i1 - i2) < 1e-4 and 0.0 <= i1 <= 1.0 + 1e-5


def test_nms_rotated_greedy_and_ge_threshold():
    boxes = torch.tensor([[0.0, 0.0, 1.0, 1.0, 0.0], [0.5, 0.0, 1.0, 1.0, 0.0], [10.0, 10.0, 2.0, 2.0, 45.0],
                          [0.0, 0.0, 1.0, 1.0, 0.0]])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.95])
    # iou(0,1) = 1/3: suppressed at thr 0.3, kept at 0.35; box 3 (== box 0) has the top score
    assert d2ops.nms_rotated(boxes, scores, 0.3).tolist() == [3, 2]
    assert d2ops.nms_rotated(boxes, scores, 0.35).tolist() == [3, 1, 2]
    # CPU semantics: iou >= thr suppresses (threshold exactly at the IoU value)
    iou = float(d2ops.pairwise_iou_rotated(boxes[:1], boxes[1:2]))
    assert d2ops.nms_rotated(boxes[:2], scores[:2], iou).tolist() == [0]
    # batched: different categories never suppress each other
    keep = d2ops.batched_nms_rotated(boxes, scores, torch.tensor([0, 1, 0, 2]), 0.3)
    assert sorted(keep.tolist()) == [0, 1, 2, 3]
    assert d2ops.batched_nms_rotated(torch.zeros((0, 5)), torch.zeros(0), torch.zeros(0, dtype=torch.long), 0.5).numel() == 0


def test_roi_align_rotated_known_answers():
    # on a linear ramp f(y,x) = 2x + 3y + 1 bilinear interpolation is exact and a bin average is the
    # value at the bin centre (in continuous coords: pixel centre i is at i + 0.5)
    H, W = 32, 48
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    f = (2 * xs + 3 * ys + 1)[None, None]
    cx, cy, w, h = 20.0, 14.0, 12.0, 8.0
    out = d2ops.roi_align_rotated(f, torch.tensor([[0, cx, cy, w, h, 0.0]]), (4, 6), 1.0, 2)[0, 0]
    for ph in range(4):
        for pw in range(6):
            bx = cx - w / 2 + (pw + 0.5) * w / 6 - 0.5
            by = cy - h / 2 + (ph + 0.5) * h / 4 - 0.5
            assert abs(float(out[ph, pw]) - (2 * bx + 3 * by + 1)) < 1e-3
    # angle 180: same bins visited in reverse order
    out180 = d2ops.roi_align_rotated(f, torch.tensor([[0, cx, cy, w, h, 180.0]]), (4, 6), 1.0, 2)[0, 0]
    np.testing.assert_allclose(out180.numpy(), out.flip(0, 1).numpy(), atol=1e-3)
    # angle 90 (CCW in image coords): output rows run along -x ... check centre symmetry only
    out90 = d2ops.roi_align_rotated(f, torch.tensor([[0, cx, cy, 8.0, 8.0, 90.0]]), (2, 2), 1.0, 2)[0, 0]
    assert abs(float(out90.mean()) - (2 * (cx - 0.5) + 3 * (cy - 0.5) + 1)) < 1e-3
    # spatial scale + adaptive sampling (sampling_ratio 0): ceil(roi/bins) samples, still exact on a ramp
    out_s = d2ops.roi_align_rotated(f, torch.tensor([[0, 80.0, 56.0, 48.0, 32.0, 0.0]]), (4, 6), 0.25, 0)[0, 0]
    np.testing.assert_allclose(out_s.numpy(), out.numpy(), atol=1e-3)
    # fully outside -> zeros; empty roi list -> empty output
    z = d2ops.roi_align_rotated(f, torch.tensor([[0, -100.0, -100.0, 10.0, 10.0, 0.0]]), (2, 2), 1.0, 2)
    assert float(z.abs().max()) == 0.0
    assert d2ops.roi_align_rotated(f, torch.zeros((0, 6)), (2, 2), 1.0, 2).shape == (0, 1, 2, 2)


def test_apply_deltas_identity_clamp_and_wrap():
    boxes = torch.tensor([[10.0, 20.0, 30.0, 40.0, 170.0], [5.0, 5.0, 2.0, 3.0, -179.0]])
    out = d2ops.apply_deltas_rotated(torch.zeros((2, 5)), boxes, (10, 10, 5, 5, 10))
    np.testing.assert_allclose(out.numpy(), boxes.numpy(), atol=1e-5)
    d = torch.tensor([[0.0, 0.0, 100.0, 0.0, 10.0 * math.pi * 20 / 180]])     # dw clamped, +20 deg -> wraps
    out = d2ops.apply_deltas_rotated(d, boxes[:1], (10, 10, 5, 5, 10))
    assert abs(float(out[0, 2]) - 30.0 * 1000.0 / 16) < 1e-2
    assert abs(float(out[0, 4]) - (-170.0)) < 1e-3


def test_anchor_generator_count_and_order():
    cell = d2ops.rotated_cell_anchors(16, (0.2, 0.5, 1.0), (-90, -45, 0, 45))
    assert cell.shape == (12, 5)
    assert cell[:, 4].tolist() == [-90, -45, 0, 45] * 3                   # ratio-major, angle-minor
    np.testing.assert_allclose((cell[:, 2] * cell[:, 3]).numpy(), 256.0, rtol=1e-5)
    np.testing.assert_allclose((cell[:4, 3] / cell[:4, 2]).numpy(), 0.2, rtol=1e-5)
    total = 0
    for i, (h, w) in enumerate(((256, 256), (128, 128), (64, 64), (32, 32), (16, 16))):
        a = d2ops.rotated_grid_anchors(h, w, 4 << i, cell)
        total += len(a)
        assert a[12 + 5, 0] == 4 << i and a[12 + 5, 1] == 0               # second cell: x = stride, y = 0
    assert total == 12 * 87296 == 1047552


def test_clip_only_near_horizontal_and_level_assignment():
    b = torch.tensor([[5.0, 5.0, 20.0, 4.0, 0.5], [5.0, 5.0, 20.0, 4.0, 30.0], [50.0, 95.0, 10.0, 20.0, -180.5]])
    d2ops.clip_rotated_(b, (100, 100))
    np.testing.assert_allclose(b[0].numpy(), [7.5, 5.0, 15.0, 4.0, 0.5], atol=1e-5)
    np.testing.assert_allclose(b[1].numpy(), [5.0, 5.0, 20.0, 4.0, 30.0], atol=1e-5)
    assert abs(float(b[2, 4]) - 179.5) < 1e-4                              # angle normalised, not clipped
    lv = d2ops.assign_boxes_to_levels(torch.tensor([[0, 0, 10.0, 10.0, 0], [0, 0, 224.0, 224.0, 0], [0, 0, 2000.0, 2000.0, 0],
                                                    [0, 0, 112.0, 112.0, 0]]), 2, 6)
    assert lv.tolist() == [0, 2, 4, 1]
