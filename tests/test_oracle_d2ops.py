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


# ------------------------------------------------------------------ direction-sensitive pins
def _ramp(H, W, ax, ay, c0):
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    return (ax * xs + ay * ys + c0)[None, None]


RAMP_BOXES = [
    # (cx, cy, w, h, angle), out (PH, PW), spatial scale, sampling ratio
    ((30.0, 24.0, 12.0, 8.0, 90.0), (4, 6), 1.0, 2),
    ((30.0, 24.0, 12.0, 8.0, 30.0), (4, 6), 1.0, 2),
    ((30.0, 24.0, 12.0, 8.0, -45.0), (2, 8), 1.0, 3),
    ((31.5, 22.25, 9.0, 14.0, 135.0), (7, 7), 1.0, 2),
    ((120.0, 96.0, 48.0, 32.0, 30.0), (4, 6), 0.25, 0),       # scaled level + adaptive sampling
    ((30.0, 24.0, 12.0, 8.0, -90.0), (8, 32), 1.0, 2),
]


@pytest.mark.parametrize("box,out_hw,scale,sr", RAMP_BOXES)
def test_roi_align_rotation_direction_on_a_ramp(box, out_hw, scale, sr):
    """f = 2x + 3y + 1 is NOT symmetric under x <-> y or under a sign flip of the angle: a +theta / -theta slip (or a
    swapped sin/cos term) in the sampler changes every bin.  Expected values come from the documented convention
    (tests/known_answers.py), not from either implementation."""
    from known_answers import ramp_roi_align_expected
    f = _ramp(64, 80, 2.0, 3.0, 1.0)
    got = d2ops.roi_align_rotated(f, torch.tensor([[0.0, *box]]), out_hw, scale, sr)[0, 0].numpy()
    want = ramp_roi_align_expected(box, out_hw, scale, 2.0, 3.0, 1.0)
    np.testing.assert_allclose(got, want, atol=2e-3)
    # and the mirrored angle really is a different answer (the test has teeth)
    wrong = ramp_roi_align_expected(box[:4] + (-box[4],), out_hw, scale, 2.0, 3.0, 1.0)
    assert np.abs(wrong - want).max() > 1.0


def test_roi_align_plus_90_explicit_numbers():
    """a = +90 (CCW, y down): the box's own x axis points UP in the image, its y axis points RIGHT.  2x2 bins of a
    4x4 box centred at index (10, 20) [= continuous (10.5, 20.5)] on f = 2x + 3y + 1:
    bin (ph, pw) centre offset (dx, dy) = (+-1, +-1) -> image offset (dy, -dx)."""
    f = _ramp(40, 40, 2.0, 3.0, 1.0)
    out = d2ops.roi_align_rotated(f, torch.tensor([[0, 10.5, 20.5, 4.0, 4.0, 90.0]]), (2, 2), 1.0, 2)[0, 0]
    val = lambda x, y: 2 * x + 3 * y + 1
    want = [[val(10 - 1, 20 + 1), val(10 - 1, 20 - 1)],     # ph=0 (dy=-1): x = 10 + dy ; pw=0 (dx=-1): y = 20 - dx
            [val(10 + 1, 20 + 1), val(10 + 1, 20 - 1)]]
    np.testing.assert_allclose(out.numpy(), np.array(want, dtype=np.float32), atol=1e-4)


def test_rotated_vertices_order_and_direction():
    """get_rotated_vertices at +30 degrees: pts[0] is the image of the unrotated (+w/2, +h/2) corner, pts[1] of
    (+w/2, -h/2), then their point reflections; CCW in y-down coordinates."""
    from known_answers import box_corners
    for box in ([5.0, 3.0, 4.0, 2.0, 30.0], [5.0, 3.0, 4.0, 2.0, -30.0], [0.0, 0.0, 10.0, 1.0, 90.0], [-7.0, 2.0, 3.0, 9.0, 147.0]):
        np.testing.assert_allclose(d2ops.rotated_vertices(box), box_corners(box), atol=1e-5)
    v = d2ops.rotated_vertices([0.0, 0.0, 4.0, 2.0, 90.0])
    # d2 docstring example: at 90 degrees the unrotated top-left corner (-2,-1) ends up bottom-left: (-1, +2)
    np.testing.assert_allclose(v[2], [-1.0, 2.0], atol=1e-5)


def test_iou_against_independent_float64_clipping_sweep():
    """10^4 random and near-degenerate pairs: d2's fp32 intersection-points + Graham-scan algorithm vs a float64
    Sutherland-Hodgman clipper written from the documented box convention (a sign slip in the vertex formula moves
    the boxes relative to each other whenever their centres differ, so this also pins the direction)."""
    from known_answers import iou_f64, random_box_pairs
    b1, b2, fam = random_box_pairs(10000, 2024)
    got = np.array([float(d2ops.lib().d2o_single_box_iou_rotated(d2ops._p(b1[i]), d2ops._p(b2[i]))) for i in range(len(b1))])
    want = np.array([iou_f64(b1[i], b2[i]) for i in range(len(b1))])
    err = np.abs(got - want)
    names = ["generic", "thin", "shared-edge", "identical", "1e-3deg", "nested", "concentric", "far"]
    worst = {names[f]: float(err[fam == f].max()) for f in range(8)}
    print("rotated IoU vs float64 clipping, max |err| per family:", worst)
    for k in ("generic", "thin", "identical", "1e-3deg", "nested", "concentric"):
        assert worst[k] < 1e-5, (k, worst[k])                  # measured: 2e-7 ... 2.4e-6
    assert worst["far"] == 0.0
    # collinear edges (a box against its own translate by one width): the published fp32 algorithm divides two
    # rounding residues (`t = cross / det` with det ~ 1e-5 instead of 0) and can place an "intersection" anywhere on
    # the common line, i.e. report half a box of overlap for two boxes that only touch (known d2 behaviour; 0.5 % of
    # this family).  The restatement keeps that; it is bounded here in frequency, not in value.
    e = err[fam == 2]
    assert np.median(e) < 1e-6 and (e > 1e-5).mean() < 0.01
    assert np.mean(err[fam != 2]) < 1e-6
    # teeth: the mirrored-angle convention is a different function on this sample
    flipped = np.array([iou_f64(b1[i] * [1, 1, 1, 1, -1], b2[i] * [1, 1, 1, 1, -1]) for i in range(0, 2000, 8)])
    assert np.abs(flipped - want[0:2000:8]).max() > 0.1


def test_min_area_rect_known_answers():
    """the product's host `min_area_rect` (post_processor_rotated_boxes.py:196-216 call site), pinned analytically: the
    minimum-area rectangle of a rectangle's own corners is that rectangle; of two collinear boxes, their span.  (The goldens
    of the word merge are generated with the INDEPENDENT brute force of oracle/min_area_rect.py bound as cv2.minAreaRect;
    test_min_area_rect_product_equals_the_independent_oracle below holds the two to each other.)"""
    from glass_amd.postprocess.post_processor_rotated_boxes import min_area_rect
    from known_answers import canonical_rect, rect_points
    g = np.random.default_rng(7)
    for _ in range(200):
        cx, cy = g.uniform(-100, 100, 2)
        w, h = g.uniform(5, 80), g.uniform(1, 4.9)
        a = g.uniform(-180, 180)
        pts = rect_points(cx, cy, w, h, a)
        got = canonical_rect(*min_area_rect(pts[g.permutation(4)]))
        np.testing.assert_allclose(got[:4], (cx, cy, w, h), atol=1e-6)
        d = abs(got[4] - a % 180.0)
        assert min(d, 180.0 - d) < 1e-6
        # two rectangles of the same height side by side on one axis -> the 8 points' rectangle is the union span
        gap = g.uniform(0, 30)
        t = np.radians(a)
        shift = np.array([np.cos(t), np.sin(t)]) * (w + gap)
        both = np.concatenate([pts, pts + shift])
        got = canonical_rect(*min_area_rect(both[g.permutation(8)]))
        np.testing.assert_allclose(got[:4], (cx + shift[0] / 2, cy + shift[1] / 2, 2 * w + gap, h), atol=1e-6)
        d = abs(got[4] - a % 180.0)
        assert min(d, 180.0 - d) < 1e-6
    # duplicates and collinear points
    c, s, ang = min_area_rect(np.array([[1.0, 1.0]] * 8))
    assert c == (1.0, 1.0) and s == (0.0, 0.0)
    c, s, ang = min_area_rect(np.array([[0.0, 0.0], [2.0, 2.0], [4.0, 4.0], [1.0, 1.0]] * 2))
    assert np.allclose(c, (2.0, 2.0)) and abs(max(s) - np.hypot(4, 4)) < 1e-9 and min(s) == 0.0
    # a square inside a larger rectangle's corner set changes nothing
    outer = rect_points(0, 0, 10, 4, 20)
    inner = rect_points(0, 0, 2, 2, 65)
    got = canonical_rect(*min_area_rect(np.concatenate([outer, inner])))
    np.testing.assert_allclose(got, (0, 0, 10, 4, 20), atol=1e-6)


def test_min_area_rect_product_equals_the_independent_oracle():
    """oracle/min_area_rect.py (fp64 brute force over every point pair, no hull, no code shared with the product) against the
    product's hull walk on 1000 random 8-point sets shaped like the word merge's inputs (the corners of two boxes), on
    generic point clouds, and on designed ties.  Compared as rectangles (centre, sorted sides, long-side direction mod 180):
    which of the four (w, h, angle) forms comes back is irrelevant to the caller (reference :266-283 maps all four to one
    box - checked here too).  Tie-break, documented in the oracle: the first minimal pair in input order; for EQUAL-area
    rectangles of different shape (a square and the same square turned by 45 degrees) the two implementations may
    legitimately differ, so those cases assert the AREA only."""
    from glass_amd.postprocess.post_processor_rotated_boxes import min_area_rect
    from known_answers import canonical_rect, rect_points
    from oracle.min_area_rect import min_area_rect_bruteforce
    g = np.random.default_rng(11)
    worst = 0.0
    for k in range(1000):
        if k % 2 == 0:      # two word boxes: nearly parallel, overlapping or adjacent
            cx, cy, w, h, a = g.uniform(-200, 200), g.uniform(-200, 200), g.uniform(10, 120), g.uniform(4, 30), g.uniform(-180, 180)
            p1 = rect_points(cx, cy, w, h, a)
            t = np.radians(a)
            p2 = rect_points(cx + np.cos(t) * g.uniform(0.2, 1.2) * w, cy + np.sin(t) * g.uniform(0.2, 1.2) * w + g.uniform(-3, 3),
                             w * g.uniform(0.5, 1.5), h * g.uniform(0.7, 1.3), a + g.uniform(-8, 8))
            pts = np.concatenate([p1, p2])
        else:               # generic clouds of 3..8 points
            pts = g.uniform(-50, 50, (g.integers(3, 9), 2))
        pts = pts[g.permutation(len(pts))]
        want, gap = min_area_rect_bruteforce(pts, return_gap=True)
        got = min_area_rect(pts)
        assert abs(got[1][0] * got[1][1] - want[1][0] * want[1][1]) <= 1e-9 * max(1.0, want[1][0] * want[1][1])
        if gap > 1e-9 * want[1][0] * want[1][1]:            # a unique minimum: the same rectangle
            a_, b_ = canonical_rect(*got), canonical_rect(*want)
            if abs(a_[2] - a_[3]) < 1e-9:                   # (a square: its direction is defined mod 90 only)
                continue
            d = abs(a_[4] - b_[4])
            worst = max(worst, float(np.abs(np.array(a_[:4]) - np.array(b_[:4])).max()), min(d, 180.0 - d))
    assert worst < 1e-7, worst
    # designed ties: a rectangle's corners given twice; a square + the same square turned by 45 degrees (two different
    # minimum-area rectangles of equal area: only the area is defined)
    r = rect_points(3, -2, 40, 10, 17)
    a_, b_ = canonical_rect(*min_area_rect(np.concatenate([r, r]))), canonical_rect(*min_area_rect_bruteforce(np.concatenate([r, r])))
    np.testing.assert_allclose(a_, b_, atol=1e-9)
    sq = np.concatenate([rect_points(0, 0, 10, 10, 0), rect_points(0, 0, 10, 10, 45)])
    ga, gb = min_area_rect(sq), min_area_rect_bruteforce(sq)
    assert abs(ga[1][0] * ga[1][1] - gb[1][0] * gb[1][1]) < 1e-9


def test_reference_quadrant_logic_is_invariant_to_the_rect_form():
    """reference post_processor_rotated_boxes.py:264-283: `angle = 90 - angle`, then the orientation decides which of w / h is
    the width - restated here on the four equivalent (w, h, angle) forms of one rectangle: all give the same rotated box, so
    the RotatedRect convention of the minAreaRect stand-in cannot leak into the goldens."""
    def ref_box(center, shape, angle, orientation):
        angle = 90 - angle
        diff = (orientation - angle + 180) % 360 - 180
        if -45 < diff <= 45:
            width, height = shape[1], shape[0]
        elif 45 < diff <= 135:
            width, height = shape[0], shape[1]
            angle += 90
        elif -135 < diff <= -45:
            width, height = shape[0], shape[1]
            angle -= 90
        else:
            width, height = shape[1], shape[0]
            angle += 180
        return center[0], center[1], width, height, (angle + 180) % 360 - 180
    g = np.random.default_rng(3)
    for _ in range(200):
        c, w, h, a, o = (g.uniform(0, 100), g.uniform(0, 100)), g.uniform(5, 60), g.uniform(2, 20), g.uniform(-180, 180), g.uniform(-180, 180)
        forms = [(w, h, a), (h, w, a + 90), (w, h, a + 180), (h, w, a - 90)]
        boxes = [ref_box(c, (fw, fh), fa, o) for fw, fh, fa in forms]
        for b in boxes[1:]:
            np.testing.assert_allclose(b, boxes[0], atol=1e-9)
