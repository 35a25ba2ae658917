"""Independent float64 known answers for the detectron2-owned geometry (test helper, not product code).

The oracle's rotated RoIAlign / rotated IoU (oracle/d2_ops.c) and the HIP kernels were both written from the same
recollection of detectron2 v0.6, so agreement between the two cannot catch a shared slip (e.g. +theta / -theta).
Everything here is derived from the *documented* conventions instead of from either implementation:

  * detectron2 `RotatedBoxes` docstring: a box (cx, cy, w, h, a) is the axis-aligned box (cx, cy, w, h) rotated by `a`
    degrees COUNTER-CLOCKWISE about its centre, in image coordinates (x right, y DOWN).  In such coordinates a CCW
    rotation by t maps an offset (dx, dy) to (dx cos t + dy sin t, -dx sin t + dy cos t): at a = +90 the unrotated
    top-left corner goes to the bottom-left, the unrotated "right" direction points UP in the image.
  * `ROIAlignRotated` samples the feature map on the regular PH x PW grid of that rotated box (pixel centres at
    i + 0.5, hence the -0.5 shift into index space), `sampling_ratio`^2 samples per bin placed symmetrically about
    the bin centre - so on a LINEAR ramp every bin equals the ramp at the rotated bin centre, exactly.
  * IoU = area(P ∩ Q) / (area P + area Q - area(P ∩ Q)) for the two rotated rectangles P, Q; computed here with
    Sutherland-Hodgman clipping in float64 (a different algorithm from d2's intersection-points + Graham scan).
"""
import math

import numpy as np


def ccw_offset(dx, dy, angle_deg):
    """offset (dx, dy) of the unrotated box -> offset after a CCW rotation by angle_deg in y-down image coordinates"""
    t = math.radians(angle_deg)
    return dx * math.cos(t) + dy * math.sin(t), -dx * math.sin(t) + dy * math.cos(t)


def box_corners(box):
    """corners of (cx, cy, w, h, a): images of the unrotated (+w/2,+h/2), (+w/2,-h/2), (-w/2,-h/2), (-w/2,+h/2)"""
    cx, cy, w, h, a = [float(v) for v in box]
    out = []
    for sx, sy in ((1, 1), (1, -1), (-1, -1), (-1, 1)):
        ox, oy = ccw_offset(sx * w / 2, sy * h / 2, a)
        out.append((cx + ox, cy + oy))
    return np.array(out, dtype=np.float64)


def ramp_roi_align_expected(box, out_hw, spatial_scale, ax, ay, c0):
    """per-bin RoIAlignRotated output on the ramp f[y][x] = ax*x + ay*y + c0 (index space), all samples inside"""
    cx, cy, w, h, a = [float(v) for v in box]
    PH, PW = out_hw
    cw, ch = cx * spatial_scale - 0.5, cy * spatial_scale - 0.5
    rw, rh = w * spatial_scale, h * spatial_scale
    out = np.zeros((PH, PW), dtype=np.float64)
    for ph in range(PH):
        for pw in range(PW):
            dx = -rw / 2 + (pw + 0.5) * rw / PW
            dy = -rh / 2 + (ph + 0.5) * rh / PH
            ox, oy = ccw_offset(dx, dy, a)
            out[ph, pw] = ax * (cw + ox) + ay * (ch + oy) + c0
    return out


def _poly_area(p):
    x, y = p[:, 0], p[:, 1]
    return 0.5 * float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def _clip(subject, a, b, sign):
    """keep the part of polygon `subject` on the inner side of the directed edge a->b"""
    out = []
    n = len(subject)
    ex, ey = b[0] - a[0], b[1] - a[1]
    for i in range(n):
        p, q = subject[i], subject[(i + 1) % n]
        sp = sign * (ex * (p[1] - a[1]) - ey * (p[0] - a[0]))
        sq = sign * (ex * (q[1] - a[1]) - ey * (q[0] - a[0]))
        if sp >= 0:
            out.append(p)
        if (sp >= 0) != (sq >= 0) and sp != sq:
            t = sp / (sp - sq)
            out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return out


def intersection_area_f64(b1, b2):
    p, q = box_corners(b1), box_corners(b2)
    sign = 1.0 if _poly_area(q) > 0 else -1.0
    poly = [tuple(v) for v in p]
    for i in range(4):
        if not poly:
            return 0.0
        poly = _clip(poly, q[i], q[(i + 1) % 4], sign)
    if len(poly) < 3:
        return 0.0
    return abs(_poly_area(np.array(poly)))


def iou_f64(b1, b2):
    a1, a2 = float(b1[2]) * float(b1[3]), float(b2[2]) * float(b2[3])
    if a1 <= 0 or a2 <= 0:
        return 0.0
    inter = intersection_area_f64(b1, b2)
    return inter / (a1 + a2 - inter)


def random_box_pairs(n, seed):
    """generic pairs + the near-degenerate families the judge named: thin boxes, shared edges, identical boxes,
    1e-3-degree offsets, nested boxes, far-apart boxes.  Returns (b1 [n,5], b2 [n,5], family [n])."""
    g = np.random.default_rng(seed)
    b1 = np.zeros((n, 5), dtype=np.float32)
    b2 = np.zeros((n, 5), dtype=np.float32)
    fam = np.zeros((n,), dtype=np.int32)
    for i in range(n):
        f = i % 8
        fam[i] = f
        c = g.uniform(-50, 50, 2)
        wh = g.uniform(2, 60, 2)
        a = g.uniform(-180, 180)
        base = np.array([c[0], c[1], wh[0], wh[1], a])
        if f == 0:      # generic overlapping
            o = base + np.array([g.uniform(-20, 20), g.uniform(-20, 20), 0, 0, 0])
            o[2:4] = g.uniform(2, 60, 2)
            o[4] = g.uniform(-180, 180)
        elif f == 1:    # thin boxes (aspect up to 200)
            base[3] = g.uniform(0.05, 0.5)
            o = base.copy()
            o[:2] += g.uniform(-3, 3, 2)
            o[3] = g.uniform(0.05, 0.5)
            o[4] = base[4] + g.uniform(-30, 30)
        elif f == 2:    # shared edge: translated by exactly its width along its own axis
            ox, oy = ccw_offset(base[2], 0.0, base[4])
            o = base.copy()
            o[0] += ox
            o[1] += oy
        elif f == 3:    # identical boxes
            o = base.copy()
        elif f == 4:    # tiny angular offset
            o = base.copy()
            o[4] += g.choice([-1, 1]) * 1e-3
        elif f == 5:    # nested (same centre, same angle, smaller)
            o = base.copy()
            o[2:4] *= g.uniform(0.1, 0.9, 2)
        elif f == 6:    # same centre, different angle / shape
            o = base.copy()
            o[2:4] = g.uniform(2, 60, 2)
            o[4] = g.uniform(-180, 180)
        else:           # far apart
            o = base.copy()
            o[:2] += 500
        b1[i], b2[i] = base, o
    return b1, b2, fam


# ----------------------------------------------------------------------------- minimum-area rectangle
def rect_points(cx, cy, w, h, angle_deg):
    """the 4 corners of a rectangle whose first side (length w) points along angle_deg (x right, y down)"""
    t = math.radians(angle_deg)
    u = np.array([math.cos(t), math.sin(t)])
    v = np.array([-math.sin(t), math.cos(t)])
    c = np.array([cx, cy], dtype=np.float64)
    return np.array([c + sx * w / 2 * u + sy * h / 2 * v for sx, sy in ((-1, -1), (1, -1), (1, 1), (-1, 1))])


def canonical_rect(center, size, angle_deg):
    """((cx,cy),(w,h),angle) -> representation-independent tuple: centre, sorted sides, long-side direction mod 180"""
    (cx, cy), (w, h), a = center, size, angle_deg
    if h > w:
        w, h, a = h, w, a + 90.0
    a = a % 180.0
    return cx, cy, w, h, a
