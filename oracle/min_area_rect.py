"""TEST INFRASTRUCTURE (oracle): an independent fp64 minimum-area enclosing rectangle, the stand-in for `cv2.minAreaRect`
when the reference's word post-processor is run to generate goldens (reference call site:
glass/postprocess/post_processor_rotated_boxes.py:264, `center, shape, angle = cv2.minAreaRect(np.array(polygon))`, reached
from `merge_boxes` :187-216 with the 8 corner points of two word boxes).  OpenCV is in neither image; parity with OpenCV
itself is therefore UNPINNED - what this file pins is the geometry: the rectangle returned is the minimum-area one.

Deliberately shares NO code with the product (`glass_amd.postprocess.post_processor_rotated_boxes.min_area_rect` computes a
convex hull and walks its edges; the device kernel does the same in fp32): this is the brute force over EVERY ordered pair of
distinct input points as the direction of a side - a minimum-area enclosing rectangle has a side collinear with an edge of
the convex hull (Freeman & Shapira 1975), every hull edge joins two input points, and a direction that is not a hull edge
can only give a larger-or-equal rectangle, so the minimum over all pairs is the minimum over hull edges, without ever
building the hull.

Return convention: ((cx, cy), (w, h), angle_deg) with `angle_deg` the direction of the side of length `w`, image coordinates
(the RotatedRect convention; which of the four equivalent (w, h, angle) forms comes back is irrelevant to the caller: the
reference's quadrant logic :266-283 maps all four to the same box - tests/test_oracle_d2ops.py checks that on this function).

Tie-break (documented because it is a choice): candidates are visited in pair order (i, j), i major, and a later candidate
replaces the incumbent only when its area is smaller by more than 1e-12 relative - i.e. among EQUAL-area rectangles the first
pair in input order wins.  Equal areas with DIFFERENT rectangles need a symmetric point set (e.g. the corners of a square
plus the corners of the same square turned by 45 degrees); the word post-processor's inputs (two near-parallel boxes) do not
produce them, and the golden scenes are checked for a unique minimum when they are generated."""
import numpy as np


def min_area_rect_bruteforce(points, return_gap=False):
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 2)
    n = len(pts)
    best = None
    second = np.inf          # smallest area among candidates whose RECTANGLE differs from the best one (for the uniqueness check)
    cands = []
    for i in range(n):
        for j in range(n):
            if i == j:
                continue
            e = pts[j] - pts[i]
            ln = float(np.hypot(e[0], e[1]))
            if ln == 0.0:
                continue
            ux, uy = e[0] / ln, e[1] / ln
            a = pts[:, 0] * ux + pts[:, 1] * uy            # coordinates along the side
            b = -pts[:, 0] * uy + pts[:, 1] * ux           # ... and across
            w, h = float(a.max() - a.min()), float(b.max() - b.min())
            ca, cb = (a.max() + a.min()) / 2.0, (b.max() + b.min()) / 2.0
            cx, cy = ca * ux - cb * uy, ca * uy + cb * ux
            cands.append((w * h, cx, cy, w, h, float(np.degrees(np.arctan2(uy, ux)))))
    if not cands:                                          # all points coincide
        return ((float(pts[0, 0]), float(pts[0, 1])), (0.0, 0.0), 0.0) if not return_gap else (((float(pts[0, 0]), float(pts[0, 1])), (0.0, 0.0), 0.0), np.inf)
    for c in cands:
        if best is None or c[0] < best[0] * (1.0 - 1e-12) - 1e-300:
            best = c
    if return_gap:
        def same(c):
            # the same rectangle up to the four (w, h, angle) forms and up to the float32 noise of the inputs (the parallel sides
            # of a float32 polygon are parallel to ~1e-5 degrees): centre and sides within 1e-3 px, direction within 0.01 degrees
            d = (c[5] - best[5]) % 90.0
            return (abs(c[1] - best[1]) < 1e-3 and abs(c[2] - best[2]) < 1e-3 and min(d, 90.0 - d) < 1e-2 and
                    abs(max(c[3], c[4]) - max(best[3], best[4])) < 1e-3)
        others = [c[0] for c in cands if not same(c)]
        second = min(others) if others else np.inf
    _, cx, cy, w, h, ang = best
    out = ((float(cx), float(cy)), (float(w), float(h)), float(ang))
    return (out, second - best[0]) if return_gap else out
