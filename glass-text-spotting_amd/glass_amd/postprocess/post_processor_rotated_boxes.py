"""Host-side post-processing of rotated word boxes.

Behavioural mirror of reference glass/postprocess/post_processor_rotated_boxes.py:
`PostProcessorRotatedBoxes.{__call__ :66-87, filter_small_boxes :89-94, post_process_word_preds
:96-106, merge_intersecting_boxes :108-184, _merge_rotated_boxes :187-216, boxes_to_polygons
:219-250, polygons_to_rotated_boxes :253-286}`.  This tail stays host Python (ms-scale, <=100
boxes); the rotated IoU matrix and the NMS inside the merge loop run on the HIP kernels.
`cv2.minAreaRect` (OpenCV is not installed on either box) is restated for the 8-point case by
`min_area_rect` below; the reference's follow-up orientation correction makes the result
independent of which of the four equivalent (size, angle) representations is returned.
The reference quirk that `merged_angle` is in RADIANS when scores are given (:203-206) while it
is compared with degrees (:267) is reproduced, not fixed.
"""
from __future__ import annotations

import logging
import time

import numpy as np
import torch

from ..structures.boxes import nms_rotated, pairwise_ioa_rotated
from ..structures.core import Instances, RotatedBoxes
from ..utils.registry import Registry

POST_PROCESSOR_REGISTRY = Registry("POST_PROCESSOR")


def build_post_processor(cfg, *args, **kwargs):
    return POST_PROCESSOR_REGISTRY.get(cfg.POST_PROCESSING.NAME)(cfg, *args, **kwargs)


def _convex_hull(pts: np.ndarray) -> np.ndarray:
    """Andrew monotone chain; returns hull vertices in counter-clockwise order (x right, y up)."""
    p = np.unique(pts, axis=0)
    if len(p) <= 2:
        return p
    p = p[np.lexsort((p[:, 1], p[:, 0]))]

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower, upper = [], []
    for q in p:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], q) <= 0:
            lower.pop()
        lower.append(q)
    for q in p[::-1]:
        while len(upper) >= 2 and cross(upper[-2], upper[-1], q) <= 0:
            upper.pop()
        upper.append(q)
    return np.array(lower[:-1] + upper[:-1])


def min_area_rect(points: np.ndarray):
    """Minimum-area enclosing rectangle: ((cx,cy), (size_along_angle, size_across), angle_deg), the
    OpenCV RotatedRect convention (angle = direction of the first size, image coordinates)."""
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 2)
    hull = _convex_hull(pts)
    if len(hull) == 1:
        return (float(hull[0, 0]), float(hull[0, 1])), (0.0, 0.0), 0.0
    if len(hull) == 2:
        d = hull[1] - hull[0]
        c = (hull[0] + hull[1]) / 2
        return (float(c[0]), float(c[1])), (float(np.hypot(*d)), 0.0), float(np.degrees(np.arctan2(d[1], d[0])))
    best = None
    for i in range(len(hull)):
        e = hull[(i + 1) % len(hull)] - hull[i]
        n = np.hypot(*e)
        if n == 0:
            continue
        u = e / n
        v = np.array([-u[1], u[0]])
        pu, pv = hull @ u, hull @ v
        w, h = pu.max() - pu.min(), pv.max() - pv.min()
        if best is None or w * h < best[0]:
            c = u * (pu.max() + pu.min()) / 2 + v * (pv.max() + pv.min()) / 2
            best = (w * h, c, w, h, np.degrees(np.arctan2(u[1], u[0])))
    _, c, w, h, ang = best
    return (float(c[0]), float(c[1])), (float(w), float(h)), float(ang)


@POST_PROCESSOR_REGISTRY.register()
class PostProcessorRotatedBoxes:
    def __init__(self, cfg):
        self.logger = logging.getLogger(__name__)
        pp = cfg.POST_PROCESSING
        self.skip_all = pp.SKIP_ALL
        self.minimal_ioa_thresh = 0.01
        self.class_names = list(cfg.MODEL.ROI_HEADS.CLASS_NAMES)
        self.word_ind = self.class_names.index("word")
        self.detect_threshold = pp.DETECT_THRESHOLD
        self.min_box_dim = pp.MIN_BOX_DIMENSION
        self.merge_ioa_thresh = pp.MERGE_IOA_THRESH
        self.pairs_height_ratio_thresh = pp.PAIRS_HEIGHT_RATIO_THRESH
        self.box_px_padding = pp.BOX_PX_PADDING
        self.max_input_size = cfg.INPUT.MAX_SIZE_TEST
        self.valid_score = pp.VALID_CONFIDENCE
        assert self.valid_score <= self.detect_threshold, \
            "Valid score threshold must be smaller than the other class thresholds, to prevent word-in-word  cases"
        self.max_angle_diff = pp.MAX_ANGLE_DIFF

    # ------------------------------------------------------------------ device path (default)
    text_threshold = None          # set by PostProcessorAcademic
    text_encoder = None

    def _thresholds(self):
        return [float(self.min_box_dim), float(self.valid_score), float(self.detect_threshold),
                float(self.merge_ioa_thresh), float(self.pairs_height_ratio_thresh), float(self.max_angle_diff),
                float(self.minimal_ioa_thresh), float(self.text_threshold if self.text_threshold is not None else 0.0)]

    def process_padded(self, boxes, scores, counts_dev, text, scale_xy, image_sizes, extra=None):
        from ..utils.pipeline import drive
        return drive(self.process_padded_g(boxes, scores, counts_dev, text, scale_xy, image_sizes, extra))

    def process_padded_g(self, boxes, scores, counts_dev, text, scale_xy, image_sizes, extra=None):
        """All images of a step in ONE kernel (ops.native.postprocess_words) + one read of the compact
        results.  boxes [N,K,5], scores [N,K], text [N,K,T,C]|None (None: no text filter), scale_xy [N,2]|None.
        `extra`: dict name -> padded [N,K,...] tensors gathered along with the survivors."""
        from ..ops import native as K
        use_text = text is not None and self.text_threshold is not None
        stop = self.text_encoder.character.index("[s]") if use_text else 1
        out = K.postprocess_words(boxes, scores, counts_dev, text if use_text else None, scale_xy, self._thresholds(), stop)
        from ..utils.pipeline import ReadBack, StepOutput
        if use_text:          # one read-back (generator form: utils/pipeline.py) of counts, characters and lengths
            h_count, h_char, h_len = yield ReadBack(out["count"], out["char"], out["text_len"])
            counts, chars, tlen = h_count.tolist(), h_char.numpy(), h_len.tolist()
        else:
            counts, chars, tlen = (yield ReadBack(out["count"]))[0].tolist(), None, None
        results = StepOutput()
        results.words = out                # padded device tensors of THIS call (distributed.pack_words)
        for n, size in enumerate(image_sizes):
            c = counts[n]
            r = Instances(tuple(size))
            r.pred_boxes = RotatedBoxes(out["boxes"][n, :c])
            r.scores = out["scores"][n, :c]
            r.pred_classes = torch.zeros((c,), dtype=torch.int64, device=boxes.device)
            src = out["src"][n, :c].long()
            if text is not None:
                r.pred_text_prob = text[n][src]
            for name, t in (extra or {}).items():
                if t is not None:
                    r.set(name, t[n][src])
            r.pred_polygons = out["polygons"][n, :c]
            if use_text:
                from .post_processor_academic import strip_special
                r.pred_text_scores = out["text_score"][n, :c]
                r.pred_texts = [strip_special("".join(self.text_encoder.character[int(i)] for i in chars[n, j, :tlen[n][j]]))
                                for j in range(c)]
            results.append(r)
        return results

    def __call__(self, preds: Instances, **kwargs):
        """Reference call convention (one image's Instances in, Instances out) on the device kernel."""
        if self.skip_all:
            self.logger.warning('SKIPPING POST PROCESSING - "SKIP_ALL" is "True" in config file')
            return preds
        n = len(preds)
        dev = preds.pred_boxes.tensor.device
        if n == 0 and not preds.has("pred_text_prob") and self.text_threshold is not None:
            _ = preds.pred_text_prob          # reference behaviour: AttributeError on a detection-less image
        K_ = max(n, 1)
        boxes = torch.zeros((1, K_, 5), dtype=torch.float32, device=dev)
        scores = torch.zeros((1, K_), dtype=torch.float32, device=dev)
        boxes[0, :n] = preds.pred_boxes.tensor
        scores[0, :n] = preds.scores
        text = None
        if preds.has("pred_text_prob"):
            tp = preds.pred_text_prob
            text = torch.zeros((1, K_) + tuple(tp.shape[1:]), dtype=torch.float32, device=dev)
            text[0, :n] = tp
        extra = {}
        if preds.has("orientations"):
            o = torch.zeros((1, K_, 2), dtype=torch.float32, device=dev)
            o[0, :n] = preds.orientations
            extra["orientations"] = o
        # every other per-instance field rides along and is gathered with the survivors, as the reference's
        # `preds[keep]` indexing does (pred_masks / pred_rboxes of the mask branch, user fields)
        boxlike = set()
        for name, v in preds.get_fields().items():
            if name in ("pred_boxes", "scores", "pred_classes", "pred_text_prob", "orientations") or name in extra:
                continue
            t = v.tensor if isinstance(v, RotatedBoxes) else v
            if not isinstance(t, torch.Tensor) or t.shape[0] != n:
                continue
            if isinstance(v, RotatedBoxes):
                boxlike.add(name)
            pad = torch.zeros((1, K_) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
            pad[0, :n] = t
            extra[name] = pad
        from ..ops import native as K
        cnt = K.upload([n], torch.int32, dev)
        out = self.process_padded(boxes, scores, cnt, text, None, [preds.image_size], extra)[0]
        for name in boxlike:
            out.set(name, RotatedBoxes(out.get(name)))
        return out

    # ------------------------------------------------------------------ host restatement (kept as the readable
    # statement of the semantics and as a cross-check of the kernel in tests; ~56 ms per image)
    def host_call(self, preds: Instances):
        if self.skip_all:
            self.logger.warning('SKIPPING POST PROCESSING - "SKIP_ALL" is "True" in config file')
            return preds
        t0 = time.perf_counter()
        preds = self.filter_small_boxes(preds)
        preds = self.post_process_word_preds(preds)
        self.logger.info(f"Merged and removed {len(preds)} Words")
        self.logger.info(f"Post-Process Word Time: {(time.perf_counter() - t0) * 1e3:.1f} ms")
        preds.pred_polygons = self.boxes_to_polygons(preds.pred_boxes.tensor)
        return preds

    def filter_small_boxes(self, preds: Instances):
        if len(preds) == 0:
            return preds
        boxes = preds.pred_boxes.tensor
        return preds[torch.min(boxes[:, 2], boxes[:, 3]) >= self.min_box_dim]

    def post_process_word_preds(self, preds: Instances):
        preds = preds[preds.scores >= self.valid_score]
        preds = self.merge_intersecting_boxes(preds, ioa_threshold=self.merge_ioa_thresh,
                                              pairs_height_ratio_thresh=self.pairs_height_ratio_thresh)
        return preds[preds.scores >= self.detect_threshold]

    def merge_intersecting_boxes(self, preds: Instances, ioa_threshold: float, pairs_height_ratio_thresh: float):
        """Pair logic runs on host copies (<=100 boxes): the reference writes merged boxes back with
        repeated indices (`tensor[pairs[:, 0]] = merged`, :175-176), whose result on the CPU path is
        "last pair wins" — reproduced here with an explicit in-order loop so it is deterministic on any
        device.  The IoA matrix and the 0.99 NMS run on the HIP kernels."""
        if len(preds) == 0:
            return preds
        dev = preds.pred_boxes.tensor.device
        while True:
            boxes = preds.pred_boxes.tensor.detach().cpu().clone()
            scores = preds.scores.detach().cpu()
            ioa = pairwise_ioa_rotated(boxes.to(dev), boxes.to(dev)).cpu()
            pairs = torch.nonzero(ioa.fill_diagonal_(0).triu() >= self.minimal_ioa_thresh)
            if len(pairs) == 0:
                break
            heights, angles = boxes[:, 3], boxes[:, 4]
            adiff = angles[pairs[:, 1]] - angles[pairs[:, 0]]
            adiff = torch.abs((adiff + 180) % 360 - 180)
            similar_angle = (adiff < self.max_angle_diff) | (adiff > (180 - self.max_angle_diff))
            hr = heights[pairs[:, 1]] / heights[pairs[:, 0]]
            similar_height = (pairs_height_ratio_thresh < hr) & (hr < (1 / (pairs_height_ratio_thresh + 1e-6)))
            valid_score = torch.min(scores[pairs[:, 0]], scores[pairs[:, 1]]) >= self.valid_score
            ioa_mask = ioa[pairs[:, 0], pairs[:, 1]] >= ioa_threshold
            combined = valid_score & similar_height & ioa_mask & similar_angle
            if (~combined).all():
                break
            vp = pairs[combined]
            merged = self._merge_rotated_boxes(boxes[vp[:, 0]], boxes[vp[:, 1]], scores[vp[:, 0]], scores[vp[:, 1]])
            for k in range(len(vp)):                       # tensor[vp[:, 0]] = merged   (last wins)
                boxes[vp[k, 0]] = merged[k]
            for k in range(len(vp)):                       # tensor[vp[:, 1]] = merged.clone()
                boxes[vp[k, 1]] = merged[k]
            preds.pred_boxes.tensor.copy_(boxes.to(dev))
            keep = nms_rotated(preds.pred_boxes.tensor, preds.scores, iou_threshold=0.99)
            preds = preds[keep]
        return preds

    @classmethod
    def _merge_rotated_boxes(cls, boxes1, boxes2, scores1=None, scores2=None) -> torch.Tensor:
        assert len(boxes1) == len(boxes2), "We only combine pairs of boxes, please insert same boxes lengths"
        p1, p2 = cls.boxes_to_polygons(boxes1), cls.boxes_to_polygons(boxes2)
        a1 = boxes1[:, 4] * np.pi / 180
        a2 = boxes2[:, 4] * np.pi / 180
        if scores1 is not None and scores2 is not None:
            merged_angle = torch.where(scores1 >= scores2, a1, a2)          # radians (reference quirk)
        else:
            merged_angle = torch.atan2(torch.sin(a1) + torch.sin(a2), torch.cos(a1) + torch.cos(a2)) * 180 / np.pi
        return cls.polygons_to_rotated_boxes(torch.hstack((p1, p2)), orientations=merged_angle)

    @staticmethod
    def boxes_to_polygons(boxes: torch.Tensor) -> torch.Tensor:
        n = len(boxes)
        if n == 0:
            return torch.tensor([]).reshape((0, 4, 2)).to(dtype=boxes.dtype, device=boxes.device)
        cx, cy, w, h, a = boxes.T
        t = (-a / 180) * np.pi
        poly = torch.zeros((n, 4, 2)).to(dtype=boxes.dtype, device=boxes.device)
        s, c = torch.sin(t), torch.cos(t)
        poly[:, 0, 0] = cx + (h * s - w * c) / 2
        poly[:, 1, 0] = cx + (h * s + w * c) / 2
        poly[:, 2, 0] = cx - (h * s - w * c) / 2
        poly[:, 3, 0] = cx - (h * s + w * c) / 2
        poly[:, 0, 1] = cy - (h * c + w * s) / 2
        poly[:, 1, 1] = cy - (h * c - w * s) / 2
        poly[:, 2, 1] = cy + (h * c + w * s) / 2
        poly[:, 3, 1] = cy + (h * c - w * s) / 2
        return poly

    @staticmethod
    def polygons_to_rotated_boxes(polygons: torch.Tensor, orientations: torch.Tensor = None) -> torch.Tensor:
        np_polygons = polygons.detach().cpu().numpy()
        orient = None if orientations is None else orientations.detach().cpu().numpy()
        out = torch.zeros((len(polygons), 5))
        for i, polygon in enumerate(np_polygons):
            center, shape, angle = min_area_rect(np.array(polygon))
            angle = 90 - angle
            diff = (float(orient[i]) - angle) if orient is not None else 0.0
            diff = (diff + 180) % 360 - 180
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
            angle = (angle + 180) % 360 - 180
            out[i] = torch.tensor([center[0], center[1], width, height, angle])
        return out.to(device=polygons.device, dtype=polygons.dtype)
