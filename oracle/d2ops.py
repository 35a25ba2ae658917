"""ORACLE (test infrastructure, NOT product code).

CPU restatement of the detectron2 v0.6 pieces of the GLASS inference path that are NOT
in /root/reference (detectron2==0.6 is an un-vendored dependency, reference README.md:36).
Native ops are in oracle/d2_ops.c (compiled by `oracle.build()`), the tensor-level
helpers below follow [d2-recall]:
  modeling/anchor_generator.py (RotatedAnchorGenerator), modeling/box_regression.py
  (Box2BoxTransformRotated), layers/nms.py (batched_nms_rotated), modeling/poolers.py
  (ROIPooler/assign_boxes_to_levels), modeling/proposal_generator/rrpn.py
  (find_top_rrpn_proposals), structures/rotated_boxes.py (clip/nonempty).
Reference call sites: glass/modeling/meta_arch/glass_rcnn.py:82-92,
glass/modeling/fusion/recognizers_hybrid_head.py:188-205,320,453-500,550,556,
glass/modeling/roi_heads/rotated_fast_rcnn.py:112-113,131,342.

PARITY UNPINNED for these ops by the reference itself (it has no tests/fixtures); they are
pinned by analytic known-answer tests (tests/test_oracle_d2ops.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from typing import List, Sequence, Tuple

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libd2oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "d2_ops.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", _SO, src, "-lm"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        L.d2o_single_box_iou_rotated.restype = ctypes.c_float
        L.d2o_single_box_iou_rotated.argtypes = [fp, fp]
        L.d2o_rotated_vertices.argtypes = [fp, fp]
        L.d2o_pairwise_iou_rotated.argtypes = [fp, ctypes.c_int, fp, ctypes.c_int, fp]
        L.d2o_nms_rotated.restype = ctypes.c_int
        L.d2o_nms_rotated.argtypes = [fp, fp, ctypes.c_int, ctypes.c_float,
                                      ctypes.POINTER(ctypes.c_int64)]
        L.d2o_roi_align_rotated.argtypes = [fp] + [ctypes.c_int] * 4 + [fp, ctypes.c_int, ctypes.c_float,
                                                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, fp]
        _lib = L
    return _lib


def _f32(t) -> np.ndarray:
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu().numpy()
    return np.ascontiguousarray(t, dtype=np.float32)


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


# ---------------------------------------------------------------- native wrappers
def rotated_vertices(box) -> np.ndarray:
    """(cx, cy, w, h, angle_deg) -> [4, 2] vertices in d2's order (box_iou_rotated_utils.h get_rotated_vertices)"""
    b = _f32(box).reshape(5)
    out = np.zeros((8,), dtype=np.float32)
    lib().d2o_rotated_vertices(_p(b), _p(out))
    return out.reshape(4, 2)


def pairwise_iou_rotated(b1, b2) -> torch.Tensor:
    a, b = _f32(b1).reshape(-1, 5), _f32(b2).reshape(-1, 5)
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    if out.size:
        lib().d2o_pairwise_iou_rotated(_p(a), a.shape[0], _p(b), b.shape[0], _p(out))
    return torch.from_numpy(out)


def nms_rotated(boxes, scores, thr: float) -> torch.Tensor:
    b, s = _f32(boxes).reshape(-1, 5), _f32(scores).reshape(-1)
    keep = np.zeros((b.shape[0],), dtype=np.int64)
    n = lib().d2o_nms_rotated(_p(b), _p(s), b.shape[0], float(thr),
                              keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return torch.from_numpy(keep[:n].copy())


def batched_nms_rotated(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, thr: float) -> torch.Tensor:
    """d2 layers/nms.py: shift each category by (max_c - min_c + 1) then plain NMS."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    boxes = boxes.float()
    max_c = (torch.max(boxes[:, 0], boxes[:, 1]) + torch.max(boxes[:, 2], boxes[:, 3]) / 2).max()
    min_c = (torch.min(boxes[:, 0], boxes[:, 1]) - torch.max(boxes[:, 2], boxes[:, 3]) / 2).min()
    offsets = idxs.to(boxes) * (max_c - min_c + 1)
    b = boxes.clone()
    b[:, :2] += offsets[:, None]
    return nms_rotated(b, scores, thr)


def roi_align_rotated(x: torch.Tensor, rois: torch.Tensor, out_size: Tuple[int, int], spatial_scale: float,
                      sampling_ratio: int) -> torch.Tensor:
    """x: (N,C,H,W); rois: (R,6) = (batch_idx,cx,cy,w,h,angle)."""
    xn, rn = _f32(x), _f32(rois).reshape(-1, 6)
    N, C, H, W = xn.shape
    R = rn.shape[0]
    out = np.zeros((R, C, out_size[0], out_size[1]), dtype=np.float32)
    if R:
        lib().d2o_roi_align_rotated(_p(xn), N, C, H, W, _p(rn), R, float(spatial_scale), out_size[0], out_size[1],
                                    int(sampling_ratio), _p(out))
    return torch.from_numpy(out)


# ---------------------------------------------------------------- tensor-level helpers
def rotated_cell_anchors(size: float, aspect_ratios: Sequence[float], angles: Sequence[float]) -> torch.Tensor:
    out = []
    area = size ** 2.0
    for r in aspect_ratios:
        w = math.sqrt(area / r)
        h = r * w
        out.extend([0, 0, w, h, a] for a in angles)
    return torch.tensor(out, dtype=torch.float32)


def rotated_grid_anchors(H: int, W: int, stride: int, cell: torch.Tensor, offset: float = 0.0) -> torch.Tensor:
    sx = torch.arange(offset * stride, W * stride, step=stride, dtype=torch.float32)
    sy = torch.arange(offset * stride, H * stride, step=stride, dtype=torch.float32)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    xx, yy = xx.reshape(-1), yy.reshape(-1)
    z = torch.zeros_like(xx)
    shifts = torch.stack((xx, yy, z, z, z), dim=1)
    return (shifts.view(-1, 1, 5) + cell.view(1, -1, 5)).reshape(-1, 5)


SCALE_CLAMP = math.log(1000.0 / 16)


def apply_deltas_rotated(deltas: torch.Tensor, boxes: torch.Tensor, weights: Sequence[float]) -> torch.Tensor:
    """Box2BoxTransformRotated.apply_deltas: deltas (N, k*5), boxes (N,5)."""
    assert deltas.shape[1] % 5 == 0 and boxes.shape[1] == 5
    boxes = boxes.to(deltas.dtype).unsqueeze(2)
    ctr_x, ctr_y, widths, heights, angles = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3], boxes[:, 4]
    wx, wy, ww, wh, wa = weights
    dx = deltas[:, 0::5] / wx
    dy = deltas[:, 1::5] / wy
    dw = deltas[:, 2::5] / ww
    dh = deltas[:, 3::5] / wh
    da = deltas[:, 4::5] / wa
    dw = torch.clamp(dw, max=SCALE_CLAMP)
    dh = torch.clamp(dh, max=SCALE_CLAMP)
    pred = torch.zeros_like(deltas)
    pred[:, 0::5] = dx * widths + ctr_x
    pred[:, 1::5] = dy * heights + ctr_y
    pred[:, 2::5] = torch.exp(dw) * widths
    pred[:, 3::5] = torch.exp(dh) * heights
    pa = da * 180.0 / math.pi + angles
    pred[:, 4::5] = (pa + 180.0) % 360.0 - 180.0
    return pred


def clip_rotated_(boxes: torch.Tensor, hw: Tuple[int, int], clip_angle_threshold: float = 1.0) -> torch.Tensor:
    """RotatedBoxes.clip in place on an (N,5) tensor."""
    h, w = hw
    boxes[:, 4] = (boxes[:, 4] + 180.0) % 360.0 - 180.0
    idx = torch.where(torch.abs(boxes[:, 4]) <= clip_angle_threshold)[0]
    x1 = boxes[idx, 0] - boxes[idx, 2] / 2.0
    y1 = boxes[idx, 1] - boxes[idx, 3] / 2.0
    x2 = boxes[idx, 0] + boxes[idx, 2] / 2.0
    y2 = boxes[idx, 1] + boxes[idx, 3] / 2.0
    x1.clamp_(min=0, max=w); y1.clamp_(min=0, max=h); x2.clamp_(min=0, max=w); y2.clamp_(min=0, max=h)
    boxes[idx, 0] = (x1 + x2) / 2.0
    boxes[idx, 1] = (y1 + y2) / 2.0
    boxes[idx, 2] = torch.min(boxes[idx, 2], x2 - x1)
    boxes[idx, 3] = torch.min(boxes[idx, 3], y2 - y1)
    return boxes


def assign_boxes_to_levels(boxes: torch.Tensor, min_level: int, max_level: int, canonical_box_size: int = 224,
                           canonical_level: int = 4) -> torch.Tensor:
    box_sizes = torch.sqrt(boxes[:, 2] * boxes[:, 3])
    lvl = torch.floor(canonical_level + torch.log2(box_sizes / canonical_box_size + 1e-8))
    lvl = torch.clamp(lvl, min=min_level, max=max_level)
    return lvl.to(torch.int64) - min_level


def roi_pooler(features: List[torch.Tensor], scales: Sequence[float], boxes_per_image: List[torch.Tensor],
               out_size: Tuple[int, int], sampling_ratio: int) -> torch.Tensor:
    """ROIPooler(pooler_type='ROIAlignRotated').forward; features NCHW per level."""
    rois = []
    for i, b in enumerate(boxes_per_image):
        rois.append(torch.cat([torch.full((len(b), 1), float(i)), b.float()], dim=1))
    rois = torch.cat(rois, dim=0) if rois else torch.zeros((0, 6))
    C = features[0].shape[1]
    if len(features) == 1:
        return roi_align_rotated(features[0], rois, out_size, scales[0], sampling_ratio)
    min_level = int(round(-math.log2(scales[0])))
    max_level = int(round(-math.log2(scales[-1])))
    lv = assign_boxes_to_levels(rois[:, 1:], min_level, max_level)
    out = torch.zeros((rois.shape[0], C, out_size[0], out_size[1]), dtype=torch.float32)
    for level, (f, s) in enumerate(zip(features, scales)):
        inds = torch.nonzero(lv == level).squeeze(1)
        if len(inds):
            out[inds] = roi_align_rotated(f, rois[inds], out_size, s, sampling_ratio)
    return out


def find_top_rrpn_proposals(proposals: List[torch.Tensor], logits: List[torch.Tensor],
                            image_sizes: List[Tuple[int, int]], nms_thresh: float, pre_nms_topk: int,
                            post_nms_topk: int, min_box_size: float = 0.0):
    """proposals[l]: (N, Hi*Wi*A, 5); logits[l]: (N, Hi*Wi*A). Returns per image (boxes, logits)."""
    num_images = len(image_sizes)
    topk_scores, topk_props, level_ids = [], [], []
    batch_idx = torch.arange(num_images)
    for level_id, (p, lg) in enumerate(zip(proposals, logits)):
        Hi_Wi_A = lg.shape[1]
        k = min(pre_nms_topk, Hi_Wi_A)
        lg_s, idx = lg.sort(descending=True, dim=1, stable=True)
        topk_scores.append(lg_s[batch_idx, :k])
        topk_props.append(p[batch_idx[:, None], idx[batch_idx, :k]])
        level_ids.append(torch.full((k,), level_id, dtype=torch.int64))
    topk_scores = torch.cat(topk_scores, dim=1)
    topk_props = torch.cat(topk_props, dim=1)
    level_ids = torch.cat(level_ids, dim=0)
    results = []
    for n, image_size in enumerate(image_sizes):
        boxes = topk_props[n].clone()
        scores = topk_scores[n]
        lvl = level_ids
        valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores)
        if not valid.all():
            boxes, scores, lvl = boxes[valid], scores[valid], lvl[valid]
        clip_rotated_(boxes, image_size)
        keep = (boxes[:, 2] > min_box_size) & (boxes[:, 3] > min_box_size)
        if keep.sum().item() != len(boxes):
            boxes, scores, lvl = boxes[keep], scores[keep], lvl[keep]
        keep = batched_nms_rotated(boxes, scores, lvl, nms_thresh)[:post_nms_topk]
        results.append((boxes[keep], scores[keep]))
    return results
