"""Rotated-box helpers of reference glass/structures/boxes.py:23-64 (`pairwise_ioa_rotated`,
`box_to_rbox`, `rbox_to_box`); the IoU matrix itself comes from the HIP kernel."""
from __future__ import annotations

import torch

from ..ops import native as K


def pairwise_iou_rotated(boxes1: torch.Tensor, boxes2: torch.Tensor) -> torch.Tensor:
    return K.pairwise_iou_rotated(boxes1.float().contiguous(), boxes2.float().contiguous())


def pairwise_ioa_rotated(boxes1_tensor: torch.Tensor, boxes2_tensor: torch.Tensor) -> torch.Tensor:
    """Intersection over the smaller area, (M,N); same algebra as the reference (:33-48)."""
    assert boxes1_tensor.shape[1] == 5 and boxes2_tensor.shape[1] == 5, "Input tensors don't describe rotated boxes"
    iou = pairwise_iou_rotated(boxes1_tensor, boxes2_tensor)
    area1 = boxes1_tensor[:, 2] * boxes1_tensor[:, 3]
    area2 = boxes2_tensor[:, 2] * boxes2_tensor[:, 3]
    a1 = area1.repeat(len(boxes2_tensor), 1).T
    a2 = area2.repeat(len(boxes1_tensor), 1)
    intersection = (a1 + a2) * iou / (1 + iou)
    return intersection / torch.min(a1, a2)


def nms_rotated(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """d2 layers.nms.nms_rotated on the HIP NMS kernel (single image, no clipping, no score filter)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    if n > 1024:
        raise ValueError("nms_rotated: more than 1024 boxes is not supported on this path")
    hw = torch.zeros((1, 2), dtype=torch.int32, device=boxes.device)
    _, _, idx, cnt = K.rotated_nms_select(boxes.float().contiguous().view(1, n, 5), scores.float().contiguous().view(1, n),
                                          None, None, hw, float("-inf"), float(iou_threshold), n, 0)
    return idx[0, : int(cnt[0].item())].long()


def box_to_rbox(box_tensor: torch.Tensor) -> torch.Tensor:
    """XYXY -> (cx,cy,w,h,0), computed in float64 like the reference (:51-59)."""
    dt = box_tensor.dtype
    arr = box_tensor.double().clone()
    arr[:, 2] -= arr[:, 0]
    arr[:, 3] -= arr[:, 1]
    arr[:, 0] += arr[:, 2] / 2.0
    arr[:, 1] += arr[:, 3] / 2.0
    angles = torch.zeros((arr.shape[0], 1), dtype=arr.dtype, device=arr.device)
    return torch.cat((arr, angles), dim=1).to(dtype=dt)
