"""`FastRCNNConvFCHead` (0 conv, NUM_FC fc layers + ReLU) on the MFMA GEMM kernel.

detectron2 v0.6 modeling/roi_heads/box_head.py [d2-recall], selected by reference
configs/glass_pretrain.yaml:89-97 and built at glass/modeling/fusion/recognizers_hybrid_head.py:
207-209.  d2 flattens the pooled NCHW [P,256,7,7] channel-major; the pooled tensor here is
NHWC, so fc1's weight columns are permuted (c,h,w) -> (h,w,c) once at load.
"""
from __future__ import annotations

import torch

from ...utils.module import InferenceModule

from ...checkpoint import conv_weight, dev
from ...ops import native as K
from ...structures.core import ShapeSpec
from ...utils.registry import ROI_BOX_HEAD_REGISTRY


@ROI_BOX_HEAD_REGISTRY.register()
class FastRCNNConvFCHead(InferenceModule):
    def __init__(self, cfg, input_shape: ShapeSpec):
        super().__init__()
        assert cfg.MODEL.ROI_BOX_HEAD.NUM_CONV == 0, "conv layers in the box head are not built"
        self.num_fc = cfg.MODEL.ROI_BOX_HEAD.NUM_FC
        self.fc_dim = cfg.MODEL.ROI_BOX_HEAD.FC_DIM
        self.in_shape = input_shape
        self.fcs = []

    @property
    def output_shape(self) -> ShapeSpec:
        return ShapeSpec(channels=self.fc_dim)

    def import_weights(self, sd, device, prefix: str) -> None:
        self.fcs = []
        C, H, W = self.in_shape.channels, self.in_shape.height, self.in_shape.width
        for i in range(1, self.num_fc + 1):
            w = sd[f"{prefix}fc{i}.weight"].float()
            if i == 1:
                w = w.view(w.shape[0], C, H, W).permute(0, 2, 3, 1).reshape(w.shape[0], -1)
            self.fcs.append((conv_weight(w, device), dev(sd[f"{prefix}fc{i}.bias"], device)))

    def forward_nhwc(self, pooled: torch.Tensor) -> torch.Tensor:
        """pooled [R,7,7,256] NHWC -> [R,fc_dim]."""
        x = pooled.reshape(pooled.shape[0], -1)
        for w, b in self.fcs:
            x = K.linear(x, w, b, relu=1)
        return x

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from ..backbone.resnet_fpn import as_nhwc
        return self.forward_nhwc(as_nhwc(x))


def build_box_head(cfg, input_shape):
    return ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)(cfg, input_shape)
