"""Recognizer CNNs (`CNN_V1_1`, and the unused `CNN_V2_1`) on the HIP conv kernels.

Mirrors reference glass/modeling/recognition/recognizer_backbone.py:34-81:
x1 = ReLU(BN(conv[2,1] s[2,1])), out = ReLU(BN(conv3x3(x1))) + x1 — the trailing add is the
second conv's epilogue (relu mode 2 = ReLU before the residual add).
"""
from __future__ import annotations

import torch

from ...utils.module import InferenceModule

from ...checkpoint import fold_conv
from ...ops import native as K
from ...utils.registry import Registry

RECOGNIZER_BACKBONE_REGISTRY = Registry("RECOGNIZER_BACKBONE")


def build_recognizer_backbonev2(cfg, input_shape):
    return RECOGNIZER_BACKBONE_REGISTRY.get(cfg.MODEL.ROI_RECOGNIZER_HEAD.RECOGNIZER_HEAD.BACKBONE.NAME)(cfg, input_shape)


@RECOGNIZER_BACKBONE_REGISTRY.register()
class CNN_V1_1(InferenceModule):
    def __init__(self, cfg, input_shape):
        super().__init__()
        self.channels = input_shape.channels
        self.conv_norm = cfg.MODEL.ROI_RECOGNIZER_HEAD.NORM
        self.w = {}

    def import_weights(self, sd, device, prefix: str) -> None:
        self.w = {"conv1": fold_conv(sd, prefix + "conv1", prefix + "conv1.norm", device),
                  "conv2": fold_conv(sd, prefix + "conv2", prefix + "conv2.norm", device)}

    def forward_nhwc(self, x: torch.Tensor) -> torch.Tensor:
        x1 = K.conv2d_nhwc(x, *self.w["conv1"], stride=(2, 1), relu=1)
        return K.conv2d_nhwc(x1, *self.w["conv2"], padding=1, relu=2, residual=x1, res_mode=1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from ..backbone.resnet_fpn import as_nhwc
        return self.forward_nhwc(as_nhwc(x)).permute(0, 3, 1, 2)


@RECOGNIZER_BACKBONE_REGISTRY.register()
class CNN_V2_1(CNN_V1_1):
    """reference recognizer_backbone.py:84-146 (no shipped config selects it): x1 = ReLU(BN(conv[2,1] s[2,1])),
    x12 = ReLU(BN(conv3x3(x1))) + x1, out = x12 + ReLU(BN(conv3x3(x12))) - both adds are conv epilogues."""

    def import_weights(self, sd, device, prefix: str) -> None:
        self.w = {k: fold_conv(sd, prefix + k, prefix + k + ".norm", device) for k in ("conv1", "conv2", "conv3")}

    def forward_nhwc(self, x: torch.Tensor) -> torch.Tensor:
        x1 = K.conv2d_nhwc(x, *self.w["conv1"], stride=(2, 1), relu=1)
        x12 = K.conv2d_nhwc(x1, *self.w["conv2"], padding=1, relu=2, residual=x1, res_mode=1)
        return K.conv2d_nhwc(x12, *self.w["conv3"], padding=1, relu=2, residual=x12, res_mode=1)
