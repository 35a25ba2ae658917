"""`RotatedMaskRCNNConvUpsampleHead`, inference path, on the HIP conv kernels.

Reference: glass/modeling/roi_heads/rotated_mask_head.py:409-442 (a d2 `MaskRCNNConvUpsampleHead` whose eval
forward is `layers(x)` + `mask_rcnn_inference`); layer stack per the reference YAMLs (NUM_CONV 4, CONV_DIM 256,
NORM '', one class) [d2-recall: detectron2/modeling/roi_heads/mask_head.py]:
4 x [conv3x3 + ReLU] -> ConvTranspose2d(256, 256, k=2, s=2) + ReLU -> conv1x1 -> sigmoid.

ConvTranspose2d with kernel = stride = 2 has no overlapping taps: output pixel (2h+a, 2w+b) is a 1x1 convolution
of input pixel (h, w) with the weight slice [:, :, a, b].  It runs as ONE 1x1 conv to 4*C channels (rows ordered
(a*2+b)*C + c, ReLU fused) followed by the 2x pixel shuffle kernel.
"""
from __future__ import annotations

import torch

from ...checkpoint import conv_weight, dev
from ...ops import native as K
from ...structures.core import ShapeSpec
from ...utils.module import InferenceModule
from ...utils.registry import ROI_MASK_HEAD_REGISTRY


@ROI_MASK_HEAD_REGISTRY.register()
class RotatedMaskRCNNConvUpsampleHead(InferenceModule):
    def __init__(self, cfg, input_shape: ShapeSpec):
        super().__init__()
        mc = cfg.MODEL.ROI_MASK_HEAD
        assert mc.POOLER_TYPE in ["ROIAlignRotated"], mc.POOLER_TYPE          # rotated_mask_head.py:419-420
        assert not mc.NORM, "normalised mask heads are not built (NORM '' in every reference config)"
        self.num_conv = mc.NUM_CONV
        self.conv_dim = mc.CONV_DIM
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        self.cls_agnostic = bool(mc.CLS_AGNOSTIC_MASK)
        self.in_channels = input_shape.channels
        self.convs, self.deconv, self.predictor = [], None, None

    def import_weights(self, sd, device, prefix: str) -> None:
        self.convs = []
        for k in range(1, self.num_conv + 1):
            w = sd[f"{prefix}mask_fcn{k}.weight"].float().permute(0, 2, 3, 1)          # [Cout,3,3,Cin]
            self.convs.append((conv_weight(w, device), dev(sd[f"{prefix}mask_fcn{k}.bias"], device)))
        wd = sd[prefix + "deconv.weight"].float()                                       # [Cin, Cout, 2, 2]
        cin, cout = wd.shape[0], wd.shape[1]
        w1 = wd.permute(2, 3, 1, 0).reshape(4 * cout, 1, 1, cin)                        # row = (a*2+b)*Cout + co
        b1 = sd[prefix + "deconv.bias"].float().repeat(4)
        self.deconv = (conv_weight(w1, device), dev(b1, device))
        wp = sd[prefix + "predictor.weight"].float().permute(0, 2, 3, 1)                # [classes,1,1,C]
        self.predictor = (conv_weight(wp, device), dev(sd[prefix + "predictor.bias"], device))

    def layers_nhwc(self, pooled: torch.Tensor) -> torch.Tensor:
        """pooled [R,P,P,C] NHWC -> mask logits [R,2P,2P,classes]."""
        x = pooled
        for w, b in self.convs:
            x = K.conv2d_nhwc(x, w, b, padding=1, relu=1)
        x = K.conv2d_nhwc(x, self.deconv[0], self.deconv[1], relu=1)
        x = K.pixel_shuffle2x_nhwc(x)
        return K.conv2d_nhwc(x, self.predictor[0], self.predictor[1])

    def forward_nhwc(self, pooled: torch.Tensor, pred_classes: torch.Tensor = None) -> torch.Tensor:
        """d2 mask_rcnn_inference: sigmoid, then the predicted class' channel (channel 0 when class agnostic or
        single class).  Returns pred_masks [R,1,2P,2P]."""
        R = pooled.shape[0]
        P2 = 2 * pooled.shape[1]
        if R == 0:
            return torch.zeros((0, 1, P2, P2), dtype=torch.float32, device=pooled.device)
        logits = self.layers_nhwc(pooled)                       # [R,2P,2P,classes]
        K.sigmoid_(logits)
        if logits.shape[-1] == 1 or self.cls_agnostic:
            return logits[..., 0].unsqueeze(1)
        idx = pred_classes.long().view(R, 1, 1, 1).expand(R, P2, P2, 1)
        return torch.gather(logits, 3, idx)[..., 0].unsqueeze(1)


def build_mask_head(cfg, input_shape):
    return ROI_MASK_HEAD_REGISTRY.get(cfg.MODEL.ROI_MASK_HEAD.NAME)(cfg, input_shape)
