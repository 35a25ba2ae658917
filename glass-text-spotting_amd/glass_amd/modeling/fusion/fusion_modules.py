"""Global-to-local fusion modules on HIP kernels.

Mirrors reference glass/modeling/fusion/fusion_modules.py: `MultiAspectGCAttention` (:22-157,
pooling 'att', fusion 'channel_add') and `P2P3Fusion` (:250-286), plus the registry/builder
(:10-18), and the three variants no shipped config selects - `SimpleAttention`, `LocalOnly`, `Conv1x1`
(:160-247).  The head hands every fusion module the channel-INTERLEAVED concat (local on even,
global on odd channels - free for MultiAspectGCAttention's `order`); the variants consume the reference's
plain cat(local, global), so their weight columns are permuted to the interleaved order once at load.
"""
from __future__ import annotations

from typing import Dict

import torch

from ...utils.module import InferenceModule

from ...checkpoint import conv_weight, dev, fold_conv
from ...ops import native as K
from ...utils.registry import Registry

HYBRID_FEATURE_FUSION_REGISTRY = Registry("HYBRID_FEATURE_FUSION")


def build_hybrid_feature_fusion(cfg, input_shape):
    name = cfg.MODEL.HYBRID_FUSION.NAME
    return HYBRID_FEATURE_FUSION_REGISTRY.get(name)(cfg, input_shape)


@HYBRID_FEATURE_FUSION_REGISTRY.register()
class MultiAspectGCAttention(InferenceModule):
    def __init__(self, cfg, input_shape):
        super().__init__()
        self.inplanes = cfg.MODEL.LOCAL_FEATURE_EXTRACTOR.NUM_FEATURES + input_shape.channels
        self.ratio = cfg.MODEL.HYBRID_FUSION.RATIO
        self.headers = cfg.MODEL.HYBRID_FUSION.HEADERS
        self.outplane = cfg.MODEL.HYBRID_FUSION.NUM_FEATURES
        self.fusion_type = cfg.MODEL.HYBRID_FUSION.FUSION_TYPE
        assert self.fusion_type == "channel_add", "only channel_add is built (all reference configs)"
        self.planes = int(self.inplanes * self.ratio)
        self.w: Dict[str, torch.Tensor] = {}

    def import_weights(self, sd, device, prefix: str) -> None:
        w = {}
        w["w_mask"] = dev(sd[prefix + "conv_mask.weight"].reshape(-1), device)
        w["b_mask"] = dev(sd[prefix + "conv_mask.bias"].reshape(-1), device)
        w["w1"] = dev(sd[prefix + "channel_add_conv.0.weight"].reshape(self.planes, self.inplanes), device)
        w["b1"] = dev(sd[prefix + "channel_add_conv.0.bias"], device)
        w["ln_g"] = dev(sd[prefix + "channel_add_conv.1.weight"].reshape(-1), device)
        w["ln_b"] = dev(sd[prefix + "channel_add_conv.1.bias"].reshape(-1), device)
        w["w2"] = dev(sd[prefix + "channel_add_conv.3.weight"].reshape(self.inplanes, self.planes), device)
        w["b2"] = dev(sd[prefix + "channel_add_conv.3.bias"], device)
        w["out"] = fold_conv(sd, prefix + "out", None, device)
        self.w = w

    def forward_interleaved(self, x: torch.Tensor) -> torch.Tensor:
        """x: [R,8,32,512] NHWC, channels already interleaved (x[:, order] of the reference,
        fusion_modules.py:50-53,131); modified in place.  Returns [R,8,32,256]."""
        w = self.w
        K.gc_attention_inplace(x, self.headers, w["w_mask"], w["b_mask"], w["w1"], w["b1"], w["ln_g"], w["ln_b"],
                               w["w2"], w["b2"])
        return K.conv2d_nhwc(x, *w["out"], padding=1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """reference convention: logical NCHW cat(local, global) [R,512,8,32] -> [R,256,8,32]."""
        C = x.shape[1]
        order = torch.zeros(C, dtype=torch.long)
        order[0::2] = torch.arange(C)[: C // 2]
        order[1::2] = torch.arange(C)[C // 2:]
        xi = x[:, order.to(x.device)].permute(0, 2, 3, 1).contiguous()
        return self.forward_interleaved(xi).permute(0, 3, 1, 2)


def _interleaved_position(c_total: int) -> torch.Tensor:
    """position in the interleaved layout of channel k of cat(local, global): local k -> 2k, global k -> 2k+1."""
    half = c_total // 2
    k = torch.arange(c_total)
    return torch.where(k < half, 2 * k, 2 * (k - half) + 1)


class _CatFusionBase(InferenceModule):
    """shared plumbing of the cat(local, global) variants"""

    def __init__(self, cfg, input_shape):
        super().__init__()
        self.local_ch = cfg.MODEL.LOCAL_FEATURE_EXTRACTOR.NUM_FEATURES
        self.global_ch = input_shape.channels
        self.in_channels = self.local_ch + self.global_ch
        self.out_channels = cfg.MODEL.HYBRID_FUSION.NUM_FEATURES
        assert self.local_ch == self.global_ch, "the interleaved concat assumes equally wide local / global features"
        self.w: Dict[str, torch.Tensor] = {}

    def _cols_to_interleaved(self, w2d: torch.Tensor) -> torch.Tensor:
        out = torch.empty_like(w2d)
        out[:, _interleaved_position(self.in_channels)] = w2d
        return out

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """reference convention: logical NCHW cat(local, global) -> NCHW."""
        C = x.shape[1]
        order = torch.zeros(C, dtype=torch.long)
        order[0::2] = torch.arange(C)[: C // 2]
        order[1::2] = torch.arange(C)[C // 2:]
        xi = x[:, order.to(x.device)].permute(0, 2, 3, 1).contiguous()
        return self.forward_interleaved(xi).permute(0, 3, 1, 2)


@HYBRID_FEATURE_FUSION_REGISTRY.register()
class Conv1x1(_CatFusionBase):
    """1x1 conv (no bias) over the concat, reference fusion_modules.py:222-247."""

    def import_weights(self, sd, device, prefix: str) -> None:
        w = sd[prefix + "conv.weight"].float().reshape(self.out_channels, self.in_channels)
        self.w = {"conv": conv_weight(self._cols_to_interleaved(w).reshape(self.out_channels, 1, 1, self.in_channels), device)}

    def forward_interleaved(self, x: torch.Tensor) -> torch.Tensor:
        return K.conv2d_nhwc(x, self.w["conv"], None)


@HYBRID_FEATURE_FUSION_REGISTRY.register()
class LocalOnly(_CatFusionBase):
    """keeps the local half of the concat, reference fusion_modules.py:189-219."""

    def __init__(self, cfg, input_shape):
        super().__init__(cfg, input_shape)
        assert cfg.MODEL.HYBRID_FUSION.NUM_FEATURES == self.local_ch

    def import_weights(self, sd, device, prefix: str) -> None:
        self.w = {}

    def forward_interleaved(self, x: torch.Tensor) -> torch.Tensor:
        assert x.shape[-1] == self.in_channels
        return x[..., 0::2].contiguous()


@HYBRID_FEATURE_FUSION_REGISTRY.register()
class SimpleAttention(_CatFusionBase):
    """x <- linear(x) * x over the channel axis (no bias), then a 1x1 conv (no bias), reference
    fusion_modules.py:160-186.  The linear is a 1x1 conv on the MFMA kernel; both its rows and its columns move to
    the interleaved channel order so that the gate lines up with x."""

    def import_weights(self, sd, device, prefix: str) -> None:
        C = self.in_channels
        wl = self._cols_to_interleaved(sd[prefix + "linear.weight"].float().reshape(C, C))
        wl_rows = torch.empty_like(wl)
        wl_rows[_interleaved_position(C)] = wl
        wc = self._cols_to_interleaved(sd[prefix + "conv.weight"].float().reshape(self.out_channels, C))
        self.w = {"linear": conv_weight(wl_rows.reshape(C, 1, 1, C), device), "conv": conv_weight(wc.reshape(self.out_channels, 1, 1, C), device)}

    def forward_interleaved(self, x: torch.Tensor) -> torch.Tensor:
        g = K.conv2d_nhwc(x, self.w["linear"], None)
        K.mul_(g, x)
        return K.conv2d_nhwc(g, self.w["conv"], None)


class P2P3Fusion(InferenceModule):
    """conv1x1(p2) + nearest_up2(conv1x1(p3)) — the upsample+add is the first conv's epilogue."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.w: Dict[str, tuple] = {}

    def import_weights(self, sd, device, prefix: str) -> None:
        self.w = {"conv1": fold_conv(sd, prefix + "conv1", None, device),
                  "conv2": fold_conv(sd, prefix + "conv2", None, device)}
        # [W1 | W2] as ONE 1x1 layer over cat(pool(p2), pool(up2(p3))) for the pooled form below (no bias in either conv)
        w1, w2 = self.w["conv1"][0].raw, self.w["conv2"][0].raw
        self.w["cat"] = conv_weight(torch.cat([w1, w2], dim=3), device) if self.w["conv1"][1] is None and self.w["conv2"][1] is None else None
        self.out_channels = int(w1.shape[0])

    def forward_nhwc(self, p2: torch.Tensor, p3: torch.Tensor) -> torch.Tensor:
        t = K.conv2d_nhwc(p3, *self.w["conv2"])
        return K.conv2d_nhwc(p2, *self.w["conv1"], residual=t, res_mode=2)

    def can_pool(self, p2: torch.Tensor, p3: torch.Tensor, num_rois: int, bins: int) -> bool:
        """pool-then-project pays while the pooled bins are fewer than the map's pixels (with a margin for the second pooling
        pass); fp32 routing only (the fp16 modes' oracle emulates the whole-map order of roundings)"""
        rt = K.routing_of(self.w["conv1"][0])
        return (self.w.get("cat") is not None and rt.precision == "fp32" and rt.pooled_fusion and p2.dtype == torch.float32 and
                p3.dtype == torch.float32 and p2.shape[1] == 2 * p3.shape[1] and p2.shape[2] == 2 * p3.shape[2] and num_rois > 0 and
                1.25 * num_rois * bins * 2 < p2.shape[0] * (p2.shape[1] * p2.shape[2] + p3.shape[1] * p3.shape[2]))

    def pooled_nhwc(self, p2: torch.Tensor, p3: torch.Tensor, scale: float, boxes: torch.Tensor, roi_image: torch.Tensor,
                    out_size, sampling_ratio: int, out: torch.Tensor, out_coff: int, out_cstride: int) -> torch.Tensor:
        """recognizer_pooler(P2P3Fusion(p2, p3)) (reference recognizers_hybrid_head.py:548-550 + fusion_modules.py:281-286)
        with the two stages swapped: ROIAlignRotated is a fixed linear combination of feature-map pixels and the fusion is
        conv1x1(p2) + nearest_up2(conv1x1(p3)) with no bias, norm or activation, so
            pool(W1 p2 + up2(W2 p3)) = W1 pool(p2) + W2 pool(up2(p3))      (exactly, in real arithmetic)
        and the 1x1 convolutions run on R x PH x PW pooled bins (65 K rows for 256 RoIs) instead of N x (H2 W2 + H3 W3) pixels
        (655 K for 8 images): 17 instead of 86 GFLOP per step, and the fused [N,256,256,256] map is never written.  p3 is pooled
        THROUGH the nearest upsampling (glass_roi_align_rotated_up2: the taps of the upsampled map, read at (y >> 1, x >> 1))."""
        R, C = boxes.shape[0], p2.shape[-1]
        cat = torch.empty((R, out_size[0], out_size[1], 2 * C), dtype=torch.float32, device=p2.device)
        K.roi_align_rotated([p2], [scale], boxes, roi_image, out_size, sampling_ratio, out=cat, out_coff=0)
        K.roi_align_rotated([p3], [scale], boxes, roi_image, out_size, sampling_ratio, out=cat, out_coff=C, up2=True, channels=C)
        return K.conv2d_nhwc(cat, self.w["cat"], None, out=out, out_coff=out_coff, out_cstride=out_cstride)

    def forward(self, x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
        from ..backbone.resnet_fpn import as_nhwc
        return self.forward_nhwc(as_nhwc(x1), as_nhwc(x2)).permute(0, 3, 1, 2)
