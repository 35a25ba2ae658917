"""Local-crop feature extractor ("Res34"-style ResNet on 128x128 word crops) on the HIP conv kernel.

Mirrors reference glass/modeling/fusion/local_feature_extraction.py:22-29 (`ResNetFeatureExtractor`
= `ResNet(3, 256, BasicBlock, [1,2,5,3])`, forward :153-188, BasicBlock :290-323) and its
registry/builder (:9-18).  This is the single largest cost of the path (6.38 GMAC per RoI), so
all RoIs of all images of a step run as ONE batch through each layer.
The last conv can write straight into the channel-interleaved fusion input (out/out_coff/
out_cstride), which makes the reference's `torch.cat((local, global), 1)[:, order]`
(recognizers_hybrid_head.py:560, fusion_modules.py:131) free.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from ...utils.module import InferenceModule

from ...checkpoint import fold_conv
from ...ops import native as K
from ...utils.registry import Registry

LOCAL_FEATURE_EXTRACTOR_REGISTRY = Registry("LOCAL_FEATURE_EXTRACTOR")

_LAYERS = ((1, 1), (2, 2), (3, 5), (4, 3))


def build_hybrid_feature_extractor(cfg, input_shape):
    name = cfg.MODEL.LOCAL_FEATURE_EXTRACTOR.NAME
    out_channels = cfg.MODEL.LOCAL_FEATURE_EXTRACTOR.NUM_FEATURES
    return LOCAL_FEATURE_EXTRACTOR_REGISTRY.get(name)(3, out_channels)


@LOCAL_FEATURE_EXTRACTOR_REGISTRY.register()
class ResNetFeatureExtractor(InferenceModule):
    def __init__(self, input_channel, output_channel=512):
        super().__init__()
        assert input_channel == 3
        self.output_channel = output_channel
        self.w: Dict[str, tuple] = {}

    def import_weights(self, sd, device, prefix: str) -> None:
        p = prefix + "ConvNet."
        w = {}
        w["conv0_1"] = fold_conv(sd, p + "conv0_1", p + "bn0_1", device)
        w["conv0_2"] = fold_conv(sd, p + "conv0_2", p + "bn0_2", device)
        for li, nblk in _LAYERS:
            # layer3 / layer4 (and conv3) run behind maxpool3 = MaxPool2d(2, (2, 1), (0, 1)) (reference :123): width W/4 + 1,
            # i.e. 33 for the 128-pixel crops of every config - a 4 k + 1 map: these layers carry the last-column strip weights
            rg = li >= 3
            for b in range(nblk):
                q = f"{p}layer{li}.{b}."
                w[f"l{li}.{b}.conv1"] = fold_conv(sd, q + "conv1", q + "bn1", device, ragged=rg)
                w[f"l{li}.{b}.conv2"] = fold_conv(sd, q + "conv2", q + "bn2", device, ragged=rg)
                if (q + "downsample.0.weight") in sd:
                    w[f"l{li}.{b}.down"] = fold_conv(sd, q + "downsample.0", q + "downsample.1", device)
            if li < 4:
                w[f"conv{li}"] = fold_conv(sd, f"{p}conv{li}", f"{p}bn{li}", device, ragged=rg)
        w["conv4_1"] = fold_conv(sd, p + "conv4_1", p + "bn4_1", device)
        self.w = w

    def forward_nhwc(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, out_coff: int = 0,
                     out_cstride: int = 1) -> torch.Tensor:
        """x: [R,128,128,4] (NHWC4 crops) -> [R,8,32,256] (or into `out`)."""
        w = self.w
        pools = {1: ((2, 2), (2, 2), (0, 0)), 2: ((2, 2), (2, 2), (0, 0)), 3: ((2, 2), (2, 1), (0, 1))}
        fused_stem = x.shape[0] > 0 and w["conv0_1"][1] is not None and w["conv0_2"][1] is not None and \
            K.local_stem_supported(x, w["conv0_1"][0], w["conv0_2"][0])
        if fused_stem:
            # conv0_1 + conv0_2 + maxpool1 in one kernel (csrc/local_stem.hip): the 16- and 32-channel maps stay on the CU
            x = K.local_stem_fused(x, *w["conv0_1"], *w["conv0_2"])
        else:
            x = K.conv2d_nhwc(x, *w["conv0_1"], padding=1, relu=1, out_dtype=K.act_dtype(w["conv0_1"][0]))      # entry of the fp16-storage chain
            x = K.conv2d_nhwc(x, *w["conv0_2"], padding=1, relu=1)
        for li, nblk in _LAYERS:
            if li in pools and x.shape[0] > 0 and not (fused_stem and li == 1):
                x = K.maxpool2d_nhwc(x, *pools[li])
            for b in range(nblk):
                key = f"l{li}.{b}."
                res = K.conv2d_nhwc(x, *w[key + "down"]) if (key + "down") in w else x
                o = K.conv2d_nhwc(x, *w[key + "conv1"], padding=1, relu=1)
                x = K.conv2d_nhwc(o, *w[key + "conv2"], padding=1, relu=1, residual=res, res_mode=1)
            if li < 4:
                x = K.conv2d_nhwc(x, *w[f"conv{li}"], padding=1, relu=1)
        # exit: fp32 (the fusion attention and everything after it read fp32)
        return K.conv2d_nhwc(x, *w["conv4_1"], stride=(2, 1), relu=1, out=out, out_coff=out_coff, out_cstride=out_cstride,
                             out_dtype=torch.float32)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        """reference call convention: logical NCHW [R,3,128,128] in, logical NCHW out."""
        x = input.permute(0, 2, 3, 1)
        if x.shape[-1] == 3:
            x = torch.nn.functional.pad(x, (0, 1))
        return self.forward_nhwc(x.contiguous()).permute(0, 3, 1, 2)


# reference local_feature_extraction.py:33-42: its constructor calls `super(ResNet_FeatureExtractorV2, self)` - a name that
# does not exist (NameError) - so no reference config can select it either
LOCAL_FEATURE_EXTRACTOR_REGISTRY.register_unbuilt(
    "ResNetFeatureExtractorV2", "its reference constructor raises NameError (local_feature_extraction.py:36), no config uses it")
