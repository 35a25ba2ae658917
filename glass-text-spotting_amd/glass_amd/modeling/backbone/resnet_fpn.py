"""ResNet-50 + FPN backbone on the HIP conv kernel (NHWC, BN folded).

Mirrors detectron2 v0.6 `build_resnet_fpn_backbone` as selected by reference
configs/glass_pretrain.yaml:41-54 (R-50, STRIDE_IN_1X1=True, FPN 256 ch with norm,
FUSE_TYPE sum, LastLevelMaxPool) and called at reference
glass/modeling/meta_arch/glass_rcnn.py:83 [d2-recall for the d2-owned structure].

MI355X-first choices: every conv is one launch of the fp32-MFMA implicit-GEMM kernel with
bias/ReLU/residual fused in its epilogue; the FPN top-down `lateral + upsample2x(prev)` is the
lateral conv's epilogue (res_mode 2), so no separate upsample/add pass touches HBM; p6 is a
strided view, not a kernel.
"""
from __future__ import annotations

from typing import Dict

import torch

from ...utils.module import InferenceModule

from ...checkpoint import fold_conv
from ...ops import native as K
from ...structures.core import ShapeSpec
from ...utils.registry import BACKBONE_REGISTRY

_STAGES = (("res2", 3), ("res3", 4), ("res4", 6), ("res5", 3))


def as_nchw_view(x_nhwc: torch.Tensor) -> torch.Tensor:
    """Logical NCHW view (channels_last strides) of an NHWC tensor — the layout the reference's
    module boundaries speak, without moving memory."""
    return x_nhwc.permute(0, 3, 1, 2)


def as_nhwc(x: torch.Tensor) -> torch.Tensor:
    """NHWC-contiguous tensor for a logical NCHW input; free if x is already channels_last."""
    v = x.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


class ResNetFPN(InferenceModule):
    size_divisibility = 32

    def __init__(self, cfg):
        super().__init__()
        assert cfg.MODEL.RESNETS.DEPTH == 50, "only R-50 is built"
        assert cfg.MODEL.RESNETS.STRIDE_IN_1X1, "STRIDE_IN_1X1=False is not built"
        assert cfg.MODEL.FPN.FUSE_TYPE == "sum"
        self.out_channels = cfg.MODEL.FPN.OUT_CHANNELS
        self._out_features = ["p2", "p3", "p4", "p5", "p6"]
        self._strides = {f"p{i}": 2 ** i for i in range(2, 7)}
        self.w: Dict[str, tuple] = {}

    def output_shape(self) -> Dict[str, ShapeSpec]:
        return {n: ShapeSpec(channels=self.out_channels, stride=self._strides[n]) for n in self._out_features}

    # ------------------------------------------------------------------ weights
    def import_weights(self, sd: Dict[str, torch.Tensor], device, prefix: str = "backbone.") -> None:
        bu = prefix + "bottom_up."
        w = {}
        w["stem"] = fold_conv(sd, bu + "stem.conv1", bu + "stem.conv1.norm", device)
        for sname, nblk in _STAGES:
            for b in range(nblk):
                q = f"{bu}{sname}.{b}."
                for cname in ("conv1", "conv2", "conv3", "shortcut"):
                    if (q + cname + ".weight") in sd:
                        w[f"{sname}.{b}.{cname}"] = fold_conv(sd, q + cname, q + cname + ".norm", device)
                if f"{sname}.{b}.shortcut" in w:
                    # relu(conv3(out) + shortcut(x)) as ONE dual-source launch: [W_shortcut | W_conv3] along Cin, the two folded biases summed
                    (wsc, bsc), (w3, b3) = w[f"{sname}.{b}.shortcut"], w[f"{sname}.{b}.conv3"]
                    dual = K.prepare_dual_weights(wsc, w3)
                    if dual is not None and bsc is not None and b3 is not None:
                        w[f"{sname}.{b}.dual"] = (dual, (bsc + b3).contiguous())
        for lvl in (2, 3, 4, 5):
            w[f"lat{lvl}"] = fold_conv(sd, f"{prefix}fpn_lateral{lvl}", f"{prefix}fpn_lateral{lvl}.norm", device)
            w[f"out{lvl}"] = fold_conv(sd, f"{prefix}fpn_output{lvl}", f"{prefix}fpn_output{lvl}.norm", device)
        self.w = w

    # ------------------------------------------------------------------ forward
    def forward_nhwc(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        """x: [N,Hp,Wp,4] normalised NHWC4 batch -> {p2..p6} NHWC tensors."""
        w = self.w
        # entry of the conv chain: in 'fp16s' mode the stem writes fp16 and every later conv / pool follows its input's dtype
        if w["stem"][1] is not None and K.backbone_stem_supported(x, w["stem"][0]):
            # conv 7x7 s2 + BN + ReLU + max-pool 3x3 s2 in one kernel (csrc/backbone_stem.hip): the [N,H/2,W/2,64] map stays on the CU
            x = K.backbone_stem_fused(x, *w["stem"])
        else:
            x = K.conv2d_nhwc(x, *w["stem"], stride=2, padding=3, relu=1, out_dtype=K.act_dtype(w["stem"][0]))
            x = K.maxpool2d_nhwc(x, 3, 2, 1)
        feats = {}
        for sname, nblk in _STAGES:
            for b in range(nblk):
                stride = 2 if (b == 0 and sname != "res2") else 1
                key = f"{sname}.{b}."
                out = K.conv2d_nhwc(x, *w[key + "conv1"], stride=stride, relu=1)
                out = K.conv2d_nhwc(out, *w[key + "conv2"], padding=1, relu=1)
                dual = w.get(key + "dual")
                if dual is not None and K.dual_supported(x, out, dual[0], stride):
                    # first block of a stage: the shortcut conv rides in conv3's k-loop (x then conv2's output), its map never reaches HBM
                    x = K.conv1x1_dual_nhwc(x, out, dual[0], dual[1], stride=stride, relu=1)
                    continue
                sc = K.conv2d_nhwc(x, *w[key + "shortcut"], stride=stride) if (key + "shortcut") in w else x
                x = K.conv2d_nhwc(out, *w[key + "conv3"], relu=1, residual=sc, res_mode=1)
            feats[sname] = x
        res: Dict[str, torch.Tensor] = {}
        prev = None
        for lvl in (5, 4, 3, 2):
            if prev is None:
                prev = K.conv2d_nhwc(feats[f"res{lvl}"], *w[f"lat{lvl}"])
            else:
                prev = K.conv2d_nhwc(feats[f"res{lvl}"], *w[f"lat{lvl}"], residual=prev, res_mode=2)
            res[f"p{lvl}"] = K.conv2d_nhwc(prev, *w[f"out{lvl}"], padding=1)
        # LastLevelMaxPool = max_pool2d(kernel 1, stride 2): a pure subsample -> strided copy
        res["p6"] = res["p5"][:, ::2, ::2, :].contiguous()
        return {k: res[k] for k in self._out_features}

    def forward(self, x_nchw: torch.Tensor) -> Dict[str, torch.Tensor]:
        """d2 call convention: logical NCHW in, dict of logical NCHW (channels_last) out."""
        x = as_nhwc(x_nchw)
        if x.shape[-1] == 3:
            x = torch.nn.functional.pad(x, (0, 1))
        return {k: as_nchw_view(v) for k, v in self.forward_nhwc(x).items()}


@BACKBONE_REGISTRY.register()
def build_resnet_fpn_backbone(cfg, input_shape=None):
    return ResNetFPN(cfg)
