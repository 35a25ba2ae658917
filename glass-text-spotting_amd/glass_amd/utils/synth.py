"""Deterministic synthetic checkpoint / image / box generators.

There is no network, so neither the reference's released checkpoints nor its datasets
exist on either box.  `make_state_dict` emits a detectron2-keyed state dict with the
exact parameter names and shapes of the reference model (the key list of SURVEY.md §8b;
names follow reference glass/modeling/fusion/local_feature_extraction.py:103-132,
fusion_modules.py:48-65, recognition/recognizer_encoder.py:105-127,
recognition/prediction_aster.py:233-284, roi_heads/rotated_fast_rcnn.py:536-550 and
detectron2 v0.6's ResNet/FPN/RPN/FastRCNNConvFCHead module names [d2-recall]).

All draws come from one CPU `torch.Generator`, in a fixed order, so the same seed gives
bit-identical tensors here and on the GPU box.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Tuple

import torch


class _Gen:
    def __init__(self, seed: int):
        self.g = torch.Generator(device="cpu")
        self.g.manual_seed(seed)

    def normal(self, shape, std=1.0, mean=0.0):
        return torch.randn(shape, generator=self.g, dtype=torch.float32) * std + mean

    def uniform(self, shape, lo, hi):
        return torch.rand(shape, generator=self.g, dtype=torch.float32) * (hi - lo) + lo


def _conv(sd, g, name, cout, cin, kh, kw, bias=False, gain=1.0):
    fan_in = cin * kh * kw
    sd[name + ".weight"] = g.normal((cout, cin, kh, kw), std=gain * math.sqrt(2.0 / fan_in))
    if bias:
        sd[name + ".bias"] = g.normal((cout,), std=0.1)


def _bn(sd, g, name, c, gamma_scale=1.0, track=False):
    sd[name + ".weight"] = g.uniform((c,), 0.5, 1.5) * gamma_scale
    sd[name + ".bias"] = g.normal((c,), std=0.1)
    sd[name + ".running_mean"] = g.normal((c,), std=0.1)
    sd[name + ".running_var"] = g.uniform((c,), 0.5, 1.5)
    if track:
        sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def _linear(sd, g, name, cout, cin, gain=1.0):
    sd[name + ".weight"] = g.normal((cout, cin), std=gain * math.sqrt(2.0 / cin))
    sd[name + ".bias"] = g.normal((cout,), std=0.1)


# ResNet-50 stage table: (name, blocks, bottleneck width, out channels)
RESNET50_STAGES = (("res2", 3, 64, 256), ("res3", 4, 128, 512), ("res4", 6, 256, 1024),
                   ("res5", 3, 512, 2048))
# local extractor ("Res34"-style): BasicBlock counts [1,2,5,3], widths 64/128/256/256
LOCAL_LAYERS = ((1, 64), (2, 128), (5, 256), (3, 256))
RES_BRANCH_GAMMA = 0.25   # last norm of every residual branch: keeps activations O(1..10)


def make_state_dict(seed: int = 1234, num_classes: int = 1, num_chars: int = 97,
                    num_anchors: int = 12, box_pool: int = 7, fc_dim: int = 2048,
                    parts: Tuple[str, ...] = ("backbone", "rpn", "box", "recog")) -> "OrderedDict[str, torch.Tensor]":
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    # each part has its own stream so a test can build only what it needs and still get
    # the same tensors as the full model
    if "backbone" in parts:
        g = _Gen(seed)
        p = "backbone.bottom_up."
        # pixel values are O(100): scale the stem so the trunk runs at O(1)
        _conv(sd, g, p + "stem.conv1", 64, 3, 7, 7, gain=1.0 / 64)
        _bn(sd, g, p + "stem.conv1.norm", 64)
        cin = 64
        for sname, nblk, width, cout in RESNET50_STAGES:
            for b in range(nblk):
                q = f"{p}{sname}.{b}."
                if cin != cout:
                    _conv(sd, g, q + "shortcut", cout, cin, 1, 1)
                    _bn(sd, g, q + "shortcut.norm", cout)
                _conv(sd, g, q + "conv1", width, cin, 1, 1)
                _bn(sd, g, q + "conv1.norm", width)
                _conv(sd, g, q + "conv2", width, width, 3, 3)
                _bn(sd, g, q + "conv2.norm", width)
                _conv(sd, g, q + "conv3", cout, width, 1, 1)
                _bn(sd, g, q + "conv3.norm", cout, gamma_scale=RES_BRANCH_GAMMA)
                cin = cout
        for lvl, c in zip((2, 3, 4, 5), (256, 512, 1024, 2048)):
            _conv(sd, g, f"backbone.fpn_lateral{lvl}", 256, c, 1, 1, gain=0.7)
            _bn(sd, g, f"backbone.fpn_lateral{lvl}.norm", 256)
            _conv(sd, g, f"backbone.fpn_output{lvl}", 256, 256, 3, 3, gain=0.7)
            _bn(sd, g, f"backbone.fpn_output{lvl}.norm", 256)
    if "rpn" in parts:
        g = _Gen(seed + 1)
        p = "proposal_generator.rpn_head."
        _conv(sd, g, p + "conv", 256, 256, 3, 3, bias=True, gain=0.7)
        _conv(sd, g, p + "objectness_logits", num_anchors, 256, 1, 1, bias=True)
        _conv(sd, g, p + "anchor_deltas", num_anchors * 5, 256, 1, 1, bias=True, gain=0.15)
    if "box" in parts:
        g = _Gen(seed + 2)
        _linear(sd, g, "roi_heads.box_head.fc1", fc_dim, 256 * box_pool * box_pool, gain=0.7)
        _linear(sd, g, "roi_heads.box_head.fc2", fc_dim, fc_dim, gain=0.7)
        _linear(sd, g, "roi_heads.box_predictor.cls_score", num_classes + 1, fc_dim)
        _linear(sd, g, "roi_heads.box_predictor.bbox_pred", num_classes * 5, fc_dim, gain=0.5)
        _linear(sd, g, "roi_heads.box_predictor.orientation_pred", 4, fc_dim)
    if "recog" in parts:
        g = _Gen(seed + 3)
        _conv(sd, g, "roi_heads.recognizer_feature_fusion.conv1", 256, 256, 1, 1, gain=0.7)
        _conv(sd, g, "roi_heads.recognizer_feature_fusion.conv2", 256, 256, 1, 1, gain=0.7)
        p = "roi_heads.hybrid_net.ConvNet."
        _conv(sd, g, p + "conv0_1", 16, 3, 3, 3, gain=1.0 / 64)
        _bn(sd, g, p + "bn0_1", 16, track=True)
        _conv(sd, g, p + "conv0_2", 32, 16, 3, 3)
        _bn(sd, g, p + "bn0_2", 32, track=True)
        cin = 32
        for li, (nblk, planes) in enumerate(LOCAL_LAYERS, start=1):
            for b in range(nblk):
                q = f"{p}layer{li}.{b}."
                _conv(sd, g, q + "conv1", planes, cin, 3, 3)
                _bn(sd, g, q + "bn1", planes, track=True)
                _conv(sd, g, q + "conv2", planes, planes, 3, 3)
                _bn(sd, g, q + "bn2", planes, gamma_scale=RES_BRANCH_GAMMA, track=True)
                if cin != planes:
                    _conv(sd, g, q + "downsample.0", planes, cin, 1, 1)
                    _bn(sd, g, q + "downsample.1", planes, track=True)
                cin = planes
            if li < 4:
                _conv(sd, g, f"{p}conv{li}", planes, planes, 3, 3, gain=0.7)
                _bn(sd, g, f"{p}bn{li}", planes, track=True)
        _conv(sd, g, p + "conv4_1", 256, 256, 2, 2, gain=0.7)
        _bn(sd, g, p + "bn4_1", 256, track=True)
        p = "roi_heads.fusion_net."
        _conv(sd, g, p + "conv_mask", 1, 64, 1, 1, bias=True)
        _conv(sd, g, p + "channel_add_conv.0", 256, 512, 1, 1, bias=True)
        sd[p + "channel_add_conv.1.weight"] = g.uniform((256, 1, 1), 0.5, 1.5)
        sd[p + "channel_add_conv.1.bias"] = g.normal((256, 1, 1), std=0.1)
        _conv(sd, g, p + "channel_add_conv.3", 512, 256, 1, 1, bias=True)
        _conv(sd, g, p + "out", 256, 512, 3, 3, bias=True, gain=0.7)
        p = "roi_heads.recognizer_head.backbone."
        _conv(sd, g, p + "conv1", 256, 256, 2, 1, gain=0.7)
        _bn(sd, g, p + "conv1.norm", 256)
        _conv(sd, g, p + "conv2", 256, 256, 3, 3, gain=0.7)
        _bn(sd, g, p + "conv2.norm", 256)
        for layer in range(2):
            q = f"roi_heads.recognizer_head.encoder.bilsm_stack.{layer}."
            for sfx in ("", "_reverse"):
                sd[q + f"rnn.weight_ih_l0{sfx}"] = g.normal((1024, 256), std=1.0 / 16)
                sd[q + f"rnn.weight_hh_l0{sfx}"] = g.normal((1024, 256), std=1.0 / 16)
                sd[q + f"rnn.bias_ih_l0{sfx}"] = g.normal((1024,), std=0.3)
                sd[q + f"rnn.bias_hh_l0{sfx}"] = g.normal((1024,), std=0.3)
            sd[q + "linear.weight"] = g.normal((256, 512), std=1.0 / 16)
            sd[q + "linear.bias"] = g.normal((256,), std=0.1)
        q = "roi_heads.recognizer_head.decoder.recognizer.decoder."
        for nm in ("sEmbed", "xEmbed"):
            sd[q + f"attention_unit.{nm}.weight"] = g.normal((256, 256), std=1.0 / 16)
            sd[q + f"attention_unit.{nm}.bias"] = g.normal((256,), std=0.1)
        sd[q + "attention_unit.wEmbed.weight"] = g.normal((1, 256), std=0.25)
        sd[q + "attention_unit.wEmbed.bias"] = g.normal((1,), std=0.1)
        sd[q + "tgt_embedding.weight"] = g.normal((num_chars, 256), std=1.0)
        sd[q + "gru.weight_ih_l0"] = g.normal((768, 512), std=1.0 / 16)
        sd[q + "gru.weight_hh_l0"] = g.normal((768, 256), std=1.0 / 16)
        sd[q + "gru.bias_ih_l0"] = g.normal((768,), std=0.1)
        sd[q + "gru.bias_hh_l0"] = g.normal((768,), std=0.1)
        sd[q + "fc.weight"] = g.normal((num_chars, 256), std=0.5)
        sd[q + "fc.bias"] = g.normal((num_chars,), std=0.1)
        sd[q + "temperature"] = torch.ones(1)
    if "mask" in parts:        # rotated mask head; own stream, so the other parts are unchanged
        g = _Gen(seed + 4)
        p = "roi_heads.mask_head."
        for k in range(1, 5):
            _conv(sd, g, p + f"mask_fcn{k}", 256, 256, 3, 3, bias=True)
        # ConvTranspose2d weight is [Cin, Cout, 2, 2]; each output pixel sees ONE tap -> fan_in = Cin
        sd[p + "deconv.weight"] = g.normal((256, 256, 2, 2), std=math.sqrt(2.0 / 256))
        sd[p + "deconv.bias"] = g.normal((256,), std=0.1)
        _conv(sd, g, p + "predictor", num_classes, 256, 1, 1, bias=True, gain=2.0)
    return sd


def make_image(index: int, height: int, width: int) -> torch.Tensor:
    """uint8 HWC BGR image: U{0..255} noise smoothed by a 5x5 box filter (image seed =
    1000 + global image index)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(1000 + index)
    x = torch.randint(0, 256, (1, 3, height + 4, width + 4), generator=g, dtype=torch.int32).float()
    x = torch.nn.functional.avg_pool2d(x, 5, stride=1)
    # widen the contrast the box filter removed, keep 0..255
    x = ((x - 127.5) * 3.0 + 127.5).clamp_(0, 255)
    return x[0].permute(1, 2, 0).round().to(torch.uint8).contiguous()


def make_boxes(index: int, count: int, height: int, width: int) -> torch.Tensor:
    """`count` rotated word boxes (cx,cy,w,h,angle_deg) (box seed = 2000 + index):
    w~U(40,300), h~U(16,64), angle ~ 70% N(0,10deg) / 30% U(-180,180)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(2000 + index)
    u = torch.rand((count, 6), generator=g)
    n = torch.randn((count,), generator=g)
    cx = (0.1 + 0.8 * u[:, 0]) * width
    cy = (0.1 + 0.8 * u[:, 1]) * height
    w = 40 + 260 * u[:, 2]
    h = 16 + 48 * u[:, 3]
    ang = torch.where(u[:, 4] < 0.7, n * 10.0, u[:, 5] * 360.0 - 180.0)
    return torch.stack([cx, cy, w, h, ang], dim=1).float()


def make_text_image(index: int, height: int, width: int, words: int = 0) -> torch.Tensor:
    """uint8 HWC BGR image with text-like structure instead of filtered noise: a dark, slowly varying background and bright
    anti-aliased strokes (2-5 px wide line segments grouped into "letters" along rotated baselines), i.e. large flat regions,
    sharp edges and a strong DC component per tile - the input statistics that stress a Winograd transform (cancellation of
    large, nearly equal values).  Seed = 3000 + index."""
    g = torch.Generator(device="cpu")
    g.manual_seed(3000 + index)
    yy, xx = torch.meshgrid(torch.arange(height, dtype=torch.float32), torch.arange(width, dtype=torch.float32), indexing="ij")
    img = 30.0 + 25.0 * torch.sin(xx / 97.0 + 0.3) * torch.cos(yy / 131.0) + 4.0 * torch.rand((height, width), generator=g)
    n_words = words or max(8, (height * width) // 20000)
    u = torch.rand((n_words, 6), generator=g)
    for i in range(n_words):
        cx, cy = float(u[i, 0]) * width, float(u[i, 1]) * height
        h = 14.0 + 46.0 * float(u[i, 2])
        letters = 3 + int(float(u[i, 3]) * 8)
        ang = (float(u[i, 4]) - 0.5) * (0.5 if u[i, 5] < 0.7 else 6.28)
        ca, sa = math.cos(ang), math.sin(ang)
        bright = 190.0 + 65.0 * float(u[i, 5])
        thick = max(1.2, h / 9.0)
        s = torch.rand((letters, 3, 4), generator=g)                 # 3 strokes per letter: (x0, y0, x1, y1) in the letter box
        for k in range(letters):
            ox = (k - letters / 2.0) * 0.7 * h
            for j in range(3):
                lx0, ly0 = ox + 0.6 * h * float(s[k, j, 0]), (float(s[k, j, 1]) - 0.5) * h
                lx1, ly1 = ox + 0.6 * h * float(s[k, j, 2]), (float(s[k, j, 3]) - 0.5) * h
                x0, y0 = cx + ca * lx0 - sa * ly0, cy + sa * lx0 + ca * ly0
                x1, y1 = cx + ca * lx1 - sa * ly1, cy + sa * lx1 + ca * ly1
                a0, a1 = int(max(0, min(x0, x1) - thick - 2)), int(min(width, max(x0, x1) + thick + 3))
                b0, b1 = int(max(0, min(y0, y1) - thick - 2)), int(min(height, max(y0, y1) + thick + 3))
                if a1 <= a0 or b1 <= b0:
                    continue
                px, py = xx[b0:b1, a0:a1], yy[b0:b1, a0:a1]
                dx, dy = x1 - x0, y1 - y0
                t = (((px - x0) * dx + (py - y0) * dy) / max(dx * dx + dy * dy, 1e-6)).clamp_(0, 1)
                dist = torch.sqrt((px - x0 - t * dx) ** 2 + (py - y0 - t * dy) ** 2)
                cov = (thick * 0.5 + 0.5 - dist).clamp_(0, 1)        # anti-aliased coverage
                img[b0:b1, a0:a1] = torch.maximum(img[b0:b1, a0:a1], cov * bright + (1 - cov) * img[b0:b1, a0:a1])
    bgr = torch.stack([img * 0.9, img, img * 0.8 + 10.0], dim=2).clamp_(0, 255)
    return bgr.round().to(torch.uint8).contiguous()


def widen_dynamic_range(sd, seed: int = 77, conv_gain=(1.5, 2.5), gamma=(1.0, 3.0), var=(0.1, 0.6), head_gain: float = 1.0 / 12):
    """A copy of a `make_state_dict` checkpoint re-scaled towards what a TRAINED checkpoint looks like to the kernels (VERDICT r3
    #4): deep-stage activations of 1e1 ... 1e2 instead of O(1).  For res4 / res5 and the FPN (+ layer3 / layer4 of the local
    extractor): 3x3 and lateral conv weights x U(conv_gain) per layer, the BatchNorm in front of every 3x3 conv gets gamma x
    U(gamma) and running_var ~ U(var) per channel (eval BN scale = gamma / sqrt(var): up to ~9x; on the seed-1234 checkpoint the
    3x3 layers of res4 / res5 then read inputs of up to 150 ... 420, the pyramid levels reach 300 ... 440 (sigma ~50), the local
    extractor's layer3 / layer4 3x3 inputs 40 ... 100 - measured with the oracle), while the norm that closes a
    residual branch is scaled DOWN by the branch's mean gain so the trunk does not explode - the 3x3 (Winograd) layers see
    large, strongly positive inputs and produce outputs that are brought back to O(1..10) behind them, which is where an
    absolute error of a transform-domain kernel would show.  The heads that READ the (now ~10x larger) pyramid - RPN conv,
    box-head fc1, P2P3 fusion - are scaled by `head_gain`, as training would have done: logits stay O(1..10)."""
    g = _Gen(seed)
    out = OrderedDict((k, v.clone()) for k, v in sd.items())

    def boost_bn(name):
        c = out[name + ".weight"].shape[0]
        gm = g.uniform((c,), *gamma)
        out[name + ".weight"] *= gm
        out[name + ".running_var"] = g.uniform((c,), *var)
        return float(gm.mean()) / math.sqrt(0.5 * (var[0] + var[1]))

    def boost_conv(name):
        k = float(g.uniform((1,), *conv_gain))
        out[name + ".weight"] *= k
        return k

    p = "backbone.bottom_up."
    for sname, nblk, _w, _c in RESNET50_STAGES[2:]:                       # res4, res5
        for b in range(nblk):
            q = f"{p}{sname}.{b}."
            if (q + "conv1.weight") not in out:
                continue
            gain = boost_bn(q + "conv1.norm")                            # conv2's (3x3) input
            gain *= boost_conv(q + "conv2")
            gain *= boost_bn(q + "conv2.norm")
            out[q + "conv3.norm.weight"] /= gain                         # the branch returns to the trunk at its old scale
    for lvl in (2, 3, 4, 5):
        if f"backbone.fpn_lateral{lvl}.weight" in out:
            boost_conv(f"backbone.fpn_lateral{lvl}")                     # the 3x3 output convs read these (x upsampled sums)
            boost_conv(f"backbone.fpn_output{lvl}")
    for name in ("proposal_generator.rpn_head.conv", "roi_heads.box_head.fc1", "roi_heads.recognizer_feature_fusion.conv1",
                 "roi_heads.recognizer_feature_fusion.conv2", "roi_heads.mask_head.mask_fcn1"):
        if (name + ".weight") in out:
            out[name + ".weight"] *= head_gain
    p = "roi_heads.hybrid_net.ConvNet."
    for li, (nblk, _planes) in enumerate(LOCAL_LAYERS, start=1):
        if li < 3:
            continue
        for b in range(nblk):
            q = f"{p}layer{li}.{b}."
            if (q + "conv1.weight") not in out:
                continue
            gain = boost_conv(q + "conv1") * boost_bn(q + "bn1")          # conv2's (3x3) input
            gain *= boost_conv(q + "conv2")
            out[q + "bn2.weight"] /= gain
    return out


def pattern_text(N: int, K: int, T: int = 26, C: int = 97, stop_index: int = 94) -> torch.Tensor:
    """Peaked character distributions [N,K,T,C] from integer arithmetic (exactly reproducible on any host): probability 0.9 at
    class (7 n + 3 k + 5 t) % C, the stop symbol at step (k % 7) + 2, 0.1 / (C - 1) elsewhere.  Input of the word post-processor's
    regression fixture (scripts/make_pp_regression.py, tests/golden/postprocess_words_regression.npz)."""
    n, k, t = torch.meshgrid(torch.arange(N), torch.arange(K), torch.arange(T), indexing="ij")
    peak = (7 * n + 3 * k + 5 * t) % C
    peak = torch.where(t == (k % 7) + 2, torch.full_like(peak, stop_index), peak)
    text = torch.full((N, K, T, C), 0.1 / (C - 1), dtype=torch.float32)
    text.scatter_(3, peak.unsqueeze(-1), 0.9)
    return text
