"""detectron2-keyed state-dict importer: fold BatchNorm, re-lay weights for the HIP kernels.

The reference loads a torch `.pth` `{"model": state_dict}` with d2 module names through
`DetectionCheckpointer(model).load(path)` (reference glass/inference/glass_runner.py:59-60).
This module accepts the same key set and produces, once at load time, the
device tensors the kernels read:
  conv  [Cout,Cin,KH,KW] (+BN eval)  ->  w [Cout,KH,KW,Cin4] fp32 (BN scale folded, Cin padded
                                         to a multiple of 4 with zeros), bias [Cout]; wrapped in a
                                         ConvWeight together with its packed kernel forms (conv_weight)
  linear / recurrent                 ->  see `pack_kblocked` in ops/native.py
All tensor arithmetic here is one-off load-time plumbing on CPU (float64 fold, fp32 store).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

BN_EPS = 1e-5


def load_checkpoint_file(path: str) -> Dict[str, torch.Tensor]:
    """torch .pth with {"model": sd} (d2 convention) or a bare state dict."""
    obj = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(obj, dict) and "model" in obj and isinstance(obj["model"], dict):
        obj = obj["model"]
    return {k: (v if isinstance(v, torch.Tensor) else torch.as_tensor(v)) for k, v in obj.items()}


def fold_conv(sd: Dict[str, torch.Tensor], conv: str, norm: Optional[str], device, cin_pad: int = 4,
              eps: float = BN_EPS, ragged: bool = False) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """conv `conv`.weight[/bias] followed by eval-mode BatchNorm `norm`.* -> (w_khwc, bias).  `ragged`: the layer runs on maps
    of odd width (the local extractor's 16 x 33): it gets the last-column strip weights next to its Winograd forms
    (ops.native.conv2d_nhwc).  (Every 3x3 layer of this model has stride 1 - STRIDE_IN_1X1 puts a block's stride on its 1x1
    conv - so there is no stride to tell the packer about; ops.native.prepare_conv_weights keeps the argument for callers
    that have one.)"""
    w = sd[conv + ".weight"].double()
    b = sd[conv + ".bias"].double() if (conv + ".bias") in sd else None
    if norm is not None and (norm + ".weight") in sd:
        g, beta = sd[norm + ".weight"].double(), sd[norm + ".bias"].double()
        mean, var = sd[norm + ".running_mean"].double(), sd[norm + ".running_var"].double()
        scale = g / torch.sqrt(var + eps)
        w = w * scale.view(-1, 1, 1, 1)
        b = (beta - mean * scale) if b is None else (beta + (b - mean) * scale)
    cout, cin, kh, kw = w.shape
    cin4 = (cin + cin_pad - 1) // cin_pad * cin_pad
    out = torch.zeros((cout, kh, kw, cin4), dtype=torch.float32)
    out[..., :cin] = w.permute(0, 2, 3, 1).float()
    return conv_weight(out, device, ragged), (None if b is None else b.float().contiguous().to(device))


def dev(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().float().contiguous().to(device)


def conv_weight(w: torch.Tensor, device, ragged: bool = False):
    """[Cout,KH,KW,Cin] (conv) or [Nout,K] (linear) -> ops.native.ConvWeight on `device`: the raw fp32 tensor plus every
    packed form the layer's kernels stream (Winograd U, pointwise / fp16 MFMA fragment order), built here, ONCE, for the conv
    routing of the model being loaded (ops.native.packing_for: its precision decides the forms, and the Routing itself is stamped
    on the ConvWeight so that the layer's launches follow it).  CPU `device` (host-only tests): raw tensor only."""
    from .ops import native as K
    return K.prepare_conv_weights(dev(w, device), ragged=ragged)
