"""ctypes bindings of include/glass_hip.h over torch device tensors.

torch is used only for device memory and the current HIP stream.  Every wrapper refuses
CPU tensors: there is no fallback path.
"""
from __future__ import annotations

import ctypes
import math
from typing import List, Optional, Sequence, Tuple

import torch

from .._lib import GlassLibraryError, check, lib

c_int, c_float, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, c_int) for n in (
        "N", "H", "W", "Cin", "Cout", "KH", "KW", "stride_h", "stride_w", "pad_h", "pad_w", "Ho", "Wo",
        "ldx", "ldy", "y_coff", "y_cstride", "relu", "res_mode", "ldr")]


class RoiAlignDesc(ctypes.Structure):
    _fields_ = [("num_levels", c_int), ("feat", c_void_p * 5), ("H", c_int * 5), ("W", c_int * 5), ("ld", c_int * 5),
                ("scale", c_float * 5), ("min_level", c_int), ("C", c_int), ("PH", c_int), ("PW", c_int),
                ("sampling_ratio", c_int), ("ldy", c_int), ("y_coff", c_int), ("y_cstride", c_int)]


class DecoderWeights(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("sW", "sB", "wW", "wB", "emb", "w_ih", "w_hh", "b_ih", "b_hh", "fcW", "fcB")] + \
               [("temperature", c_float)]


def _dev(t: torch.Tensor, name: str = "tensor") -> int:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise GlassLibraryError(f"{name} must live on a HIP device (got {getattr(t, 'device', type(t))}); "
                                "the GLASS hot path has no CPU fallback")
    return t.data_ptr()


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise GlassLibraryError(f"{name} must be contiguous float32 (got {t.dtype}, contiguous={t.is_contiguous()})")
    return t


def stream_handle() -> int:
    return torch.cuda.current_stream().cuda_stream


def _pair(v) -> Tuple[int, int]:
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def conv_out_size(H, W, KH, KW, stride, padding):
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    return (H + 2 * ph - KH) // sh + 1, (W + 2 * pw - KW) // sw + 1


def conv2d_nhwc(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, stride=1, padding=0,
                relu: int = 0, residual: Optional[torch.Tensor] = None, res_mode: int = 0,
                out: Optional[torch.Tensor] = None, out_coff: int = 0, out_cstride: int = 1,
                cin: Optional[int] = None) -> torch.Tensor:
    """y = act(conv(x, w) + bias [+ residual]).  x [N,H,W,ldx] NHWC, w [Cout,KH,KW,Cin]."""
    _f32c(x, "x"); _f32c(w, "w")
    N, H, W, ldx = x.shape
    Cout, KH, KW, Cin = w.shape
    if cin is not None and cin != Cin:
        raise GlassLibraryError(f"cin={cin} does not match weight Cin={Cin}")
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    Ho, Wo = conv_out_size(H, W, KH, KW, stride, padding)
    if out is None:
        out = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    else:
        _f32c(out, "out")
        if tuple(out.shape[:3]) != (N, Ho, Wo):
            raise GlassLibraryError(f"out has shape {tuple(out.shape)}, expected ({N},{Ho},{Wo},*)")
    d = ConvDesc(N, H, W, Cin, Cout, KH, KW, sh, sw, ph, pw, Ho, Wo, ldx, out.shape[3], out_coff, out_cstride, relu,
                 res_mode if residual is not None else 0, residual.shape[-1] if residual is not None else 0)
    if residual is not None:
        _f32c(residual, "residual")
    check(lib().glass_conv2d_nhwc(ctypes.byref(d), c_void_p(_dev(x, "x")), c_void_p(_dev(w, "w")),
                                  c_void_p(_dev(bias, "bias") if bias is not None else None),
                                  c_void_p(_dev(residual, "residual") if residual is not None else None),
                                  c_void_p(_dev(out, "out")), c_void_p(stream_handle())), "glass_conv2d_nhwc")
    return out


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: int = 0,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [M,K] @ w[Nout,K]^T + bias on the same MFMA kernel (H = W = KH = KW = 1)."""
    M, K = x.shape
    y = conv2d_nhwc(x.view(M, 1, 1, K), w.view(w.shape[0], 1, 1, K), bias, relu=relu,
                    out=None if out is None else out.view(M, 1, 1, -1))
    return y.view(M, -1)


def maxpool2d_nhwc(x: torch.Tensor, kernel, stride, padding=0) -> torch.Tensor:
    _f32c(x, "x")
    N, H, W, C = x.shape
    KH, KW = _pair(kernel)
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    Ho, Wo = (H + 2 * ph - KH) // sh + 1, (W + 2 * pw - KW) // sw + 1
    y = torch.empty((N, Ho, Wo, C), dtype=torch.float32, device=x.device)
    check(lib().glass_maxpool2d_nhwc(c_void_p(_dev(x)), c_void_p(_dev(y)), N, H, W, C, KH, KW, sh, sw, ph, pw, Ho, Wo,
                                     c_void_p(stream_handle())), "glass_maxpool2d_nhwc")
    return y


def preprocess_image(chw: torch.Tensor, mean: Sequence[float], std: Sequence[float], batch: torch.Tensor, n: int) -> None:
    """(chw - mean)/std -> slot n of the zero-padded NHWC4 batch [N,Hp,Wp,4]."""
    _f32c(chw, "image"); _f32c(batch, "batch")
    _, H, W = chw.shape
    _, Hp, Wp, four = batch.shape
    assert four == 4
    m = (c_float * 3)(*[float(v) for v in mean])
    s = (c_float * 3)(*[float(v) for v in std])
    check(lib().glass_preprocess_image(c_void_p(_dev(chw)), H, W, m, s, c_void_p(_dev(batch)), int(n), Hp, Wp,
                                       c_void_p(stream_handle())), "glass_preprocess_image")


def image_u8hwc_to_chw(img: torch.Tensor, out_hw: Tuple[int, int], flip_channels: bool = False) -> torch.Tensor:
    if img.dtype != torch.uint8 or not img.is_contiguous():
        raise GlassLibraryError("image must be contiguous uint8 HWC")
    H, W, C = img.shape
    assert C == 3
    out = torch.empty((3, out_hw[0], out_hw[1]), dtype=torch.float32, device=img.device)
    check(lib().glass_image_u8hwc_to_chw_resized(c_void_p(_dev(img)), H, W, c_void_p(_dev(out)), out_hw[0], out_hw[1],
                                                 int(flip_channels), c_void_p(stream_handle())),
          "glass_image_u8hwc_to_chw_resized")
    return out


def roi_align_rotated(feats: List[torch.Tensor], scales: Sequence[float], boxes: torch.Tensor, batch_idx: torch.Tensor,
                      out_size: Tuple[int, int], sampling_ratio: int, channels: Optional[int] = None,
                      out: Optional[torch.Tensor] = None, out_coff: int = 0, out_cstride: int = 1) -> torch.Tensor:
    """feats: NHWC level tensors; boxes [R,5] float32; batch_idx [R] int32. Returns [R,PH,PW,C]."""
    R = boxes.shape[0]
    C = channels if channels is not None else feats[0].shape[-1]
    PH, PW = out_size
    if out is None:
        out = torch.empty((R, PH, PW, C), dtype=torch.float32, device=feats[0].device)
    d = RoiAlignDesc()
    d.num_levels = len(feats)
    for i, (f, s) in enumerate(zip(feats, scales)):
        _f32c(f, f"feat[{i}]")
        d.feat[i] = _dev(f)
        d.H[i], d.W[i], d.ld[i] = f.shape[1], f.shape[2], f.shape[3]
        d.scale[i] = float(s)
    d.min_level = int(round(-math.log2(scales[0])))
    d.C, d.PH, d.PW, d.sampling_ratio = C, PH, PW, int(sampling_ratio)
    d.ldy, d.y_coff, d.y_cstride = out.shape[-1], out_coff, out_cstride
    if R > 0:
        _f32c(boxes, "boxes")
        if batch_idx.dtype != torch.int32:
            raise GlassLibraryError("batch_idx must be int32")
        check(lib().glass_roi_align_rotated(ctypes.byref(d), c_void_p(_dev(boxes)), c_void_p(_dev(batch_idx)), R,
                                            c_void_p(_dev(out)), c_void_p(stream_handle())), "glass_roi_align_rotated")
    return out
