"""ctypes bindings of include/glass_hip.h over torch device tensors.

torch is used only for device memory and the current HIP stream.  Every wrapper refuses
CPU tensors: there is no fallback path.
"""
from __future__ import annotations

import collections
import ctypes
import math
import os
import threading
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


def _fhc(t: torch.Tensor, name: str) -> torch.Tensor:
    """contiguous float32 or float16 (fp16 storage mode) activation tensor"""
    if t.dtype not in (torch.float32, torch.float16) or not t.is_contiguous():
        raise GlassLibraryError(f"{name} must be contiguous float32 / float16 (got {t.dtype}, contiguous={t.is_contiguous()})")
    return t


def upload(data, dtype: torch.dtype, device) -> torch.Tensor:
    """A few host integers / floats -> device tensor WITHOUT draining the stream.  `torch.tensor(data, device=...)` and
    `.to(device)` of pageable memory are blocking copies: the host waits for everything queued on the stream before it
    (the step had ~8 of them = 8 hidden host/GPU synchronisations).  Staged through pinned memory the copy is just one
    more stream-ordered command (the caching host allocator keeps the staging block alive until the copy has run)."""
    t = torch.as_tensor(data, dtype=dtype)
    device = torch.device(device)
    if device.type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def stream_handle() -> int:
    return torch.cuda.current_stream().cuda_stream


def _pair(v) -> Tuple[int, int]:
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def conv_out_size(H, W, KH, KW, stride, padding):
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    return (H + 2 * ph - KH) // sh + 1, (W + 2 * pw - KW) // sw + 1


_PRECISIONS = ("fp32", "fp16", "fp16s")


_RNN_LAYOUTS = ((1, 1), (2, 1), (2, 2))       # (directions, RoI groups) per workgroup the library builds


def _parse_rnn(v):
    """"steps" | "persistent" | "2x1" | "2x2" | "1x1" | one of the (dirs, groups) tuples the library builds - anything else
    fails HERE (at Routing construction / replace), not inside the first launch"""
    if isinstance(v, str) and v in ("steps", "persistent"):
        return v
    try:
        nd, ng = (int(t) for t in (v if isinstance(v, (tuple, list)) else str(v).split("x")))
    except (TypeError, ValueError):
        nd = ng = -1
    if (nd, ng) not in _RNN_LAYOUTS:
        raise GlassLibraryError(f"unknown recurrent routing {v!r} (steps | persistent | 2x1 | 2x2 | 1x1)")
    return (nd, ng)


def _parse_split(v) -> int:
    try:
        n = int(v)
    except (TypeError, ValueError):
        n = -1
    if n not in (0, 6, 9):
        raise GlassLibraryError(f"unknown 1x1 split routing {v!r} (0 | 9 | 6)")
    return n


class Routing:
    """Which kernel a conv / linear launch takes, as VALUES that travel with the layer (no global state, re-entrant
    per model): a model builds one `Routing` at construction (environment variables GLASS_* give the defaults, `MODEL.
    CONV_PRECISION` the precision), `load_state_dict` stamps it on every `ConvWeight` it packs (`packing_for`), and
    `conv2d_nhwc` reads the routing OF THE WEIGHT IT IS HANDED - two models of different precision can be driven from two host
    threads, or interleaved segment by segment by `run_pipelined`, without touching anything shared.
      precision   'fp32' (the reference's arithmetic: fp32 MFMA / Winograd), 'fp16' (operands rounded to fp16, fp16 MFMA, fp32
                  accumulate, fp32 storage) or 'fp16s' (the same arithmetic, conv-path activations STORED as fp16; configs[4])
      winograd    eligible 3x3 / stride-1 layers take the Winograd kernels (GLASS_WINOGRAD=0: direct kernel everywhere)
      f43         ... F(4x4,3x3) where it pays (GLASS_WINOGRAD43=0: F(2x2,3x3) only)
      pw          the weight-streaming 1x1 kernel: True (the end-to-end rule), False, or "all" (every supported layer)
      h16         fp16 modes: the packed-weight fp16-MFMA kernel (GLASS_CONV_H16=0: fp32 template with fp16 operands)
      local_stem  the fused conv0_1 + conv0_2 + maxpool kernel of the local extractor (GLASS_LOCAL_STEM=0: three launches)
      stem        the fused 7x7 conv + ReLU + max-pool kernel of the ResNet stem (GLASS_BACKBONE_STEM=0: two launches)
      ragged      maps of width 4 k + 1 on the F(4x4) kernel: full tile columns there + the last pixel column as a strip
                  convolution (GLASS_W43_RAGGED=0: a whole extra tile column, as in rounds 2-3)
      rnn         the recurrent encoder: (directions, RoI groups) a workgroup of the one-launch-per-layer BiLSTM kernel
                  interleaves - "1x1" (default: shortest chain, 3 % off the one-step-at-a-time latency), "2x1", "2x2",
                  "persistent" = the library's default - or "steps" (GLASS_RNN=steps: one launch per time step,
                  csrc/recognition.hip); outputs are bit-identical
      splitk      implicit-GEMM launches with few output pixels and a long K (one image per step: the box head's fc layers on
                  100 rows, res4 / res5 3x3, the 11-row predictors) as a split-K launch + ordered reduction (GLASS_SPLITK=0)
      small_grid  3x3 layers whose F(4x4) grid does not fill the chip pick their Winograd form by rounds x workgroup time,
                  incl. the F(2x2) body + strip form for odd widths (GLASS_SMALL_GRID=0: the batch-8 rules only)
      split       1x1 layers (Cin % 32 == 0, Cout % 128 == 0) on the bf16 matrix cores with EXACT fp32 products: each fp32
                  operand as three bf16 pieces, 9 = all nine piece products (the sum an fp32 fma chain accumulates, in another
                  order; the default), 6 = without the three products below 2^-23 (opt-in, measurement only), 0 = the fp32-MFMA
                  kernels (GLASS_PW_SPLIT=0 | 6 | 9; csrc/pointwise_split.hip).  CAVEAT: +-inf / NaN operands (and |v| > 3.39e38)
                  do not split (h = inf, v - h = NaN): one non-finite activation makes every output channel of its pixel NaN,
                  where the fp32 kernels would propagate it only through the products it takes part in - same "non-finite in,
                  non-finite out" contract, coarser; a forward pass on finite weights and images never produces one
      pooled_fusion  P2P3Fusion's two 1x1 convolutions AFTER the recognizer pooler (on the pooled bins) instead of on the whole
                  p2 / p3 maps - RoIAlign and the fusion are both linear (GLASS_POOLED_FUSION=0: whole-map fusion, then pool)"""
    __slots__ = ("precision", "winograd", "f43", "pw", "h16", "local_stem", "stem", "ragged", "pooled_fusion", "rnn", "splitk", "small_grid", "split")

    def __init__(self, precision: Optional[str] = None, winograd: Optional[bool] = None, f43: Optional[bool] = None, pw=None,
                 h16: Optional[bool] = None, local_stem: Optional[bool] = None, stem: Optional[bool] = None,
                 ragged: Optional[bool] = None, pooled_fusion: Optional[bool] = None, rnn=None,
                 splitk: Optional[bool] = None, small_grid: Optional[bool] = None, split: Optional[int] = None):
        e = os.environ.get
        self.precision = precision or e("GLASS_CONV_PRECISION", "fp32")
        if self.precision not in _PRECISIONS:
            raise GlassLibraryError(f"unknown conv precision {self.precision!r}")
        self.winograd = (e("GLASS_WINOGRAD", "1") != "0") if winograd is None else bool(winograd)
        self.f43 = (e("GLASS_WINOGRAD43", "1") != "0") if f43 is None else bool(f43)
        self.pw = {"0": False, "all": "all"}.get(e("GLASS_POINTWISE", "1"), True) if pw is None else (pw if pw == "all" else bool(pw))
        self.h16 = (e("GLASS_CONV_H16", "1") != "0") if h16 is None else bool(h16)
        self.local_stem = (e("GLASS_LOCAL_STEM", "1") != "0") if local_stem is None else bool(local_stem)
        self.stem = (e("GLASS_BACKBONE_STEM", "1") != "0") if stem is None else bool(stem)
        self.ragged = (e("GLASS_W43_RAGGED", "1") != "0") if ragged is None else bool(ragged)
        self.pooled_fusion = (e("GLASS_POOLED_FUSION", "1") != "0") if pooled_fusion is None else bool(pooled_fusion)
        self.rnn = _parse_rnn(e("GLASS_RNN", "1x1")) if rnn is None else _parse_rnn(rnn)
        self.splitk = (e("GLASS_SPLITK", "1") != "0") if splitk is None else bool(splitk)
        self.small_grid = (e("GLASS_SMALL_GRID", "1") != "0") if small_grid is None else bool(small_grid)
        self.split = _parse_split(e("GLASS_PW_SPLIT", "9") if split is None else split)

    def replace(self, **kw) -> "Routing":
        r = Routing.__new__(Routing)
        for k in Routing.__slots__:
            setattr(r, k, kw.pop(k, getattr(self, k)))
        if kw:
            raise TypeError(f"unknown routing fields {sorted(kw)}")
        if r.precision not in _PRECISIONS:
            raise GlassLibraryError(f"unknown conv precision {r.precision!r}")
        r.split = _parse_split(r.split)
        r.rnn = _parse_rnn(r.rnn)
        return r

    @property
    def act_dtype(self) -> torch.dtype:
        """storage dtype the modules request for conv-path activations: float16 in 'fp16s' mode, else float32"""
        return torch.float16 if self.precision == "fp16s" else torch.float32

    def __repr__(self):
        return "Routing(" + ", ".join(f"{k}={getattr(self, k)!r}" for k in Routing.__slots__) + ")"


# The routing of launches that are handed RAW tensors (tests, micro-benchmarks, scripts/): the set_*() functions below edit
# it.  A model's layers never read it - their ConvWeight carries the model's own Routing.
_DEFAULT = Routing()
_TLS = threading.local()                   # per host thread: .last_path (diagnostic), .load (the enclosing packing_for)
_COUNTS = {"on_the_fly": 0}


def default_routing() -> Routing:
    return _DEFAULT


def routing_of(w) -> Routing:
    """the Routing a launch with weight `w` follows: the one stamped on its ConvWeight at load, else the raw-tensor default"""
    r = getattr(w, "routing", None)
    return r if r is not None else _DEFAULT


def set_conv_precision(precision: str) -> str:
    """Precision of launches on RAW tensors / ConvWeights prepared outside a model load (tests, scripts): 'fp32', 'fp16' or
    'fp16s' (see Routing).  Models are not affected: their precision is `MODEL.CONV_PRECISION`, carried by their weights.
    `GLASS_CONV_PRECISION` sets the initial value.  Returns the previous setting."""
    if precision not in _PRECISIONS:
        raise GlassLibraryError(f"unknown conv precision {precision!r}")
    prev = _DEFAULT.precision
    _DEFAULT.precision = precision
    return prev


def conv_precision() -> str:
    return _DEFAULT.precision


def act_dtype(w=None) -> torch.dtype:
    """storage dtype for conv-path activations under the routing of weight `w` (None: the raw-tensor default)"""
    return routing_of(w).act_dtype


def last_conv_path() -> str:
    """'pointwise' (conv1x1_pw_f32), 'winograd43' (conv3x3_wino43_f32; 'winograd43r': its full tile columns + the last pixel column as a strip convolution), 'fused_stem', 'winograd128' / 'winograd' (conv3x3_wino128_f32 / conv3x3_wino_f32), 'direct', 'direct_fp16' (fp32 template, fp16 operands) or 'packed_fp16' (conv_h16_kernel): which kernel
    the most recent conv2d_nhwc call OF THIS HOST THREAD launched (bench/profiling aid)."""
    return getattr(_TLS, "last_path", "direct")


def set_winograd(enabled: bool) -> bool:
    """raw-tensor default: route eligible 3x3 convolutions through glass_conv3x3_winograd_nhwc (default on; GLASS_WINOGRAD=0
    turns it off).  Returns the previous setting."""
    prev = _DEFAULT.winograd
    _DEFAULT.winograd = bool(enabled)
    return prev


def set_winograd43(enabled: bool) -> bool:
    """raw-tensor default: let eligible layers (Cout % 128 == 0, Cin % 32 == 0, see _use_f43) take the F(4x4,3x3) kernel
    (default on; GLASS_WINOGRAD43=0 turns it off: F(2x2,3x3) everywhere).  Returns the previous setting."""
    prev = _DEFAULT.f43
    _DEFAULT.f43 = bool(enabled)
    return prev


def set_conv_h16(enabled: bool) -> bool:
    """raw-tensor default: let fp16-input convolutions with Cin % 64 == 0 and Cout % 64 == 0 take the packed-weight fp16 kernel
    (glass_conv2d_nhwc_h16_packed; default on, GLASS_CONV_H16=0: the fp32 template with fp16 operands everywhere).
    Returns the previous setting."""
    prev = _DEFAULT.h16
    _DEFAULT.h16 = bool(enabled)
    return prev


def set_pointwise(enabled: bool) -> bool:
    """raw-tensor default: let eligible 1x1 convolutions (Cin % 32 == 0, Cout % 128 == 0) take the weight-streaming GEMM kernel
    (default on; GLASS_POINTWISE=0: the implicit-GEMM kernel everywhere).  Returns the previous setting."""
    prev = _DEFAULT.pw
    _DEFAULT.pw = enabled if enabled == "all" else bool(enabled)      # "all": every supported layer (tests, micro-benchmarks)
    return prev


def set_pw_split(products: int) -> int:
    """raw-tensor default: 1x1 layers (Cin % 32 == 0, Cout % 128 == 0, grid >= 1.5 x the chip: _use_split) on the bf16-split kernel with 9 (the exact
    product, default) or 6 piece products, 0: the fp32-MFMA kernels (see Routing.split).  Returns the previous setting."""
    prev = _DEFAULT.split
    _DEFAULT.split = _parse_split(products)
    return prev


def winograd_pack(w: torch.Tensor, f43=False) -> torch.Tensor:
    """w [Cout,3,3,Cin] -> packed U for glass_conv3x3_winograd_nhwc (16*Cout*Cin floats) or, with f43 True, for
    glass_conv3x3_winograd43_nhwc (36*Cout*Cin floats); f43 == "pw": w [Cout,1,1,Cin] -> the fragment-ordered weights of
    glass_conv1x1_pointwise_nhwc; f43 == "h16": w [Cout,KH,KW,Cin] -> the fp16 fragment-ordered weights of
    glass_conv2d_nhwc_h16_packed."""
    _f32c(w, "w")
    Cout, KH, KW, Cin = w.shape
    L = lib()
    if f43 == "col1":
        # the last-pixel-column strip of a 3x3 / pad-1 layer: output column W-1 reads input columns W-2, W-1 (W is padding),
        # i.e. taps kw = 0, 1 -> a KH = 3, KW = 1 convolution over channels (kw, cin): [Cout,3,1,2*Cin]
        return w[:, :, 0:2, :].reshape(Cout, 3, 1, 2 * Cin).contiguous()
    if f43 == "h16":
        u = torch.empty((int(L.glass_conv_h16_weight_halves(Cout, KH, KW, Cin)),), dtype=torch.float16, device=w.device)
        check(L.glass_conv_h16_pack_weights(c_void_p(_dev(w, "w")), Cout, KH, KW, Cin, c_void_p(_dev(u)), c_void_p(stream_handle())),
              "glass_conv_h16_pack_weights")
        return u
    if f43 == "pws":
        u = torch.empty((int(L.glass_pointwise_split_weight_bytes(Cout, Cin)),), dtype=torch.uint8, device=w.device)
        check(L.glass_pointwise_split_pack_weights(c_void_p(_dev(w, "w")), Cout, Cin, c_void_p(_dev(u)), c_void_p(stream_handle())),
              "glass_pointwise_split_pack_weights")
        return u
    if f43 == "pw":
        nf, fn, what = L.glass_pointwise_weight_floats, L.glass_pointwise_pack_weights, "glass_pointwise_pack_weights"
    elif f43:
        nf, fn, what = L.glass_winograd43_weight_floats, L.glass_winograd43_pack_weights, "glass_winograd43_pack_weights"
    else:
        nf, fn, what = L.glass_winograd_weight_floats, L.glass_winograd_pack_weights, "glass_winograd_pack_weights"
    u = torch.empty((int(nf(Cout, Cin)),), dtype=torch.float32, device=w.device)
    check(fn(c_void_p(_dev(w, "w")), Cout, Cin, c_void_p(_dev(u)), c_void_p(stream_handle())), what)
    return u


def _use_f43(N: int, H: int, W: int, Cout: int, Cin: int, body: bool = False) -> bool:
    """F(4x4,3x3) pays when the 4x4 tiling does not waste much of the map (H, W rounded up to multiples of 4 vs 2)
    and the grid still fills the chip (16 tiles x 128 channels per workgroup; 32 x 64 for the narrow shape).  `body`: only the
    full tile columns run on this kernel (width 4 k + 1: the last column is a strip convolution)."""
    t4 = ((H + 3) // 4) * (W // 4 if body else (W + 3) // 4)
    waste = (t4 * 16.0) / (((H + 1) // 2) * ((W + 1) // 2) * 4.0)
    wide = Cout % 128 == 0 and Cin % 32 == 0
    blocks = ((N * t4 + 15) // 16) * (Cout // 128) if wide else ((N * t4 + 31) // 32) * (Cout // 64)
    return waste <= 1.25 and blocks >= 192


NUM_CUS = 256
_F43_SPLITK = os.environ.get("GLASS_F43_SPLITK", "1") != "0"        # (A/B switch of the F(4x4) split-K routing, read once)
_SPLIT_LONGK = os.environ.get("GLASS_SPLIT_LONGK", "1") != "0"      # (A/B switch of _use_split's long-k rule, read once)


def _use_split(px: int, Cin: int, Cout: int) -> bool:
    """the bf16-split 1x1 kernel pays when its 64-pixel x 128-channel blocks fill the chip one and a half times (below that the
    per-workgroup weight stream is not amortised and the implicit-GEMM kernel's smaller tiles / split-K win: 1024 -> 256 at 64 x 64
    is 0.036 -> 0.050 ms) and, for the two-k-tile layers (Cin 64: prologue + epilogue per block are most of its life), only on big
    maps (64 -> 256 + residual: 0.93x at 8 x 256 x 256, 1.10x at 1 x 256 x 256) - profiles/r05_pw_split.txt, per-layer tables"""
    blocks = -(-px // 64) * (Cout // 128)
    if _SPLIT_LONGK and 192 <= blocks < 384 and Cin >= 512:
        # round 6: layers that fill the chip only 0.75 - 1.5 times but run a LONG k-loop per block (>= 16 k-tiles: the weight stream of a
        # block is amortised over its own k-loop, not over many blocks) - the box head's fc1 / fc2 at 800 rows, the FPN lateral on res5,
        # the recurrent layers' input projections: 13 - 25 % faster than the implicit-GEMM kernel (scripts/exp_fc1_split.py)
        return True
    return blocks >= 384 and (Cin >= 128 or px >= 262144)


def _small_grid_3x3(N: int, H: int, W: int, Cout: int, Cin: int, can_body: bool, f43_ok: bool):
    """3x3 / stride-1 layers whose F(4x4) grid does NOT fill the chip (`_use_f43` said no) - every trunk layer once ONE image
    is in flight (the reference predictor's batch, glass_runner.py:93-96): pick the Winograd form by ROUNDS x the time one
    workgroup takes.  All of these kernels run one workgroup per CU, so a launch costs ceil(workgroups / 256) rounds of a
    workgroup's duration, which is linear in Cin (measured on MI355X, profiles/r05_conv_table_b1_before.txt and the batch-8
    tables, scripts/exp_small_grid.py: F(4x4) wide 10 + 0.34 Cin us, narrow 9 + 0.56 Cin, F(2x2) 128-channel 9 + 0.30 Cin, F(2x2)
    64-channel 3 Cin).
    Returns (kind, body): kind "f43" | "f22", body True = full tile columns + the last-column strip (odd widths; ~20 us)."""
    def rounds(blocks):
        return -(-blocks // NUM_CUS)
    cands = []
    wide43 = Cout % 128 == 0 and Cin % 32 == 0
    wide22 = Cout % 128 == 0 and Cin % 32 == 0
    for body in ((False, True) if can_body else (False,)):
        strip = 20.0 if body else 0.0
        if f43_ok and (not body or W % 4 == 1):
            t4 = N * ((H + 3) // 4) * (W // 4 if body else (W + 3) // 4)
            blocks = -(-t4 // 16) * (Cout // 128) if wide43 else -(-t4 // 32) * (Cout // 64)
            cands.append((rounds(blocks) * ((10 + 0.34 * Cin) if wide43 else (9 + 0.56 * Cin)) + strip, "f43", body))
        t2 = N * ((H + 1) // 2) * (W // 2 if body else (W + 1) // 2)
        blocks = -(-t2 // 32) * (Cout // 128) if wide22 else -(-t2 // 64) * (Cout // 64)
        cands.append((rounds(blocks) * ((9 + 0.30 * Cin) if wide22 else 3.0 * Cin) + strip, "f22", body))
    t, kind, body = min(cands)
    return kind, body, t


def _TLS_force_f43k():
    return getattr(_TLS, "force_f43k", None)


def _f43_splitk_plan(N: int, H: int, W: int, Cout: int, Cin: int, can_body: bool):
    """Split-K plan of the F(4x4,3x3) kernel (glass_conv3x3_winograd43_splitk_nhwc, wide shape) for a 3x3 / stride-1 layer whose
    16-tile x 128-channel grid leaves most of the chip idle - one image in flight (reference glass_runner.py:93-96): res4 / res5 3x3
    on 64 x 64 / 32 x 32 maps are 32 / 16 workgroups of 8 / 16 k-tiles.  Model fitted to scripts/exp_f43_splitk.py on MI355X
    (profiles/r06_f43_splitk.txt): a launch takes ceil(workgroups / 256) rounds of 17 + 7.5 us per k-tile a workgroup walks,
    + 8 + 0.3 us per slice for the reduction launch, + the slices' partial outputs once out and once back at ~12 TB/s (they stay in L2 / Infinity
    Cache at these sizes), + ~25 us for the last-column strip in `body` form.
    -> (splits, body, modelled us) or None."""
    if Cout % 128 or Cin % 32:
        return None
    nk = Cin // 32
    best = None
    for body in ((False, True) if (can_body and W % 4 == 1) else (False,)):
        t4 = N * ((H + 3) // 4) * (W // 4 if body else (W + 3) // 4)
        blocks = -(-t4 // 16) * (Cout // 128)
        if blocks > NUM_CUS // 4:       # a quarter of the chip or less: with more, and two steps in flight, the other step's kernels
            continue                    # already fill the idle CUs and the split only adds a launch (8-image bench: 310.0 vs 309.0)
        for sl in (2, 3, 4, 6, 8, 12, 16, 24, 32):
            if nk % sl or blocks * sl > NUM_CUS:
                continue
            t = -(-(blocks * sl) // NUM_CUS) * (17.0 + 7.5 * nk / sl) + 8.0 + 0.3 * sl + 2.0 * sl * N * H * W * Cout * 4 / 12e6 + (25.0 if body else 0.0)
            if best is None or t < best[2]:
                best = (sl, body, t)
    return best


class ConvWeight:
    """One conv / linear layer's weights as the kernels read them: `raw` [Cout,KH,KW,Cin] fp32 (what the implicit-GEMM kernel
    takes) plus the packed forms the other kernels stream (`packs`: False -> F(2x2,3x3) U, True -> F(4x4,3x3) U, "pw" ->
    fragment-ordered 1x1 weights, "h16" -> fp16 fragment-ordered weights), built ONCE by prepare_conv_weights when the
    checkpoint is loaded and owned by the layer: they live and die with the model, no global cache, no first-launch
    synchronisation, nothing keyed by a device address (checkpoint row).  `routing`: the owning model's
    Routing (None for weights prepared outside a model load: such launches follow the raw-tensor default).  `version` is
    `raw._version` at packing time: an in-place edit of `raw` afterwards (weight surgery, `copy_`) makes `_packed` re-pack
    instead of streaming stale packs."""
    __slots__ = ("raw", "packs", "routing", "version")

    def __init__(self, raw: torch.Tensor, routing: Optional[Routing] = None):
        self.raw = _f32c(raw, "w")
        self.packs = {}
        self.routing = routing
        # (tensors created under torch.inference_mode() carry no version counter: loading a model inside inference_mode()
        #  must work - such weights cannot be edited in place anyway, so there is no staleness to detect)
        self.version = None if raw.is_inference() else raw._version

    @property
    def shape(self):
        return self.raw.shape

    @property
    def device(self):
        return self.raw.device

    def nbytes(self) -> int:
        return self.raw.numel() * 4 + sum(u.numel() * u.element_size() for u in self.packs.values())


class packing_for:
    """`with packing_for(routing):` around a model's import_weights (this host thread): prepare_conv_weights packs for THAT
    routing's precision ('fp32': Winograd / pointwise forms, 'fp16' / 'fp16s': the fp16 fragment form), stamps the routing on
    every ConvWeight it builds, and skips its per-call stream synchronisation - the caller synchronises once when every layer
    is packed (GeneralizedRCNN.load_state_dict).  A plain precision string is accepted for a default-switch Routing."""

    def __init__(self, routing):
        self.routing = routing if isinstance(routing, Routing) else Routing(precision=routing)

    def __enter__(self):
        self.prev = getattr(_TLS, "load", None)
        _TLS.load = self.routing
        return self

    def __exit__(self, *exc):
        _TLS.load = self.prev
        return False


def _probe_desc(Cout: int, KH: int, KW: int, Cin: int) -> ConvDesc:
    """a small stride-1 'same' layer with these channel counts: what the *_supported() entry points need to say whether a
    weight CAN take a kernel (map size and batch decide at launch whether it does)"""
    ph, pw = KH // 2, KW // 2
    return ConvDesc(1, 16, 16, Cin, Cout, KH, KW, 1, 1, ph, pw, 16 + 2 * ph - KH + 1, 16 + 2 * pw - KW + 1, Cin, Cout, 0, 1, 0, 0, 0)


def pack_kinds(Cout: int, KH: int, KW: int, Cin: int, precision: str, stride=1, ragged: bool = False, split: int = 0) -> list:
    """which packed forms a [Cout,KH,KW,Cin] weight can be asked for under `precision` (the routing of conv2d_nhwc);
    "all": every form the layer supports (micro-benchmarks that force kernels across precisions).  `stride` != 1 rules the
    Winograd forms out (stride-2 3x3 layers were carrying 52/9 of their size in packs no launch could take)."""
    if precision == "all":
        return pack_kinds(Cout, KH, KW, Cin, "fp32", stride, ragged, split=9) + pack_kinds(Cout, KH, KW, Cin, "fp16", stride)
    L, d = lib(), _probe_desc(Cout, KH, KW, Cin)
    kinds = []
    if precision == "fp32":
        if KH == 3 and KW == 3 and _pair(stride) == (1, 1):
            if L.glass_winograd_supported(ctypes.byref(d)):
                kinds.append(False)
            if L.glass_winograd43_supported(ctypes.byref(d)):
                kinds.append(True)
                if ragged and (2 * Cin) % 32 == 0:
                    kinds.append("col1")     # the layer meets maps of width 4 k + 1 (caller's hint): strip weights of the last column
        if KH == 1 and KW == 1 and L.glass_pointwise_supported(ctypes.byref(d)):
            kinds.append("pw")
        if split and KH == 1 and KW == 1 and L.glass_pointwise_split_supported(ctypes.byref(d)):
            kinds.append("pws")
    elif Cin % 64 == 0 and L.glass_conv_h16_supported(ctypes.byref(d), 1):
        kinds.append("h16")
    return kinds


def prepare_conv_weights(w: torch.Tensor, precision: Optional[str] = None, stride=1, ragged: bool = False) -> ConvWeight:
    """w [Cout,KH,KW,Cin] (or [Nout,K] for a linear layer) fp32 on the device -> ConvWeight with every packed form its layer
    can be routed to under `precision` (default: the enclosing packing_for(), else the raw-tensor default precision).
    `stride`: the layer's stride when the caller knows it (a stride-2 3x3 layer needs no Winograd form); `ragged`: the layer
    runs on maps of width 4 k + 1 (the local extractor's 16 x 33 maps) and wants the last-column strip weights.  Load-time plumbing:
    ~10 pack kernels per MB of weights, once per model."""
    if isinstance(w, ConvWeight):
        return w
    if w.dim() == 2:
        w = w.view(w.shape[0], 1, 1, w.shape[1])
    load = getattr(_TLS, "load", None)
    cw = ConvWeight(w, routing=load)
    precision = precision or (load.precision if load is not None else _DEFAULT.precision)
    Cout, KH, KW, Cin = w.shape
    if w.is_cuda:
        for kind in pack_kinds(Cout, KH, KW, Cin, precision, stride, ragged, split=(load.split if load is not None else _DEFAULT.split)):
            cw.packs[kind] = winograd_pack(w, kind)
        if cw.packs and load is None:
            torch.cuda.current_stream().synchronize()      # other streams (pipelined steps) may launch with it next
    return cw


def packs_on_the_fly() -> int:
    """how many conv launches had to pack their weights at launch time (a raw tensor instead of a ConvWeight, a ConvWeight
    prepared for another precision, or one whose raw tensor was edited in place after packing): 0 on the model path - tests
    assert it.  Raw-tensor callers that launch the same layer repeatedly should call prepare_conv_weights once."""
    return _COUNTS["on_the_fly"]


def _packed(w, wt: torch.Tensor, kind) -> torch.Tensor:
    if isinstance(w, ConvWeight):
        if w.version is not None and not wt.is_inference() and w.version != wt._version:
            # `raw` was written in place since it was packed: every pack is stale.  Re-pack the forms the layer had, on this
            # launch's stream (stream order covers this launch; a caller that edits weights while OTHER streams are launching
            # with them has to synchronise itself, as with any tensor)
            for k in list(w.packs):
                w.packs[k] = winograd_pack(wt, k)
            w.version = wt._version
            _COUNTS["on_the_fly"] += 1
        u = w.packs.get(kind)
        if u is not None:
            return u
    # raw tensors (tests, micro-benchmarks) and layers prepared for another precision: pack for this launch only - same
    # stream as the launch, so stream order is all the synchronisation it needs, and nothing outlives the call
    _COUNTS["on_the_fly"] += 1
    return winograd_pack(wt, kind)


def conv2d_nhwc(x: torch.Tensor, w, bias: Optional[torch.Tensor] = None, *, stride=1, padding=0,
                relu: int = 0, residual: Optional[torch.Tensor] = None, res_mode: int = 0,
                out: Optional[torch.Tensor] = None, out_coff: int = 0, out_cstride: int = 1,
                cin: Optional[int] = None, winograd: Optional[bool] = None, out_dtype: Optional[torch.dtype] = None,
                precision: Optional[str] = None, routing: Optional[Routing] = None) -> torch.Tensor:
    """y = act(conv(x, w) + bias [+ residual]).  x [N,H,W,ldx] NHWC, w [Cout,KH,KW,Cin]: a ConvWeight (the model path: packed
    forms built at load) or a plain device tensor (packed per launch where the chosen kernel needs it).
    3x3/stride 1/pad 1 layers that glass_winograd_supported() accepts go through the Winograd kernel
    (winograd=None: follow the routing; True/False force F(2x2,3x3) on / off for this call, "f43" forces the F(4x4,3x3)
    kernel, "f22r" F(2x2,3x3) on the full tile columns of an odd-width map + the last-column strip).  Which kernel runs is decided by `routing` (default: the Routing stamped on `w` at model load, else the
    raw-tensor default) with `precision` overriding its precision for this call - nothing process-global is read for a
    model's layers."""
    wt = w.raw if isinstance(w, ConvWeight) else _f32c(w, "w")
    rt = routing if routing is not None else routing_of(w)
    if precision is not None and precision != rt.precision:
        rt = rt.replace(precision=precision)
    _fhc(x, "x")
    N, H, W, ldx = x.shape
    Cout, KH, KW, Cin = wt.shape
    if cin is not None and cin != Cin:
        raise GlassLibraryError(f"cin={cin} does not match weight Cin={Cin}")
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    Ho, Wo = conv_out_size(H, W, KH, KW, stride, padding)
    if out is None:
        # the output follows the input's storage dtype unless told otherwise (fp16 tensors exist only in 'fp16s' mode)
        out = torch.empty((N, Ho, Wo, Cout), dtype=out_dtype or x.dtype, device=x.device)
    else:
        _fhc(out, "out")
        if tuple(out.shape[:3]) != (N, Ho, Wo):
            raise GlassLibraryError(f"out has shape {tuple(out.shape)}, expected ({N},{Ho},{Wo},*)")
    if out_coff < 0 or out_cstride < 1 or out_coff + (Cout - 1) * out_cstride >= out.shape[3]:
        raise GlassLibraryError(f"output channel window (offset {out_coff}, {Cout} channels, stride {out_cstride}) does not "
                                f"fit the {out.shape[3]} channels of out")
    if residual is not None and residual.shape[-1] < Cout:
        raise GlassLibraryError(f"residual has {residual.shape[-1]} channels, conv produces {Cout}")
    if ldx < Cin:
        raise GlassLibraryError(f"x has {ldx} channels, weight expects {Cin}")
    d = ConvDesc(N, H, W, Cin, Cout, KH, KW, sh, sw, ph, pw, Ho, Wo, ldx, out.shape[3], out_coff, out_cstride, relu,
                 res_mode if residual is not None else 0, residual.shape[-1] if residual is not None else 0)
    if residual is not None:
        _fhc(residual, "residual")
    def launch(entry: str, path: str, xt: torch.Tensor, weights: torch.Tensor, flags: Optional[int] = None) -> torch.Tensor:
        """one conv entry of the library: (desc, x, w | packed weights, bias, residual, y[, flags], stream)"""
        _TLS.last_path = path
        args = [ctypes.byref(d), c_void_p(_dev(xt, "x")), c_void_p(_dev(weights, "w")),
                c_void_p(_dev(bias, "bias") if bias is not None else None),
                c_void_p(_dev(residual, "residual") if residual is not None else None), c_void_p(_dev(out, "out"))]
        if flags is not None:
            args.append(int(flags))
        check(getattr(lib(), entry)(*args, c_void_p(stream_handle())), entry)
        return out

    any_half = x.dtype == torch.float16 or out.dtype == torch.float16 or (residual is not None and residual.dtype == torch.float16)
    if any_half:
        if rt.precision != "fp16s" or winograd:
            raise GlassLibraryError("float16 activation tensors need conv precision 'fp16s' (and no forced Winograd)")
        flags = (1 if x.dtype == torch.float16 else 0) | (2 if out.dtype == torch.float16 else 0) | \
                (4 if residual is not None and residual.dtype == torch.float16 else 0)
        # fp16 input, Cin / Cout multiples of 64: the kernel built for the fp16 matrix cores (csrc/conv_h16.hip, weights
        # pre-rounded and packed once); everything else - fp32 entries, the 4/16/32-channel first layers, the narrow heads -
        # stays on the fp32 template with fp16 operands
        if rt.h16 and lib().glass_conv_h16_supported(ctypes.byref(d), int(flags)):
            return launch("glass_conv2d_nhwc_h16_packed", "packed_fp16", x, _packed(w, wt, "h16"), flags)
        return launch("glass_conv2d_nhwc_h16", "direct_fp16", x, wt, flags)
    if rt.precision in ("fp16", "fp16s") and not winograd:
        # fp32 tensors in an fp16 mode (every layer of 'fp16', the fp32-input layers of 'fp16s': fusion conv, fc1 / fc2): where
        # a conv does enough work per input element, round the input to fp16 ONCE (glass_cast_f32_to_f16 - the rounding the
        # template applies while staging) and run the fp16-MFMA kernel on it; fp32 output and residual as they are
        if (rt.h16 and KH * KW * Cout >= 512 and x.numel() > 0 and
                lib().glass_conv_h16_supported(ctypes.byref(d), 1)):
            xh = torch.empty(x.shape, dtype=torch.float16, device=x.device)
            check(lib().glass_cast_f32_to_f16(c_void_p(_dev(x, "x")), c_void_p(_dev(xh)), ctypes.c_int64(x.numel()),
                                              c_void_p(stream_handle())), "glass_cast_f32_to_f16")
            return launch("glass_conv2d_nhwc_h16_packed", "packed_fp16", xh, _packed(w, wt, "h16"), 1)
        return launch("glass_conv2d_nhwc_f16", "direct_fp16", x, wt)
    # the weight-streaming 1x1 kernel on the wide layers with long k-loops.  Layer by layer (scripts/bench_conv.py) it is
    # ahead for every Cin >= 256 (256->1024 and 256->256 x1.08, 1024->256 x1.09, 512->128 x1.11, 512->2048 @32x32 x1.07;
    # behind on 128->512 x0.92 and 64->256 x0.87), but routing the narrow / small ones to it (Cout 128, 8192-pixel maps)
    # LOWERS the end-to-end rate: 274.1 images/s with this rule, 271 without the kernel, 267 with "Cin >= 256, >= 8192
    # pixels" (three alternating runs each, same box).  Round 4: the res5-size layers with BOTH channel counts >= 512 (8192
    # pixels at batch 8: 2048->512, 512->2048 +res, 1024->2048 s2, 1024->512 s2) are 7-13 % faster on it and worth +0.2 % end
    # to end (five alternating runs: 280.3 vs 279.7); the Cout = 128 layers still are not (279.5), and below 8192 pixels the
    # per-workgroup weight stream is not amortised (batch 2: 2-4x SLOWER than the direct kernel).
    px = N * Ho * Wo
    if winograd in ("pws9", "pws6"):                 # forced: the exact-product bf16-split 1x1 kernel (tests, scripts)
        if not lib().glass_pointwise_split_supported(ctypes.byref(d)):
            raise GlassLibraryError(f"winograd={winograd!r} but glass_pointwise_split_supported() rejects this layer")
        return launch("glass_conv1x1_pointwise_split_nhwc", "pointwise_split", x, _packed(w, wt, "pws"), int(winograd[3]))
    if (winograd is None and rt.split and KH == 1 and KW == 1 and _use_split(px, Cin, Cout) and
            lib().glass_pointwise_split_supported(ctypes.byref(d)) and (not isinstance(w, ConvWeight) or "pws" in w.packs)):
        return launch("glass_conv1x1_pointwise_split_nhwc", "pointwise_split", x, _packed(w, wt, "pws"), rt.split)
    if (winograd is None and rt.pw and KH == 1 and KW == 1 and
            (rt.pw == "all" or (Cin >= 256 and Cout >= 256 and px >= 16384) or (Cin >= 512 and Cout >= 512 and px >= 8192))
            and lib().glass_pointwise_supported(ctypes.byref(d))):
        return launch("glass_conv1x1_pointwise_nhwc", "pointwise", x, _packed(w, wt, "pw"))
    use_wino = rt.winograd if winograd is None else winograd
    if winograd is None and use_wino and KH == 3:
        # the Winograd kernel runs one 64-tile x 64-channel workgroup per CU: below ~96 workgroups (FPN p6, batch-2 res5)
        # the direct kernel's smaller tiles fill the chip better (measured 0.68-0.82x vs 1.15x at 128 workgroups)
        use_wino = ((N * ((H + 1) // 2) * ((W + 1) // 2) + 63) // 64) * (Cout // 64) >= 96
    # (a model's layer takes the split only if its load prepared the strip weights - fold_conv(..., ragged=True) - so that no
    #  launch of the model path ever packs; raw tensors (tests, scripts) get them packed for the launch)
    # odd widths (the local extractor's 16 x 33 maps): full tile columns on a Winograd kernel + the last pixel column as a strip
    # convolution.  A model's layer takes the split only if its load prepared the strip weights - fold_conv(..., ragged=True) -
    # so that no launch of the model path ever packs; raw tensors (tests, scripts) get them packed for the launch.
    strip_ok = (rt.ragged and KH == 3 and KW == 3 and W % 2 == 1 and W >= 5 and (2 * Cin) % 32 == 0 and out_cstride == 1 and
                ldx == Cin and res_mode in (0, 1) and x.numel() > 0 and (not isinstance(w, ConvWeight) or "col1" in w.packs))
    ragged = strip_ok and W % 4 == 1
    f43 = winograd == "f43" or (winograd is None and rt.f43 and use_wino and KH == 3 and _use_f43(N, H, W, Cout, Cin, ragged))
    f22_body = False
    t_alt = None                                    # modelled time of the launch the layer takes otherwise (small grids only)
    if winograd == "f22r":                          # forced: F(2x2) on the full tile columns + the last-column strip (tests)
        if not strip_ok:
            raise GlassLibraryError("winograd='f22r' needs an odd width >= 5, dense input, unit channel stride and res_mode 0/1")
        f22_body = True
    if (winograd is None and use_wino and not f43 and KH == 3 and KW == 3 and rt.small_grid and
            lib().glass_winograd_supported(ctypes.byref(d))):
        # the F(4x4) grid does not fill the chip (one image in flight): rounds x workgroup time decides
        kind, body, t_alt = _small_grid_3x3(N, H, W, Cout, Cin, strip_ok,
                                            rt.f43 and bool(lib().glass_winograd43_supported(ctypes.byref(d))))
        f43, ragged, f22_body = kind == "f43", kind == "f43" and body, kind == "f22" and body
    # round 6: the same layers - and the ones the rule above already sent to the implicit-GEMM kernel - as a SPLIT-K launch of the
    # F(4x4) kernel when the model says it is >= 15 % faster than what they take otherwise
    if (_F43_SPLITK and winograd is None and rt.winograd and rt.f43 and rt.small_grid and rt.splitk and rt.precision == "fp32" and KH == 3 and KW == 3 and
            _pair(stride) == (1, 1) and x.dtype == torch.float32 and out.dtype == torch.float32 and out_cstride == 1 and res_mode in (0, 1) and
            not _use_f43(N, H, W, Cout, Cin, strip_ok and W % 4 == 1) and (not isinstance(w, ConvWeight) or True in w.packs)):
        plan = _f43_splitk_plan(N, H, W, Cout, Cin, strip_ok and (not isinstance(w, ConvWeight) or "col1" in w.packs))
        if _TLS_force_f43k() is not None:           # (scripts/exp_f43_splitk.py: force a slice count to fit the model; 0 = never)
            fs = _TLS_force_f43k()                  # slices, or (slices, body)
            fs = fs if isinstance(fs, tuple) else (fs, False)
            plan, t_alt = ((int(fs[0]), bool(fs[1]) and strip_ok and W % 4 == 1, 0.0) if fs[0] else None), None
        if plan is not None:
            sl, body, t_sk = plan
            if t_alt is None:
                t_alt = _splitk_slices(rt, N * Ho * Wo, KH * KW * Cin, Cin, Cout, want_time=True)[1] if not use_wino else None
            if (t_alt is None or t_sk < (0.8 if body else 0.85) * t_alt) and lib().glass_winograd43_splitk_supported(ctypes.byref(d), sl):
                nbytes = int(lib().glass_winograd43_splitk_workspace_bytes(ctypes.byref(d), sl))
                ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
                check(lib().glass_conv3x3_winograd43_splitk_nhwc(
                    ctypes.byref(d), c_void_p(_dev(x, "x")), c_void_p(_dev(_packed(w, wt, True), "w")),
                    c_void_p(_dev(bias, "bias") if bias is not None else None),
                    c_void_p(_dev(residual, "residual") if residual is not None else None), c_void_p(_dev(out, "out")), int(sl), int(body),
                    c_void_p(_dev(ws)), ctypes.c_int64(nbytes), c_void_p(stream_handle())), "glass_conv3x3_winograd43_splitk_nhwc")
                if body:
                    _last_column_strip(x, _packed(w, wt, "col1"), bias, residual, out, d, out_coff)
                _TLS.last_path = "winograd43k"        # (bench / profiling: F(4x4) split-K + reduction)
                return out
    if use_wino and f43 and KH == 3 and KW == 3 and lib().glass_winograd43_supported(ctypes.byref(d)):
        if ragged:
            # width 4 k + 1: the F(4x4) kernel on the k full tile columns - 32 instead of 36 tiles per 16 x 33 map, and e.g.
            # 1024 instead of 1152 workgroups = 4 instead of 4.5 rounds on 256 CUs - and the last pixel column as a KH = 3,
            # KW = 1 convolution over the last two input columns seen as 2*Cin channels (x, y and the residual re-viewed as
            # [N,H,1,W*ld] rows with a channel offset; no copy)
            launch("glass_conv3x3_winograd43_body_nhwc", "winograd43", x, _packed(w, wt, True))
            _last_column_strip(x, _packed(w, wt, "col1"), bias, residual, out, d, out_coff)
            _TLS.last_path = "winograd43r"          # (bench / profiling: F(4x4) body + last-column strip)
            return out
        return launch("glass_conv3x3_winograd43_nhwc", "winograd43", x, _packed(w, wt, True))
    if winograd == "f43":
        raise GlassLibraryError("winograd='f43' but glass_winograd43_supported() rejects this layer")
    if use_wino and KH == 3 and KW == 3 and lib().glass_winograd_supported(ctypes.byref(d)):
        path = "winograd128" if lib().glass_winograd_block_channels(Cout, Cin) == 128 else "winograd"
        if f22_body:
            launch("glass_conv3x3_winograd_body_nhwc", path, x, _packed(w, wt, False))
            _last_column_strip(x, _packed(w, wt, "col1"), bias, residual, out, d, out_coff)
            _TLS.last_path = path + "r"             # (F(2x2) body + last-column strip)
            return out
        return launch("glass_conv3x3_winograd_nhwc", path, x, _packed(w, wt, False))
    if winograd:
        raise GlassLibraryError("winograd=True but glass_winograd_supported() rejects this layer")
    splits = _splitk_slices(rt, N * Ho * Wo, KH * KW * Cin, Cin, Cout) if (res_mode in (0, 1) or residual is None) else 0
    if splits > 1 and out.dtype == torch.float32 and x.dtype == torch.float32 and lib().glass_conv2d_splitk_supported(ctypes.byref(d), splits):
        # few output pixels and a long K (one image in flight: box head fc layers, res4 / res5 3x3, the 11-row predictors):
        # the k-tiles as `splits` independent slices of workgroups, partial sums through a workspace, ordered reduction
        nbytes = int(lib().glass_conv2d_splitk_workspace_bytes(ctypes.byref(d), splits))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
        _TLS.last_path = "direct"
        check(lib().glass_conv2d_nhwc_splitk(ctypes.byref(d), c_void_p(_dev(x, "x")), c_void_p(_dev(wt, "w")),
                                             c_void_p(_dev(bias, "bias") if bias is not None else None),
                                             c_void_p(_dev(residual, "residual") if residual is not None else None),
                                             c_void_p(_dev(out, "out")), splits, c_void_p(_dev(ws)), ctypes.c_int64(nbytes),
                                             c_void_p(stream_handle())), "glass_conv2d_nhwc_splitk")
        return out
    return launch("glass_conv2d_nhwc", "direct", x, wt)


_DUAL = os.environ.get("GLASS_PW_DUAL", "1") != "0"                # (A/B switch of the shortcut-into-conv3 fusion, read once)


def prepare_dual_weights(w1, w2) -> Optional["ConvWeight"]:
    """the [Cout,1,1,Cin1+Cin2] concatenation (w1 first) of two 1x1 layers that write the SAME output - a bottleneck block's
    shortcut and conv3 - with the bf16-split pack of conv1x1_dual_nhwc; None when the load's routing has no exact split kernel
    (fp16 modes, split 0 / 6) or the shapes do not fit it.  Load-time plumbing next to prepare_conv_weights."""
    r1, r2 = _raw(w1), _raw(w2)
    load = getattr(_TLS, "load", None)
    rt = load if load is not None else _DEFAULT
    if (not _DUAL or rt.precision != "fp32" or rt.split != 9 or not r1.is_cuda or r1.shape[1:3] != (1, 1) or r2.shape[1:3] != (1, 1) or
            r1.shape[0] != r2.shape[0] or r1.shape[0] % 128 or r1.shape[3] % 32 or r2.shape[3] % 32):
        return None
    cw = ConvWeight(torch.cat([r1, r2], dim=3).contiguous(), routing=load)
    cw.packs["pws"] = winograd_pack(cw.raw, "pws")
    if load is None:
        torch.cuda.current_stream().synchronize()
    return cw


def dual_supported(x1: torch.Tensor, x2: torch.Tensor, w, stride: int = 1, routing: Optional[Routing] = None) -> bool:
    """does conv1x1_dual_nhwc take this pair (and does the routing want it: same grid rule as the single-source split kernel)"""
    if w is None or not isinstance(w, ConvWeight) or "pws" not in w.packs:
        return False
    rt = routing if routing is not None else routing_of(w)
    if rt.precision != "fp32" or rt.split != 9 or x1.dtype != torch.float32 or x2.dtype != torch.float32:
        return False
    N, H, W, ldx = x1.shape
    Cout, Cin2 = w.raw.shape[0], x2.shape[3]
    Cin1 = w.raw.shape[3] - Cin2
    Ho, Wo = conv_out_size(H, W, 1, 1, stride, 0)
    if Cin1 <= 0 or ldx != Cin1 or tuple(x2.shape[:3]) != (N, Ho, Wo) or not _use_split(N * Ho * Wo, Cin1 + Cin2, Cout):
        return False
    d = ConvDesc(N, H, W, Cin1, Cout, 1, 1, stride, stride, 0, 0, Ho, Wo, ldx, Cout, 0, 1, 0, 0, 0)
    return bool(lib().glass_pointwise_split_dual_supported(ctypes.byref(d), Cin2, x2.shape[3]))


def conv1x1_dual_nhwc(x1: torch.Tensor, x2: torch.Tensor, w: "ConvWeight", bias: Optional[torch.Tensor] = None, *, stride: int = 1,
                      relu: int = 0) -> torch.Tensor:
    """y = act([x1 strided | x2] . w^T + bias): a bottleneck block's `relu(conv3(out) + shortcut(x))` (detectron2 BottleneckBlock behind
    reference glass/modeling/meta_arch/glass_rcnn.py:83 [d2-recall]) as ONE launch of the exact bf16-split kernel with one accumulator -
    x1 [N,H,W,Cin1] the block input (the shortcut's operand, `stride`), x2 [N,Ho,Wo,Cin2] conv2's output, `w` from prepare_dual_weights,
    `bias` the sum of the two folded biases.  The [N,Ho,Wo,Cout] shortcut map is neither written nor read back."""
    _f32c(x1, "x1"); _f32c(x2, "x2")
    N, H, W, ldx = x1.shape
    Cout, Cin2 = w.raw.shape[0], x2.shape[3]
    Cin1 = w.raw.shape[3] - Cin2
    Ho, Wo = conv_out_size(H, W, 1, 1, stride, 0)
    if tuple(x2.shape[:3]) != (N, Ho, Wo) or ldx != Cin1:
        raise GlassLibraryError(f"conv1x1_dual_nhwc: x1 {tuple(x1.shape)} (stride {stride}) and x2 {tuple(x2.shape)} do not share an output grid / weight {tuple(w.raw.shape)}")
    out = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=x1.device)
    d = ConvDesc(N, H, W, Cin1, Cout, 1, 1, stride, stride, 0, 0, Ho, Wo, ldx, Cout, 0, 1, relu, 0, 0)
    _TLS.last_path = "pointwise_split"
    check(lib().glass_conv1x1_pointwise_split_dual_nhwc(
        ctypes.byref(d), c_void_p(_dev(x1, "x1")), c_void_p(_dev(x2, "x2")), int(Cin2), int(x2.shape[3]), c_void_p(_dev(_packed(w, w.raw, "pws"), "w")),
        c_void_p(_dev(bias, "bias") if bias is not None else None), c_void_p(_dev(out, "out")), c_void_p(stream_handle())),
        "glass_conv1x1_pointwise_split_dual_nhwc")
    return out


def _last_column_strip(x: torch.Tensor, wcol: torch.Tensor, bias, residual, out: torch.Tensor, d: ConvDesc, out_coff: int) -> None:
    """output column W-1 of the 3x3 / pad-1 layer `d`: glass_conv2d_nhwc with KH 3, KW 1, pad (1, 0) on rows re-viewed as ONE
    pixel of W*ld channels - input channels [(W-2)*ldx, W*ldx) = the last two columns (needs ldx == Cin: dense rows), output
    channels [(W-1)*ldy + y_coff, ...) and the residual likewise.  Same stream, after the body launch."""
    N, H, W, Cin, Cout = d.N, d.H, d.W, d.Cin, d.Cout
    assert d.ldx == Cin, "the caller routes only dense inputs here"
    ds = ConvDesc(N, H, 1, 2 * Cin, Cout, 3, 1, 1, 1, 1, 0, H, 1, W * d.ldx, W * d.ldy, (W - 1) * d.ldy + out_coff, 1, d.relu,
                  d.res_mode, W * d.ldr if d.res_mode else 0)
    isz = x.element_size()
    xp = _dev(x, "x") + (W - 2) * d.ldx * isz
    rp = (_dev(residual, "residual") + (W - 1) * d.ldr * residual.element_size()) if (residual is not None and d.res_mode) else None
    check(lib().glass_conv2d_nhwc(ctypes.byref(ds), c_void_p(xp), c_void_p(_dev(wcol, "w")),
                                  c_void_p(_dev(bias, "bias") if bias is not None else None), c_void_p(rp), c_void_p(_dev(out, "out")),
                                  c_void_p(stream_handle())), "glass_conv2d_nhwc(last column)")


def _splitk_slices(rt: Routing, M: int, Ktot: int, Cin: int, Cout: int, want_time: bool = False):
    """k-slices for an implicit-GEMM launch that is latency-bound behind a long k-loop (0: a single slice).  Cost model from
    scripts/exp_small_grid.py on MI355X (64 x 64 tiles, 8 workgroups resident per CU; 128 x 32 for Cout <= 32): a workgroup
    alone on its CU takes ~0.75 us per 32-deep k-tile (load -> LDS -> barrier latency), w > 1 workgroups per CU take
    ~0.3 + 0.45 w us per k-tile of all of them (the CU retires one k-tile per ~0.45 us), so
    T(s) = nk / s * per_ktile(tiles * s / 256)  +  the partial sums once out and once back at ~4 TB/s  +  ~5 us for the
    reduction launch.  The split is taken when it saves >= 15 %.
    (box head fc1, 12544 -> 2048: 100 rows 298 -> 64 us with 8 slices, 800 rows 424 -> 348; res5 3x3 at 32 x 32: 112 -> 47.)"""
    if not rt.splitk or rt.precision != "fp32" or Cin % 32 != 0 or M <= 0:
        return (0, float("inf")) if want_time else 0
    nk = Ktot // 32
    if nk < 16:
        return (0, float("inf")) if want_time else 0
    tiles = -(-M // 128) if Cout <= 32 else -(-M // 64) * -(-Cout // 64)

    def t(s):
        wpc = max(1.0, tiles * s / NUM_CUS)
        extra = 0.0 if s == 1 else 2.0 * s * M * Cout * 4 / 4e6 + 5.0 + 0.3 * s
        return nk / s * (0.75 if wpc <= 1.0 else 0.3 + 0.45 * wpc) + extra
    best, tbest = 0, 0.85 * t(1)
    for s in (2, 3, 4, 6, 7, 8, 9, 12, 14, 16, 18, 24, 28, 32):
        if nk % s == 0 and nk // s >= 4 and t(s) < tbest:
            best, tbest = s, t(s)
    if want_time:
        return best, (tbest if best else t(1))
    return best


def linear(x: torch.Tensor, w, bias: Optional[torch.Tensor] = None, relu: int = 0,
           out: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None, precision: Optional[str] = None,
           routing: Optional[Routing] = None) -> torch.Tensor:
    """x [M,K] @ w[Nout,K]^T + bias on the same MFMA kernel (H = W = KH = KW = 1); few-row / long-K fp32 layers go through
    glass_conv2d_nhwc_splitk (conv2d_nhwc routes them)."""
    M, K = x.shape
    y = conv2d_nhwc(x.view(M, 1, 1, K), w if isinstance(w, ConvWeight) else w.view(w.shape[0], 1, 1, K), bias, relu=relu,
                    out=None if out is None else out.view(M, 1, 1, -1), out_dtype=out_dtype, precision=precision, routing=routing)
    return y.view(M, -1)


def _raw(w) -> torch.Tensor:
    return w.raw if isinstance(w, ConvWeight) else w


def local_stem_supported(x: torch.Tensor, w1, w2) -> bool:
    """the fused conv0_1 + conv0_2 + maxpool kernel takes fp32 NHWC4 crops with H, W multiples of 32 (precision fp32, or
    fp16s: the fp16-storage arithmetic, fp16 output)"""
    rt = routing_of(w1)
    w1, w2 = _raw(w1), _raw(w2)
    return (rt.precision in ("fp32", "fp16s") and rt.local_stem and x.dtype == torch.float32 and
            x.dim() == 4 and x.shape[-1] == 4 and tuple(w1.shape) == (16, 3, 3, 4) and tuple(w2.shape) == (32, 3, 3, 16) and
            bool(lib().glass_local_stem_supported(int(x.shape[1]), int(x.shape[2]))))


def local_stem_fused(x: torch.Tensor, w1, b1: torch.Tensor, w2, b2: torch.Tensor) -> torch.Tensor:
    """x [R,H,W,4] -> maxpool2x2(relu(conv3x3(relu(conv3x3(x, w1) + b1), w2) + b2)) [R,H/2,W/2,32] in one kernel; in 'fp16s'
    mode with the fp16 roundings of the unfused fp16-storage chain and an fp16 output."""
    h16 = routing_of(w1).precision == "fp16s"
    w1, w2 = _raw(w1), _raw(w2)
    for t, n in ((x, "x"), (w1, "w1"), (b1, "b1"), (w2, "w2"), (b2, "b2")):
        _f32c(t, n)
    R, H, W, _ = x.shape
    y = torch.empty((R, H // 2, W // 2, 32), dtype=torch.float16 if h16 else torch.float32, device=x.device)
    fn = lib().glass_local_stem_fused_h16 if h16 else lib().glass_local_stem_fused
    check(fn(c_void_p(_dev(x)), c_void_p(_dev(w1)), c_void_p(_dev(b1)), c_void_p(_dev(w2)), c_void_p(_dev(b2)), c_void_p(_dev(y)),
             R, H, W, c_void_p(stream_handle())), "glass_local_stem_fused")
    return y


def backbone_stem_supported(x: torch.Tensor, w) -> bool:
    """the fused 7x7 conv + ReLU + 3x3 max-pool kernel takes fp32 NHWC4 batches with H, W multiples of 4 under an fp32 routing"""
    rt = routing_of(w)
    wt = _raw(w)
    return (rt.precision == "fp32" and rt.stem and x.dtype == torch.float32 and x.dim() == 4 and x.shape[-1] == 4 and
            tuple(wt.shape) == (64, 7, 7, 4) and x.shape[0] > 0 and
            bool(lib().glass_backbone_stem_supported(int(x.shape[1]), int(x.shape[2]))))


def backbone_stem_fused(x: torch.Tensor, w, bias: torch.Tensor) -> torch.Tensor:
    """x [N,H,W,4] -> maxpool3x3s2p1(relu(conv7x7s2p3(x, w) + bias)) [N,H/4,W/4,64] in one kernel (csrc/backbone_stem.hip)"""
    wt = _raw(w)
    for t, n in ((x, "x"), (wt, "w"), (bias, "bias")):
        _f32c(t, n)
    N, H, W, _ = x.shape
    y = torch.empty((N, H // 4, W // 4, 64), dtype=torch.float32, device=x.device)
    check(lib().glass_backbone_stem_fused(c_void_p(_dev(x)), c_void_p(_dev(wt)), c_void_p(_dev(bias)), c_void_p(_dev(y)), N, H, W,
                                          c_void_p(stream_handle())), "glass_backbone_stem_fused")
    _TLS.last_path = "fused_stem"
    return y


def maxpool2d_nhwc(x: torch.Tensor, kernel, stride, padding=0) -> torch.Tensor:
    _fhc(x, "x")
    N, H, W, C = x.shape
    KH, KW = _pair(kernel)
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    Ho, Wo = (H + 2 * ph - KH) // sh + 1, (W + 2 * pw - KW) // sw + 1
    y = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
    fn = lib().glass_maxpool2d_nhwc_h16 if x.dtype == torch.float16 else lib().glass_maxpool2d_nhwc
    check(fn(c_void_p(_dev(x)), c_void_p(_dev(y)), N, H, W, C, KH, KW, sh, sw, ph, pw, Ho, Wo, c_void_p(stream_handle())),
          "glass_maxpool2d_nhwc")
    return y


def pixel_shuffle2x_nhwc(x: torch.Tensor) -> torch.Tensor:
    """x [N,H,W,4C] (channel = (a*2+b)*C + c) -> [N,2H,2W,C]: the scatter half of ConvTranspose2d(k=2, s=2)."""
    _f32c(x, "x")
    N, H, W, C4 = x.shape
    C = C4 // 4
    y = torch.empty((N, 2 * H, 2 * W, C), dtype=torch.float32, device=x.device)
    check(lib().glass_pixel_shuffle2x_nhwc(c_void_p(_dev(x)), c_void_p(_dev(y)), N, H, W, C, c_void_p(stream_handle())),
          "glass_pixel_shuffle2x_nhwc")
    return y


def mul_(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _f32c(a, "a"); _f32c(b, "b")
    if a.shape != b.shape:
        raise GlassLibraryError(f"mul_: shapes {tuple(a.shape)} vs {tuple(b.shape)}")
    check(lib().glass_mul_inplace(c_void_p(_dev(a)), c_void_p(_dev(b)), ctypes.c_int64(a.numel()), c_void_p(stream_handle())),
          "glass_mul_inplace")
    return a


def sigmoid_(x: torch.Tensor) -> torch.Tensor:
    _f32c(x, "x")
    check(lib().glass_sigmoid_inplace(c_void_p(_dev(x)), ctypes.c_int64(x.numel()), c_void_p(stream_handle())),
          "glass_sigmoid_inplace")
    return x


def paste_rotated_masks(masks: torch.Tensor, boxes: torch.Tensor, image_hw: Tuple[int, int], threshold: float = 0.5) -> torch.Tensor:
    """masks [R,M,M] float, boxes [R,5] -> bool [R,H,W] (threshold >= 0) or uint8 [R,H,W] (threshold < 0)."""
    _f32c(masks, "masks"); _f32c(boxes, "boxes")
    R, M, M2 = masks.shape
    if M != M2:
        raise GlassLibraryError("Only square mask predictions are supported")
    H, W = int(image_hw[0]), int(image_hw[1])
    out = torch.empty((R, H, W), dtype=torch.uint8, device=masks.device)
    if R:
        check(lib().glass_paste_rotated_masks(c_void_p(_dev(masks)), c_void_p(_dev(boxes)), R, M, H, W, c_float(threshold),
                                              c_void_p(_dev(out)), c_void_p(stream_handle())), "glass_paste_rotated_masks")
    return out.view(torch.bool) if threshold >= 0 else out


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
                      out: Optional[torch.Tensor] = None, out_coff: int = 0, out_cstride: int = 1, up2: bool = False) -> torch.Tensor:
    """feats: NHWC level tensors; boxes [R,5] float32; batch_idx [R] int32. Returns [R,PH,PW,C].  up2: the level tensors are
    HALF resolution and are pooled through nearest x2 upsampling (`scales` describe the upsampled map)."""
    R = boxes.shape[0]
    C = channels if channels is not None else feats[0].shape[-1]
    PH, PW = out_size
    if out is None:
        out = torch.empty((R, PH, PW, C), dtype=torch.float32, device=feats[0].device)
    d = RoiAlignDesc()
    d.num_levels = len(feats)
    half = feats[0].dtype == torch.float16
    for i, (f, s) in enumerate(zip(feats, scales)):
        _fhc(f, f"feat[{i}]")
        if f.dtype != feats[0].dtype:
            raise GlassLibraryError("all pyramid levels of one RoIAlign call must share a dtype")
        d.feat[i] = _dev(f)
        d.H[i], d.W[i], d.ld[i] = f.shape[1] * (2 if up2 else 1), f.shape[2] * (2 if up2 else 1), f.shape[3]
        d.scale[i] = float(s)
    d.min_level = int(round(-math.log2(scales[0])))
    d.C, d.PH, d.PW, d.sampling_ratio = C, PH, PW, int(sampling_ratio)
    d.ldy, d.y_coff, d.y_cstride = out.shape[-1], out_coff, out_cstride
    if R > 0:
        _f32c(boxes, "boxes")
        if batch_idx.dtype != torch.int32:
            raise GlassLibraryError("batch_idx must be int32")
        if up2 and half:
            raise GlassLibraryError("roi_align_rotated(up2=True) takes fp32 levels")
        fn = lib().glass_roi_align_rotated_up2 if up2 else (lib().glass_roi_align_rotated_h16 if half else lib().glass_roi_align_rotated)
        check(fn(ctypes.byref(d), c_void_p(_dev(boxes)), c_void_p(_dev(batch_idx)), R, c_void_p(_dev(_f32c(out, "out"))),
                 c_void_p(stream_handle())), "glass_roi_align_rotated")
    return out


# --------------------------------------------------------------------------- proposals
def _i32(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.int32 or not t.is_contiguous():
        raise GlassLibraryError(f"{name} must be contiguous int32")
    return t


class RpnLevel(ctypes.Structure):
    _fields_ = [("logits", c_void_p), ("deltas", c_void_p), ("cell_anchors", c_void_p)] + \
               [(n, c_int) for n in ("ldl", "ldd", "H", "W", "stride", "topk", "slot_off")]


def rpn_topk_decode(levels: Sequence[dict], N: int, A: int, anchor_offset: float, weights: Sequence[float],
                    out_boxes: torch.Tensor, out_scores: torch.Tensor, out_level: torch.Tensor) -> None:
    """levels: dicts with logits/deltas (device tensors whose data_ptr is the first element; may be channel
    slices of one head tensor), ldl/ldd (pixel strides), H, W, stride, cell_anchors, topk, slot_off."""
    L = len(levels)
    arr = (RpnLevel * L)()
    for i, lv in enumerate(levels):
        arr[i].logits, arr[i].deltas = _dev(lv["logits"]), _dev(lv["deltas"])
        arr[i].cell_anchors = _dev(lv["cell_anchors"])
        for k in ("ldl", "ldd", "H", "W", "stride", "topk", "slot_off"):
            setattr(arr[i], k, int(lv[k]))
    nbytes = int(lib().glass_rpn_workspace_bytes(N, L))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=out_boxes.device)
    w = (c_float * 5)(*[float(v) for v in weights])
    check(lib().glass_rpn_topk_decode(arr, L, N, A, c_float(anchor_offset), w, int(out_boxes.shape[1]),
                                      c_void_p(_dev(out_boxes)), c_void_p(_dev(out_scores)), c_void_p(_dev(out_level)),
                                      c_void_p(_dev(ws)), ctypes.c_int64(nbytes), c_void_p(stream_handle())),
          "glass_rpn_topk_decode")


NMS_CLIP, NMS_DROP_EMPTY = 1, 2


def zeros_views(fields, device) -> list:
    """[(shape, dtype float32 | int32) ...] -> zeroed tensors that are views of ONE flat buffer: one fill launch instead of one
    per output (padded outputs whose tails the kernels do not write must read zero; every view starts 16-byte aligned)"""
    sizes = [(int(math.prod(shape)) + 3) // 4 * 4 for shape, _ in fields]
    flat = torch.zeros((sum(sizes),), dtype=torch.float32, device=device)
    out, off = [], 0
    for (shape, dt), n in zip(fields, sizes):
        v = flat[off:off + int(math.prod(shape))]
        out.append((v if dt == torch.float32 else v.view(torch.int32)).view(shape))
        off += n
    return out


def rotated_nms_select(boxes: torch.Tensor, scores: torch.Tensor, cat: Optional[torch.Tensor],
                       valid_count: Optional[torch.Tensor], image_hw: torch.Tensor, score_thresh: float, nms_thresh: float,
                       post_topk: int, flags: int):
    """boxes [N,S,5], scores [N,S] -> (out_boxes [N,K,5], out_scores [N,K], out_index [N,K], out_count [N])."""
    _f32c(boxes, "boxes"); _f32c(scores, "scores"); _i32(image_hw, "image_hw")
    N, S = scores.shape
    dev = boxes.device
    ob, os_, oi, oc = zeros_views([((N, post_topk, 5), torch.float32), ((N, post_topk), torch.float32), ((N, post_topk), torch.int32),
                                   ((N,), torch.int32)], dev)
    check(lib().glass_rotated_nms_select(
        c_void_p(_dev(boxes)), c_void_p(_dev(scores)), c_void_p(_dev(_i32(cat, "cat")) if cat is not None else None),
        c_void_p(_dev(_i32(valid_count, "valid_count")) if valid_count is not None else None), N, S,
        c_void_p(_dev(image_hw)), c_float(score_thresh), c_float(nms_thresh), int(post_topk), int(flags),
        c_void_p(_dev(ob)), c_void_p(_dev(os_)), c_void_p(_dev(oi)), c_void_p(_dev(oc)), c_void_p(stream_handle())),
        "glass_rotated_nms_select")
    return ob, os_, oi, oc


def pairwise_iou_rotated(b1: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    _f32c(b1, "boxes1"); _f32c(b2, "boxes2")
    out = torch.zeros((b1.shape[0], b2.shape[0]), dtype=torch.float32, device=b1.device)
    if out.numel():
        check(lib().glass_pairwise_iou_rotated(c_void_p(_dev(b1)), b1.shape[0], c_void_p(_dev(b2)), b2.shape[0],
                                               c_void_p(_dev(out)), c_void_p(stream_handle())), "glass_pairwise_iou_rotated")
    return out


def box_decode(cls_logits: torch.Tensor, deltas: torch.Tensor, orient_logits: torch.Tensor, proposals: torch.Tensor,
               weights: Sequence[float]):
    R = proposals.shape[0]
    dev = proposals.device
    ob = torch.empty((R, 5), dtype=torch.float32, device=dev)
    fg = torch.empty((R,), dtype=torch.float32, device=dev)
    orr = torch.empty((R, 2), dtype=torch.float32, device=dev)
    w = (c_float * 5)(*[float(v) for v in weights])
    if R:
        for t, nm in ((cls_logits, "cls"), (deltas, "deltas"), (orient_logits, "orient"), (proposals, "proposals")):
            _f32c(t, nm)
        check(lib().glass_box_decode(c_void_p(_dev(cls_logits)), c_void_p(_dev(deltas)), c_void_p(_dev(orient_logits)),
                                     c_void_p(_dev(proposals)), R, w, c_void_p(_dev(ob)), c_void_p(_dev(fg)),
                                     c_void_p(_dev(orr)), c_void_p(stream_handle())), "glass_box_decode")
    return ob, fg, orr


# --------------------------------------------------------------------------- recognition
def pack_kblocked(w: torch.Tensor) -> torch.Tensor:
    """W [rows, K] -> packed [K/4, rows, 4] (see include/glass_hip.h); load-time plumbing."""
    rows, K = w.shape
    assert K % 4 == 0
    return w.reshape(rows, K // 4, 4).permute(1, 0, 2).contiguous()


def gc_attention_inplace(x: torch.Tensor, heads: int, w_mask, b_mask, w1, b1, ln_g, ln_b, w2, b2) -> torch.Tensor:
    """x [R,H,W,C] interleaved cat(local,global); modified in place."""
    _f32c(x, "x")
    R, H, W, C = x.shape
    P = w1.shape[0]
    if R:
        check(lib().glass_gc_attention_inplace(
            c_void_p(_dev(x)), R, H * W, C, int(heads), P, *[c_void_p(_dev(_f32c(t, "w"))) for t in
                                                           (w_mask, b_mask, w1, b1, ln_g, ln_b, w2, b2)],
            c_void_p(stream_handle())), "glass_gc_attention_inplace")
    return x


def mean_over_h(x: torch.Tensor) -> torch.Tensor:
    _f32c(x, "x")
    R, H, W, C = x.shape
    y = torch.empty((R, W, C), dtype=torch.float32, device=x.device)
    check(lib().glass_mean_over_h(c_void_p(_dev(x)), c_void_p(_dev(y)), R, H, W, C, c_void_p(stream_handle())),
          "glass_mean_over_h")
    return y


def new_handoff_status(device) -> torch.Tensor:
    """a zeroed device int for the `status` argument of bilstm_recurrence / attention_decode: the persistent kernels OR a bit
    into it when one of their bounded in-kernel waits gives up (bit 0 BiLSTM, bit 1 decoder) - the outputs of that call are
    then undefined.  The caller reads it back with whatever it reads back next (`HandoffGuard`)."""
    return torch.zeros((1,), dtype=torch.int32, device=device)


class HandoffGuard:
    """Fail-safe of the one-launch recurrent kernels in the product path (VERDICT r5 #2 / ADVICE r5 medium): a recognizer call
    made under a guard hands `guard.status` to its persistent launches and leaves in `guard.retry` a closure that re-runs the
    SAME encoder + decoder on the step kernels (`Routing.rnn = "steps"`, the kernels the persistent ones are tested against).
    Whoever owns the step's next host read-back adds `guard.status` to it and calls `guard.resolve(host_value)`:
    0 -> None; non-zero -> the re-computed text probabilities (logged once; after `STICKY_AFTER` give-ups of one owner the owner
    is told to stay on the step kernels: `guard.owner.rnn_override = "steps"`)."""
    STICKY_AFTER = 3
    __slots__ = ("status", "retry", "owner")

    def __init__(self, device, owner=None):
        self.status = new_handoff_status(device)
        self.retry = None
        self.owner = owner

    def resolve(self, host_status) -> Optional[torch.Tensor]:
        st = int(host_status)
        if st == 0 or self.retry is None:
            return None
        _note_handoff_giveup(st, self.owner)
        return self.retry()


_HANDOFF_LOG = {"count": 0}


def handoff_giveups() -> int:
    """how many recognizer calls of this process were re-run on the step kernels after a persistent kernel gave up a hand-off"""
    return _HANDOFF_LOG["count"]


def _note_handoff_giveup(st: int, owner) -> None:
    import logging
    _HANDOFF_LOG["count"] += 1
    log = logging.getLogger("glass_amd")
    n = getattr(owner, "rnn_giveups", 0) + 1 if owner is not None else _HANDOFF_LOG["count"]
    if owner is not None:
        owner.rnn_giveups = n
    if _HANDOFF_LOG["count"] == 1:
        log.warning("a persistent recurrent kernel gave up an in-kernel hand-off (status %d: bit 0 BiLSTM, bit 1 decoder) - the "
                    "workgroups of a chain were not co-resident in time (shared GPU, CU mask, debugger?).  The step's encoder / "
                    "decoder was re-run on the step kernels; results are unaffected.", st)
    if owner is not None and n == HandoffGuard.STICKY_AFTER and getattr(owner, "rnn_override", None) is None:
        owner.rnn_override = "steps"
        log.warning("%d hand-off give-ups: this model stays on the step kernels (Routing.rnn = 'steps') from here on", n)


def recurrence_test_hook(spin_limit: int = 0, withhold_ticket: int = -1) -> None:
    """TEST HOOK (glass_recurrence_test_hook): bound of the persistent kernels' waits in sweeps (0 = built-in) and the start
    ticket of a workgroup that never publishes (-1 = none); process-wide, for launches issued afterwards."""
    check(lib().glass_recurrence_test_hook(ctypes.c_int64(int(spin_limit)), int(withhold_ticket)), "glass_recurrence_test_hook")


def bilstm_recurrence(xg: torch.Tensor, w_hh_packed: torch.Tensor, hidden: int, mode=None,
                      status: Optional[torch.Tensor] = None) -> torch.Tensor:
    """xg [R,T,2,4*Hd] -> out [R,T,2*Hd].  `mode`: "steps" (one launch per time step, glass_bilstm_recurrence), "persistent" or
    (dirs, groups) per workgroup (ONE launch per layer, glass_bilstm_recurrence_persistent; bit-identical outputs); None = the
    raw-tensor default routing's `rnn`.  `status`: int32 [1] device word of this call's hand-off status (new_handoff_status)."""
    _f32c(xg, "xg"); _f32c(w_hh_packed, "w_hh_packed")
    R, T = xg.shape[0], xg.shape[1]
    out = torch.empty((R, T, 2 * hidden), dtype=torch.float32, device=xg.device)
    if R == 0:
        return out
    mode = _DEFAULT.rnn if mode is None else mode
    if mode != "steps":
        nd, ng = (0, 0) if mode == "persistent" else mode
        nbytes = int(lib().glass_bilstm_persistent_workspace_bytes(R, hidden))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=xg.device)
        check(lib().glass_bilstm_recurrence_persistent(c_void_p(_dev(xg)), c_void_p(_dev(w_hh_packed)), c_void_p(_dev(out)), R, T,
                                                       hidden, int(nd), int(ng),
                                                       c_void_p(_dev(_i32(status, "status")) if status is not None else None),
                                                       c_void_p(_dev(ws)), ctypes.c_int64(nbytes),
                                                       c_void_p(stream_handle())), "glass_bilstm_recurrence_persistent")
        return out
    nbytes = int(lib().glass_bilstm_workspace_bytes(R, hidden))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=xg.device)
    check(lib().glass_bilstm_recurrence(c_void_p(_dev(xg)), c_void_p(_dev(w_hh_packed)), c_void_p(_dev(out)), R, T, hidden,
                                        c_void_p(_dev(ws)), ctypes.c_int64(nbytes), c_void_p(stream_handle())),
          "glass_bilstm_recurrence")
    return out


def recurrence_status(reset: bool = True) -> int:
    """the device's sticky hand-off status (synchronises the device): 0 = every in-kernel wait of the persistent recurrent
    kernels was met; bit 0 / bit 1 = an LSTM / decoder hand-off gave up (outputs of that call are garbage)"""
    st = ctypes.c_int(0)
    check(lib().glass_recurrence_status(ctypes.byref(st), int(bool(reset))), "glass_recurrence_status")
    return int(st.value)


def attention_decode(x: torch.Tensor, xproj: torch.Tensor, weights: dict, roi_image: torch.Tensor, num_images: int,
                     num_classes: int, max_len: int, eos: int, mode=None, status: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x, xproj [R,T,D]; weights: dict of packed device tensors + 'temperature' float.  `mode` as in bilstm_recurrence:
    "steps" = two launches per decoding step (glass_attention_decode); anything else = ONE launch for all steps
    (glass_attention_decode_persistent: needs weights["sW_rm"], weights["emb_gi"] and a shape it supports, else the step
    kernels run); None = the raw-tensor default routing's `rnn`."""
    _f32c(x, "x"); _f32c(xproj, "xproj"); _i32(roi_image, "roi_image")
    R, T, D = x.shape
    out = torch.empty((R, max_len, num_classes), dtype=torch.float32, device=x.device)
    if R == 0:
        return out
    pred = torch.empty((R, max_len), dtype=torch.int32, device=x.device)
    w = DecoderWeights()
    for n in ("sW", "sB", "wW", "wB", "emb", "w_ih", "w_hh", "b_ih", "b_hh", "fcW", "fcB"):
        setattr(w, n, _dev(_f32c(weights[n], n)))
    w.temperature = float(weights["temperature"])
    mode = _DEFAULT.rnn if mode is None else mode
    if (mode != "steps" and "sW_rm" in weights and "emb_gi" in weights and
            lib().glass_decode_persistent_supported(T, D, int(num_classes), int(max_len))):
        nbytes = int(lib().glass_decode_persistent_workspace_bytes(R))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
        check(lib().glass_attention_decode_persistent(
            c_void_p(_dev(x)), c_void_p(_dev(xproj)), ctypes.byref(w), c_void_p(_dev(_f32c(weights["sW_rm"], "sW_rm"))),
            c_void_p(_dev(_f32c(weights["emb_gi"], "emb_gi"))), c_void_p(_dev(roi_image)), R, int(num_images), T, D, int(num_classes),
            int(max_len), int(eos), c_void_p(_dev(out)), c_void_p(_dev(pred)),
            c_void_p(_dev(_i32(status, "status")) if status is not None else None), c_void_p(_dev(ws)), ctypes.c_int64(nbytes),
            c_void_p(stream_handle())), "glass_attention_decode_persistent")
        return out
    nbytes = int(lib().glass_decode_workspace_bytes(R, D))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    check(lib().glass_attention_decode(c_void_p(_dev(x)), c_void_p(_dev(xproj)), ctypes.byref(w), c_void_p(_dev(roi_image)),
                                       R, int(num_images), T, D, int(num_classes), int(max_len), int(eos),
                                       c_void_p(_dev(out)), c_void_p(_dev(pred)), c_void_p(_dev(ws)), ctypes.c_int64(nbytes),
                                       c_void_p(stream_handle())),
          "glass_attention_decode")
    return out


def attention_decode_step(x: torch.Tensor, xproj: torch.Tensor, weights: dict, h: torch.Tensor, y_prev: torch.Tensor,
                          num_classes: int):
    """one decoder step (reference DecoderUnit.forward): x, xproj [R,T,D], state h [R,D], y_prev int32 [R] ->
    (logits [R,C], probabilities [R,C], next state [R,D])."""
    _f32c(x, "x"); _f32c(xproj, "xproj"); _f32c(h, "h"); _i32(y_prev, "y_prev")
    R, T, D = x.shape
    logits = torch.empty((R, num_classes), dtype=torch.float32, device=x.device)
    probs = torch.empty((R, num_classes), dtype=torch.float32, device=x.device)
    h_out = torch.empty((R, D), dtype=torch.float32, device=x.device)
    if R == 0:
        return logits, probs, h_out
    w = DecoderWeights()
    for n in ("sW", "sB", "wW", "wB", "emb", "w_ih", "w_hh", "b_ih", "b_hh", "fcW", "fcB"):
        setattr(w, n, _dev(_f32c(weights[n], n)))
    w.temperature = float(weights["temperature"])
    nbytes = int(lib().glass_decode_step_workspace_bytes(R, D))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    check(lib().glass_attention_decode_step(c_void_p(_dev(x)), c_void_p(_dev(xproj)), ctypes.byref(w), R, T, D, int(num_classes),
                                            c_void_p(_dev(h)), c_void_p(_dev(y_prev)), c_void_p(_dev(h_out)), c_void_p(_dev(logits)),
                                            c_void_p(_dev(probs)), c_void_p(_dev(ws)), ctypes.c_int64(nbytes), c_void_p(stream_handle())),
          "glass_attention_decode_step")
    return logits, probs, h_out


def detections_finalize(boxes: torch.Tensor, scores: torch.Tensor, orient: Optional[torch.Tensor],
                        text: Optional[torch.Tensor], counts: torch.Tensor, roi_start: Optional[torch.Tensor],
                        scale_xy: torch.Tensor, out_hw: torch.Tensor, min_box_dim: float, do_filter_small: bool):
    """Batched meta-arch postprocess. boxes [N,K,5] ... -> (boxes, scores, orient|None, text|None, counts) padded."""
    N, K, _ = boxes.shape
    dev = boxes.device
    TC = int(text.shape[1] * text.shape[2]) if text is not None else 0
    ob, os_, oo, ot, oc = zeros_views([((N, K, 5), torch.float32), ((N, K), torch.float32),
                                       ((N, K, 2) if orient is not None else (0,), torch.float32),
                                       ((N, K) + tuple(text.shape[1:]) if text is not None else (0,), torch.float32),
                                       ((N,), torch.int32)], dev)
    oo = oo if orient is not None else None
    ot = ot if text is not None else None
    _f32c(boxes, "boxes"); _f32c(scores, "scores"); _i32(counts, "counts"); _f32c(scale_xy, "scale_xy"); _i32(out_hw, "out_hw")
    opt = lambda t: c_void_p(_dev(t)) if t is not None else c_void_p(None)
    check(lib().glass_detections_finalize(
        c_void_p(_dev(boxes)), c_void_p(_dev(scores)), opt(orient), opt(text), c_void_p(_dev(counts)), opt(roi_start),
        c_void_p(_dev(scale_xy)), c_void_p(_dev(out_hw)), N, K, TC, c_float(min_box_dim), int(bool(do_filter_small)),
        c_void_p(_dev(ob)), c_void_p(_dev(os_)), opt(oo), opt(ot), c_void_p(_dev(oc)), c_void_p(stream_handle())),
        "glass_detections_finalize")
    return ob, os_, oo, ot, oc


def text_argmax(text: torch.Tensor, counts: torch.Tensor):
    """text [N,K,T,C] probability rows, counts int32 [N] -> (argmax int32 [N,K,T], max float32 [N,K,T]) of the rows of the
    boxes k < counts[n] (the others are left uninitialised).  reference text_encoder.py:81-151 `preds_prob.max(dim=2)`."""
    _f32c(text, "text"); _i32(counts, "counts")
    N, K, T, C = (int(v) for v in text.shape)
    arg = torch.empty((N, K, T), dtype=torch.int32, device=text.device)
    mx = torch.empty((N, K, T), dtype=torch.float32, device=text.device)
    check(lib().glass_text_argmax(c_void_p(_dev(text)), c_void_p(_dev(counts)), N, K, T, C, c_void_p(_dev(arg)), c_void_p(_dev(mx)),
                                  c_void_p(stream_handle())), "glass_text_argmax")
    return arg, mx


def pack_word_records(words: dict, max_det: int, steps: int) -> torch.Tensor:
    """the padded outputs of postprocess_words -> [N, 1 + max_det (16 + steps)] float32 word records (one launch)"""
    N, Kk = words["scores"].shape
    Tw = int(words["char"].shape[2])
    rec = torch.empty((N, 1 + max_det * (16 + steps)), dtype=torch.float32, device=words["scores"].device)
    check(lib().glass_pack_word_records(c_void_p(_dev(_f32c(words["boxes"], "boxes"))), c_void_p(_dev(_f32c(words["scores"], "scores"))),
                                        c_void_p(_dev(_f32c(words["text_score"], "text_score"))), c_void_p(_dev(_f32c(words["polygons"], "polygons"))),
                                        c_void_p(_dev(_i32(words["text_len"], "text_len"))), c_void_p(_dev(_i32(words["char"], "char"))),
                                        c_void_p(_dev(_i32(words["count"], "count"))), N, Kk, Tw, int(max_det), int(steps),
                                        c_void_p(_dev(rec)), c_void_p(stream_handle())), "glass_pack_word_records")
    return rec


def postprocess_words(boxes: torch.Tensor, scores: torch.Tensor, counts: torch.Tensor, text: Optional[torch.Tensor],
                      scale_xy: Optional[torch.Tensor], thresholds8: Sequence[float], stop_index: int) -> dict:
    """boxes [N,K,5], scores [N,K], counts int32 [N], text [N,K,T,C]|None -> dict of padded outputs (views of ONE zeroed
    buffer: one fill launch instead of eight)."""
    _f32c(boxes, "boxes"); _f32c(scores, "scores"); _i32(counts, "counts")
    N, K, _ = boxes.shape
    dev = boxes.device
    T, C = (int(text.shape[2]), int(text.shape[3])) if text is not None else (1, 1)
    fields = (("boxes", (N, K, 5), torch.float32), ("scores", (N, K), torch.float32), ("polygons", (N, K, 4, 2), torch.float32),
              ("src", (N, K), torch.int32), ("char", (N, K, T), torch.int32), ("text_score", (N, K), torch.float32),
              ("text_len", (N, K), torch.int32), ("count", (N,), torch.int32))
    sizes = [int(math.prod(shape)) for _, shape, _ in fields]
    flat = torch.zeros((sum(sizes),), dtype=torch.float32, device=dev)
    out, off = {}, 0
    for (name, shape, dt), n in zip(fields, sizes):
        v = flat[off:off + n]
        out[name] = (v if dt == torch.float32 else v.view(torch.int32)).view(shape)
        off += n
    thr = (c_float * 8)(*[float(v) for v in thresholds8])
    opt = lambda t: c_void_p(_dev(t)) if t is not None else c_void_p(None)
    arg = mx = None
    if text is not None:
        if T > 64:
            raise ValueError(f"postprocess_words: T={T} decoding steps (max 64)")
        arg, mx = text_argmax(text, counts)
    if scale_xy is not None:
        _f32c(scale_xy, "scale_xy")
    check(lib().glass_postprocess_words(
        c_void_p(_dev(boxes)), c_void_p(_dev(scores)), c_void_p(_dev(counts)), opt(arg), opt(mx), opt(scale_xy), N, K, T, thr,
        int(stop_index), c_void_p(_dev(out["boxes"])), c_void_p(_dev(out["scores"])), c_void_p(_dev(out["polygons"])),
        c_void_p(_dev(out["src"])), c_void_p(_dev(out["char"])), c_void_p(_dev(out["text_score"])),
        c_void_p(_dev(out["text_len"])), c_void_p(_dev(out["count"])), c_void_p(stream_handle())), "glass_postprocess_words")
    return out
