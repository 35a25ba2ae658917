"""Package power and shader clock while ONE 1x1 layer runs back to back on the fp32-MFMA kernel and on the bf16-split kernel
(nine and six products): is the split kernel held by the power cap?   python scripts/exp_pw_split_power.py [seconds]"""
import os, sys
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ["3"])
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import power_clock_log as P   # noqa: E402  (its sampler; SECONDS from argv)
import torch                  # noqa: E402
from glass_amd.ops import native as K   # noqa: E402
dev = torch.device("cuda:0")
_print = print


def quiet_leg(name, fn, flop):
    """P.leg prints every sample: keep the summary lines"""
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        P.leg(name, fn, flop)
    for ln in buf.getvalue().splitlines():
        if ln.startswith("==") or "steady" in ln:
            _print(ln)


for (N, H, W, Cin, Cout, st, has_res) in [(8, 256, 256, 256, 256, 1, 1), (8, 64, 64, 1024, 2048, 2, 0)]:
    x = torch.randn((N, H, W, Cin), device=dev)
    w_raw = torch.randn((Cout, 1, 1, Cin), device=dev) * (1.0 / Cin ** 0.5)
    w = K.prepare_conv_weights(w_raw, "fp32")
    w.packs["pws"] = K.winograd_pack(w_raw, "pws")
    Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
    res = torch.randn((N, Ho, Wo, Cout), device=dev) if has_res else None
    y = torch.empty((N, Ho, Wo, Cout), device=dev)
    for force in (None, "pws9", "pws6"):
        quiet_leg(f"[{N},{H},{W},{Cin}]->{Cout} s{st} {force or 'fp32 MFMA (routed)'}",
                  lambda: K.conv2d_nhwc(x, w, None, stride=st, relu=1, residual=res, res_mode=1 if has_res else 0, out=y, winograd=force,
                                            routing=K.default_routing().replace(split=0) if force is None else None),
                  2.0 * N * Ho * Wo * Cin * Cout)
