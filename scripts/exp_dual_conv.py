"""How much would fusing a ResNet stage's first-block shortcut conv into its conv3 (+residual) save?  Times, per stage, the two
launches the step issues today against ONE 1x1 conv of K = K_conv3 + K_shortcut with the same output (the fused kernel's
work: it reads both inputs once and never writes / re-reads the shortcut tensor)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
import torch
from glass_amd.ops import native as K
dev = torch.device("cuda:0")


def t(fn, it=30):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


tot_now = tot_fused = 0.0
# (stage, N, H_in, W_in, C_in (block input), mid channels, C_out, stride of the block)
for name, N, H, W, Cin, mid, Cout, s in (("res2", 8, 256, 256, 64, 64, 256, 1), ("res3", 8, 256, 256, 256, 128, 512, 2),
                                         ("res4", 8, 128, 128, 512, 256, 1024, 2), ("res5", 8, 64, 64, 1024, 512, 2048, 2)):
    Ho, Wo = H // s, W // s
    x = torch.randn((N, H, W, Cin), device=dev)
    a = torch.randn((N, Ho, Wo, mid), device=dev)
    wsc = K.prepare_conv_weights(torch.randn((Cout, 1, 1, Cin), device=dev) * 0.05)
    w3 = K.prepare_conv_weights(torch.randn((Cout, 1, 1, mid), device=dev) * 0.05)
    b = torch.randn((Cout,), device=dev)
    cat = torch.randn((N, Ho, Wo, mid + Cin), device=dev)
    wcat = K.prepare_conv_weights(torch.randn((Cout, 1, 1, mid + Cin), device=dev) * 0.05)
    sc = K.conv2d_nhwc(x, wsc, b, stride=s)
    t_sc = t(lambda: K.conv2d_nhwc(x, wsc, b, stride=s))
    p_sc = K.last_conv_path()
    t_c3 = t(lambda: K.conv2d_nhwc(a, w3, b, relu=1, residual=sc, res_mode=1))
    p_c3 = K.last_conv_path()
    t_f = t(lambda: K.conv2d_nhwc(cat, wcat, b, relu=1))
    p_f = K.last_conv_path()
    print(f"{name}: shortcut {t_sc:.3f} ms ({p_sc}) + conv3+res {t_c3:.3f} ms ({p_c3}) = {t_sc + t_c3:.3f}  vs  one K={mid + Cin} conv {t_f:.3f} ms ({p_f})")
    tot_now += t_sc + t_c3; tot_fused += t_f
print(f"total per step: now {tot_now:.3f} ms, fused estimate {tot_fused:.3f} ms, saving {tot_now - tot_fused:.3f} ms")
