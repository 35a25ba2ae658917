import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "glass-text-spotting_amd"))
import torch
from glass_amd.ops import native as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [(8, 64, 64, 256, 256), (256, 16, 33, 256, 256), (8, 128, 128, 128, 128), (64, 8, 32, 512, 256), (8, 256, 256, 64, 64)]
bad = 0
for (N, H, W, Cin, Cout) in shapes:
    x = torch.randn((N, H, W, Cin), device=dev)
    w = torch.randn((Cout, 3, 3, Cin), device=dev) * 0.05
    b = torch.randn((Cout,), device=dev)
    r = torch.randn((N, H, W, Cout), device=dev)
    ref = K.conv2d_nhwc(x, w, b, padding=1, relu=1, residual=r, res_mode=1, winograd=True).clone()
    refd = K.conv2d_nhwc(x, w, b, padding=1, relu=1, residual=r, res_mode=1, winograd=False).clone()
    for it in range(40):
        y = K.conv2d_nhwc(x, w, b, padding=1, relu=1, residual=r, res_mode=1, winograd=True)
        yd = K.conv2d_nhwc(x, w, b, padding=1, relu=1, residual=r, res_mode=1, winograd=False)
        if not torch.equal(y, ref) or not torch.equal(yd, refd):
            bad += 1
    print((N, H, W, Cin, Cout), "max|wino-direct|/max", float((ref - refd).abs().max() / refd.abs().max()), "nondeterministic runs so far:", bad)
print("BAD", bad)
