"""Which kernels write more bytes than their output?  Run under `rocprofv3 --pmc WRITE_SIZE --kernel-trace` (and FETCH_SIZE):
every kernel below produces (or copies) the SAME 8 x 256 x 256 x 256 fp32 tensor = 536.9 MB, three launches each."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "glass-text-spotting_amd"))
import torch
from glass_amd.ops import native as K
dev = torch.device("cuda:0")
N, H, W, C = 8, 256, 256, 256
x = torch.randn((N, H, W, C), device=dev)
y = torch.empty((N, H, W, C), device=dev)
r = torch.randn((N, H, W, C), device=dev)
b = torch.randn((C,), device=dev)
w3 = K.prepare_conv_weights(torch.randn((C, 3, 3, C), device=dev) * 0.05, "fp32")
w1 = K.prepare_conv_weights(torch.randn((C, 1, 1, C), device=dev) * 0.05, "fp32")
R0 = K.default_routing()
for _ in range(3):
    K.conv2d_nhwc(x, w3, b, padding=1, relu=1, out=y)                                      # conv3x3_wino43_f32 (FPN p2 layer)
    K.conv2d_nhwc(x, w3, b, padding=1, relu=1, out=y, residual=r, res_mode=1)             # ... with a residual
    K.conv2d_nhwc(x, w3, b, padding=1, relu=1, out=y, winograd=True)                       # conv3x3_wino128_f32
    K.conv2d_nhwc(x, w3, b, padding=1, relu=1, out=y, winograd=False)                      # conv_igemm_f32 128x128
    K.conv2d_nhwc(x, w1, b, relu=1, out=y)                                                 # conv1x1_pw_f32
    K.conv2d_nhwc(x, w1, b, relu=1, out=y, routing=R0.replace(pw=False))                   # conv_igemm_f32 on the 1x1
    y.copy_(x)                                                                             # a plain device copy (calibration)
    K.maxpool2d_nhwc(x, 1, 1)                                                              # maxpool k1 s1 = a copy through our own kernel
torch.cuda.synchronize()
print("done")
