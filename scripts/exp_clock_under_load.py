import os, sys, subprocess, time, threading
sys.path.insert(0, "glass-text-spotting_amd")
import torch
from glass_amd.ops import native as K
dev = torch.device("cuda:0")
mode = sys.argv[1]
N, H, W, C = 8, 256, 256, 256
w = K.prepare_conv_weights(torch.randn((C, 3, 3, C), device=dev) * 0.05, "all")
b = torch.randn((C,), device=dev)
if mode == "fp16":
    K.set_conv_precision("fp16s")
    x = torch.randn((N, H, W, C), device=dev).half(); y = torch.empty((N, H, W, C), device=dev, dtype=torch.float16)
else:
    x = torch.randn((N, H, W, C), device=dev); y = torch.empty((N, H, W, C), device=dev)
f = lambda: K.conv2d_nhwc(x, w, b, padding=1, relu=1, out=y)
f(); torch.cuda.synchronize()
out = []
def smi():
    time.sleep(1.0)
    for _ in range(3):
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        out.append([l for l in r.splitlines() if "sclk" in l or "Power" in l or "mclk" in l])
        time.sleep(0.5)
t = threading.Thread(target=smi); t.start()
t0 = time.time(); n = 0
while time.time() - t0 < 4.0:
    for _ in range(50): f()
    torch.cuda.synchronize(); n += 50
el = time.time() - t0
t.join()
print(mode, K.last_conv_path(), f"{el / n * 1e3:.3f} ms/launch")
for o in out: print(o)
