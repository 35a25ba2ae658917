"""HBM ceiling probe for the 1x1+residual epilogue pattern: c = relu(a + b) and a plain copy, 537 MB tensors."""
import torch
dev = torch.device("cuda:0")
n = 8 * 256 * 256 * 256
a = torch.randn(n, device=dev); b = torch.randn(n, device=dev); c = torch.empty_like(a)
def timed(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
ms = timed(lambda: torch.add(a, b, out=c))
print(f"add  (2 reads + 1 write of {n*4/1e6:.0f} MB): {ms:.3f} ms  {3*n*4/ms/1e9:.2f} TB/s")
ms = timed(lambda: c.copy_(a))
print(f"copy (1 read + 1 write): {ms:.3f} ms  {2*n*4/ms/1e9:.2f} TB/s")
ms = timed(lambda: torch.relu_(c))
print(f"relu_ in place (1 read + 1 write same lines): {ms:.3f} ms  {2*n*4/ms/1e9:.2f} TB/s")
ms = timed(lambda: a.sum())
print(f"sum (1 read): {ms:.3f} ms  {n*4/ms/1e9:.2f} TB/s")
