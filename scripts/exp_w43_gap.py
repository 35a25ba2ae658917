"""Where the F(4x4) kernel's time goes between "0.67 by the phase stamps" and "0.55 measured" (first half).
Runs the instrumented instantiation (GLASS_W43_ABL=4) once per layer shape; every workgroup records its phase stamps
(s_memtime), its life on the 100 MHz counter all CUs share (s_memrealtime) and the CU it ran on (HW_ID / XCC_ID), and the
library appends the raw records to $GLASS_W43_DBG_DUMP.  This script turns them into the per-shape table of
profiles/r06_w43_gap.txt:   span = launch skew + sum of block lives + dispatch gaps + tail, per CU, against the MFMA issue time.
    GLASS_W43_ABL=4 GLASS_W43_DBG_DUMP=/tmp/w43.bin python scripts/exp_w43_gap.py > gpurun_out/w43_gap.txt
"""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glass-text-spotting_amd"))
import numpy as np
import torch
from glass_amd.ops import native as K

assert os.environ.get("GLASS_W43_ABL") == "4" and os.environ.get("GLASS_W43_DBG_DUMP"), __doc__
dump = os.environ["GLASS_W43_DBG_DUMP"]
dev = torch.device("cuda:0")
SHADER_MHZ = float(os.environ.get("W43_SHADER_MHZ", "0"))       # from scripts/micro/clock_calib (0: report in ticks only)
LAYERS = [("FPN p2 out 256->256 @ 8x256x256", 8, 256, 256, 256, 256, False),
          ("FPN p3 out 256->256 @ 8x128x128", 8, 128, 128, 256, 256, False),
          ("local l3 256->256 @ 256x16x33 +res (full grid)", 256, 16, 33, 256, 256, True),
          ("fusion 512->256 @ 256x8x32", 256, 8, 32, 512, 256, False),
          ("local l2 128->128 @ 256x32x32", 256, 32, 32, 128, 128, False),
          ("res2 64->64 @ 8x256x256 (narrow)", 8, 256, 256, 64, 64, False),
          ("local l1 64->64 @ 256x64x64 +res (narrow)", 256, 64, 64, 64, 64, True)]


def read_records(path):
    out = []
    with open(path, "rb") as f:
        while True:
            hdr = f.read(32)
            if len(hdr) < 32:
                break
            nblk, nk, tiles_n, wide = struct.unpack("4q", hdr)
            out.append((nblk, nk, tiles_n, wide, np.frombuffer(f.read(nblk * 64), dtype=np.uint64).reshape(nblk, 8).astype(np.int64)))
    return out


print("# conv3x3_wino43_f32: per-shape accounting of one launch (instrumented instantiation, ~1 % slower than the product's)")
print("# ticks = s_memtime; us = s_memrealtime (100 MHz, shared by all CUs); MFMA issue time = nk x 576 (288 narrow) x 32 shader cycles per wave")
for name, N, H, W, Cin, Cout, res in LAYERS:
    x = torch.randn((N, H, W, Cin), device=dev)
    w = K.prepare_conv_weights(torch.randn((Cout, 3, 3, Cin), device=dev) * 0.05, "all")
    b = torch.randn((Cout,), device=dev)
    y = torch.empty((N, H, W, Cout), device=dev)
    r = torch.randn((N, H, W, Cout), device=dev) if res else None
    f = lambda: K.conv2d_nhwc(x, w, b, padding=1, relu=1, out=y, residual=r, res_mode=1 if res else 0, winograd="f43")
    f(); torch.cuda.synchronize()                       # warm (its record is skipped)
    open(dump, "wb").close()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    recs = read_records(dump)
    assert len(recs) >= 1, "no record dumped (GLASS_W43_ABL=4 build path not taken?)"
    nblk, nk, tiles_n, wide, h = recs[-1]
    ev_us = e0.elapsed_time(e1) * 1e3
    life = (h[:, 5] - h[:, 4]) / 100.0                                    # us
    span = (h[:, 5].max() - h[:, 4].min()) / 100.0
    # (s_memtime is a per-XCD counter: stamps of different blocks are not comparable, durations within a block are)
    tick_mhz = float((h[:, 3] - h[:, 0]).sum()) / float(life.sum())
    pro, loop, epi = (h[:, 1] - h[:, 0]).mean(), (h[:, 2] - h[:, 1]).mean(), (h[:, 3] - h[:, 2]).mean()
    cu_key = (h[:, 7] & 0xF) * 65536 + (h[:, 6] & 0xFF00)               # (xcc, se, sh, cu)
    keys = np.unique(cu_key)
    skew, busy, gaps, tail, cnt = [], [], [], [], []
    t0, t1 = h[:, 4].min(), h[:, 5].max()
    for k in keys:
        m = cu_key == k
        st, en = np.sort(h[m, 4]), np.sort(h[m, 5])
        cnt.append(int(m.sum()))
        skew.append((st[0] - t0) / 100.0)
        busy.append(float(((h[m, 5] - h[m, 4]) / 100.0).sum()))
        gaps.append(float(np.maximum(st[1:] - en[:-1], 0).sum() / 100.0) if len(st) > 1 else 0.0)
        tail.append((t1 - en[-1]) / 100.0)
    mfmas = nk * (576 if wide else 288)
    ex_flop = 2.0 * N * ((H + 3) // 4) * ((W + 3) // 4) * 36 * Cout * Cin
    print(f"\n## {name}: {nblk} workgroups = {nblk / 256:.2f} rounds of 256, nk = {nk}, {'wide 16 tiles x 128 ch' if wide else 'narrow 32 tiles x 64 ch'}")
    print(f"   launch by HIP events {ev_us:8.1f} us; first start -> last end {span:8.1f} us; executed {ex_flop / ev_us / 1e6:6.1f} TFLOP/s = {ex_flop / ev_us / 1e6 / 157.3:.3f} of the 2.4 GHz peak")
    print(f"   s_memtime (= shader cycles: scripts/micro/clock_calib.hip) ran at {tick_mhz:7.1f} MHz over the blocks' lives: THE SHADER CLOCK UNDER THIS KERNEL")
    print(f"   block life {life.mean():7.2f} us (min {life.min():.2f}, max {life.max():.2f}) = prologue {pro / tick_mhz:6.2f} + k-loop {loop / tick_mhz:7.2f} ({loop / nk / tick_mhz:.2f} per k-tile) + epilogue {epi / tick_mhz:6.2f} us")
    mf_us = mfmas * 32.0 / tick_mhz
    cyc = (h[:, 3] - h[:, 0]).mean()
    print(f"   cycles per block {cyc:9.0f} = prologue {pro:7.0f} + k-loop {loop:8.0f} ({loop / nk:7.0f} per k-tile, MFMA issue {mfmas / nk * 32:.0f} = {mfmas * 32 / loop:.3f} dense) + epilogue {epi:7.0f}; "
          f"MFMA issue {mfmas * 32} cycles = {mfmas * 32 / cyc:.3f} of the block")
    print(f"   => executed fraction of the 2.4 GHz peak = {mfmas * 32 / cyc:.3f} (cycles) x {tick_mhz / 2400:.3f} (clock {tick_mhz:.0f} / 2400) x {np.mean(busy) / span:.3f} (CU busy share of the span) "
          f"x {span / ev_us:.3f} (span / launch-to-launch by events) = {mfmas * 32 / cyc * tick_mhz / 2400 * np.mean(busy) / span * span / ev_us:.3f}")
    print(f"   CUs that ran blocks: {len(keys)}; blocks per CU min {min(cnt)} / mean {np.mean(cnt):.2f} / max {max(cnt)}")
    print(f"   per CU, mean (max): launch skew {np.mean(skew):6.2f} ({max(skew):.2f}) us | sum of block lives {np.mean(busy):8.2f} ({max(busy):.2f}) | "
          f"dispatch gaps between its blocks {np.mean(gaps):6.2f} ({max(gaps):.2f}) | idle tail {np.mean(tail):6.2f} ({max(tail):.2f})")
    tot = np.mean(skew) + np.mean(busy) + np.mean(gaps) + np.mean(tail)
    print(f"   span {span:.1f} us = skew {np.mean(skew) / span:.3f} + busy {np.mean(busy) / span:.3f} + gaps {np.mean(gaps) / span:.3f} + tail {np.mean(tail) / span:.3f} (sum {tot / span:.3f}); "
          f"events - span = {ev_us - span:.1f} us (launch + drain)")
