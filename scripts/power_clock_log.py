"""Package power and shader clock at ~10 Hz while ONE kernel family runs back to back (VERDICT r4 #4b: put the
"power-bound" claim on file).  Run on the GPU box:

    python scripts/power_clock_log.py [seconds per kernel] > gpurun_out/power_clock.txt

Sampler: the amdgpu hwmon files (power1_average / power1_input in microwatt, freq1_input in Hz) read directly - a
`rocm-smi` process per sample takes 0.3 s and perturbs the host; `rocm-smi --showpower --showclocks` is called once per
kernel as a cross-check and its lines are printed verbatim.  Kernels: the F(4x4) Winograd 3x3 layer of FPN p2
(conv3x3_wino43_f32), a res4-size 1x1 layer (conv1x1_pw_f32), the same 3x3 layer in the fp16-storage mode
(conv_h16_kernel), and an idle leg."""
import glob
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "glass-text-spotting_amd"))
import torch  # noqa: E402
from glass_amd.ops import native as K  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
dev = torch.device("cuda:0")


def hwmon_files():
    """the hwmon directory of the GPU this process computes on: the box exposes every card of the node under /sys, the
    container sees one - match the PCI address torch reports for cuda:0"""
    out = {}
    pr = torch.cuda.get_device_properties(0)
    bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', 0):02x}:{getattr(pr, 'pci_device_id', 0):02x}"
    cards = [c for c in glob.glob("/sys/class/drm/card*") if bdf in os.path.realpath(os.path.join(c, "device"))]
    print("cuda:0 PCI", bdf, "-> drm cards", cards, "(of", len(glob.glob("/sys/class/drm/card*/device/hwmon")), "with hwmon)")
    for hw in [h for c in cards for h in glob.glob(os.path.join(c, "device/hwmon/hwmon*"))]:
        for name in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "temp2_input", "power1_cap"):
            p = os.path.join(hw, name)
            if os.path.exists(p):
                out.setdefault(name, p)
    return out


FILES = hwmon_files()
SMI = []          # (leg, t, text): a rocm-smi sample every ~0.7 s from its own thread (the tool takes ~0.3 s per call)


def smi_loop(stop, name, t0):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        except Exception as e:   # noqa: BLE001
            r = f"rocm-smi unavailable: {e}"
        pw = [ln.split(":")[-1].strip() for ln in r.splitlines() if "Power (W)" in ln]
        sc = [ln.split("(")[-1].rstrip(")") for ln in r.splitlines() if "sclk clock level" in ln]
        SMI.append((name, time.time() - t0, pw[0] if pw else "?", sc[0] if sc else "?"))
        stop.wait(0.4)


def read(name):
    try:
        with open(FILES[name]) as f:
            return int(f.read().strip())
    except Exception:   # noqa: BLE001 - a missing sensor is reported as None
        return None


def sample_loop(stop, rows):
    t0 = time.time()
    while not stop.is_set():
        pw = read("power1_average") if "power1_average" in FILES else read("power1_input")
        rows.append((time.time() - t0, pw, read("freq1_input"), read("freq2_input"), read("temp2_input") or read("temp1_input")))
        time.sleep(0.1)


def leg(name, fn, flop_per_call):
    stop, rows = threading.Event(), []
    th = threading.Thread(target=sample_loop, args=(stop, rows))
    if fn is not None:
        fn()
        torch.cuda.synchronize()
    th.start()
    t0 = time.time()
    ts = threading.Thread(target=smi_loop, args=(stop, name, t0))
    ts.start()
    n = 0
    smi = ""
    while time.time() - t0 < SECONDS:
        if fn is None:
            time.sleep(0.2)
        else:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            n += 50
    el = time.time() - t0
    stop.set()
    th.join()
    ts.join()
    print(f"== {name}: {n} launches in {el:.2f} s" + (f" = {el / n * 1e3:.4f} ms/launch, {flop_per_call * n / el / 1e12:.1f} TFLOP/s (direct count)" if n else ""))
    print("   t_s   power_W   sclk_MHz  mclk_MHz  temp_C")
    for t, pw, f1, f2, tc in rows:
        print(f"  {t:5.2f}  {pw / 1e6 if pw else float('nan'):8.1f}  {f1 / 1e6 if f1 else float('nan'):8.0f}  {f2 / 1e6 if f2 else float('nan'):8.0f}  "
              f"{tc / 1e3 if tc else float('nan'):6.1f}")
    steady = [r for r in rows if r[0] > 1.0 and r[1]]
    if steady:
        print(f"   steady (t > 1 s): power mean {sum(r[1] for r in steady) / len(steady) / 1e6:.0f} W, "
              f"sclk mean {sum((r[2] or 0) for r in steady) / len(steady) / 1e6:.0f} MHz "
              f"(min {min((r[2] or 0) for r in steady) / 1e6:.0f}, max {max((r[2] or 0) for r in steady) / 1e6:.0f})")
    mine = [x for x in SMI if x[0] == name]
    print("   rocm-smi (--showpower --showclocks), one line per call: t_s  package power W  sclk")
    for _, t, pw, sc in mine:
        print(f"     {t:5.2f}  {pw:>8s}  {sc}")


def leg_bench(name, argv):
    """the sampler around the real workload: bench.py's own step loop in a child process"""
    stop, rows = threading.Event(), []
    th = threading.Thread(target=sample_loop, args=(stop, rows))
    th.start()
    t0 = time.time()
    ts = threading.Thread(target=smi_loop, args=(stop, name, t0))
    ts.start()
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py")] + argv,
                       capture_output=True, text=True)
    stop.set()
    th.join()
    ts.join()
    print(f"== {name}: python bench.py {' '.join(argv)} -> {r.stdout.strip()[:160]}")
    print("   (model build + warm-up first, the timed loop is the high-power stretch at the end)")
    print("   t_s   power_W   sclk_MHz")
    for t, pw, f1, f2, tc in rows[::2]:
        print(f"  {t:5.2f}  {pw / 1e6 if pw else float('nan'):8.1f}  {f1 / 1e6 if f1 else float('nan'):8.0f}")
    print("   rocm-smi, one line per call: t_s  package power W  sclk")
    for _, t, pw, sc in [x for x in SMI if x[0] == name]:
        print(f"     {t:5.2f}  {pw:>8s}  {sc}")


def main():
    print("hwmon files:", {k: v for k, v in FILES.items()})
    cap = read("power1_cap")
    print("power cap:", cap / 1e6 if cap else None, "W")
    leg("idle", None, 0)
    N, H, W, C = 8, 256, 256, 256
    x = torch.randn((N, H, W, C), device=dev)
    y = torch.empty((N, H, W, C), device=dev)
    b = torch.randn((C,), device=dev)
    w3 = K.prepare_conv_weights(torch.randn((C, 3, 3, C), device=dev) * 0.05, "all")
    leg("conv3x3 256->256 at 8x256x256 (FPN p2 output layer), fp32", lambda: K.conv2d_nhwc(x, w3, b, padding=1, relu=1, out=y), 2.0 * N * H * W * C * 9 * C)
    print("   path:", K.last_conv_path())
    N1, H1, C1, C2 = 8, 64, 1024, 256
    x1 = torch.randn((N1, H1, H1, C1), device=dev)
    y1 = torch.empty((N1, H1, H1, C2), device=dev)
    w1 = K.prepare_conv_weights(torch.randn((C2, 1, 1, C1), device=dev) * 0.05, "all")
    b1 = torch.randn((C2,), device=dev)
    leg("conv1x1 1024->256 at 8x64x64 (res4 conv1), fp32", lambda: K.conv2d_nhwc(x1, w1, b1, relu=1, out=y1), 2.0 * N1 * H1 * H1 * C1 * C2)
    print("   path:", K.last_conv_path())
    K.set_conv_precision("fp16s")
    xh, yh = x.half(), torch.empty((N, H, W, C), device=dev, dtype=torch.float16)
    w3h = K.prepare_conv_weights(torch.randn((C, 3, 3, C), device=dev) * 0.05, "all")
    leg("conv3x3 256->256 at 8x256x256, fp16 storage mode", lambda: K.conv2d_nhwc(xh, w3h, b, padding=1, relu=1, out=yh), 2.0 * N * H * W * C * 9 * C)
    print("   path:", K.last_conv_path())
    K.set_conv_precision("fp32")
    del x, y, xh, yh, x1, y1
    torch.cuda.empty_cache()
    leg_bench("the bench step loop, fp32 (300 steps of 8 images)", ["--steps", "300", "--warmup", "3", "--no-extras", "--no-cpu-baseline"])
    leg("idle again", None, 0)


if __name__ == "__main__":
    main()
