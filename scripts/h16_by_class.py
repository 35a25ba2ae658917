"""conv_h16_kernel by layer class from a bench.py --conv-table file of an fp16s run: time, algorithmic TFLOP/s (of the 2500 fp16
peak) and algorithmic HBM traffic (fp16 in + out [+ residual] + weights; of 8 TB/s) per class - which roofline, if any, binds.
    python scripts/h16_by_class.py profiles/r05_conv_table_textocr_fp16s.txt"""
import collections, re, sys
rows = []
for ln in open(sys.argv[1]):
    m = re.match(r'\s*([\d.]+) ms\s+([\d.]+) TF/s\s+(\S+)\s+x\[([^\]]*)\] -> Cout (\d+) k(\d+)x(\d+) s(.*)', ln)
    if not m:
        continue
    ms, tf, path, x, co, kh, kw, rest = m.groups()
    N, H, W, C = [int(v) for v in x.split(',')]
    rows.append(dict(ms=float(ms), path=path, N=N, H=H, W=W, C=C, co=int(co), k=int(kh), res='+res' in rest,
                     s2=rest.strip().startswith('2') or '(2' in rest))
h = [r for r in rows if r['path'] == 'packed_fp16']


def cls(r, px):
    if r['k'] == 3 and r['C'] >= 256 and px >= 2e5:
        return 'A 3x3, Cin >= 256, >= 200 K pixels (FPN / RPN / res3-4 / local l3-4)'
    if r['k'] == 3:
        return 'B 3x3, Cin < 256 or small maps'
    if r['res']:
        return 'C 1x1 + residual'
    return 'D 1x1, Cin <= 128' if r['C'] <= 128 else 'E 1x1, Cin >= 256'


agg = collections.OrderedDict()
for r in h:
    px = r['N'] * r['H'] * r['W'] / (4 if r['s2'] else 1)
    fl = 2 * px * r['co'] * r['k'] * r['k'] * r['C']
    by = (r['N'] * r['H'] * r['W'] * r['C'] + px * r['co'] * (2 if r['res'] else 1)) * 2 + r['co'] * r['k'] * r['k'] * r['C'] * 2
    a = agg.setdefault(cls(r, px), [0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += r['ms']; a[2] += fl; a[3] += by
print(f"# {sys.argv[1]}: {len(h)} conv_h16_kernel launches, {sum(r['ms'] for r in h):.2f} ms per step")
print(f"{'class':70s} {'n':>3s} {'ms':>7s} {'TFLOP/s':>8s} {'of 2500':>8s} {'alg TB/s':>9s} {'of 8':>6s}")
for k, a in sorted(agg.items()):
    print(f"{k:70s} {a[0]:3d} {a[1]:7.3f} {a[2] / a[1] / 1e9:8.0f} {a[2] / a[1] / 1e9 / 2500:8.3f} {a[3] / a[1] / 1e9:9.2f} {a[3] / a[1] / 1e9 / 8:6.2f}")
