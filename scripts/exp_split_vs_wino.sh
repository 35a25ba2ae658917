#!/bin/bash
# does the F(4x4) kernel run slower in a step whose 1x1 layers are on the bf16-split kernel?  serial conv meter of bench.py, alternating, one box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for i in 1 2 3; do
  for sp in 0 9; do
    GLASS_PW_SPLIT=$sp python bench.py --no-cpu-baseline --no-extras --steps 40 --conv-table gpurun_out/ct_${sp}_$i.txt > gpurun_out/ct_${sp}_$i.json 2>/dev/null
    python - <<PY
import re, json, collections
tot = collections.Counter()
for l in open("gpurun_out/ct_${sp}_$i.txt"):
    m = re.match(r"\s*([\d.]+) ms\s+([\d.]+) TF/s\s+(\w+)", l)
    if m: tot[m.group(3)] += float(m.group(1))
d = json.load(open("gpurun_out/ct_${sp}_$i.json"))
print("split=$sp run $i: value %.1f  frac %.3f | " % (d["value"], d["roofline"]["frac"]) + "  ".join(f"{k} {v:.2f}" for k, v in sorted(tot.items())) + "  | sum %.2f" % sum(tot.values()))
PY
  done
done
