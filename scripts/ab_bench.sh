#!/bin/bash
# End-to-end A/B of bench.py configurations given as environment assignments, three alternating rounds each (box-to-box and
# thermal drift is +-1 %, larger than most routing effects - alternate, never compare across gpurun calls):
#   gpurun -- 'bash scripts/ab_bench.sh "GLASS_POINTWISE=1" "GLASS_POINTWISE=0"'
# usage: ab.sh "VAR=val VAR2=val" "VAR=val" ...   (each arg one configuration; 3 alternating rounds)
for i in 1 2 3; do
  for cfg in "$@"; do
    echo -n "[$cfg] "
    env $cfg timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"
  done
done
