"""stdin: one bench.py JSON line -> 'value ms_per_step latency' (helper of the scripts/r06_*_ab.sh A/B scripts)"""
import json, sys
d = json.loads(sys.stdin.read())
print(round(d["value"], 1), round(d["ms_per_step"], 2), round(d.get("latency_ms_per_step") or 0.0, 3))
