"""Prints chosen fields of a bench.py JSON line read from stdin:  python bench.py ... | python tools/line_fields.py value ms_per_step rank0_stage_ms"""
import json, sys
line = json.loads([x for x in sys.stdin.read().strip().splitlines() if x.startswith("{")][-1])
def get(d, path):
    for k in path.split("."):
        d = d.get(k) if isinstance(d, dict) else None
    return d
print(*[get(line, f) for f in sys.argv[1:]])
