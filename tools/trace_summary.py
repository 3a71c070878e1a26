"""Per-kernel per-step summary from a rocprofv3 kernel trace directory."""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(f'{sys.argv[1]}/*/*_kernel_trace.csv')[0]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = [r for r in csv.DictReader(open(f)) if 'anonymous' in r['Kernel_Name']]
d = defaultdict(list)
for r in rows:
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
    d[nm].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    per = len(v) // steps
    print(f"{k:40s} n={len(v):4d} per-step={sum(v)/steps/1e3:7.3f} ms  last: {[round(x,1) for x in v[-per:]] if per <= 8 else ''}")
