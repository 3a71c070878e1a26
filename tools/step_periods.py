"""Per bench step, from a rocprofv3 kernel trace of bench.py: the GPU span (gray start -> warp end), the idle stretch before the
next step's gray kernel, and the period (gray start -> next gray start).  python3 tools/step_periods.py DIR [first_n_steps]"""
import csv, glob, sys
f = glob.glob(f'{sys.argv[1]}/**/*_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
g = [i for i, r in enumerate(rows) if 'gray_area' in nm(r)]
limit = int(sys.argv[2]) if len(sys.argv) > 2 else len(g)
out = []
for a, b in zip(g[:-1], g[1:]):
    w = max(j for j in range(a, b) if nm(rows[j]).startswith('warp_kernel'))
    s0, e_w, s1 = int(rows[a]['Start_Timestamp']), int(rows[w]['End_Timestamp']), int(rows[b]['Start_Timestamp'])
    out.append(((e_w - s0) / 1e3, (s1 - e_w) / 1e3, (s1 - s0) / 1e3))
out = out[:limit]
print("step: GPU span us / idle before the next step us / period us")
for i, (sp, idle, per) in enumerate(out):
    print(f"  {i:3d}  {sp:8.1f}  {idle:7.1f}  {per:8.1f}")
