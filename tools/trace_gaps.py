"""GPU idle time between consecutive libvstab kernels of the LAST bench step, from a rocprofv3 kernel trace."""
import csv, glob, sys
f = glob.glob(f'{sys.argv[1]}/**/*_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'anonymous' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
# last step = from the last gray kernel onwards
starts = [i for i, r in enumerate(rows) if 'gray_area' in nm(r)]
seg = rows[starts[-1]:]
t0 = int(seg[0]['Start_Timestamp'])
busy = 0; prev_end = None; gaps = []
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy += e - s
    if prev_end is not None and s - prev_end > 3000:
        gaps.append(((s - prev_end) / 1e3, prev_name, nm(r)))
    prev_end, prev_name = e, nm(r)
span = (int(seg[-1]['End_Timestamp']) - t0) / 1e3
print(f"last step: {len(seg)} kernels, span {span:.0f} us, busy {busy/1e3:.0f} us, idle {span - busy/1e3:.0f} us")
for g in sorted(gaps, reverse=True)[:12]:
    print(f"  gap {g[0]:7.1f} us  after {g[1][:36]:36s} before {g[2][:36]}")
