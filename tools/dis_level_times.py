"""Per-launch durations of the DIS kernels of the LAST bench step, in launch order, from a rocprofv3 kernel trace:
     rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 5 --warmup 2 --no-extras --no-checks --cpu-frames 0
     python3 tools/dis_level_times.py DIR"""
import csv, glob, sys
f = glob.glob(f'{sys.argv[1]}/**/*_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'anonymous' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
starts = [i for i, r in enumerate(rows) if 'gray_area' in nm(r)]
seg = rows[starts[-1]:]
t0 = int(seg[0]['Start_Timestamp'])
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id', '?'):>3}  {nm(r)[:60]}  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?'))}")
