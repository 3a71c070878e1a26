"""Every kernel of the LAST step of a Flow -> Motion Apply chain (bench.py --workload c3 | c5) in time order, with the idle time before each
and up to the next step's first kernel -- from   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --workload c5 --steps 4 --warmup 1
    python3 tools/chain_timeline.py DIR"""
import csv, glob, sys
ev = []
for f in glob.glob(f'{sys.argv[1]}/**/*_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name.split('(')[0][:60]))
ev.sort()
starts = [i for i, e in enumerate(ev) if 'gray_area' in e[2]]
a, b = starts[-2], starts[-1]          # the second-to-last step, up to the first kernel of the last one
t0, prev_end, idle = ev[a][0], ev[a][0], 0.0
for s, e, name in ev[a:b + 1]:
    gap = (s - prev_end) / 1e3
    if gap > 0: idle += gap
    if gap > 15 or (e - s) > 200e3 or name.startswith(('warp', 'plan', 'fit', 'gray')):
        print(f"{(s - t0) / 1e3:9.1f} us  gap {gap:7.1f}  +{(e - s) / 1e3:9.1f} us  {name}")
    prev_end = max(prev_end, e)
print(f"step period {(ev[b][0] - t0) / 1e3:.1f} us, of it idle (no kernel in flight) {idle:.1f} us")
