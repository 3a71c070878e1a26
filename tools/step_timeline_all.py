"""Everything the GPU did in the LAST bench step, in time order: every kernel (the library's, torch's, the runtime's fill /
copy kernels; memory copies too if the trace has them -- rocprofv3's --memory-copy-trace crashed on this pool), with the idle time before each -- from a rocprofv3 trace taken with
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 5 --warmup 2 --no-extras --no-checks --cpu-frames 0
    python3 tools/step_timeline_all.py DIR"""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(f'{d}/**/*_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K', name.split('(')[0][:70], r.get('Queue_Id', '?')))
for f in glob.glob(f'{d}/**/*_memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C', f"copy {r.get('Direction', '?')} {r.get('Size', r.get('Bytes', '?'))} B", '-'))
ev.sort()
starts = [i for i, e in enumerate(ev) if e[2] == 'K' and 'gray_area' in e[3]]
last_warp = max(i for i, e in enumerate(ev) if e[2] == 'K' and e[3].startswith('warp_kernel'))
# bench.py runs four more passes after its timed steps (three with every stage's events, one with the DIS detail events): the
# last TIMED step is the fifth from the end (argv[2] = how many steps to go back; 0 = the very last one)
back = int(sys.argv[2]) if len(sys.argv) > 2 else 4
first = starts[-1 - back]
warps = [i for i, e in enumerate(ev) if e[2] == 'K' and e[3].startswith('warp_kernel') and i > first]
last_warp = warps[0]
seg = ev[first:last_warp + 1]
t0 = seg[0][0]
prev_end = t0
idle = 0.0
for s, e, kind, name, q in seg:
    gap = (s - prev_end) / 1e3
    if gap > 0: idle += gap
    print(f"{(s - t0) / 1e3:9.1f} us  gap {gap:6.1f}  +{(e - s) / 1e3:8.1f} us  {kind} q{q:>2}  {name}")
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, idle (no kernel or copy in flight on any queue) {idle:.1f} us")
