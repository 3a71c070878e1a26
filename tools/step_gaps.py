import csv, glob, sys
f = glob.glob(f'{sys.argv[1]}/**/*_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
g = [i for i, r in enumerate(rows) if 'gray_area' in nm(r)]
w = [i for i, r in enumerate(rows) if nm(r).startswith('warp_kernel')]
# steady-state: idle between the warp's end of step k and the gray kernel's start of step k+1
gaps = []
for i in g[1:]:
    prev_w = max(j for j in w if j < i)
    gaps.append((int(rows[i]['Start_Timestamp']) - int(rows[prev_w]['End_Timestamp'])) / 1e3)
    between = [nm(rows[j]) for j in range(prev_w + 1, i)]
per = [(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3 for a, b in zip(g[:-1], g[1:])]
print("step period (gray start to gray start), last 10:", [round(x) for x in per[-10:]])
print("idle warp end -> next gray start, last 10:", [round(x) for x in gaps[-10:]])
print("kernels in between (last):", between)
