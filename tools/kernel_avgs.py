"""Average duration per kernel from a rocprofv3 --stats directory: python tools/kernel_avgs.py DIR [substr ...]"""
import csv, glob, sys
f = glob.glob(f"{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
keys = sys.argv[2:]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "anonymous" in n and (not keys or any(k in n for k in keys)):
        print(f'{n.replace("(anonymous namespace)::", "").split("(")[0]:40s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"]) / 1e3:9.1f} us')
