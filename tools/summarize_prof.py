"""Condense a rocprofv3 --kernel-trace --stats output directory into profiles/<name>.md (+ csv of our kernels)."""
import csv
import glob
import sys
from pathlib import Path

src, name = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
stats = glob.glob(f"{src}/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(stats)))
ours = [r for r in rows if "anonymous namespace" in r["Name"]]
out = Path("profiles") / f"{name}.md"
total = sum(float(r["TotalDurationNs"]) for r in ours)
with open(out, "w") as f:
    f.write(f"# {name}\n\n{note}\n\nSource: `rocprofv3 --kernel-trace --stats` ({Path(stats).name}); only libvstab kernels listed "
            f"(torch kernels in the trace are the synthetic-clip generator, outside the timed region).\n\n")
    f.write("| kernel | calls | total ms | avg us | min us | max us | share of libvstab time |\n|---|---:|---:|---:|---:|---:|---:|\n")
    for r in ours:
        nm = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        f.write(f"| `{nm}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
                f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {100*float(r['TotalDurationNs'])/total:.1f}% |\n")
with open(Path("profiles") / f"{name}_kernel_stats.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in ours:
        w.writerow(r)
print(open(out).read())
