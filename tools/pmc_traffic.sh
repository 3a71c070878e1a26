#!/bin/bash
# HBM traffic per kernel of the Flow step (and the blur warp): one bounded rocprofv3 pass per counter (FETCH_SIZE and
# WRITE_SIZE do not fit one pass), --pmc with --kernel-trace only, program directly after `--`.
#   usage: tools/pmc_traffic.sh <tag>   -> gpurun_out/<tag>_hbm_traffic.csv (per-dispatch means, KB as the counters report)
cd /tmp && export TMPDIR=/tmp
# one HIP stream under counter collection: the profiler serialises dispatches, and a kernel queued behind an event of the
# library's second (preparation) stream can then wait for a kernel the serialiser holds back -- a pass that hangs after
# "[pmc_target] clip ready" (profiles/r03_pmc_stuck_pass.md)
export VSTAB_DIS_PREP_STREAM=0
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
OUT=$R/gpurun_out/pmc_traffic_$TAG; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $R/tools/pmc_target.py > $OUT/$c.log 2>&1 || { echo "pass failed: $c"; exit 1; }
done
python3 - <<PY
import csv, glob, re, collections
def short(name):
    m = re.search(r"(level_kernel<\d+>|pis4_kernel<\d+>|warp_(?:blur_)?kernel<[^>]*>|gray_area_int_kernel<[^>]*>|fit_kernel|area_u8_kernel|area_general_rows_kernel)", name)
    return m.group(1).replace(", ", ",") if m else None
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k and r["Counter_Name"] == c:
                rows[k][c].append((int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
with open("$R/gpurun_out/${TAG}_hbm_traffic.csv", "w") as out:
    out.write("kernel,dispatches,FETCH_SIZE_KB_raw_mean,WRITE_SIZE_KB_mean,FETCH_SIZE_KB_raw_largest_grid,WRITE_SIZE_KB_largest_grid\n")
    for k, cs in sorted(rows.items()):
        def mean(v): return sum(x[1] for x in v) / len(v) if v else float("nan")
        def big(v):
            if not v: return float("nan")
            g = max(x[0] for x in v); w = [x[1] for x in v if x[0] == g]; return sum(w) / len(w)
        # level / pis kernels: all levels share a grid size, so "largest grid" = mean; kept for the single-launch kernels
        out.write(f"{k},{len(cs['FETCH_SIZE'])},{mean(cs['FETCH_SIZE']):.1f},{mean(cs['WRITE_SIZE']):.1f},{big(cs['FETCH_SIZE']):.1f},{big(cs['WRITE_SIZE']):.1f}\n")
print(open("$R/gpurun_out/${TAG}_hbm_traffic.csv").read())
PY
