#!/bin/bash
# HBM traffic per kernel of the Flow step (and the blur warp): one bounded rocprofv3 pass per counter (FETCH_SIZE and
# WRITE_SIZE do not fit one pass), --pmc with --kernel-trace only, program directly after `--`.
#   usage: tools/pmc_traffic.sh <tag>   -> gpurun_out/<tag>_hbm_traffic.csv (per-dispatch means, KB as the counters report)
R=$GRAFT_REPO_ROOT
source $R/tools/pmc_lib.sh
pmc_prepare
TAG=${1:-r04}
OUT=$R/gpurun_out/pmc_traffic_$TAG
declare -A DIRS
for c in FETCH_SIZE WRITE_SIZE; do
  pmc_pass $OUT $c 200 "$c" python3 $R/tools/pmc_target.py || exit 1
  DIRS[$c]=$PMC_DIR
done
FETCH_DIR=${DIRS[FETCH_SIZE]} WRITE_DIR=${DIRS[WRITE_SIZE]} \
python3 - <<PY
import csv, glob, re, collections, os, json, hashlib
def short(name):
    m = re.search(r"(level_kernel<\d+>|pis4_kernel<\d+>|warp_(?:blur_)?kernel<[^>]*>|gray_area_int_kernel<[^>]*>|fit_kernel|area_u8_kernel|area_general_rows_kernel)", name)
    return m.group(1).replace(", ", ",") if m else None
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.environ["FETCH_DIR" if c == "FETCH_SIZE" else "WRITE_DIR"] + "/**/*counter_collection.csv", recursive=True):
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
# the warp kernel's record for bench.py's roofline.traffic, tied to the kernel sources it was collected on
k = next((k for k in rows if k.startswith("warp_kernel<0,0,true")), None)
if k:
    def mean(v): return sum(x[1] for x in v) / len(v)
    fetch_kb, write_kb = mean(rows[k]["FETCH_SIZE"]), mean(rows[k]["WRITE_SIZE"])
    h = hashlib.sha256()
    for name in ("vstab_warp.hip", "vstab_internal.h"):
        h.update(open(f"$R/comfyui-video-stabilizer_amd/csrc/{name}", "rb").read())
    rec = {"kernel": f"warp_kernel<bilinear,q5,mask> (rocprof name: {k})", "frames": 256, "size": [1920, 1080],
           "FETCH_SIZE_KB_raw": round(fetch_kb, 1), "WRITE_SIZE_KB": round(write_kb, 1),
           "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md, HBM section) -> read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 taken as is",
           "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024, "algorithmic_bytes_per_launch": 28 * 1920 * 1080 * 256,
           "kernel_source_sha256": h.hexdigest(),
           "command": "tools/pmc_traffic.sh $TAG: rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/pmc_target.py, and a separate pass with --pmc WRITE_SIZE; mean over the Flow passes' dispatches of this kernel"}
    json.dump(rec, open("$R/gpurun_out/${TAG}_warp_traffic.json", "w"), indent=1)
    print(json.dumps(rec))
PY
