#!/bin/bash
# PMC view of the VALU-bound kernels (DIS level_fused / pis2, blur warp): one bounded rocprofv3 pass per counter group,
# --pmc with --kernel-trace only (no other trace domain), program directly after `--`.
#   usage: tools/pmc_dis.sh <tag>      -> gpurun_out/<tag>_pmc_kernels.csv (+ the raw per-dispatch csv of each pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
OUT=$R/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_dis_$i -- python3 $R/tools/pmc_target.py > $OUT/pass$i.log 2>&1 || echo "pass failed: $grp"
  f=$(find /tmp/pmc_dis_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/pass${i}_counter_collection.csv
  k=$(find /tmp/pmc_dis_$i -name "*kernel_trace.csv" | head -1)
  [ -n "$k" ] && cp $k $OUT/pass${i}_kernel_trace.csv
done
python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/pass*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        for key in ("level_fused_kernel", "pis2_kernel", "warp_kernel", "gray_area_int_kernel", "fit_kernel"):
            if key in name:
                short = name.split("(")[0][-60:]
                rows[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/pass1_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][-60:]
        dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open("$R/gpurun_out/${TAG}_pmc_kernels.csv", "w") as out:
    counters = sorted({c for v in rows.values() for c in v})
    out.write("kernel,dispatches,avg_us_under_pmc," + ",".join(counters) + "\n")
    for name, cs in sorted(rows.items()):
        d = dur.get(name, [])
        out.write(f"{name},{max(len(v) for v in cs.values())},{(sum(d)/len(d) if d else 0):.1f}," + ",".join(f"{sum(cs[c])/len(cs[c]):.6g}" if cs.get(c) else "" for c in counters) + "\n")
print(open("$R/gpurun_out/${TAG}_pmc_kernels.csv").read())
PY
