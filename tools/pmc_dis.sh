#!/bin/bash
# PMC view of the VALU-bound kernels (DIS level_kernel / pis4, blur warp): one bounded rocprofv3 pass per counter group,
# --pmc with --kernel-trace only (no other trace domain), program directly after `--`.
#   usage: tools/pmc_dis.sh <tag>      -> gpurun_out/<tag>_pmc_kernels.csv (+ the raw per-dispatch csv of each pass)
cd /tmp && export TMPDIR=/tmp
# one HIP stream under counter collection: the profiler serialises dispatches, and a kernel queued behind an event of the
# library's second (preparation) stream can then wait for a kernel the serialiser holds back -- a pass that hangs after
# "[pmc_target] clip ready" (profiles/r03_pmc_stuck_pass.md)
export VSTAB_DIS_PREP_STREAM=0
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
OUT=$R/gpurun_out/pmc_$TAG; rm -rf $OUT /tmp/pmc_dis_*; mkdir -p $OUT
i=0
# small groups (a six-counter pass once sat until its timeout on this pool: profiles/r03_pmc_stuck_pass.md); a failed pass
# ends the script (no further GPU step after a timeout) AFTER showing what the target had printed: pmc_target.py reports
# its progress and the library's device status word (VSTAB_STATUS_PIS_TIMEOUT would show as a VstabError) line by line
for grp in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_dis_$i -- python3 $R/tools/pmc_target.py > $OUT/pass$i.log 2>&1 || { echo "pass failed (rc $?): $grp"; echo "--- last lines of $OUT/pass$i.log"; tail -n 25 $OUT/pass$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
import re
def short(name):
    m = re.search(r"(level_kernel<\d+>|pis4_kernel<\d+>|warp_kernel<[^>]*>|gray_area_int_kernel<[^>]*>|fit_kernel|tensor_h_kernel|area_u8_kernel)", name)
    return m.group(1).replace(", ", ",") if m else None
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("/tmp/pmc_dis_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            # the finest-level dispatches dominate: keep every dispatch, the table reports the per-dispatch mean of the
            # LARGEST grid size of each kernel (= finest level) separately from the mean over all
            rows[k][r["Counter_Name"]].append((int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
dur = collections.defaultdict(list)
for f in sorted(glob.glob("/tmp/pmc_dis_1/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            dur[k].append((int(r.get("Grid_Size", 0) or r.get("Grid_Size_X", 0) or 0), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
def biggest(pairs):
    g = max(p[0] for p in pairs)
    v = [p[1] for p in pairs if p[0] == g]
    return sum(v) / len(v), len(v)
rows = {k: {c: biggest(v) for c, v in cs.items()} for k, cs in rows.items()}
dur = {k: biggest(v) for k, v in dur.items()}
with open("$R/gpurun_out/${TAG}_pmc_kernels.csv", "w") as out:
    counters = sorted({c for v in rows.values() for c in v})
    out.write("kernel(largest grid = finest level),dispatches,avg_us_under_pmc," + ",".join(counters) + "\n")
    for name, cs in sorted(rows.items()):
        d = dur.get(name, (0.0, 0))
        out.write(f"{name},{max(v[1] for v in cs.values())},{d[0]:.1f}," + ",".join(f"{cs[c][0]:.6g}" if cs.get(c) else "" for c in counters) + "\n")
print(open("$R/gpurun_out/${TAG}_pmc_kernels.csv").read())
PY
