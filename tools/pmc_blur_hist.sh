#!/bin/bash
# EXECUTED-instruction histogram of the motion-blur warp kernels from the SQ_INSTS_* counters (what the S-loop really
# issues, by class), per pixel-sample.  Bounded passes, --pmc with --kernel-trace only, program directly after `--`.
#   usage: tools/pmc_blur_hist.sh <tag>   -> gpurun_out/<tag>_blur_hist.md
R=$GRAFT_REPO_ROOT
source $R/tools/pmc_lib.sh
TAG=${1:-r04}
OUT=$R/gpurun_out/pmc_blur_$TAG
DIRS=""
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  pmc_pass $OUT group$i 200 "$grp" python3 $R/tools/pmc_blur_target.py || exit 1
  DIRS="$DIRS $PMC_DIR"
  [ $i -eq 1 ] && FIRST_LOG=$PMC_LOG
done
PMC_DIRS="$DIRS" FIRST_LOG="$FIRST_LOG" \
python3 - <<PY
import csv, glob, collections, re, os
dirs = os.environ["PMC_DIRS"].split()
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(f_ for d in dirs for f_ in glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        m = re.search(r"warp_(?:blur_)?kernel<([^>]*)>", r["Kernel_Name"])
        if m and ("blur" in r["Kernel_Name"] or "true" in m.group(1).split(",")[2]):
            rows[m.group(1).replace(" ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(dirs[0] + "/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        m = re.search(r"warp_(?:blur_)?kernel<([^>]*)>", r["Kernel_Name"])
        if m: dur[m.group(1).replace(" ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
samples = {}
for line in open(os.environ["FIRST_LOG"]):
    m = re.search(r"\[pmc_blur_target\] (\w+) S=(\d+) .*pixel-samples per pass (\d+)", line)
    if m: samples["1" if m.group(1) == "bicubic" else "0"] = int(m.group(3))
out = open("$R/gpurun_out/${TAG}_blur_hist.md", "w")
for k, cs in sorted(rows.items()):
    ps = samples.get(k.split(",")[0])
    out.write(f"\n### warp_kernel<{k}>  ({len(dur[k])} launches, avg {sum(dur[k]) / max(len(dur[k]), 1):.0f} us under PMC; {ps} pixel-samples per launch)\n\n")
    out.write("| counter | per launch | wave-instructions per pixel-sample x 64 lanes / 2 px per thread = per pixel-sample |\n|---|---|---|\n")
    for c, v in sorted(cs.items()):
        mean = sum(v) / len(v)
        out.write(f"| {c} | {mean:.4g} | {mean * 64 / ps:.2f} |\n" if ps else f"| {c} | {mean:.4g} | |\n")
out.close()
print(open("$R/gpurun_out/${TAG}_blur_hist.md").read())
PY
