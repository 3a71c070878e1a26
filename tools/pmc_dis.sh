#!/bin/bash
# PMC view of the latency / VALU-bound kernels (DIS level_kernel / pis4, the warps): one bounded rocprofv3 pass per counter
# group through tools/pmc_lib.sh (unique logs, evidence of a failed pass kept, stop at the first failure).
#   usage: tools/pmc_dis.sh <tag>      -> gpurun_out/<tag>_pmc_kernels.csv + gpurun_out/<tag>_pmc_kernels.md
# The .md states the utilisation figures in the units the guide gives (MI355X_MICROARCH.md: SIMD-32, a wave64 VALU
# instruction issues over 2 cycles; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count QUAD-cycles; SQ_LDS_BANK_CONFLICT and
# SQ_LDS_IDX_ACTIVE count LDS-array cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs):
#   VALU issue utilisation = SQ_INSTS_VALU / (1024 SIMDs x cycles / k),  cycles = GRBM_GUI_ACTIVE / 8,
#   k = 2 (the guide's issue rate) and k = 2.6 (tools/probes/pk_rate_probe: measured cycles per v_mul_f32 per SIMD)
# -- NOT "SQ_ACTIVE_INST_VALU x 4 / (1024 x cycles)", the formula of rounds 2-3, which exceeds 1 on the blur kernel.
R=$GRAFT_REPO_ROOT
source $R/tools/pmc_lib.sh
pmc_prepare
TAG=${1:-r04}
OUT=$R/gpurun_out/pmc_$TAG
DIRS=""
i=0
for grp in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  pmc_pass $OUT group$i 200 "$grp" python3 $R/tools/pmc_target.py || exit 1
  DIRS="$DIRS $PMC_DIR"
done
PMC_DIRS="$DIRS" python3 - <<PY
import csv, glob, collections, os, re
def short(name):
    m = re.search(r"(level_kernel<\d+>|pis4_kernel<\d+>|warp_(?:blur_)?kernel<[^>]*>|gray_area_int_kernel<[^>]*>|fit_kernel|plan_kernel)", name)
    return m.group(1).replace(", ", ",") if m else None
dirs = os.environ["PMC_DIRS"].split()
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                rows[k][r["Counter_Name"]].append((int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"]), int(r.get("Dispatch_Id", 0) or 0)))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(dirs[0] + "/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            dur[k].append((int(r.get("Grid_Size", 0) or r.get("Grid_Size_X", 0) or 0), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
# level / pis kernels are launched once per pyramid level with the SAME grid: the finest level is the dispatch with the
# largest counter value, so those kernels are reported as the SUM over one clip's four levels and as the largest dispatch
def per_clip(v, levels):
    vals = [x[1] for x in v]
    clips = max(1, len(vals) // levels)
    return sum(vals) / clips, (sum(sorted(vals)[-clips:]) / clips)
out_rows = []
for k, cs in sorted(rows.items()):
    levels = 4 if k.startswith(("level_kernel<0>", "pis4_kernel")) else 1
    rec = {"kernel": k, "levels_per_clip": levels}
    for c, v in cs.items():
        rec[c], rec[c + ":largest"] = per_clip(v, levels)
    dv = [x[1] for x in dur.get(k, [])]
    if dv:
        clips = max(1, len(dv) // levels)
        rec["us_under_pmc"] = sum(dv) / clips
        rec["us_under_pmc:largest"] = sum(sorted(dv)[-clips:]) / clips
    out_rows.append(rec)
cols = sorted({c for r in out_rows for c in r if c not in ("kernel",)})
with open("$R/gpurun_out/${TAG}_pmc_kernels.csv", "w") as out:
    out.write("kernel," + ",".join(cols) + "\n")
    for r in out_rows:
        out.write(r["kernel"] + "," + ",".join(f"{r[c]:.6g}" if isinstance(r.get(c), float) else str(r.get(c, "")) for c in cols) + "\n")
with open("$R/gpurun_out/${TAG}_pmc_kernels.md", "w") as md:
    md.write("| kernel (per clip; level/pis4: sum of the four levels) | us under PMC | VALU wave-instr | VALU util k=2 | k=2.6 | waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) | LDS conflict share (BANK_CONFLICT / IDX_ACTIVE) | LDS array busy (IDX_ACTIVE / cycles) | VMEM rd+wr wave-instr |\n|---|---|---|---|---|---|---|---|---|\n")
    for r in out_rows:
        g = r.get("GRBM_GUI_ACTIVE")
        if not g: continue
        cyc = g / 8.0
        iv = r.get("SQ_INSTS_VALU", 0.0)
        u2, u26 = iv / (1024 * cyc / 2.0), iv / (1024 * cyc / 2.6)
        wait = r.get("SQ_WAIT_ANY", 0.0) / max(r.get("SQ_WAVE_CYCLES", 1.0), 1.0)
        idx = r.get("SQ_LDS_IDX_ACTIVE", 0.0)
        conf = r.get("SQ_LDS_BANK_CONFLICT", 0.0) / idx if idx else float("nan")
        busy = idx / (256 * cyc) if idx else float("nan")
        vm = r.get("SQ_INSTS_VMEM_RD", 0.0) + r.get("SQ_INSTS_VMEM_WR", 0.0)
        md.write(f"| {r['kernel']} | {r.get('us_under_pmc', 0):.0f} | {iv:.4g} | {u2:.2f} | {u26:.2f} | {wait:.2f} | {conf:.2f} | {busy:.2f} | {vm:.4g} |\n")
print(open("$R/gpurun_out/${TAG}_pmc_kernels.md").read())
PY
