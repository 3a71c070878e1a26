#!/bin/bash
# PMC counters for the DIS kernels (one --pmc pass per group; no trace domains combined with --pmc).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/pmc_dis
mkdir -p $OUT
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_IFETCH"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-frames 0 > $OUT/$tag.log 2>&1 || echo "group failed: $grp"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        full = r["Kernel_Name"]
        k = "pis_kernel<4>" if "pis_kernel<4>" in full else "pis_kernel<1>" if "pis_kernel<1>" in full else "level_fused" if "level_fused" in full else None
        if k and (k != "level_fused" or int(r["LDS_Block_Size"]) > 60000):   # fused: finest level only
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in agg:
        print(k, {c: f"{v/cnt[(k,c)]:.4g}" for c, v in agg[k].items()})
PY
