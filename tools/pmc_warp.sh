#!/bin/bash
# PMC counters of the warp kernel (microbench, 64 frames), one bounded rocprofv3 pass per group.
cd /tmp && export TMPDIR=/tmp
# one HIP stream under counter collection: the profiler serialises dispatches, and a kernel queued behind an event of the
# library's second (preparation) stream can then wait for a kernel the serialiser holds back -- a pass that hangs after
# "[pmc_target] clip ready" (profiles/r03_pmc_stuck_pass.md)
export VSTAB_DIS_PREP_STREAM=0
R=$GRAFT_REPO_ROOT
OUT=/tmp/pmc_warp; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $R/tools/warp_microbench.py --n 64 --reps 3 > $OUT/g$i.log 2>&1 || echo "group failed: $grp"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "warp_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    print({c: f"{v/cnt[c]:.4g}" for c, v in agg.items()})
PY
