#!/bin/bash
# PMC counters of the warp kernel (microbench, 64 frames), one bounded rocprofv3 pass per group.
R=$GRAFT_REPO_ROOT
source $R/tools/pmc_lib.sh
OUT=$R/gpurun_out/pmc_warp
DIRS=""
i=0
for grp in "$@"; do
  i=$((i+1))
  pmc_pass $OUT group$i 150 "$grp" python3 $R/tools/warp_microbench.py --n 64 --reps 3 || exit 1
  DIRS="$DIRS $PMC_DIR"
done
PMC_DIRS="$DIRS" \
python3 - <<PY
import csv, glob, collections, os
for f in sorted(f_ for d in os.environ["PMC_DIRS"].split() for f_ in glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "warp_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    print({c: f"{v/cnt[c]:.4g}" for c, v in agg.items()})
PY
