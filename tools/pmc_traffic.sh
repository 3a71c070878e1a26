#!/bin/bash
# HBM traffic of the warp kernel: one bounded rocprofv3 pass per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_traffic; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $R/tools/warp_microbench.py --n 256 --reps 2 > $OUT/$c.log 2>&1 || echo "pass failed: $c"
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  head -1 $f > $OUT/r01_warp_pmc_$(echo $c | tr A-Z a-z).csv
  grep "warp_kernel" $f >> $OUT/r01_warp_pmc_$(echo $c | tr A-Z a-z).csv
done
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$OUT/r01_warp_pmc_*.csv")):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))]
    print(f.split("/")[-1], len(vals), "dispatches, mean", sum(vals)/len(vals), "KB")
PY
