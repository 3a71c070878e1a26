#!/bin/bash
# usage: tools/r05_trees_stats.sh <tag>: rocprofv3 --stats of the bench step for the round-4 tree and this tree on ONE box
cd $GRAFT_REPO_ROOT
TAG=$1
for tree in .ab_r04 . .ab_r04 .; do
  name=$(echo $tree | tr -d './'); [ -z "$name" ] && name=new
  D=/tmp/prof_${TAG}_${name}_$RANDOM
  ( cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/$tree/bench.py --steps 20 --warmup 5 --no-extras --no-checks --cpu-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_${name}_bench.log 2>&1 ) || { echo "FAILED $tree"; tail -5 gpurun_out/${TAG}_${name}_bench.log; exit 1; }
  echo "==== $tree: $(python3 -c "import json; l=json.loads(open('gpurun_out/${TAG}_${name}_bench.log').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['config']['stage_ms'])")"
  python3 - <<PY
import csv, glob, re
f = glob.glob("$D/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.reader(open(f)):
    m = re.search(r"\(anonymous namespace\)::([A-Za-z0-9_]+(<[^>]*>)?)", r[0]) if r else None
    if m and float(r[3]) > 8000: print(f"   {m.group(1):40s} calls {r[1]:>5s} avg us {float(r[3]) / 1e3:9.1f}")
PY
done
