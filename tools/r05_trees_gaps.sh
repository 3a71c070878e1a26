#!/bin/bash
# usage: tools/r05_trees_gaps.sh <tag>: span / inter-step idle / period of the bench steps, round-4 tree and this tree, ONE box
cd $GRAFT_REPO_ROOT
TAG=$1
for tree in .ab_r04 . .ab_r04 .; do
  name=$(echo $tree | tr -d './'); [ -z "$name" ] && name=new
  D=/tmp/gaps_${TAG}_${name}_$RANDOM
  ( cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/$tree/bench.py --steps 12 --warmup 3 --no-extras --no-checks --cpu-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_${name}_bench.log 2>&1 ) || { echo "FAILED $tree"; exit 1; }
  echo "==== $tree: $(python3 -c "import json; l=json.loads(open('gpurun_out/${TAG}_${name}_bench.log').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'])")"
  python3 tools/step_periods.py $D 15
done
