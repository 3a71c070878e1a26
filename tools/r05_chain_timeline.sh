#!/bin/bash
# usage: tools/r05_chain_timeline.sh c3|c5: where the GPU idles inside one step of the chain
cd $GRAFT_REPO_ROOT
W=${1:-c5}
D=$GRAFT_REPO_ROOT/gpurun_out/chain_${W}_trace
( cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/chain_${W}_bench.log 2>&1 ) || { echo "FAILED trace"; tail -20 $GRAFT_REPO_ROOT/gpurun_out/chain_${W}_bench.log; exit 1; }
python3 tools/chain_timeline.py $D | tee gpurun_out/chain_${W}_timeline.txt
find $D -name "*.csv" -size +1M -delete
