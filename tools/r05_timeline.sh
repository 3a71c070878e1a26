#!/bin/bash
# usage: tools/r05_timeline.sh <tag>: the kernels of one timed C2 step, in order, with the idle time before each
cd $GRAFT_REPO_ROOT
TAG=$1
D=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace
( cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-extras --no-checks --cpu-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace_bench.log 2>&1 ) || { echo "FAILED trace"; tail -20 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace_bench.log; exit 1; }
python3 tools/step_timeline_all.py $D > gpurun_out/${TAG}_timeline.txt 2>&1
find $D -name "*.csv" -size +1M -delete
cat gpurun_out/${TAG}_timeline.txt
