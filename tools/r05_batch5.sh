#!/bin/bash
# usage: tools/r05_batch5.sh <tag>: round-4 tree vs this tree on ONE box (bench line without extras, twice each, interleaved),
# then the whole GPU suite, homography phases, C3 / C5 chains and the timeline of a timed step
cd $GRAFT_REPO_ROOT
TAG=$1; L=comfyui-video-stabilizer_amd/lib
OUT=gpurun_out/${TAG}_batch.log; : > $OUT
echo "==== r04 tree vs this tree: bench.py --steps 20 --warmup 5 --no-extras --no-checks --cpu-frames 0" >> $OUT
for rep in 1 2; do
  for tree in .ab_r04 .; do
    timeout -k 10 300 python $tree/bench.py --steps 20 --warmup 5 --no-extras --no-checks --cpu-frames 0 > gpurun_out/${TAG}_ab.json 2> gpurun_out/${TAG}_ab.err || { tail -20 gpurun_out/${TAG}_ab.err >> $OUT; cat $OUT; exit 1; }
    python - >> $OUT <<PY
import json
l = json.loads(open("gpurun_out/${TAG}_ab.json").read().strip().splitlines()[-1])
print("$tree", l["value"], l["ms_per_step"], l["config"]["stage_ms"])
PY
  done
done
echo "==== GPU suite" >> $OUT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1; RC=$?
tail -8 gpurun_out/${TAG}_tests.log >> $OUT
[ $RC -eq 0 ] || { cat $OUT; exit 1; }
echo "==== homography phases" >> $OUT
VSTAB_LIB=$GRAFT_REPO_ROOT/$L/libvstab_htrace.so timeout -k 10 300 python tools/homography_phases.py >> $OUT 2>&1 || { echo FAILED htrace >> $OUT; cat $OUT; exit 1; }
for wl in c3 c5; do
  echo "==== bench $wl" >> $OUT
  timeout -k 10 600 python bench.py --workload $wl --steps 5 --warmup 2 > gpurun_out/${TAG}_bench_$wl.log 2> gpurun_out/${TAG}_bench_$wl.err || { tail -30 gpurun_out/${TAG}_bench_$wl.err >> $OUT; cat $OUT; exit 1; }
  python - >> $OUT <<PY
import json
l = json.loads(open("gpurun_out/${TAG}_bench_$wl.log").read().strip().splitlines()[-1])
print(l["value"], l["ms_per_step"], l["config"].get("rank0_stage_ms"), l["config"].get("rank0_host_ms"))
PY
done
echo "==== timeline of a timed step" >> $OUT
D=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace
( cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-extras --no-checks --cpu-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace_bench.log 2>&1 ) || { echo "FAILED trace" >> $OUT; cat $OUT; exit 1; }
python3 tools/step_timeline_all.py $D >> $OUT 2>&1
find $D -name "*.csv" -size +1M -delete
cat $OUT
