#!/bin/bash
# usage: tools/r05_batch4.sh <tag>: whole GPU suite, homography phases, bench C2 / C3 / C5 lines, whole-step timeline
cd $GRAFT_REPO_ROOT
TAG=$1; L=comfyui-video-stabilizer_amd/lib
OUT=gpurun_out/${TAG}_batch.log; : > $OUT
echo "==== GPU suite" >> $OUT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1; RC=$?
tail -15 gpurun_out/${TAG}_tests.log >> $OUT
[ $RC -eq 0 ] || { cat $OUT; exit 1; }
echo "==== homography phases" >> $OUT
VSTAB_LIB=$GRAFT_REPO_ROOT/$L/libvstab_htrace.so timeout -k 10 300 python tools/homography_phases.py >> $OUT 2>&1 || { echo FAILED htrace >> $OUT; cat $OUT; exit 1; }
echo "==== bench C2" >> $OUT
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_c2.log 2> gpurun_out/${TAG}_bench_c2.err || { tail -30 gpurun_out/${TAG}_bench_c2.err >> $OUT; cat $OUT; exit 1; }
python - >> $OUT <<PY
import json
l = json.loads(open("gpurun_out/${TAG}_bench_c2.log").read().strip().splitlines()[-1])
print(l["value"], l["ms_per_step"], l["config"]["stage_ms"], l["config"]["dis_ms"], l["config"]["device_plan"], l["roofline"]["frac"], l.get("parity_at_size", {}).get("bit_equal"))
PY
for wl in c3 c5; do
  echo "==== bench $wl" >> $OUT
  timeout -k 10 600 python bench.py --workload $wl --steps 5 --warmup 2 > gpurun_out/${TAG}_bench_$wl.log 2> gpurun_out/${TAG}_bench_$wl.err || { tail -30 gpurun_out/${TAG}_bench_$wl.err >> $OUT; cat $OUT; exit 1; }
  python - >> $OUT <<PY
import json
l = json.loads(open("gpurun_out/${TAG}_bench_$wl.log").read().strip().splitlines()[-1])
print(l["value"], l["ms_per_step"], l["config"].get("rank0_stage_ms"), l["config"].get("rank0_host_ms"))
PY
done
echo "==== whole-step timeline" >> $OUT
D=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace
( cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-extras --no-checks --cpu-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace_bench.log 2>&1 ) || { echo "FAILED trace" >> $OUT; cat $OUT; exit 1; }
python3 tools/step_timeline_all.py $D >> $OUT 2>&1
find $D -name "*.csv" -size +1M -delete
cat $OUT
