#!/bin/bash
# usage: tools/r05_batch3.sh <tag> : sparse upsample A/B + parity, homography phases, whole-step timeline, cost of the timers
cd $GRAFT_REPO_ROOT
TAG=$1; L=comfyui-video-stabilizer_amd/lib
OUT=gpurun_out/${TAG}_batch.log; : > $OUT
echo "==== DIS stage times: base (HEAD~) vs this tree" >> $OUT
timeout -k 10 400 python tools/ab_dis.py $L/libvstab_base.so $L/libvstab.so >> $OUT 2>&1 || { echo FAILED ab >> $OUT; cat $OUT; exit 1; }
echo "==== parity" >> $OUT
timeout -k 10 600 python -m pytest tests/test_dis_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -3 >> $OUT
echo "==== homography phases" >> $OUT
VSTAB_LIB=$GRAFT_REPO_ROOT/$L/libvstab_htrace.so timeout -k 10 300 python tools/homography_phases.py >> $OUT 2>&1 || { echo FAILED htrace >> $OUT; cat $OUT; exit 1; }
echo "==== timers on/off" >> $OUT
timeout -k 10 300 python tools/timing_cost.py >> $OUT 2>&1 || { echo FAILED timing >> $OUT; cat $OUT; exit 1; }
echo "==== whole-step timeline" >> $OUT
D=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace
( cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-extras --no-checks --cpu-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace_bench.log 2>&1 ) || { echo "FAILED trace" >> $OUT; cat $OUT; exit 1; }
python3 tools/step_timeline_all.py $D >> $OUT 2>&1
find $D -name "*.csv" -size +1M -delete
cat $OUT
