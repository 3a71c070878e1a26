#!/bin/bash
# Round-5 starting point inside ONE gpurun call: per-phase times of the fused level kernel at every level (trace build),
# and the per-launch timeline of one C2 step (255 pairs) from a kernel trace.  usage: tools/r05_baseline.sh <tag> <trace-variant>
cd $GRAFT_REPO_ROOT
TAG=$1; V=$2
L=comfyui-video-stabilizer_amd/lib
OUT=gpurun_out/${TAG}_baseline.log; : > $OUT
echo "==== trace $V" >> $OUT
VSTAB_LIB=$GRAFT_REPO_ROOT/$L/libvstab_$V.so timeout -k 10 240 python tools/fused_phases.py >> $OUT 2>&1 || { echo "FAILED trace" >> $OUT; cat $OUT; exit 1; }
echo "==== timeline of one C2 step" >> $OUT
cd /tmp && export TMPDIR=/tmp
D=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-extras --no-checks --cpu-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace_bench.log 2>&1 || { echo "FAILED trace bench" >> $GRAFT_REPO_ROOT/$OUT; exit 1; }
cd $GRAFT_REPO_ROOT
python3 tools/dis_level_times.py $D >> $OUT 2>&1
find $D -name "*.csv" -size +2M -delete
cat $OUT
