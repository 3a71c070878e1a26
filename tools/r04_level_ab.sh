#!/bin/bash
# Round-4 A/B of level_kernel variants inside ONE gpurun call (box-to-box variance ~5 %): per-phase times of the trace
# builds (tools/fused_phases.py), DIS stage time of the plain builds (tools/ab_dis.py), and the DIS parity tests on the
# candidate.  usage: tools/r04_level_ab.sh <tag> <trace variants...> -- <plain variants...>
cd $GRAFT_REPO_ROOT
TAG=$1; shift
OUT=gpurun_out/${TAG}_level_ab.log; : > $OUT
L=comfyui-video-stabilizer_amd/lib
while [ "$1" != "--" ] && [ -n "$1" ]; do
  echo "==== trace $1" >> $OUT
  VSTAB_LIB=$L/libvstab_$1.so timeout -k 10 240 python tools/fused_phases.py >> $OUT 2>&1 || { echo "FAILED $1" >> $OUT; exit 1; }
  shift
done
shift
PL=""
for v in "$@"; do [ "$v" = "main" ] || PL="$PL $L/libvstab_$v.so"; done
echo "==== stage times" >> $OUT
timeout -k 10 400 python tools/ab_dis.py $L/libvstab.so $PL >> $OUT 2>&1 || { echo "FAILED ab_dis" >> $OUT; exit 1; }
for v in "$@"; do
  echo "==== parity tests on $v" >> $OUT
  F=$L/libvstab_$v.so; [ "$v" = "main" ] && F=$L/libvstab.so
  VSTAB_LIB=$F timeout -k 10 600 python -m pytest tests/test_dis_gpu.py -x -q -m gpu 2>&1 | tail -3 >> $OUT
done
cat $OUT
