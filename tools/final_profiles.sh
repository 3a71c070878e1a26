#!/bin/bash
# Round-end evidence bundle on one MI355X box (one gpurun call): the driver-style bench line, rocprofv3 --stats of the
# same command, the timeline of a timed step, the C3 / C5 chains (plain and C5 in a world-1 RCCL group), the one-GPU
# measurements behind the multi-GPU estimate, the analytic-accuracy table, the round-4 tree beside this one (if .ab_r04 is
# there).   usage: tools/final_profiles.sh <tag>   (writes gpurun_out/<tag>_*)
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_c2.log 2> $O/${TAG}_bench_c2.err || exit 1
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$TAG && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $R/bench.py --steps 20 --warmup 5 --no-extras --no-checks --cpu-frames 0 > $O/${TAG}_bench_c2_prof.log 2>&1) || exit 1
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_c2_kernel_stats.csv
python3 tools/step_timeline_all.py /tmp/prof_$TAG > $O/${TAG}_step_timeline.txt 2>&1
python3 tools/step_periods.py /tmp/prof_$TAG 12 > $O/${TAG}_step_periods.txt 2>&1
if [ -d .ab_r04 ]; then
  for rep in 1 2; do for tree in .ab_r04 .; do
    python $tree/bench.py --steps 20 --warmup 5 --no-extras --no-checks --cpu-frames 0 2>/dev/null | python3 -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tree', l['value'], l['ms_per_step'], l['config']['stage_ms'])"
  done; done > $O/${TAG}_trees_ab.txt
fi
python bench.py --workload c5 --steps 5 --warmup 2 > $O/${TAG}_bench_c5_plain.log 2>&1 || exit 1
python bench.py --workload c3 --steps 5 --warmup 2 > $O/${TAG}_bench_c3_plain.log 2>&1 || exit 1
python bench.py --workload c5 --gpus 1 --force-dist --steps 5 --warmup 2 > $O/${TAG}_bench_c5_dist1.log 2>&1 || exit 1
python bench.py --gpus 1 --total-frames 1024 --steps 10 --warmup 3 --no-extras --no-checks --cpu-frames 0 > $O/${TAG}_c4_single_gpu.log 2>&1 || exit 1
for W in 8 4 2; do for DP in 1 0; do python tools/emulate_world.py --world $W --frames $((1024 / W)) --device-plan $DP 2>&1 | tail -1; done; done > $O/${TAG}_emulate_c4.log
python bench.py --gpus 1 --force-dist --total-frames 128 --steps 10 --warmup 3 --no-extras --no-checks --cpu-frames 0 > $O/${TAG}_c4_dist1_128.log 2>&1 || exit 1
python tools/analytic_accuracy.py --frames 48 > $O/${TAG}_analytic.log 2>&1 || exit 1
if [ -f comfyui-video-stabilizer_amd/lib/libvstab_htrace.so ]; then
  VSTAB_LIB=$R/comfyui-video-stabilizer_amd/lib/libvstab_htrace.so python tools/homography_phases.py > $O/${TAG}_homography_phases.log 2>&1
fi
echo bundle done
