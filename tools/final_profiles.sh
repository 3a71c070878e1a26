#!/bin/bash
# Round-end evidence bundle on one MI355X box (one gpurun call): the driver-style bench line, rocprofv3 --stats of the
# same command, the C5 workload (plain and in a world-1 RCCL group), the one-GPU measurements behind the multi-GPU
# estimate, the analytic-accuracy table.   usage: tools/final_profiles.sh <tag>   (writes gpurun_out/<tag>_*)
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_c2.log 2>&1 || exit 1
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$TAG && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $R/bench.py --steps 20 --warmup 5 --no-extras --no-checks --cpu-frames 0 > $O/${TAG}_bench_c2_prof.log 2>&1) || exit 1
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_c2_kernel_stats.csv
python bench.py --workload c5 --steps 5 --warmup 2 > $O/${TAG}_bench_c5_plain.log 2>&1 || exit 1
python bench.py --workload c3 --steps 5 --warmup 2 > $O/${TAG}_bench_c3_plain.log 2>&1 || exit 1
python bench.py --workload c5 --gpus 1 --force-dist --steps 5 --warmup 2 > $O/${TAG}_bench_c5_dist1.log 2>&1 || exit 1
python bench.py --gpus 1 --total-frames 1024 --steps 10 --warmup 3 --no-extras --no-checks --cpu-frames 0 > $O/${TAG}_c4_single_gpu.log 2>&1 || exit 1
for W in 8 4 2; do for DP in 1 0; do python tools/emulate_world.py --world $W --frames $((1024 / W)) --device-plan $DP 2>&1 | tail -1; done; done > $O/${TAG}_emulate_c4.log
python bench.py --gpus 1 --force-dist --total-frames 128 --steps 10 --warmup 3 --no-extras --no-checks --cpu-frames 0 > $O/${TAG}_c4_dist1_128.log 2>&1 || exit 1
python tools/analytic_accuracy.py --frames 48 > $O/${TAG}_analytic.log 2>&1 || exit 1
echo bundle done
