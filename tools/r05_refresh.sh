#!/bin/bash
# after a change of the warp sources' hash inputs: whole GPU suite, the traffic passes (-> <tag>_warp_traffic.json), the driver-style bench line
cd $GRAFT_REPO_ROOT
TAG=${1:-r05b}
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1; RC=$?
tail -4 gpurun_out/${TAG}_tests.log
[ $RC -eq 0 ] || exit 1
tools/pmc_traffic.sh $TAG > gpurun_out/${TAG}_pmc_traffic.out 2>&1 || { tail -30 gpurun_out/${TAG}_pmc_traffic.out; exit 1; }
cp gpurun_out/${TAG}_warp_traffic.json profiles/warp_traffic.json
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_c2.log 2> gpurun_out/${TAG}_bench_c2.err || exit 1
tail -c 1500 gpurun_out/${TAG}_bench_c2.log
