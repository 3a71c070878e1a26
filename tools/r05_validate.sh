#!/bin/bash
# Round-5 validation inside ONE gpurun call: the GPU tests touched by the failure-agreement / hooks / verdict changes, the
# default bench line, then the counter passes (once; stops at the first failure).  usage: tools/r05_validate.sh <tag>
cd $GRAFT_REPO_ROOT
TAG=$1
timeout -k 10 900 python -m pytest tests/test_sharded_gpu.py tests/test_device_plan_gpu.py tests/test_dis_gpu.py tests/test_nodes_gpu.py tests/test_fit_gpu.py -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1 || { tail -40 gpurun_out/${TAG}_tests.log; exit 1; }
tail -3 gpurun_out/${TAG}_tests.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.log 2> gpurun_out/${TAG}_bench.err || { tail -30 gpurun_out/${TAG}_bench.err; exit 1; }
tail -c 3000 gpurun_out/${TAG}_bench.log
tools/pmc_traffic.sh $TAG > gpurun_out/${TAG}_pmc_traffic.out 2>&1 || { tail -40 gpurun_out/${TAG}_pmc_traffic.out; exit 1; }
tail -5 gpurun_out/${TAG}_pmc_traffic.out
tools/pmc_dis.sh $TAG > gpurun_out/${TAG}_pmc_dis.out 2>&1 || { tail -40 gpurun_out/${TAG}_pmc_dis.out; exit 1; }
tail -12 gpurun_out/${TAG}_pmc_dis.out
