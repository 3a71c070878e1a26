#!/bin/bash
# usage: tools/r05_final.sh <tag>: whole GPU suite, then the evidence bundle, then the counter passes (stop at the first failure)
cd $GRAFT_REPO_ROOT
TAG=$1
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1; RC=$?
tail -4 gpurun_out/${TAG}_tests.log
[ $RC -eq 0 ] || exit 1
timeout -k 10 900 tools/final_profiles.sh $TAG > gpurun_out/${TAG}_bundle.out 2>&1 || { tail -20 gpurun_out/${TAG}_bundle.out; exit 1; }
tail -2 gpurun_out/${TAG}_bundle.out
tools/pmc_traffic.sh $TAG > gpurun_out/${TAG}_pmc_traffic.out 2>&1 || { tail -30 gpurun_out/${TAG}_pmc_traffic.out; exit 1; }
tools/pmc_dis.sh $TAG > gpurun_out/${TAG}_pmc_dis.out 2>&1 || { tail -30 gpurun_out/${TAG}_pmc_dis.out; exit 1; }
tail -12 gpurun_out/${TAG}_pmc_dis.out
cat gpurun_out/${TAG}_trees_ab.txt gpurun_out/${TAG}_emulate_c4.log
