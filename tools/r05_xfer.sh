#!/bin/bash
# coded node-boundary transfers: tests, rates, the Flow node end to end
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 900 python -m pytest tests/test_xfer_gpu.py tests/test_warp_gpu.py -x -q -k "xfer or coded or mask or upload or entry_point or soft" > $O/xfer_tests.log 2>&1 || { tail -30 $O/xfer_tests.log; exit 1; }
tail -3 $O/xfer_tests.log
timeout -k 10 900 python tools/xfer_rate.py > $O/r05_xfer_rate.log 2>&1 || { tail -30 $O/r05_xfer_rate.log; exit 1; }
tail -30 $O/r05_xfer_rate.log
