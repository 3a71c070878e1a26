#!/bin/bash
# the whole GPU suite under each of the library's alternate forms (same bits are claimed for all of them); lists what fails under each
# (tests that assert WHICH form ran -- "device_plan used", "coded chunks" -- fail by construction under the switch that turns it off)
cd $GRAFT_REPO_ROOT
for v in "VSTAB_DEVICE_PLAN=0" "VSTAB_XFER_CODED=0" "VSTAB_DIS_SPLIT=1" "VSTAB_DIS_PYRAMID_TAIL=0" "VSTAB_BLUR_FAST=0"; do
  echo "== $v"
  env $v timeout -k 10 900 python -m pytest tests -q -m gpu --maxfail=30 -W ignore::DeprecationWarning -p no:cacheprovider > gpurun_out/switch_$(echo $v | tr '=' '_').log 2>&1; grep -E "^FAILED" gpurun_out/switch_$(echo $v | tr '=' '_').log
  tail -1 gpurun_out/switch_$(echo $v | tr '=' '_').log
done
