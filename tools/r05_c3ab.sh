#!/bin/bash
# the C3 chain (Flow perspective -> Motion Apply bicubic / 17) with the Flow half's plan formed on the host / on the device
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for dp in 0 1; do
echo -n "device_plan=$dp "; VSTAB_DEVICE_PLAN=$dp python bench.py --workload c3 --steps 5 --warmup 2 2>/dev/null | python3 tools/line_fields.py value ms_per_step config.rank0_stage_ms config.rank0_device_plan config.rank0_host_ms.flow_half
done; done
