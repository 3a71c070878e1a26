#!/bin/bash
# The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: GPU sanitizers are not available
# on this pool).  Builds oracle/_build_san/libvstab_oracle.so and runs the CPU tests that exercise the checker on it.
#   usage: tools/oracle_sanitize.sh            (from the repo root, no GPU needed)
set -e
cd "$(dirname "$0")/.."
make -C oracle sanitize
export VSTAB_ORACLE_LIB=$PWD/oracle/_build_san/libvstab_oracle.so
export LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=4
python -m pytest tests/test_abi_cpu.py tests/test_e2e_golden_cpu.py tests/test_dis_sum_order_cpu.py tests/test_host_golden.py tests/test_oracle_edge_cpu.py tests/test_analytic_cpu.py tests/test_referee_cpu.py tests/test_cv2_tier_cpu.py -q -x -k "not reference_check_scripts and not launcher and not one_hip_runtime" "$@"
