#!/bin/bash
# Shared by every PMC script (pmc_traffic.sh, pmc_dis.sh, pmc_blur_hist.sh, pmc_warp.sh): one bounded rocprofv3 counter pass.
#
#   source tools/pmc_lib.sh
#   pmc_pass <outdir> <label> <timeout_s> "<counters>" <program> [args...]     -> sets PMC_DIR (rocprofv3 output of the pass)
#
# Rules a pass follows (profiles/r03_pmc_stuck_pass.md: two passes were killed at their time limit in rounds 2 and 3 and
# the second one's log was later overwritten by a successful re-run):
#   * --pmc with --kernel-trace only, the program directly after `--` (gpurun refuses other combinations);
#   * every pass writes a log and an output directory of its OWN: names carry the label, a timestamp and the shell's pid,
#     nothing is ever removed or overwritten, no `rm -rf` of a results directory;
#   * a pass that fails or is killed at its limit has its log AND whatever rocprofv3 had written (the kernel trace shows the
#     last dispatched kernel per queue) copied to <outdir>/failed_<label>_<stamp>/ before anything else happens, its last
#     lines are printed, and the function returns non-zero: the caller stops (no further GPU step after a timeout);
#   * one HIP stream under counter collection (VSTAB_DIS_PREP_STREAM=0): the profiler serialises dispatches, and a kernel
#     queued behind an event of another queue can wait for a kernel the serialiser holds back.  That cause is a reading of
#     two incidents, not a demonstrated one; nothing here tries to make it happen again.
export VSTAB_DIS_PREP_STREAM=0
# the clip of tools/pmc_target.py is synthesised ONCE, unprofiled, and parked under /tmp: the profiled process then loads it
# with a few copies instead of ~10^4 torch dispatches under counter collection (tools/pmc_make_clip.py)
pmc_prepare() {
  ( cd "$GRAFT_REPO_ROOT" && timeout -k 10 300 python3 tools/pmc_make_clip.py 256 ) || { echo "pmc_prepare: could not park the clip (the target will synthesise it itself)"; return 0; }
}
pmc_pass() {
  local out=$1 label=$2 limit=$3 counters=$4; shift 4
  local stamp; stamp=$(date +%Y%m%d-%H%M%S)_$$
  mkdir -p "$out"
  PMC_LOG="$out/${label}_${stamp}.log"
  PMC_DIR="/tmp/pmc_${label}_${stamp}"
  ( cd /tmp && TMPDIR=/tmp timeout -k 10 "$limit" rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$PMC_DIR" -- "$@" ) > "$PMC_LOG" 2>&1
  local rc=$?
  if [ $rc -ne 0 ]; then
    local keep="$out/failed_${label}_${stamp}"
    mkdir -p "$keep"
    cp "$PMC_LOG" "$keep/" 2>/dev/null
    find "$PMC_DIR" -name "*.csv" -size -8M -exec cp {} "$keep/" \; 2>/dev/null
    echo "PMC pass FAILED (rc $rc; 124/137 = killed at its ${limit}s limit): $label [$counters]"
    echo "--- evidence kept in $keep; last lines of the log:"
    tail -n 25 "$PMC_LOG"
    return 1
  fi
  return 0
}
