#!/bin/bash
# A/B of two builds of libvstab.so on the GPU box: runs "$@" with the variant build at $AB_LIB loaded INSTEAD of the
# shipped library, through the loader's own override (native.py: VSTAB_LIB).  The shipped file is never touched, so a
# timeout / kill of the wrapped command cannot leave a variant behind for the driver's GPUTEST / BENCH.
#   AB_LIB=comfyui-video-stabilizer_amd/lib/libvstab_before.so tools/ab_lib.sh python bench.py ...
set -e
test -f "$GRAFT_REPO_ROOT/$AB_LIB" || { echo "ab_lib.sh: $GRAFT_REPO_ROOT/$AB_LIB does not exist"; exit 2; }
VSTAB_LIB="$GRAFT_REPO_ROOT/$AB_LIB" "$@"
