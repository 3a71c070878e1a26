#!/bin/bash
# A/B of two builds of libvstab.so on the GPU box: runs "$@" with the library at $AB_LIB swapped in, then restores.
#   AB_LIB=comfyui-video-stabilizer_amd/lib/libvstab_before.so tools/ab_lib.sh python bench.py ...
L=$GRAFT_REPO_ROOT/comfyui-video-stabilizer_amd/lib
cp $L/libvstab.so /tmp/libvstab_current.so && cp $GRAFT_REPO_ROOT/$AB_LIB $L/libvstab.so
"$@"; rc=$?
cp /tmp/libvstab_current.so $L/libvstab.so
exit $rc
