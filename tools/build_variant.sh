#!/bin/bash
# Build a variant of libvstab.so next to the shipped one (A/B measurements on the GPU box; *.so is git-ignored):
#   tools/build_variant.sh <name> [git-rev|-] [EXTRA flags]   ->  comfyui-video-stabilizer_amd/lib/libvstab_<name>.so
# git-rev: take csrc/ from that revision instead of the working tree ("-" = working tree).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; REV=${2:--}; shift; shift || true
T=/tmp/vstab_variant_$NAME
rm -rf $T && mkdir -p $T/pkg $T/include
if [ "$REV" = "-" ]; then
  cp -r $R/comfyui-video-stabilizer_amd/csrc $T/pkg/csrc; cp $R/include/vstab.h $T/include/
else
  (cd $R && git archive $REV comfyui-video-stabilizer_amd/csrc include/vstab.h) | tar -x -C $T
  mv $T/comfyui-video-stabilizer_amd/csrc $T/pkg/csrc
fi
rm -rf $T/pkg/csrc/build
make -C $T/pkg/csrc -j8 EXTRA="$*" 2>&1 | grep -E "error|Error" || true
cp $T/pkg/lib/libvstab.so $R/comfyui-video-stabilizer_amd/lib/libvstab_$NAME.so
ls -la $R/comfyui-video-stabilizer_amd/lib/libvstab_$NAME.so
