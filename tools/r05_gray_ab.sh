#!/bin/bash
cd $GRAFT_REPO_ROOT
L=comfyui-video-stabilizer_amd/lib
timeout -k 10 600 python tools/ab_step.py $L/libvstab_gray0.so $L/libvstab_gray1.so $L/libvstab_gray2.so > gpurun_out/$1_gray_ab.log 2>&1
cat gpurun_out/$1_gray_ab.log
