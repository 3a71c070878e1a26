cd $GRAFT_REPO_ROOT
L=comfyui-video-stabilizer_amd/lib
for rep in 1 2; do
echo "main fused:";  python tools/ab_dis.py $L/libvstab.so 2>&1 | grep dis
echo "main split:";  VSTAB_DIS_SPLIT=1 python tools/ab_dis.py $L/libvstab.so 2>&1 | grep dis
echo "t512 split:";  VSTAB_DIS_SPLIT=1 python tools/ab_dis.py $L/libvstab_t512.so 2>&1 | grep dis
done
VSTAB_DIS_SPLIT=1 VSTAB_LIB=$L/libvstab_t512.so python -m pytest tests/test_dis_gpu.py -x -q -m gpu 2>&1 | tail -2
