cd $GRAFT_REPO_ROOT
O=gpurun_out/r04zz
mkdir -p $O
for s in kitti360 kitti_raw; do BTS_RENDER_LIB=behindthescenes_amd/variants/libbts_ticks.so timeout 200 python tools/bwd_ticks.py $s 2>&1 | grep -v amdgpu.ids | tee -a $O/bwd_ticks.txt; done
