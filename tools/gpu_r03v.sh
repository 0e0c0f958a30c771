cd $GRAFT_REPO_ROOT
O=gpurun_out/r03v
mkdir -p $O
BTS_RENDER_LIB=$PWD/behindthescenes_amd/variants/libbts_ticks.so python tools/rowsb_ticks.py 48 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/rowsb_ticks_k48.txt
