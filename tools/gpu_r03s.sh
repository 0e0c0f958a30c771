cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s
mkdir -p $O
LIB_AB_PASSES=3 python tools/bwd_ab.py re10k 48 default rowsbnopin 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/bwd_ab_pin.txt
LIB_AB_PASSES=2 python tools/bwd_ab.py re10k 128 default rowsbnopin 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/bwd_ab_pin.txt
