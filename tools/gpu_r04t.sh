cd $GRAFT_REPO_ROOT
O=gpurun_out/r04t
mkdir -p $O
for s in kitti360 re10k; do timeout 300 python tools/section_probe_train.py $s 2>&1 | grep -v amdgpu.ids | tee -a $O/section_probe_train.txt; done
timeout 300 python tools/section_probe.py 2>&1 | grep -A1 "^both" | tee -a $O/section_probe_train.txt
