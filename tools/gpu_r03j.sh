cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j
mkdir -p $O
for v in latea lateb lateb0 lateb3 lateg; do
  BTS_RENDER_LIB=$PWD/behindthescenes_amd/variants/libbts_$v.so python tools/late_probe.py 100 re10k_nv2 2>&1 | grep -v amdgpu.ids | tail -3 >> $O/late_probe.txt
done
cat $O/late_probe.txt
