cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i
mkdir -p $O
for v in fetchlate latedelay lateall; do
  BTS_RENDER_LIB=$PWD/behindthescenes_amd/variants/libbts_$v.so python tools/late_probe.py 150 re10k_nv2 2>&1 | grep -v amdgpu.ids >> $O/late_probe.txt
done
python tools/late_probe.py 150 re10k_nv2 2>&1 | grep -v amdgpu.ids >> $O/late_probe.txt
BTS_RENDER_LIB=$PWD/behindthescenes_amd/variants/libbts_fetchlate.so python tools/late_probe.py 150 cfg2_nv1_oneray 2>&1 | grep -v amdgpu.ids >> $O/late_probe.txt
cat $O/late_probe.txt
timeout 300 python -m pytest tests -m gpu -q --timeout 300 --tb=short -k "density_noise" 2>&1 | tail -3
