cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k
mkdir -p $O
# 1. the hardware behaviour in isolation
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma_overtake.hip -o /tmp/lds_dma_overtake 2>/dev/null && timeout 120 /tmp/lds_dma_overtake 2000 > $O/lds_dma_overtake.txt 2>&1
cat $O/lds_dma_overtake.txt
# 2. the late order (now the default) with one dependency per row piece: 200 launches of every RE10K case, and the eval frame
for c in re10k_nv2 re10k_k128 k12_idle_lanes_re10k cfg2_nv1_oneray; do
  python tools/late_probe.py 200 $c 2>&1 | grep -v amdgpu.ids | tail -4 >> $O/late_probe.txt
done
BTS_RENDER_LIB=$PWD/behindthescenes_amd/variants/libbts_fetchearly.so python tools/late_probe.py 100 re10k_nv2 2>&1 | grep -v amdgpu.ids | tail -2 >> $O/late_probe.txt
cat $O/late_probe.txt
# 3. A/B of the two orders on the eval frame, one box
python tools/lib_ab.py default fetchearly 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/lib_ab.txt
# 4. the whole GPU suite
timeout 900 python -m pytest tests -m gpu -q --timeout 600 --tb=short -rf --durations=8 2>&1 | tail -30 > $O/pytest_gpu.txt
tail -15 $O/pytest_gpu.txt
python bench.py > $O/bench_eval.json 2> $O/bench_eval.err; cat $O/bench_eval.json
