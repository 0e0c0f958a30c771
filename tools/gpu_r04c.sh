cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c
mkdir -p $O
echo "=== current lib" > $O/packed.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q --timeout 600 --tb=short -rf -k "short_rays or k32_packed" 2>&1 | grep -E "passed|failed|FAILED|Error|rel.max|assert " | head -40 >> $O/packed.txt
echo "=== r03base lib" >> $O/packed.txt
BTS_RENDER_LIB=$PWD/behindthescenes_amd/variants/libbts_r03base.so BTS_ALLOW_OLDER_ABI=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q --timeout 600 --tb=short -rf -k "short_rays or k32_packed" 2>&1 | grep -E "passed|failed|FAILED|Error|rel.max|assert " | head -40 >> $O/packed.txt
cat $O/packed.txt
timeout 600 python -m pytest tests/test_gpu_ddp.py -q --timeout 600 --tb=long -rf -k "ddp_gradients or ddp_over" 2>&1 | tail -60 > $O/ddp.txt; tail -60 $O/ddp.txt
for w in kitti_raw; do timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$w standalone', j['ms_per_step'], j['roofline']['fwd_ms'], j['roofline']['bwd_ms'])"; done
