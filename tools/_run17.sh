export TMPDIR=/tmp
V=$PWD/behindthescenes_amd/variants/libbts_gatherlds.so
BTS_RENDER_LIB=$V timeout 900 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_grad.py tests/test_gpu_train_step.py -x -q 2>&1 | tail -4
timeout 300 python tools/lib_ab.py --learn-empty default gatherlds 2>&1 | tail -2
echo "default: $(timeout 200 python tools/bwd_probe.py 5 2>&1 | tail -1)"
echo "gatherlds: $(BTS_RENDER_LIB=$V timeout 200 python tools/bwd_probe.py 5 2>&1 | tail -1)"
