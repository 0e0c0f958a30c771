export TMPDIR=/tmp
V=$PWD/behindthescenes_amd/variants/libbts_gatherlds.so
BTS_RENDER_LIB=$V timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -5
timeout 300 python tools/lib_ab.py --learn-empty default gatherlds 2>&1 | tail -6
