cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h
mkdir -p $O
BTS_RENDER_LIB=$PWD/behindthescenes_amd/variants/libbts_fetchlate.so timeout 600 python -m pytest tests/test_gpu_determinism.py -m gpu -q --timeout 600 --tb=line -k "bit_deterministic and re10k" 2>&1 | grep -v "^$" | tail -12 > $O/determinism_fetchlate.txt
cat $O/determinism_fetchlate.txt
LIB_AB_PASSES=2 python tools/lib_ab.py --learn-empty default fetchlate > $O/lib_ab_fetchlate.txt 2>&1; cat $O/lib_ab_fetchlate.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=6 -rf --tb=short 2>&1 | grep -v "^$" | tail -60 > $O/pytest_gpu.txt
grep -n "^FAILED\|passed\|failed" $O/pytest_gpu.txt | tail
python tools/bwd_probe.py 5 re10k 48 2>&1 | tail -1
python tools/bwd_probe.py 5 re10k 128 2>&1 | tail -1
