cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a
mkdir -p $O
./tools/ubench/fast_math_check > $O/fast_math_check.txt 2>&1; tail -30 $O/fast_math_check.txt
LIB_AB_PASSES=3 python tools/lib_ab.py --learn-empty r03base default > $O/lib_ab.txt 2>&1; cat $O/lib_ab.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --tb=short -rf --durations=6 -x 2>&1 | tail -40 > $O/pytest_gpu.txt
tail -25 $O/pytest_gpu.txt
python tools/section_probe.py > $O/section_probe.txt 2>&1; tail -12 $O/section_probe.txt
timeout 900 python bench.py > $O/bench_eval.json 2> $O/bench_eval.err; tail -c 3000 $O/bench_eval.json; tail -5 $O/bench_eval.err
