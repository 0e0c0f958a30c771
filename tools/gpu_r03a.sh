set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout 1500 python -m pytest tests -m gpu -q -x --durations=25 -rs 2>&1 | tail -70 > gpurun_out/r03a/pytest_gpu.txt
tail -5 gpurun_out/r03a/pytest_gpu.txt
timeout 300 python bench.py --steps 20 --warmup 3 > gpurun_out/r03a/bench_eval.json 2> gpurun_out/r03a/bench_eval.err
for w in train kitti_raw re10k; do
  timeout 300 python bench.py --workload $w --steps 10 --warmup 2 > gpurun_out/r03a/bench_$w.json 2> gpurun_out/r03a/bench_$w.err
done
timeout 300 python bench.py --workload re10k --samples 128 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r03a/bench_re10k_k128.json 2> gpurun_out/r03a/bench_re10k_k128.err
tail -c 600 gpurun_out/r03a/bench_*.json
