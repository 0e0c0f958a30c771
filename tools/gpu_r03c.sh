set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=12 -rs 2>&1 | tail -120 > $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
timeout 300 python bench.py --workload re10k --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_re10k_k48.json 2> $O/bench_re10k_k48.err
timeout 300 python bench.py --workload kitti_raw --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_kitti_raw.json 2> $O/bench_kitti_raw.err
timeout 300 python bench.py --workload profile --steps 10 --warmup 2 > $O/bench_profile.json 2> $O/bench_profile.err
tail -c 900 $O/bench_profile.json; tail -3 $O/bench_profile.err
python tools/bwd_probe.py 5 re10k 48 2>&1 | tail -2
python tools/bwd_probe.py 5 re10k 128 2>&1 | tail -2
python tools/bwd_probe.py 5 kitti_raw 2>&1 | tail -2
bash tools/profile.sh r03c_re10k bwd_re10k 48 > $O/profile_re10k.log 2>&1
tail -12 $O/profile_re10k.log
