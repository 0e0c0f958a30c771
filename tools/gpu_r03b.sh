set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=15 -rs 2>&1 | tail -150 > $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
for k in 48 128; do
  timeout 300 python bench.py --workload re10k --samples $k --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_re10k_k$k.json 2> $O/bench_re10k_k$k.err
done
export TMPDIR=/tmp
R=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_re10k -o trace -- python $R/bench.py --workload re10k --steps 5 --warmup 2 --no-cpu-baseline > $R/$O/trace_re10k.log 2>&1)
find $O/trace_re10k -type f ! -name "*kernel_stats.csv" -delete
cat $O/trace_re10k/*/*kernel_stats.csv | head -15
tail -c 700 $O/bench_re10k_k48.json $O/bench_re10k_k128.json
