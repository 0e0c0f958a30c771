export TMPDIR=/tmp
R=$PWD; mkdir -p gpurun_out/final
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/bwd_trace -o t -- python $R/tools/bwd_probe.py 5 > $R/gpurun_out/final/bwd_probe_cold.txt 2>&1
cd $R; tail -2 gpurun_out/final/bwd_probe_cold.txt
find gpurun_out/final/bwd_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -5 {} | cut -c1-120'
bash tools/profile.sh r02f fwd 2>&1 | tail -6
bash tools/profile.sh r02f_train train 2>&1 | tail -14
timeout 600 python bench.py > gpurun_out/final/bench_eval.json 2> gpurun_out/final/bench_eval.err; tail -c 1500 gpurun_out/final/bench_eval.json
timeout 600 python bench.py --workload train > gpurun_out/final/bench_train.json 2> gpurun_out/final/bench_train.err; tail -c 1200 gpurun_out/final/bench_train.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 200 python tools/bwd_probe.py 5 2>&1 | tail -2 | tee gpurun_out/final/bwd_probe_warm.txt
find gpurun_out/final gpurun_out/prof_r02f gpurun_out/prof_r02f_train -type f ! -name "*.csv" ! -name "*.json" ! -name "*.txt" ! -name "*.log" ! -name "*.err" -delete
