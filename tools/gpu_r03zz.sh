cd $GRAFT_REPO_ROOT
O=gpurun_out/r03zz
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 600 --tb=short -rf --durations=4 2>&1 | tail -14 > $O/pytest_gpu.txt
tail -8 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --no-cpu-baseline > $O/bench_eval_quick.json 2>/dev/null; python -c "
import json; j=json.loads([l for l in open('$O/bench_eval_quick.json') if l.startswith('{')][0]); print('eval %.4g rays/s, step %.3f ms, kernel %.3f ms'%(j['value'], j['ms_per_step'], j['roofline']['kernel_ms']))"
