cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03d
python tools/debug_query.py > gpurun_out/r03d/debug_query.txt 2>&1
tail -30 gpurun_out/r03d/debug_query.txt
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "field_query or occupancy or golden" 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r03d/pytest_sel.txt
grep -n "^FAILED\|passed\|failed" gpurun_out/r03d/pytest_sel.txt | tail
