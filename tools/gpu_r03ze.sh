cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ze
mkdir -p $O
timeout 280 python -m pytest tests -x -q -m gpu --timeout 280 --tb=short -rf 2>&1 | tail -8 > $O/pytest_gpu.txt
tail -6 $O/pytest_gpu.txt
