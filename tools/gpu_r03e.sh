cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
timeout 900 python -m pytest tests -m gpu -q --timeout 600 --tb=short -rf -k "real_training or multiscale or occupancy or oracle_autograd or train_step" 2>&1 | grep -v "^$" > gpurun_out/r03e/pytest_sel.txt
grep -n "^FAILED\|passed\|failed\|Error\|assert" gpurun_out/r03e/pytest_sel.txt | tail -40
