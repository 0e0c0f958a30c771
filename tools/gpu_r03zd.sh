cd $GRAFT_REPO_ROOT
O=gpurun_out/r03zd
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_protocol.py tests/test_gpu_parity.py -m gpu -q --timeout 300 --tb=short -rf -k "reference_golden and (noise or occupancy)" 2>&1 | tail -15 | tee $O/pytest_new_goldens.txt
