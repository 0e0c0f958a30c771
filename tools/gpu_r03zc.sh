cd $GRAFT_REPO_ROOT
O=gpurun_out/r03zc
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 --tb=short -rf -k "occupancy_profile_vs_reference_golden" 2>&1 | tail -15 | tee $O/pytest_profile_golden.txt
