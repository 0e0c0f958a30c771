cd $GRAFT_REPO_ROOT
O=gpurun_out/r03w
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 600 --tb=short -rf --durations=6 2>&1 | tail -30 > $O/pytest_gpu.txt
tail -12 $O/pytest_gpu.txt
bash tools/profile.sh r03w fwd > $O/profile_fwd.log 2>&1; tail -3 $O/profile_fwd.log
bash tools/profile.sh r03w_re10k bwd_re10k > $O/profile_bwd_re10k.log 2>&1; tail -6 $O/profile_bwd_re10k.log
bash tools/profile.sh r03w_train train > $O/profile_train.log 2>&1; tail -12 $O/profile_train.log
