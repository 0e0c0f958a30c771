cd $GRAFT_REPO_ROOT
for m in "fwd" "train" "bwd_kitti_raw" "bwd_re10k 48" "bwd_re10k 128"; do
  bash tools/profile.sh r04zz $m > gpurun_out/prof_r04zz_$(echo $m | tr ' ' '_').log 2>&1
  tail -9 gpurun_out/prof_r04zz_$(echo $m | tr ' ' '_').log | cut -c1-260
done
ls gpurun_out/prof_r04zz
