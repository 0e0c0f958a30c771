#!/bin/bash
# The round's closing GPU call, the LAST action of a round (no commit to bench.py, behindthescenes_amd/ or tests/ after it): the whole GPU
# suite exactly as the driver runs it (-x), smoke(), the default bench line, the same line under torch.distributed.run at N = 1 (as the
# driver launches N > 1) and steady-state kernel tables of the fused training steps and the eval frame.  The log names the tree it ran on.
#   (first, on the same tree: bash tools/gpu.sh <tag> profstep:train profstep:kitti_raw profstep:re10k profstep:re10k:128 -> copy the
#    traffic_step_*.json into profiles/<tag>/ and commit, so that the bench line below carries this tree's traffic_ratio)
#   gpurun --timeout 2700 -- 'bash tools/final_run.sh <tag> <git hash>'      -> gpurun_out/<tag>/ ; copy into profiles/<tag>/
cd ${GRAFT_REPO_ROOT:-.}
TAG=${1:-final}
mkdir -p gpurun_out/$TAG
echo "tree: ${2:-unknown}  ($(date -u +%FT%TZ); sha256 of bench.py $(sha256sum bench.py | cut -c1-16), of libbts_render.so $(sha256sum behindthescenes_amd/libbts_render.so | cut -c1-16))" | tee gpurun_out/$TAG/tree.txt
timeout 1700 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | tail -5 > gpurun_out/$TAG/pytest_gpu_x.txt; tail -3 gpurun_out/$TAG/pytest_gpu_x.txt
bash tools/gpu.sh $TAG smoke bench
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG/bench_torchrun_n1.json 2> gpurun_out/$TAG/bench_torchrun_n1.err; python tools/bench_summary.py gpurun_out/$TAG/bench_torchrun_n1.json | head -3
bash tools/gpu.sh $TAG steptrace:train:handover_kernel:--workload,train,--no-cpu-baseline,--no-other-layout,--steps,6,--warmup,3 steptrace:kitti_raw:handover_kernel:--workload,kitti_raw,--no-cpu-baseline,--no-other-layout,--steps,6,--warmup,3 steptrace:re10k:handover_kernel:--workload,re10k,--no-cpu-baseline,--no-other-layout,--steps,6,--warmup,3 steptrace:eval:gen_rays_kernel:--workload,eval,--no-cpu-baseline,--steps,12,--warmup,4 > gpurun_out/$TAG/steptraces.log 2>&1
