#!/bin/bash
# The round's closing GPU call: the whole GPU suite, smoke(), the default bench line, the same line under torch.distributed.run at N = 1
# (as the driver launches N > 1) and steady-state kernel tables of the fused training steps and the eval frame.
#   gpurun --timeout 2700 -- 'bash tools/final_run.sh <tag>'      -> gpurun_out/<tag>/ ; copy into profiles/<tag>/
cd ${GRAFT_REPO_ROOT:-.}
TAG=${1:-final}
bash tools/gpu.sh $TAG pytest smoke bench
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG/bench_torchrun_n1.json 2> gpurun_out/$TAG/bench_torchrun_n1.err; python tools/bench_summary.py gpurun_out/$TAG/bench_torchrun_n1.json | head -3
bash tools/gpu.sh $TAG steptrace:train:handover_kernel:--workload,train,--no-cpu-baseline,--steps,6,--warmup,3 steptrace:kitti_raw:handover_kernel:--workload,kitti_raw,--no-cpu-baseline,--steps,6,--warmup,3 steptrace:re10k:handover_kernel:--workload,re10k,--no-cpu-baseline,--steps,6,--warmup,3 steptrace:eval:gen_rays_kernel:--workload,eval,--no-cpu-baseline,--steps,12,--warmup,4 > gpurun_out/$TAG/steptraces.log 2>&1
