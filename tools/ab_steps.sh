#!/bin/bash
# A/B of two builds of the library on the training workloads within ONE gpurun call (box-to-box spread is +-4 %):
#   bash tools/ab_steps.sh <tag> <variant name under behindthescenes_amd/variants> [passes]
# -> gpurun_out/<tag>/ab_steps.txt: ms_per_step / kernel_ms of `default` and the variant, interleaved
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; VAR=$2; PASSES=${3:-2}
O=gpurun_out/$TAG; mkdir -p $O
: > $O/ab_steps.txt
for p in $(seq $PASSES); do
  for wl in train kitti_raw re10k "re10k --samples 128"; do
    for lib in default $VAR; do
      if [ $lib = default ]; then ENVV=""; else ENVV="BTS_RENDER_LIB=$PWD/behindthescenes_amd/variants/libbts_$lib.so BTS_ALLOW_LIB_OVERRIDE=1"; fi
      env $ENVV python bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline --no-others $AB_EXTRA 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']
print('$lib'.ljust(10), '$wl'.ljust(22), 'ms/step %.4f  kernel_ms %.4f  fwd %.4f  bwd %.4f' % (j['ms_per_step'], r['kernel_ms'], r.get('fwd_ms',0), r.get('bwd_ms',0)))" >> $O/ab_steps.txt
    done
  done
done
cat $O/ab_steps.txt
