#!/bin/bash
# A/B of ONE build with and without an environment switch on the training workloads within one gpurun call:
#   bash tools/ab_env.sh <tag> <VAR=value> [passes] [extra bench args]      -> gpurun_out/<tag>/ab_env.txt
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; SW=$2; PASSES=${3:-2}; EXTRA=${4:-}
O=gpurun_out/$TAG; mkdir -p $O
: > $O/ab_env.txt
for p in $(seq $PASSES); do
  for wl in train kitti_raw re10k; do
    for mode in on off; do
      if [ $mode = on ]; then ENVV="BTS_AB_DUMMY=1"; else ENVV="$SW"; fi
      env $ENVV python bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline --no-others --no-other-layout $EXTRA 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']
print(('default' if '$mode'=='on' else '$SW').ljust(22), '$wl'.ljust(10), 'ms/step %.4f  kernel_ms %.4f  fwd %.4f  bwd %.4f' % (j['ms_per_step'], r['kernel_ms'], r.get('fwd_ms',0), r.get('bwd_ms',0)))" >> $O/ab_env.txt
    done
  done
done
cat $O/ab_env.txt
