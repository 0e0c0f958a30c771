#!/bin/bash
# A/B of several builds of the library on the training workloads within ONE gpurun call:
#   bash tools/ab_variants.sh <tag> "<variant> <variant> ..." [passes]      ("default" = the product library)
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; VARS=$2; PASSES=${3:-2}
O=gpurun_out/$TAG; mkdir -p $O
: > $O/ab_variants.txt
for p in $(seq $PASSES); do
  for wl in train kitti_raw re10k; do
    for lib in $VARS; do
      if [ $lib = default ]; then ENVV="BTS_AB_DUMMY=1"; else ENVV="BTS_RENDER_LIB=$PWD/behindthescenes_amd/variants/libbts_$lib.so BTS_ALLOW_LIB_OVERRIDE=1"; fi
      env $ENVV python bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline --no-others --no-other-layout 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']
print('$lib'.ljust(10), '$wl'.ljust(10), 'ms/step %.4f  kernel_ms %.4f  fwd %.4f  bwd %.4f' % (j['ms_per_step'], r['kernel_ms'], r.get('fwd_ms',0), r.get('bwd_ms',0)))" >> $O/ab_variants.txt
    done
  done
done
cat $O/ab_variants.txt
