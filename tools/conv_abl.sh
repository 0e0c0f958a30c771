#!/bin/bash
# Timing ablations of the decoder tail's bf16 forward kernel (csrc/bts_conv.hip, -DBTS_CONV_ABL=<bits>: 1 no split arithmetic, 2 one
# piece's loads per tile, 4 no stores).  Build the variants first, here:
#   for v in 1 2 4 7; do python -m behindthescenes_amd.build --tag cabl$v -DBTS_CONV_ABL=$v; done
# then on the GPU box:   gpurun -- 'bash tools/conv_abl.sh'      (results of round 5: profiles/r05o/README.txt)
cd ${GRAFT_REPO_ROOT:-.}
for v in default cabl1 cabl2 cabl4 cabl7; do
  if [ $v = default ]; then L=$PWD/behindthescenes_amd/libbts_render.so; else L=$PWD/behindthescenes_amd/variants/libbts_$v.so; fi
  [ -f $L ] || continue
  echo "== $v"; BTS_RENDER_LIB=$L BTS_ALLOW_LIB_OVERRIDE=1 python tools/conv_probe.py 5 2>&1 | grep -v amdgpu.ids
done
