#!/bin/bash
# Static instruction histogram of the eval render kernel (render_kernel_p<64,64,0,1,true,true,false>) from a line-table build of ONE
# translation unit -- no GPU needed.   usage: tools/eval_kernel_static.sh [extra hipcc flags]   (output dir /tmp/ek)
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/ek && cd /tmp/ek
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -fno-gpu-rdc -munsafe-fp-atomics \
  -gline-tables-only -save-temps=obj "$@" -c $REPO/behindthescenes_amd/csrc/bts_fwd_proj.hip -o /tmp/ek/bts_fwd_proj.o
S=/tmp/ek/bts_fwd_proj-hip-amdgcn-amd-amdhsa-gfx950.s
python $REPO/tools/check_pk_opsel.py $S
python $REPO/tools/asm_line_hist.py $S render_kernel_pILi64ELi64ELi0ELi1ELb1ELb1ELb0 loop
awk '/^_ZN3bts15render_kernel_pILi64ELi64ELi0ELi1ELb1ELb1ELb0EEEvNS_9FwdParamsE:/{f=1} f&&/\.(sgpr_spill_count|vgpr_spill_count|vgpr_count|sgpr_count)|; (NumVgprs|NumSgprs|ScratchSize|Occupancy|SGPRSpill|VGPRSpill)/{print} /\.end_amdhsa_kernel/{if(f) exit}' $S | head
grep -A60 "_ZN3bts15render_kernel_pILi64ELi64ELi0ELi1ELb1ELb1ELb0EEEvNS_9FwdParamsE$" $S | grep -E "sgpr_spill|vgpr_spill|\.vgpr_count|\.sgpr_count" | head
