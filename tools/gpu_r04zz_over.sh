cd $GRAFT_REPO_ROOT
O=gpurun_out/r04zz
for rep in 1 2; do for lib in "" over2 over4; do L=""; if [ -n "$lib" ]; then L=behindthescenes_amd/variants/libbts_$lib.so; fi; for w in train kitti_raw; do BTS_RENDER_LIB=$L timeout 200 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$w', '${lib:-default}', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))" | tee -a $O/scatter_oversub.txt; done; done; done
