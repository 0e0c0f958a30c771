cd $GRAFT_REPO_ROOT
O=gpurun_out/r04zz
mkdir -p $O
for rep in 1 2; do for lib in "" rounds3; do L=""; if [ -n "$lib" ]; then L=behindthescenes_amd/variants/libbts_$lib.so; fi; BTS_RENDER_LIB=$L timeout 200 python bench.py --workload re10k --samples 128 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('re10k_k128', '${lib:-default}', round(j['ms_per_step'],3), round(j['roofline']['fwd_ms'],3), round(j['roofline']['bwd_ms'],3))" | tee -a $O/scatter_rounds3.txt; done; done
