cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g
mkdir -p $O
for b in lds_dma_ring_hd32 lds_dma_ringfetch_late_hd32 lds_dma_ringfetch_latewait_all_hd32 lds_dma_ring_hd64 lds_dma_ringfetch_late_hd64; do
  timeout 300 tools/ubench/$b 16 >> $O/lds_dma_ring.txt 2>&1
done
cat $O/lds_dma_ring.txt
LIB_AB_PASSES=2 python tools/bwd_ab.py re10k 48 default b1 b2 b3 b4 b5 b6 b35 > $O/bwd_ab_re10k.txt 2>&1
cat $O/bwd_ab_re10k.txt
timeout 600 python -m pytest tests -m gpu -q --timeout 600 --tb=short -rf -k "handover or two_pass" 2>&1 | grep -v "^$" | tail -30 > $O/pytest_sel.txt
grep -n "^FAILED\|passed\|failed\|gradient" $O/pytest_sel.txt | tail
for f in "" "--no-fused-handover"; do
  timeout 300 python bench.py --workload train --encoder monodepth2 $f --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_train_md2$f.json 2> $O/bench_train_md2$f.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/bench_train_md2$f.json") if l.startswith("{")][0]); print("MD2 [$f] ms/step %.3f value %.4g peak %.2f GB"%(j["ms_per_step"],j["value"],j["config"]["peak_hbm_bytes"]/1e9))
except Exception as e:
    print("MD2 [$f] failed", e); print(open("$O/bench_train_md2$f.err").read()[-1500:])
PY
done
