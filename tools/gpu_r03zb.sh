cd $GRAFT_REPO_ROOT
O=gpurun_out/r03zb
mkdir -p $O
for f in "" "--no-fused-handover"; do
  timeout 300 python bench.py --workload train --encoder monodepth2 $f --steps 6 --warmup 2 --no-cpu-baseline > "$O/bench_train_md2$f.json" 2> "$O/bench_train_md2$f.err"
done
timeout 300 python bench.py --workload re10k --encoder monodepth2 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_re10k_md2.json 2> $O/bench_re10k_md2.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload re10k --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_re10k_torchrun.json 2> $O/bench_re10k_torchrun.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][0])
        print(f.split("/")[-1], "value %.4g ms/step %.3f peak %.2f GB"%(j["value"], j["ms_per_step"], j["config"].get("peak_hbm_bytes",0)/1e9))
    except Exception as e:
        print(f, "ERR", e)
PY
