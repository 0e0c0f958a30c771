cd $GRAFT_REPO_ROOT
O=gpurun_out/r04zz
mkdir -p $O
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_torchrun_n1.json 2> $O/bench_torchrun_n1.err
python - <<PY
import json
j=json.loads([l for l in open("$O/bench_torchrun_n1.json") if l.startswith("{")][0])
print("eval", j["value"], j["ms_per_step"], j["n_gpus"])
print("ddp_train", json.dumps(j.get("ddp_train") or (j.get("others") or {}).get("ddp_train"))[:900])
print(list((j.get("others") or {}).keys()))
PY
tail -3 $O/bench_torchrun_n1.err
