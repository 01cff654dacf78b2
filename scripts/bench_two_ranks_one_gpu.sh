cd $GRAFT_REPO_ROOT
export GGAD_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 20 --warmup 5 --nodes 200000 --entries 4000000 --chunk 30 --e2e-steps 60 --e2e-reps 2 > gpurun_out/r06_bench_2ranks_one_gpu.json 2> gpurun_out/r06_bench_2ranks_one_gpu.err
echo rc $?
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_bench_2ranks_one_gpu.json') if l.startswith('{')][-1])
print(json.dumps(d['multi_gpu'], indent=1)[:3000]); print(d['value'], d['value_repeats']['median'])
PY
