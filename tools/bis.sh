cd $GRAFT_REPO_ROOT
timeout -k 10 900 python bench.py --config wopad16384k8 --also none --steps 1 --batch 32 --cpu-seconds 0 > gpurun_out/bench_k8.json 2> gpurun_out/bench_k8.err
python - <<PY
import json
for l in open('gpurun_out/bench_k8.json'):
    if l.startswith('{'):
        d=json.loads(l); print('k8', d['value'], d['check'], d['ms_per_layer'], d['config'])
PY
tail -3 gpurun_out/bench_k8.err
timeout -k 10 900 python -m pytest tests/test_gpu_nets.py -x -q -k eight_primes 2>&1 | tail -3
