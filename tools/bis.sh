cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --cpu-seconds 0 > gpurun_out/bis_D.json 2> gpurun_out/bis_D.err
python - <<PY
import json
for l in open('gpurun_out/bis_D.json'):
    if l.startswith('{'):
        d=json.loads(l); print('D', d['value'], d['check'], d['ms_per_layer']); a=d.get('also'); print(a if not a else (a[0]['value'], a[0]['check'], a[0]['ms_per_layer']) if isinstance(a,list) else a)
PY
tail -3 gpurun_out/bis_D.err
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -5 gpurun_out/gpu_tests.log
