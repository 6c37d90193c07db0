cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py --config approx8192 --also none --steps 2 --cpu-seconds 0 > gpurun_out/bis_a.json 2> gpurun_out/bis_a.err
python - <<PY
import json
for f in ('a',):
  for l in open('gpurun_out/bis_%s.json'%f):
    if l.startswith('{'):
        d=json.loads(l); print(f, d['value'], d['check']['all_ok'], d['check']['golden_match'], d['ms_per_layer'], d['mac_kernel_per_layer'], d['data'])
PY
tail -n 2 gpurun_out/bis_a.err
