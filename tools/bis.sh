cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py --config approx8192 --also none --steps 2 --cpu-seconds 0 > gpurun_out/bis_a.json 2> gpurun_out/bis_a.err
timeout -k 10 500 python bench.py --config wopad16384 --also none --steps 1 --batch 96 --cpu-seconds 0 > gpurun_out/bis_w.json 2> gpurun_out/bis_w.err
python - <<PY
import json
for f in ('a','w'):
  for l in open('gpurun_out/bis_%s.json'%f):
    if l.startswith('{'):
        d=json.loads(l); print(f, d['value'], d['check']['all_ok'], d['check']['golden_match'], d['ms_per_layer'], d['mac_kernel_per_layer'])
PY
tail -n 2 gpurun_out/bis_a.err gpurun_out/bis_w.err
