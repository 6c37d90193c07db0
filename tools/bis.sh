cd $GRAFT_REPO_ROOT
for ch in 32 64 128; do
timeout -k 10 400 python bench.py --cpu-seconds 0 --also none --unfused-images 0 --chunk $ch --steps 3 > gpurun_out/bis_$ch.json 2> gpurun_out/bis_$ch.err
python - <<PY
import json
for l in open('gpurun_out/bis_$ch.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$ch', d['value'], d['check']['all_ok'], d['ms_per_layer'])
PY
done
for ch in 64; do
timeout -k 10 400 python bench.py --config approx8192 --cpu-seconds 0 --also none --unfused-images 0 --chunk $ch --steps 1 > gpurun_out/bis_a$ch.json 2> gpurun_out/bis_a$ch.err
python - <<PY
import json
for l in open('gpurun_out/bis_a$ch.json'):
    if l.startswith('{'):
        d=json.loads(l); print('approx $ch', d['value'], d['check']['all_ok'], d['ms_per_layer'])
PY
tail -n 2 gpurun_out/bis_a$ch.err
done
