cd $GRAFT_REPO_ROOT
for ms in 16 8; do
CRC_MFMA_MIN_STEPS=$ms timeout -k 10 400 python bench.py --config approx8192 --also none --cpu-seconds 0 --unfused-images 0 --batch 256 > gpurun_out/bis_a$ms.json 2> gpurun_out/bis_a$ms.err
python - <<PY
import json
for l in open('gpurun_out/bis_a$ms.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$ms', d['value'], d['check']['all_ok'], d['check']['golden_match'], d['ms_per_layer'])
PY
tail -2 gpurun_out/bis_a$ms.err
done
