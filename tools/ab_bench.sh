cd $GRAFT_REPO_ROOT
for v in 1 2; do
CRC_MFMA_VARIANT=$v timeout -k 10 400 python bench.py --cpu-seconds 0 --unfused-images 0 --steps 3 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
python3 - <<PY
import json
for l in open("gpurun_out/ab_$v.json"):
    if l.startswith("{"):
        d=json.loads(l); print("variant $v", d["value"], d["check"]["all_ok"], d["ms_per_layer"]); print("   also", d["also"][0]["value"], d["also"][0]["check"]["all_ok"], d["also"][0]["ms_per_layer"])
PY
done
