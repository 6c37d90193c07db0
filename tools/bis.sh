cd $GRAFT_REPO_ROOT; O=gpurun_out/final; mkdir -p $O
timeout -k 10 500 python bench.py --steps 5 --warmup 1 > $O/bench_tiny4096.json 2> $O/bench_tiny4096.err
python3 - <<PY
import json
for f in ("bench_tiny4096",):
    for l in open("gpurun_out/final/%s.json"%f):
        if l.startswith("{"):
            d=json.loads(l); print(f, d["value"], d["check"]["all_ok"], d["check"]["golden_match"], d["ms_per_layer"], d["roofline"]["frac"], d["roofline"]["launch_ms"]); print("also", d["also"][0]["value"], d["also"][0]["check"]["all_ok"], d["also"][0]["ms_per_layer"])
PY
tail -n 2 $O/bench_tiny4096.err
