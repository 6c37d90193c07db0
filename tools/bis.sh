cd $GRAFT_REPO_ROOT; O=gpurun_out/final; mkdir -p $O
timeout -k 10 500 python bench.py --steps 5 --warmup 1 > $O/bench_tiny4096.json 2> $O/bench_tiny4096.err
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 - <<PY
import json
for f in ("bench_tiny4096","bench_default"):
    for l in open("gpurun_out/final/%s.json"%f):
        if l.startswith("{"):
            d=json.loads(l); print(f, d["value"], d["check"], d["ms_per_layer"], d["roofline"]); print("also", d["also"][0]["value"], d["also"][0]["check"]["all_ok"]); print(d["cpu_baseline"])
PY
tail -n 2 $O/bench_default.err
