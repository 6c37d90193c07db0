cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
for ch in 0 1000; do echo "CRC_SQ_CHUNK=$ch"; CRC_SQ_CHUNK=$ch CRC_BENCH_SQ_POOL=1 python tools/bench_square.py 16384 4 2500 2>&1 | grep -v amdgpu; done > $O/chunk16.txt 2>&1; cat $O/chunk16.txt
for ch in 750 1500; do echo "CRC_SQ_CHUNK=$ch"; CRC_SQ_CHUNK=$ch CRC_BENCH_SQ_POOL=1 python tools/bench_square.py 8192 3 6000 2>&1 | grep -v amdgpu; done > $O/chunk8.txt 2>&1; cat $O/chunk8.txt
( time timeout -k 10 900 python bench.py --steps 5 --warmup 1 --stream-inputs both ) > $O/bench.json 2> $O/bench.err; tail -5 $O/bench.err
python3 - <<'PY'
import json
for l in open("gpurun_out/r05f/bench.json"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["value"], d["check"], d["ms_per_layer"], d.get("streamed"))
        for a in d.get("also", []): print("  also", a["value"], a["check"]["all_ok"], a["ms_per_layer"], a.get("streamed"))
PY
