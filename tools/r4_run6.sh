cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
( time timeout -k 10 900 python -m pytest tests/test_gpu_limb.py tests/test_gpu_layers.py -x -q -m gpu ) > $O/tests.log 2>&1; echo "pytest exit $?" >> $O/tests.log; tail -8 $O/tests.log
(python tools/check_conv1.py 4096 2 32 tiny; python tools/check_conv1.py 4096 2 128 tiny; python tools/check_conv1.py 8192 3 16 approx) 2>&1 | grep -v "amdgpu\|^[EW]2" > $O/conv1.txt; cat $O/conv1.txt
( time timeout -k 10 600 python bench.py --also approx8192 --cpu-seconds 0 --steps 3 ) > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; tail -3 $O/bench.err
python3 - <<PY
import json
for l in open("$O/bench.json"):
    if l.startswith("{"):
        d=json.loads(l)
        for e in [d]+d.get("also",[]):
            print(e["value"], e["ms_per_layer"], e["check"]["all_ok"])
PY
