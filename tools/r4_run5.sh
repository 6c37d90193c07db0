cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
( time timeout -k 10 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_host_cpp.py tests/test_gpu_limb.py -x -q -m gpu --durations=8 ) > $O/tests.log 2>&1; echo "pytest exit $?" >> $O/tests.log; tail -22 $O/tests.log
( time timeout -k 10 900 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; tail -5 $O/bench.err
python3 - <<PY
import json
for l in open("$O/bench.json"):
    if l.startswith("{"):
        d=json.loads(l)
        for e in [d]+d.get("also",[]):
            print(e["value"], e["ms_per_step"], e["ms_per_layer"], e["check"], "\n   roof", {k:e["roofline"][k] for k in ("bound","frac","kernel","launch_ms")} if e["roofline"] else None, "\n   cpu", (e["cpu_baseline"] or {}).get("value"), "setup", e["setup_s"], "hbm", e["hbm_plan"]["parameters"]>>30, e["hbm_plan"]["activation_slots"]>>30)
PY
