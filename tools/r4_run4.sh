cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
timeout -k 10 120 ./tools/lds_dma_unaligned > $O/dma.txt 2>&1; cat $O/dma.txt
( time timeout -k 10 900 python -m pytest tests/test_gpu_limb.py tests/test_gpu_threads.py -x -q -m gpu ) > $O/tests_limb.log 2>&1; echo "pytest exit $?" >> $O/tests_limb.log; tail -30 $O/tests_limb.log
( time timeout -k 10 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_host_cpp.py -x -q -m gpu -k "approx or wopad256 or tiny256 or api or example" ) > $O/tests_nets.log 2>&1; echo "pytest exit $?" >> $O/tests_nets.log; tail -15 $O/tests_nets.log
( time timeout -k 10 600 python bench.py --config approx8192 --also none --steps 2 --host-cpp 0 ) > $O/bench_approx.json 2> $O/bench_approx.err; tail -3 $O/bench_approx.err
python3 - <<PY
import json
for l in open("$O/bench_approx.json"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_layer"], d["check"]["all_ok"], d.get("mfma_useful_frac_per_layer"))
PY
