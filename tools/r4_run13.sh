set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4m; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_host_cpp.py -x -q -k "streamed or tilewise" > $O/tests.log 2>&1; echo "pytest exit $?" >> $O/tests.log
timeout -k 10 700 python bench.py --config wopad16384k8 --also none --steps 1 --batch 96 --cpu-seconds 0 > $O/bench_wopad16384k8_b96.json 2> $O/bench_wopadk8.err
echo "bench exit $?"
tail -3 $O/tests.log; tail -5 $O/bench_wopadk8.err; cat $O/bench_wopad16384k8_b96.json
