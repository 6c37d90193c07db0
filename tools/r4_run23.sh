set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4w; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_nets.py tests/test_gpu_host_cpp.py -x -q -k "one_key_switch or fused or tilewise or full_size" > $O/tests.log 2>&1; echo "pytest exit $?" >> $O/tests.log; tail -15 $O/tests.log
grep -q "pytest exit 0" $O/tests.log || exit 1
timeout -k 10 600 python bench.py --config approx8192 --also wopad16384 --steps 2 --also-steps 2 --cpu-seconds 0 > $O/bench_approx_wopad.json 2> $O/bench.err
tail -3 $O/bench.err
