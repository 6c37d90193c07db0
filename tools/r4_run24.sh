set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4x; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_nets.py tests/test_gpu_host_cpp.py -x -q -k "square or one_key_switch or fused or tilewise" > $O/tests.log 2>&1; echo "pytest exit $?" >> $O/tests.log; tail -15 $O/tests.log
grep -q "pytest exit 0" $O/tests.log || exit 1
(timeout -k 10 200 python tools/bench_square_pool.py 8192 3 32; timeout -k 10 200 python tools/bench_square_pool.py 16384 4 6) 2>&1 | grep -v amdgpu > $O/sqpool.txt
cat $O/sqpool.txt
