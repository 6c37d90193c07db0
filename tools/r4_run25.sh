set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4y; mkdir -p $O
L=crcnn_amd/lib
cp $L/libcrcnn_hip.so /tmp/new.so
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "square" > $O/tests.log 2>&1; echo "pytest exit $?" >> $O/tests.log; tail -3 $O/tests.log
grep -q "pytest exit 0" $O/tests.log || exit 1
for rep in 1 2; do
for v in new prev; do
  if [ $v = new ]; then cp /tmp/new.so $L/libcrcnn_hip.so; else cp $L/libcrcnn_hip_prev.so $L/libcrcnn_hip.so; fi
  echo "== $v"; (timeout -k 10 200 python tools/bench_square_pool.py 8192 3 32; timeout -k 10 200 python tools/bench_square_pool.py 16384 4 6) 2>&1 | grep "square_pool_relin"
done
done > $O/ab.txt 2>&1
cp /tmp/new.so $L/libcrcnn_hip.so
cat $O/ab.txt
