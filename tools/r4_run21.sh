set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4u; mkdir -p $O
L=crcnn_amd/lib
cp $L/libcrcnn_hip.so /tmp/new.so
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "square or relin" > $O/tests.log 2>&1; echo "pytest exit $?" >> $O/tests.log; tail -2 $O/tests.log
for rep in 1 2; do
for v in new prev; do
  if [ $v = new ]; then cp /tmp/new.so $L/libcrcnn_hip.so; else cp $L/libcrcnn_hip_prev.so $L/libcrcnn_hip.so; fi
  for cfg in "8192 3 1250" "16384 4 512"; do
    echo "== $v $cfg"; timeout -k 10 120 python tools/bench_square.py $cfg 2>&1 | grep -v amdgpu
  done
done
done > $O/ab.txt 2>&1
cp /tmp/new.so $L/libcrcnn_hip.so
grep -E "^==|us/ct" $O/ab.txt
