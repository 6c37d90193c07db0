set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s; mkdir -p $O
L=crcnn_amd/lib
cp $L/libcrcnn_hip.so /tmp/new.so
for rep in 1 2; do
for v in new prev; do
  if [ $v = new ]; then cp /tmp/new.so $L/libcrcnn_hip.so; else cp $L/libcrcnn_hip_prev.so $L/libcrcnn_hip.so; fi
  for g in "aconv2 64 3 limbk" "conv2p 128 3 limbk" "fc3 128 3 limbk"; do
    echo "== $v $g"; timeout -k 10 120 python tools/bench_mac.py $g 2>&1 | grep -v amdgpu
  done
done
done > $O/ab.txt 2>&1
cp /tmp/new.so $L/libcrcnn_hip.so
grep -E "^==|ms/launch" $O/ab.txt
