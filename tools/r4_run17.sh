set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4q; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_limb.py -x -q > $O/tests.log 2>&1; echo "pytest exit $?" >> $O/tests.log
tail -3 $O/tests.log
grep -q "pytest exit 0" $O/tests.log || exit 1
for g in "aconv2 64 3 limb" "aconv2 64 3 limbk" "conv2p 128 3 limbk" "fc3 128 3 limbk" "afc3 64 3 limb"; do
  echo "== $g"; timeout -k 10 120 python tools/bench_mac.py $g 2>&1 | grep -v amdgpu
done > $O/ab.txt 2>&1
grep -E "^==|ms/launch" $O/ab.txt
