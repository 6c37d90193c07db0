cd $GRAFT_REPO_ROOT
CRC_MFMA_VARIANT=2 timeout -k 10 300 python -m pytest tests/test_gpu_limb.py -x -q 2>&1 | tail -2
for v in 1 2; do for g in conv2p fc3 afc3 aconv2; do
echo -n "variant $v: "; CRC_MFMA_VARIANT=$v timeout -k 10 200 python tools/bench_mac.py $g 32 2 limbk 2>&1 | grep -v amdgpu | tail -1
done; done
