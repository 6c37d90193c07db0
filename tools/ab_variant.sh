cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_limb.py -x -q 2>&1 | tail -2
for g in "conv2p 128" "fc3 128" "conv2p 32"; do
timeout -k 10 200 python tools/bench_mac.py $g 2 limbk 2>&1 | grep -v amdgpu | tail -1
done
