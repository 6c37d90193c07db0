cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i; mkdir -p $O
(python tools/check_conv1.py 4096 2 128 tiny; python tools/check_conv1.py 8192 3 16 approx) 2>&1 | grep -v "amdgpu\|^[EW]2" > $O/conv1.txt; cat $O/conv1.txt
( time timeout -k 10 900 python -m pytest tests/test_gpu_limb.py -x -q -m gpu ) > $O/tests.log 2>&1; echo "pytest exit $?" >> $O/tests.log; tail -4 $O/tests.log
bash tools/pmc_conv1_issue.sh 2>&1 | tail -2
