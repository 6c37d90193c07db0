# the whole GPU suite with the slowest cases listed (what profiles/rNN_gpu_tests.txt holds)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/suite; mkdir -p $O
( time timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=25 ) > $O/tests.log 2>&1; echo "pytest exit $?" >> $O/tests.log
tail -8 $O/tests.log
