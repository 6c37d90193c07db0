cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
( time timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=25 ) > $O/tests.log 2>&1; echo "pytest exit $?" >> $O/tests.log; tail -40 $O/tests.log
( time timeout -k 10 700 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; tail -c 3000 $O/bench.json
