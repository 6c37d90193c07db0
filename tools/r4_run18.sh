set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4r; mkdir -p $O
( time timeout -k 10 900 python bench.py --steps 5 --warmup 1 ) > $O/bench_default_invocation.json 2> $O/bench_default.err
tail -4 $O/bench_default.err
