set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4z; mkdir -p $O
timeout -k 10 500 python bench.py --config approx8192 --also none --steps 1 --batch 128 --python-twin --cpu-seconds 0 > $O/bench_approx_twin.json 2> $O/bench.err
echo "exit $?"; tail -3 $O/bench.err
