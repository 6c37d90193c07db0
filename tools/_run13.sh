cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
( time timeout -k 10 600 python -m pytest tests -x -q -m gpu --durations=15 ) > $O/gpu_tests.txt 2>&1; echo "pytest exit $?" >> $O/gpu_tests.txt; tail -5 $O/gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
for cfg in "8192 3 2500" "16384 4 1250"; do tag=$(echo $cfg | tr ' ' '_')
  CRC_BENCH_SQ_POOL=1 bash tools/pmc_square.sh "$cfg" pool_$tag > $O/pmc_square_pool_$tag.json 2> $O/pmc_square_pool_$tag.err
done
( time timeout -k 10 700 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_invocation.json 2> $O/bench_driver.err; tail -4 $O/bench_driver.err
