cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05p; mkdir -p $O
(CRC_F64_WAVE=23 timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_nets.py -x -q -k "square or eight or approx8192_t42 or wopad16384_t44" 2>&1 | tail -4) > $O/tests.txt; cat $O/tests.txt
for rep in 1 2; do
for w in 7 23; do echo "CRC_F64_WAVE=$w"; (CRC_F64_WAVE=$w CRC_BENCH_SQ_POOL=1 python tools/bench_square.py 8192 3 5000; CRC_F64_WAVE=$w CRC_BENCH_SQ_POOL=1 python tools/bench_square.py 16384 4 2500) 2>&1 | grep -v amdgpu; done
done > $O/ab.txt 2>&1; cat $O/ab.txt
