cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_layers.py -x -q -m gpu > $O/tests.log 2>&1; echo "pytest exit $?" >> $O/tests.log; tail -5 $O/tests.log
(for cfg in "8192 3 1250" "8192 4 1250" "16384 4 512" "16384 8 256" "4096 2 2048"; do echo "== $cfg"; python tools/bench_square.py $cfg 2>&1 | grep -v amdgpu; done) > $O/sq.txt 2>&1
cat $O/sq.txt
bash tools/prof_square.sh "8192 3 1250" r4b > $O/prof.txt 2>&1; cat $O/prof.txt
bash tools/prof_square.sh "16384 4 512" r4b16 > $O/prof16.txt 2>&1; cat $O/prof16.txt
