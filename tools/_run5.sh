cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
free -g > $O/mem.txt; df -h /dev/shm >> $O/mem.txt; nproc >> $O/mem.txt
(timeout -k 10 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_host_cpp.py -x -q --durations=8 2>&1 | tail -25) > $O/tests.txt
cat $O/tests.txt
timeout -k 10 100 ./tools/ntt_butterfly_ab > $O/ntt_butterfly_ab.txt 2>&1; cat $O/ntt_butterfly_ab.txt
for ch in 0 1000 2000; do echo "CRC_SQ_CHUNK=$ch"; CRC_SQ_CHUNK=$ch CRC_BENCH_SQ_POOL=1 python tools/bench_square.py 8192 3 5000 2>&1 | grep -v amdgpu; done > $O/chunk.txt 2>&1; cat $O/chunk.txt
cat $O/mem.txt
