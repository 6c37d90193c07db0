set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4n; mkdir -p $O
timeout -k 10 300 python -m cProfile -o $O/prof.out -m pytest tests/test_gpu_nets.py -x -q -k "layerwise and wopad16384_t44 or configs0_in_full or ntt_resident and wopad16384_t44" > $O/tests.log 2>&1
python - <<'PY' > gpurun_out/r4n/top.txt
import pstats
p=pstats.Stats('gpurun_out/r4n/prof.out'); p.sort_stats('tottime').print_stats(35)
PY
tail -3 $O/tests.log
