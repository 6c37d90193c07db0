set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4t; mkdir -p $O
for rep in 1 2; do for pf in 0 1; do for cfg in "8192 3 1250" "16384 4 512"; do
  echo "== prefetch=$pf $cfg"; CRC_SQ_PREFETCH=$pf timeout -k 10 120 python tools/bench_square.py $cfg 2>&1 | grep -v amdgpu
done; done; done > $O/ab.txt 2>&1
grep -E "^==|us/ct" $O/ab.txt
