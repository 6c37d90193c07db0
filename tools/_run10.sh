cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05k; mkdir -p $O
(timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_nets.py -x -q -k "square or eight or approx8192 or wopad16384 or resident" 2>&1 | tail -6) > $O/tests.txt; cat $O/tests.txt
for rep in 1 2; do
for w in 0 1; do echo "CRC_F64_WAVE=$w"; (CRC_F64_WAVE=$w CRC_BENCH_SQ_POOL=1 python tools/bench_square.py 8192 3 5000; CRC_F64_WAVE=$w CRC_BENCH_SQ_POOL=1 python tools/bench_square.py 16384 4 2500) 2>&1 | grep -v amdgpu; done
done > $O/ab.txt 2>&1; cat $O/ab.txt
summ() { python3 - "$1" "$2" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
cts = int(sys.argv[2]) * 4
tot = 0
for r in csv.DictReader(open(f)):
    us = float(r["TotalDurationNs"]) / 1e3 / cts; tot += us
    if us > 0.02: print(f'{r["Name"][:70]:70s} {us:7.3f} us/ct')
print(f'{"sum":70s} {tot:7.3f} us/ct')
PY
}
for w in 0 1; do
CRC_F64_WAVE=$w CRC_BENCH_SQ_POOL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p8_$w -o p -- python3 tools/bench_square.py 8192 3 5000 > $O/p8_$w.log 2>&1
echo "== 8192/3 CRC_F64_WAVE=$w"; summ $O/p8_$w 5000
CRC_F64_WAVE=$w CRC_BENCH_SQ_POOL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p16_$w -o p -- python3 tools/bench_square.py 16384 4 2500 > $O/p16_$w.log 2>&1
echo "== 16384/4 CRC_F64_WAVE=$w"; summ $O/p16_$w 2500
done
