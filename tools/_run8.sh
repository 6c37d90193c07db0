cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05h; mkdir -p $O
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
for rep in 1 2; do
(cd .ab_base && CRC_BENCH_SQ_POOL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/base$rep -o b -- python3 tools/bench_square.py 8192 3 5000 > $O/base$rep.log 2>&1)
echo "== baseline (round start) $rep"; summ $O/base$rep 5000
CRC_BENCH_SQ_POOL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/new$rep -o n -- python3 tools/bench_square.py 8192 3 5000 > $O/new$rep.log 2>&1
echo "== new $rep"; summ $O/new$rep 5000
done
for cfg in "16384 4 2500"; do
(cd .ab_base && CRC_BENCH_SQ_POOL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/base16 -o b -- python3 tools/bench_square.py $cfg > $O/base16.log 2>&1)
echo "== baseline 16384"; summ $O/base16 2500
CRC_BENCH_SQ_POOL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/new16 -o n -- python3 tools/bench_square.py $cfg > $O/new16.log 2>&1
echo "== new 16384"; summ $O/new16 2500
done
for rep in 1 2; do
echo "== baseline"; (cd .ab_base && CRC_BENCH_SQ_POOL=1 python tools/bench_square.py 8192 3 5000; CRC_BENCH_SQ_POOL=1 python tools/bench_square.py 16384 4 2500) 2>&1 | grep -v amdgpu
echo "== new"; (CRC_BENCH_SQ_POOL=1 python tools/bench_square.py 8192 3 5000; CRC_BENCH_SQ_POOL=1 python tools/bench_square.py 16384 4 2500) 2>&1 | grep -v amdgpu
done
# kernel traces of the measured path itself (bench_host)
K=/tmp/crc_keep; rm -rf $K; mkdir -p $K
CRC_BENCH_KEEP=$K timeout -k 10 500 python bench.py --also approx8192 --cpu-seconds 0 --batch 256 --also-batch 96 --also-steps 1 --stream-inputs none > $O/prof_prepare.json 2> $O/prof_prepare.err
for cfg in tiny4096 approx8192; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$cfg -o $cfg -- $(cat $K/$cfg/cmd.txt) > $O/prof_$cfg.log 2>&1
done
rm -rf $K
ls $O
