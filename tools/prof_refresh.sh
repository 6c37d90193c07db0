cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
K=/tmp/keepr; rm -rf $K; mkdir -p $K
CRC_BENCH_KEEP=$K python3 bench.py --config approx4096r --steps 1 --cpu-seconds 0 --also none --latency off > /dev/null 2>/tmp/e.err || tail -3 /tmp/e.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_refresh -o r -- $(cat $K/approx4096r/cmd_approx4096r.txt) > gpurun_out/prof_refresh.log 2>&1
python3 - <<'P'
import csv,glob
f=glob.glob('gpurun_out/prof_refresh/**/r_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:28]: print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {100*float(r['TotalDurationNs'])/tot:5.1f}")
P
