# per-kernel rocprof stats of Square + relinearise: tools/prof_square.sh "<n> <k> <cts>" <tag> [relin_path]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CFG=${1:-"8192 3 1250"}; TAG=${2:-sq}; export CRC_RELIN_PATH=${3:-0}
O=gpurun_out/prof_$TAG; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $TAG -- python3 tools/bench_square.py $CFG > $O/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
cts = int("$CFG".split()[2]) * 4          # bench_square runs the sequence 4 times (1 warm-up + 3 timed)
tot = 0
for r in rows:
    us = float(r["TotalDurationNs"]) / 1e3 / cts
    tot += us
    if us > 0.005: print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s}  {us:7.3f} us/ct')
print(f'{"sum":90s}              {tot:7.3f} us/ct')
PY
