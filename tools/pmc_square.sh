# HBM traffic of the Square + relinearise sequence from the PMC counters (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one pass):
#   tools/pmc_square.sh "<n> <k> <cts>" <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CFG=${1:-"8192 3 1250"}; TAG=${2:-sq}
O=gpurun_out/pmc_$TAG; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 tools/bench_square.py $CFG > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 tools/bench_square.py $CFG > $O/write.log 2>&1
python3 tools/pmc_square_summary.py $O "$CFG" > $O/summary.json
cat $O/summary.json
