cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05j
timeout -k 10 120 ./tools/f64_row_timeline 8192 4096 4 > gpurun_out/r05j/f64_row_timeline.txt 2>&1
grep -v "cycles of wave" gpurun_out/r05j/f64_row_timeline.txt
