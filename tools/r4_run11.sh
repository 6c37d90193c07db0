cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k; mkdir -p $O
bash tools/prof_square.sh "16384 4 512" r4k_r3 > $O/prof_r3.txt 2>&1
CRC_F64_RADIX=4 bash tools/prof_square.sh "16384 4 512" r4k_r4 > $O/prof_r4.txt 2>&1
CRC_F64_RADIX=4 bash tools/prof_square.sh "8192 3 1250" r4k_r4s > $O/prof_r4s.txt 2>&1
cat $O/prof_r3.txt $O/prof_r4.txt $O/prof_r4s.txt | grep -v "at::\|rocclr\|evk_canon\|keys_f64"
