# round-4 A/B of the Square + relinearise chain: r3 library vs the fused / register-holding kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $O/test_ops.log 2>&1; echo "pytest exit $?" >> $O/test_ops.log; tail -3 $O/test_ops.log
L=crcnn_amd/lib
cp $L/libcrcnn_hip.so /tmp/new.so
(
for cfg in "8192 3 1250" "16384 4 512" "16384 8 256"; do
  echo "== $cfg"
  cp $L/libcrcnn_hip_r3.so $L/libcrcnn_hip.so; echo "r3 library"; python tools/bench_square.py $cfg 2>&1 | grep -v amdgpu
  cp /tmp/new.so $L/libcrcnn_hip.so
  echo "new default"; python tools/bench_square.py $cfg 2>&1 | grep -v amdgpu
  echo "new CRC_F64_HOLD_NT=512"; CRC_F64_HOLD_NT=512 python tools/bench_square.py $cfg 2>&1 | grep -v amdgpu
  echo "new CRC_SQ_FUSE=0"; CRC_SQ_FUSE=0 python tools/bench_square.py $cfg 2>&1 | grep -v amdgpu
done
) > $O/ab.txt 2>&1
cat $O/ab.txt
bash tools/prof_square.sh "8192 3 1250" r4a_A > $O/prof_A.txt 2>&1
CRC_F64_HOLD_NT=512 bash tools/prof_square.sh "8192 3 1250" r4a_B > $O/prof_B.txt 2>&1
cat $O/prof_A.txt $O/prof_B.txt
bash tools/pmc_square.sh "8192 3 1250" r4a_A > $O/pmc_A.json 2> $O/pmc_A.err
CRC_F64_HOLD_NT=512 bash tools/pmc_square.sh "8192 3 1250" r4a_B > $O/pmc_B.json 2> $O/pmc_B.err
python3 - <<PY
import json
for t in "AB":
    try:
        d=json.load(open("$O/pmc_%s.json"%t)); print(t, d.get("total"), {k[:44]:(v["read"],v["write"]) for k,v in d["kernels"].items()})
    except Exception as e: print(t, "pmc failed", e)
PY
