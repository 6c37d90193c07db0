#!/bin/bash
# Round 6's evidence, ONE box, one call: the driver's own bench invocation, then rocprofv3 --kernel-trace --stats over the SAME bench_host command
# lines (the program directly after `--`), then the PMC passes behind roofline.traffic on the CURRENT kernel names.  Everything lands in gpurun_out/final6;
# tools/collect_round6.py copies the summaries into profiles/r06_*.
#   tools/measure_round6.sh first    the driver's bench invocation + kernel stats + the batch-1 PMC passes on the same box (needs the kept command lines: ~12 minutes)
#   tools/measure_round6.sh second   the other PMC passes (bytes do not depend on the box) and the micro-benchmarks (~10 minutes)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final6; mkdir -p $O
P=${1:-first}
K=/tmp/crc_keep6
if [ $P = first ]; then
  # the driver's invocation; CRC_BENCH_KEEP leaves every configuration's encrypted inputs and bench_host command line behind for the profiler runs below
  rm -rf $K; mkdir -p $K
  ( time CRC_BENCH_KEEP=$K CRC_BENCH_KEEP_CONFIGS=tiny4096,approx8192,wopad16384 timeout -k 10 1100 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_invocation.json 2> $O/bench_driver_invocation.err ) 2> $O/bench_driver_invocation.time
  tail -3 $O/bench_driver_invocation.time
fi
if [ $P = first ]; then
  # per-kernel stats of the measured path itself, same box, same process image: the bench_host command of each configuration with fewer steps
  for cfg in tiny4096 approx8192 wopad16384; do
    [ -f $K/$cfg/cmd_$cfg.txt ] || continue
    cmd=$(sed -e 's/ steps=[0-9]*/ steps=3/' -e 's/ warmup=[0-9]*/ warmup=1/' -e 's/ stream_inputs=[a-z,]*//' -e 's/ stream_steps=[0-9]*//' -e 's/ plain_inputs=[^ ]*//' $K/$cfg/cmd_$cfg.txt)
    timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$cfg -o $cfg -- $cmd > $O/prof_$cfg.log 2>&1 || echo "prof $cfg failed"
  done
  # the single-image runs (batch 1): the weight-stream kernel of the dense layers
  for cfg in tiny4096 approx8192; do
    [ -f $K/$cfg/cmd_${cfg}_b1.txt ] || continue
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${cfg}_b1 -o ${cfg}_b1 -- $(cat $K/$cfg/cmd_${cfg}_b1.txt) > $O/prof_${cfg}_b1.log 2>&1 || echo "prof $cfg b1 failed"
  done
  # batch-1 fc3 of PlainModelTiny (mac_stream_kernel): 34.4 GB of weights per launch
  if [ -f $K/tiny4096/cmd_tiny4096_b1.txt ]; then
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_b1_fetch -o f -- $(cat $K/tiny4096/cmd_tiny4096_b1.txt) > $O/pmc_b1_fetch.log 2>&1
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_b1_write -o w -- $(cat $K/tiny4096/cmd_tiny4096_b1.txt) > $O/pmc_b1_write.log 2>&1
  fi
fi
if [ $P = second ]; then
  # HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes, nothing but --kernel-trace beside them): the headline kernel on its bench launch, the one-channel
  # convolution, the Square + pooled key switch sequence at both rings, and the batch-1 weight stream
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 tools/bench_mac.py conv2p 128 1 limbk > $O/pmc_fetch.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 tools/bench_mac.py conv2p 128 1 limbk > $O/pmc_write.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_c1_fetch -o f -- python3 tools/check_conv1.py 4096 2 128 tiny > $O/pmc_c1_fetch.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_c1_write -o w -- python3 tools/check_conv1.py 4096 2 128 tiny > $O/pmc_c1_write.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o s -- python3 tools/bench_mac.py conv2p 128 1 limbk > $O/pmc_sq.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_grbm -o g -- python3 tools/bench_mac.py conv2p 128 1 limbk > $O/pmc_grbm.log 2>&1
  for cfg in "8192 3 1250" "16384 4 1250"; do tag=$(echo $cfg | tr ' ' '_')
    CRC_BENCH_SQ_POOL=1 bash tools/prof_square.sh "$cfg" sqpool6_$tag 0 > $O/prof_square_pool_$tag.txt 2>&1
    CRC_BENCH_SQ_POOL=1 bash tools/pmc_square.sh "$cfg" pool6_$tag > $O/pmc_square_pool_$tag.json 2> $O/pmc_square_pool_$tag.err
  done
  for cfg in "8192 3 1250" "16384 4 512"; do tag=$(echo $cfg | tr ' ' '_')
    bash tools/pmc_square.sh "$cfg" sq6_$tag > $O/pmc_square_$tag.json 2> $O/pmc_square_$tag.err
  done
fi
if [ $P = second ]; then
  (python3 tools/bench_ntt.py 4096 2 8192; python3 tools/bench_ntt.py 8192 3 4096; python3 tools/bench_ntt.py 16384 4 1024) > $O/ntt_elementwise.txt 2>&1
  (timeout -k 10 200 python3 tools/bench_square_pool.py 8192 3 32; timeout -k 10 200 python3 tools/bench_square_pool.py 16384 4 6) 2>&1 | grep -v amdgpu > $O/square_pool.txt
  python3 tools/bench_encrypt.py > $O/device_encryptor.txt 2>&1
fi
[ $P = first ] && rm -rf $K
ls $O
