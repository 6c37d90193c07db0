set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
P=${1:-all}
if [ $P = all ] || [ $P = bench ]; then
timeout -k 10 900 python bench.py --steps 5 --warmup 1 > $O/bench_default_invocation.json 2> $O/bench_default.err
timeout -k 10 500 python bench.py --config approx8192k4 --also none --steps 1 --batch 256 > $O/bench_approx8192k4_b256.json 2> $O/bench_approx8192k4.err
timeout -k 10 700 python bench.py --config wopad16384k8 --also none --steps 1 --batch 96 --cpu-seconds 0 > $O/bench_wopad16384k8_b96.json 2> $O/bench_wopadk8.err
timeout -k 10 500 python bench.py --also none --steps 3 --python-twin --unfused-images 128 > $O/bench_tiny4096_with_python_twin.json 2> $O/bench_tiny4096_twin.err
fi
if [ $P = all ] || [ $P = prof ]; then
# per-kernel traces of the measured path itself: bench.py leaves the encrypted inputs and the bench_host command line behind (CRC_BENCH_KEEP), rocprofv3 runs that command
K=/tmp/crc_keep; rm -rf $K; mkdir -p $K
CRC_BENCH_KEEP=$K timeout -k 10 400 python bench.py --also approx8192 --cpu-seconds 0 --batch 256 --also-batch 96 --also-steps 1 > $O/prof_prepare.json 2> $O/prof_prepare.err
for cfg in tiny4096 approx8192; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$cfg -o $cfg -- $(cat $K/$cfg/cmd.txt) > $O/prof_$cfg.log 2>&1
done
rm -rf $K
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sq -o sq -- python3 tools/bench_square.py 8192 3 1250 > $O/prof_sq.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c1 -o c1 -- python3 tools/check_conv1.py 4096 2 128 tiny > $O/prof_c1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_c1_fetch -o f -- python3 tools/check_conv1.py 4096 2 128 tiny > $O/pmc_c1_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_c1_write -o w -- python3 tools/check_conv1.py 4096 2 128 tiny > $O/pmc_c1_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o s -- python3 tools/bench_mac.py conv2p 128 1 limbk > $O/pmc_sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_grbm -o g -- python3 tools/bench_mac.py conv2p 128 1 limbk > $O/pmc_grbm.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 tools/bench_mac.py conv2p 128 1 limbk > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 tools/bench_mac.py conv2p 128 1 limbk > $O/pmc_write.log 2>&1
fi
if [ $P = all ] || [ $P = square ]; then
# Square + relinearise: per-kernel stats and PMC traffic at the three rings, and the round-2 key switching (CRC_RELIN_PATH=1) beside it
for cfg in "8192 3 1250" "16384 4 512" "16384 8 256"; do tag=$(echo $cfg | tr ' ' '_')
  bash tools/prof_square.sh "$cfg" sq_$tag 0 > $O/prof_square_$tag.txt 2>&1
  CRC_SQ_PATH=1 bash tools/prof_square.sh "$cfg" sqold_$tag 1 > $O/prof_square_old_$tag.txt 2>&1
  bash tools/pmc_square.sh "$cfg" $tag > $O/pmc_square_$tag.json 2> $O/pmc_square_$tag.err
done
# the layer pair the fused networks run: Square + pooling with one key switch per pooled ciphertext (per squared ciphertext, whole images of 1250)
for cfg in "8192 3 1250" "16384 4 1250"; do tag=$(echo $cfg | tr ' ' '_')
  CRC_BENCH_SQ_POOL=1 bash tools/prof_square.sh "$cfg" sqpool_$tag 0 > $O/prof_square_pool_$tag.txt 2>&1
  CRC_BENCH_SQ_POOL=1 bash tools/pmc_square.sh "$cfg" pool_$tag > $O/pmc_square_pool_$tag.json 2> $O/pmc_square_pool_$tag.err
done
(timeout -k 10 200 python tools/bench_square_pool.py 8192 3 32; timeout -k 10 200 python tools/bench_square_pool.py 16384 4 6) 2>&1 | grep -v amdgpu > $O/square_pool.txt
(for cfg in "8192 3 1250" "8192 4 1250" "16384 4 512" "16384 8 256"; do
   echo "round-2 kernels (CRC_SQ_PATH=1 CRC_RELIN_PATH=1: SEAL's 61-bit auxiliary base, key switching over the coefficient moduli)"; CRC_SQ_PATH=1 CRC_RELIN_PATH=1 python tools/bench_square.py $cfg 2>&1 | grep -v amdgpu
   echo "key switching over two fp64 primes, SEAL's auxiliary base (CRC_SQ_PATH=1)"; CRC_SQ_PATH=1 python tools/bench_square.py $cfg 2>&1 | grep -v amdgpu
   echo "key switching and auxiliary base over fp64 primes (CRC_SQ_PATH=2)"; CRC_SQ_PATH=2 python tools/bench_square.py $cfg 2>&1 | grep -v amdgpu
   echo "default"; python tools/bench_square.py $cfg 2>&1 | grep -v amdgpu
 done) > $O/square_paths.txt
fi
if [ $P = all ] || [ $P = micro ]; then
(python tools/bench_ntt.py 4096 2 8192; python tools/bench_ntt.py 8192 3 4096; python tools/bench_ntt.py 16384 4 1024) > $O/ntt_elementwise.txt 2>&1
(python tools/bench_square.py 8192 3 1250; python tools/bench_square.py 16384 4 512; python tools/bench_square.py 8192 3 1250 0 0; python tools/bench_square.py 8192 4 1250) > $O/square.txt 2>&1
rm -f $O/mac_geometries.txt
for g in conv1 conv1p conv2 conv2p fc3 aconv1 aconv2 afc3; do python tools/bench_mac.py $g 32 2 packed 2>&1 | grep -v amdgpu >> $O/mac_geometries.txt; done
(python tools/check_conv1.py 4096 2 32 tiny; python tools/check_conv1.py 8192 3 16 approx) 2>&1 | grep -v "amdgpu\|^[EW]2" > $O/conv1.txt
(timeout -k 10 120 ./tools/mfma_shape; timeout -k 10 120 ./tools/mfma_shape zeros) > $O/mfma_shape.txt 2>&1
(python tools/bench_pack.py 8192 3 1250; python tools/bench_pack.py 16384 8 3920) 2>&1 | grep -v amdgpu > $O/pack.txt
for g in conv2p fc3 aconv2 afc3; do python tools/bench_mac.py $g 32 2 limb 2>&1 | grep -v amdgpu >> $O/mac_geometries.txt; python tools/bench_mac.py $g 32 2 limbk 2>&1 | grep -v amdgpu >> $O/mac_geometries.txt; done
fi
ls $O
