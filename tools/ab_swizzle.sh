#!/bin/bash
# same-box A/B of the LDS swizzle (-DCRC_SWZ_KEEP_BIT0): bench two configurations with the shipped library, rebuild with the macro, bench again, rebuild back
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { for cfg in approx8192 tiny4096; do python3 bench.py --config $cfg --steps 2 --cpu-seconds 0 --also none --latency off --stream-inputs none > /tmp/o.json 2>/tmp/o.err || { echo "$1 $cfg FAILED"; tail -3 /tmp/o.err; continue; }
  python3 -c "
import json; l=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); print('$1', '$cfg', l['value'], {k: round(v, 4) for k, v in l['ms_per_layer'].items()}, 'ok', l['check']['all_ok'])"; done; }
run shipped
python3 tools/bench_square_pool.py 8192 3 32 2>&1 | grep -v amdgpu | tail -3
touch crcnn_amd/csrc/ntt_device.h
make -C crcnn_amd/csrc -j16 FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DCRC_SWZ_KEEP_BIT0" > /tmp/mk.log 2>&1 || { tail -5 /tmp/mk.log; exit 1; }
run keep_bit0
python3 tools/bench_square_pool.py 8192 3 32 2>&1 | grep -v amdgpu | tail -3
python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "square or ntt or baseline_ring" 2>&1 | tail -2
