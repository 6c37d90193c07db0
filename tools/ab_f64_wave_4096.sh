#!/bin/bash
# same-box A/B of the wave-local fp64 kernels at n = 4096 (round 6: CS = 2): CRC_F64_WAVE=0 (the round-4 kernels this ring ran until now) against the default
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in 0 7; do
  echo "CRC_F64_WAVE=$w"
  CRC_F64_WAVE=$w python3 tools/bench_square_pool.py 4096 2 64 2>&1 | grep -v amdgpu | tail -3
  CRC_F64_WAVE=$w python3 bench.py --config approx4096r --steps 2 --cpu-seconds 0 --also none --latency off --stream-inputs none > /tmp/o.json 2>/tmp/o.err || { tail -3 /tmp/o.err; continue; }
  python3 -c "
import json; l=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); print('approx4096r', l['value'], {k: round(v, 4) for k, v in l['ms_per_layer'].items()}, 'ok', l['check']['all_ok'], l['check']['golden_match'])"
done
