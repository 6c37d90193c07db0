#!/bin/bash
# same-box A/B of the wave-local 64-bit row transform at n = 2048 (round 6: one cross stage): CRC_NTT_WAVE=15 (the round-4 kernel this ring ran until now) against the default
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in 15 47; do
  echo "CRC_NTT_WAVE=$w"
  CRC_NTT_WAVE=$w python3 tools/bench_ntt.py 2048 1 16384 2>&1 | grep -v amdgpu | head -3
  CRC_NTT_WAVE=$w python3 bench.py --config tiny2048r --steps 2 --cpu-seconds 0 --also none --latency off --stream-inputs none > /tmp/o.json 2>/tmp/o.err || { tail -3 /tmp/o.err; continue; }
  python3 -c "
import json; l=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); print('tiny2048r', l['value'], {k: round(v, 4) for k, v in l['ms_per_layer'].items()}, 'ok', l['check']['all_ok'], l['check']['golden_match'])"
done
