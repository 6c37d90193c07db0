#!/bin/bash
# A/B of the batch-1 dense kernels on one box: CRC_MAC_STREAM=0 (mac3_kernel) against the four shapes of mac_stream_kernel, per-layer ms of a one-image forward
# usage: tools/ab_mac_stream.sh <out file> [configs...]
out=$1; shift
cfgs=${@:-tiny4096 approx8192}
: > $out
for cfg in $cfgs; do
  for s in 0 1 2 3 4; do
    CRC_MAC_STREAM=$s python3 bench.py --config $cfg --batch 1 --chunk 1 --distinct 1 --steps 10 --warmup 2 --cpu-seconds 0 --also none --latency off --stream-inputs none > /tmp/ab_ms.json 2>/tmp/ab_ms.err || { echo "$cfg shape $s FAILED" >> $out; tail -3 /tmp/ab_ms.err >> $out; continue; }
    python3 - "$cfg" "$s" >> $out <<'P'
import json, sys
l = json.loads(open('/tmp/ab_ms.json').read().strip().splitlines()[-1])
print(sys.argv[1], 'CRC_MAC_STREAM=' + sys.argv[2], 'ms/image', round(l['ms_per_step'], 3), {k: round(v, 3) for k, v in l['ms_per_layer'].items()}, 'GB/s', l['layer_hbm_GBps'], 'ok', l['check']['all_ok'])
P
  done
done
cat $out
